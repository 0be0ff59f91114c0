// Dev probe: ablation timing of the MFMA actor front end (copy of pw_actor_front_kernel with switches).
// hipcc --offload-arch=gfx950 -O3 -o /tmp/front_probe tools/front_probe.hip && /tmp/front_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ int mfma_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

// MODE bits: 1 = skip G store, 2 = skip stage-2 MFMA, 4 = skip LDS fill, 8 = skip transpose (store acc directly, wrong layout)
template <int S1C, int MODE, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) front(const float *__restrict__ X, const float *__restrict__ frag,
                                                    const float *__restrict__ b1, const float *__restrict__ bih,
                                                    const long rows, const int D, float *__restrict__ G)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int S1 = 4 * S1C;
    float4 *f_wih = reinterpret_cast<float4 *>(smem_raw);
    float *f_w1 = reinterpret_cast<float *>(f_wih + 8 * 2 * 4 * 64);
    float *s_b1 = f_w1 + 2 * S1 * 64;
    float *s_bih = s_b1 + 64;
    float *s_t = s_bih + 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
    if (!(MODE & 4)) {
        const float4 *src = reinterpret_cast<const float4 *>(frag);
        const int n4 = 8 * 2 * 4 * 64 + (2 * S1 * 64) / 4;
        for (int f = tid; f < n4; f += WAVES * 64) f_wih[f] = src[f];
    }
    if (tid < 64) s_b1[tid] = b1[tid];
    if (tid < 256) s_bih[tid] = bih[tid];
    __syncthreads();
    const long row0 = ((long)blockIdx.x * WAVES + wave) * 32;
    if (row0 >= rows) return;
    long myrow = row0 + col;
    if (myrow >= rows) myrow = rows - 1;
    f32x16 acc1[2];
    for (int m = 0; m < 2; ++m)
        for (int r = 0; r < 16; ++r) acc1[m][r] = 0.0f;
    const float *xr = X + (size_t)myrow * D;
    float xb[S1];
#pragma unroll
    for (int s = 0; s < S1; ++s) { const int k = 2 * s + half; xb[s] = k < D ? xr[k] : 0.0f; }
#pragma unroll
    for (int s = 0; s < S1; ++s)
#pragma unroll
        for (int m = 0; m < 2; ++m)
            acc1[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(f_w1[(m * S1 + s) * 64 + lane], xb[s], acc1[m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[m][r] = fmaxf(acc1[m][r] + s_b1[m * 32 + mfma_row(r, half)], 0.0f);
    float *patch = s_t + wave * 32 * 33;
#pragma unroll 1
    for (int n = 0; n < 8; ++n) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = (MODE & 2) ? acc1[n & 1][r] : 0.0f;
        if (!(MODE & 2)) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const float4 a = f_wih[((n * 2 + m) * 4 + rq) * 64 + lane];
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, acc1[m][4 * rq + 0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, acc1[m][4 * rq + 1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, acc1[m][4 * rq + 2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, acc1[m][4 * rq + 3], acc, 0, 0, 0);
                }
        }
        if (MODE & 8) {
            if (!(MODE & 1)) {
                float4 *dst = reinterpret_cast<float4 *>(G + (size_t)(row0 + (lane >> 1)) * 256 + n * 32 + (lane & 1) * 16);
#pragma unroll
                for (int q = 0; q < 4; ++q) dst[q] = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
            } else if (acc[0] == 12345.f) G[0] = acc[1] + acc[5] + acc[9] + acc[15];
            continue;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) patch[col * 33 + mfma_row(r, half)] = acc[r];
        wave_lds_sync();
        {
            const int rr = lane >> 1, u0 = (lane & 1) * 16;
            const long orow = row0 + rr;
            if (orow < rows) {
                float4 *dst = reinterpret_cast<float4 *>(G + (size_t)orow * 256 + n * 32 + u0);
                float4 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float *src = patch + rr * 33 + u0 + 4 * q;
                    const float *bb = s_bih + n * 32 + u0 + 4 * q;
                    v[q] = make_float4(src[0] + bb[0], src[1] + bb[1], src[2] + bb[2], src[3] + bb[3]);
                }
                if (!(MODE & 1)) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) dst[q] = v[q];
                } else if (v[0].x == 12345.f) G[0] = v[1].y + v[2].z + v[3].w;
            }
        }
        wave_lds_sync();
    }
}

__global__ void fill_only(float4 *G, long n4) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) G[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

template <int MODE, int WAVES>
static void run(const char *name, const float *X, const float *frag, const float *b1, const float *bih, long rows, int D, float *G)
{
    constexpr int S1C = 2;
    const size_t shm = (size_t)8 * 2 * 4 * 64 * 16 + (size_t)(2 * 4 * S1C * 64 + 64 + 256 + WAVES * 32 * 33) * 4;
    hipFuncSetAttribute(reinterpret_cast<const void *>(front<S1C, MODE, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const long per = WAVES * 32;
    const unsigned grid = (unsigned)((rows + per - 1) / per);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((front<S1C, MODE, WAVES>), dim3(grid), dim3(WAVES * 64), shm, 0, X, frag, b1, bih, rows, D, G);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL((front<S1C, MODE, WAVES>), dim3(grid), dim3(WAVES * 64), shm, 0, X, frag, b1, bih, rows, D, G);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s waves/WG %d grid %4u : %7.2f us/launch (%s)\n", name, WAVES, grid, ms * 1000.f / 200.f, hipGetErrorString(hipGetLastError()));
}

int main()
{
    const long rows = 24576; const int D = 16;
    float *X, *frag, *b1, *bih, *G;
    hipMalloc(&X, rows * D * 4); hipMalloc(&frag, 1 << 20); hipMalloc(&b1, 256); hipMalloc(&bih, 1024); hipMalloc(&G, rows * 256 * 4);
    hipMemset(X, 0, rows * D * 4); hipMemset(frag, 0, 1 << 20); hipMemset(b1, 0, 256); hipMemset(bih, 0, 1024);
    run<0, 4>("full", X, frag, b1, bih, rows, D, G);
    run<1, 4>("no G store", X, frag, b1, bih, rows, D, G);
    run<2, 4>("no stage-2 MFMA", X, frag, b1, bih, rows, D, G);
    run<4, 4>("no LDS fill", X, frag, b1, bih, rows, D, G);
    run<8, 4>("no transpose (direct acc store)", X, frag, b1, bih, rows, D, G);
    run<9, 4>("no transpose, no store", X, frag, b1, bih, rows, D, G);
    run<1 | 4, 4>("no fill, no store", X, frag, b1, bih, rows, D, G);
    run<0, 8>("full", X, frag, b1, bih, rows, D, G);
    run<0, 2>("full", X, frag, b1, bih, rows, D, G);
    run<0, 1>("full", X, frag, b1, bih, rows, D, G);
    run<4, 1>("no LDS fill", X, frag, b1, bih, rows, D, G);
    {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const long n4 = rows * 64;
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(fill_only, dim3((unsigned)(n4 / 256)), dim3(256), 0, 0, (float4 *)G, n4);
        hipEventRecord(e0, 0);
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(fill_only, dim3((unsigned)(n4 / 256)), dim3(256), 0, 0, (float4 *)G, n4);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("plain 25 MB float4 fill: %.2f us/launch\n", ms * 5.f);
    }
    return 0;
}
