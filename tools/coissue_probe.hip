// Micro-benchmark (diagnostic, never shipped): can a wave issuing f32 MFMAs and a wave issuing vector FMAs share a
// SIMD without slowing each other?  One workgroup of 8 waves per CU (2 per SIMD: waves w and w + 4 share SIMD w & 3).
// Waves 0-3 run `mf` dependent-chain v_mfma_f32_32x32x2_f32 (2 independent chains), waves 4-7 run `vf` v_pk_fma_f32
// or v_fma_f32 (8 independent chains).  mode: 1 = matrix waves only, 2 = vector waves only, 3 = both.
// hipcc --offload-arch=gfx950 -O3 tools/coissue_probe.hip -o tools/coissue_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int packed>
__global__ void __launch_bounds__(512) probe(float *out, unsigned long long *cyc, int iters, int mode)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned long long t0 = 0, t1 = 0;
    float res = 0.f;
    if (wave < 4) {
        if (mode & 1) {
            f32x16 a, b;
            for (int q = 0; q < 16; ++q) { a[q] = 0.f; b[q] = 0.f; }
            const float x = 1.0f + lane * 1e-3f, y = 0.5f;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    a = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a, 0, 0, 0);
                    b = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, b, 0, 0, 0);
                }
            }
            res = a[0] + b[5];
            asm volatile("s_nop 0" ::"v"(res));
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        }
    } else if (mode & 2) {
        f32x2 acc[8];
        for (int q = 0; q < 8; ++q) acc[q] = f32x2{0.f, 0.f};
        const f32x2 w = {1.0001f, 0.9999f}, h = {lane * 1e-6f, 1e-6f};
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if (packed) {
                        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[q]) : "v"(w), "v"(h));
                    } else {
                        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[q].x) : "v"(w.x), "v"(h.x));
                        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[q].y) : "v"(w.y), "v"(h.y));
                    }
                }
            }
        }
        for (int q = 0; q < 8; ++q) res += acc[q].x + acc[q].y;
        asm volatile("s_nop 0" ::"v"(res));
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    }
    out[blockIdx.x * 512 + threadIdx.x] = res;
    if (blockIdx.x == 0 && lane == 0) cyc[wave] = t1 - t0;
}

int main()
{
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 64);
    const int iters = 2000;
    for (int packed = 1; packed >= 0; --packed)
        for (int mode = 1; mode <= 3; ++mode) {
            hipMemset(cyc, 0, 64);
            if (packed) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(512), 0, 0, out, cyc, iters, mode);
            else hipLaunchKernelGGL(probe<0>, dim3(256), dim3(512), 0, 0, out, cyc, iters, mode);
            hipDeviceSynchronize();
            unsigned long long c[8];
            hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
            printf("%s mode %d (%s): matrix wave 0: %6.1f cycles per MFMA | vector wave 4: %6.2f cycles per %s\n",
                   packed ? "v_pk_fma_f32" : "v_fma_f32   ", mode, mode == 1 ? "matrix only" : mode == 2 ? "vector only" : "both",
                   c[0] / (double)(iters * 16), c[4] / (double)(iters * 32 * (packed ? 1 : 2)), packed ? "v_pk_fma_f32" : "v_fma_f32");
        }
    return 0;
}
