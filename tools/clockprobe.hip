// Diagnostic (not product): shader clock under a light load + cost of dependent op chains.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/clockprobe.hip -o /tmp/clockprobe && /tmp/clockprobe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int OP>
__global__ void __launch_bounds__(64) chain(float *out, unsigned long long *stamps, int iters, float a, float b)
{
    float x = a + threadIdx.x * 1e-3f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (OP == 0) x = x * b + a;          // mul + add (no fma): 2 dependent VALU
            if (OP == 1) x = a / (x + b);        // IEEE div + add
            if (OP == 2) x = sqrtf(x + b);       // IEEE sqrt + add
            if (OP == 3) x = __shfl(x, (threadIdx.x + 1) & 63, 64) + a;   // ds_bpermute + add
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 64 + threadIdx.x] = x;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int OP>
void run(const char *name, int blocks, int iters, int ops_per_unroll)
{
    float *out; unsigned long long *st;
    hipMalloc(&out, blocks * 64 * 4); hipMalloc(&st, blocks * 16);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(chain<OP>, dim3(blocks), dim3(64), 0, 0, out, st, iters, 1.0001f, 0.999f);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(2 * blocks);
    hipMemcpy(h.data(), st, blocks * 16, hipMemcpyDeviceToHost);
    double cyc = (double)h[0], real = (double)h[1];
    printf("%-22s blocks=%5d  clock=%.0f MHz  cycles per dependent step=%.1f (%d ops)\n", name, blocks,
           cyc / real * 100.0, cyc / ((double)iters * 16), ops_per_unroll);
    hipFree(out); hipFree(st);
}

int main()
{
    for (int blocks : {100, 400, 4096}) {
        run<0>("mul+add", blocks, 20000, 2);
        run<1>("div+add", blocks, 5000, 2);
        run<2>("sqrt+add", blocks, 5000, 2);
        run<3>("bpermute+add", blocks, 5000, 2);
    }
    return 0;
}
