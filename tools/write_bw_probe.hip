// Diagnostic: what a pure store stream reaches on this chip (the rollout kernels' output is ~97 % writes).
// hipcc --offload-arch=gfx950 -O3 tools/write_bw_probe.hip -o tools/write_bw_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ void __launch_bounds__(256) fill(f4 *p, size_t n, float v)
{
    const f4 x = {v, v, v, v};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (NT) __builtin_nontemporal_store(x, p + i); else p[i] = x;
    }
}
// each wave writes one contiguous 1 KiB per store instruction, blocks of `chunk` KiB per wave visit (the obs block pattern)
template <bool NT>
__global__ void __launch_bounds__(64) fill_blocks(f4 *p, size_t n_blocks, int per_block, float v)
{
    const f4 x = {v, v, v, v};
    for (size_t b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        f4 *q = p + b * (size_t)per_block * 64;
        for (int k = 0; k < per_block; ++k) { if (NT) __builtin_nontemporal_store(x, q + k * 64 + threadIdx.x); else q[k * 64 + threadIdx.x] = x; }
    }
}
template <bool NT>
__global__ void __launch_bounds__(256) copy(const f4 *s, f4 *d, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const f4 x = s[i];
        if (NT) __builtin_nontemporal_store(x, d + i); else d[i] = x;
    }
}
int main()
{
    const size_t bytes = 8ull << 30, n = bytes / 16;
    f4 *p; if (hipMalloc(&p, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char *name, auto launch, double moved) {
        for (int i = 0; i < 2; ++i) launch();
        hipDeviceSynchronize();
        hipEventRecord(e0, nullptr);
        for (int i = 0; i < 5; ++i) launch();
        hipEventRecord(e1, nullptr); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-56s %7.0f GB/s\n", name, moved * 5 / (ms * 1e-3) / 1e9);
    };
    for (int grid : {2048, 8192, 32768}) {
        char nm[96];
        snprintf(nm, sizeof nm, "fill float4, plain stores, grid %d x 256", grid);
        time(nm, [&] { hipLaunchKernelGGL(fill<false>, dim3(grid), dim3(256), 0, nullptr, p, n, 1.0f); }, (double)bytes);
        snprintf(nm, sizeof nm, "fill float4, nt stores,    grid %d x 256", grid);
        time(nm, [&] { hipLaunchKernelGGL(fill<true>, dim3(grid), dim3(256), 0, nullptr, p, n, 1.0f); }, (double)bytes);
    }
    for (int per : {3, 19, 75}) {   // 3 KiB (C2's 48 x 16 floats), 19 KiB (N = 48 row block), 75 KiB
        char nm[96];
        snprintf(nm, sizeof nm, "1-wave workgroups, %d KiB contiguous per visit, nt", per);
        time(nm, [&] { hipLaunchKernelGGL(fill_blocks<true>, dim3(8192), dim3(64), 0, nullptr, p, n / ((size_t)per * 64), per, 2.0f); }, (double)(n / ((size_t)per * 64)) * per * 1024);
    }
    time("copy float4 (4 GiB -> 4 GiB), nt stores: read + write bytes", [&] { hipLaunchKernelGGL(copy<true>, dim3(8192), dim3(256), 0, nullptr, p, p + n / 2, n / 2); }, (double)bytes);
    return 0;
}
