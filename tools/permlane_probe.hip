// v_permlane16_swap / v_permlane32_swap (gfx950) as lane ^ 16 / lane ^ 32 exchanges (csrc/pw_kernels_actor16.hpp lane_xor16 / lane_xor32) against __shfl_xor.
// hipcc --offload-arch=gfx950 -O3 tools/permlane_probe.hip -o tools/permlane_probe.bin   ->  "mismatches 0"
#include <hip/hip_runtime.h>
__device__ __forceinline__ float xor16(float v) {
    // v_permlane16_swap: rows 1 <-> 0 and 3 <-> 2 between the two operands
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const bool odd = (threadIdx.x >> 4) & 1;
    return __uint_as_float(odd ? r[0] : r[1]);
}
__device__ __forceinline__ float xor32(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const bool hi = (threadIdx.x >> 5) & 1;
    return __uint_as_float(hi ? r[0] : r[1]);
}
__global__ void k(float *o, const float *i) {
    float v = i[threadIdx.x];
    o[threadIdx.x] = xor16(v);
    o[64 + threadIdx.x] = xor32(v);
    o[128 + threadIdx.x] = __shfl_xor(v, 16, 64);
    o[192 + threadIdx.x] = __shfl_xor(v, 32, 64);
}
int main() {
    float *i, *o; hipMalloc(&i, 256); hipMalloc(&o, 1024);
    float h[64]; for (int x = 0; x < 64; ++x) h[x] = x;
    hipMemcpy(i, h, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(o, i);
    float r[256]; hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int x = 0; x < 64; ++x) { if (r[x] != r[128 + x]) bad++; if (r[64 + x] != r[192 + x]) bad++; }
    printf("mismatches %d  (xor16 lane0 %g lane16 %g; xor32 lane0 %g lane40 %g)\n", bad, r[0], r[16], r[64], r[64 + 40]);
    return bad != 0;
}
