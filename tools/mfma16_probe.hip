// Diagnostic (never shipped): is v_mfma_f32_16x16x4_f32 bit-for-bit a chain of fused multiply-adds over k in ascending
// order (k = 0, 1, 2, 3 within an instruction, instructions in program order), as v_mfma_f32_32x32x2_f32 is?
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/mfma16_probe.hip -o tools/mfma16_probe.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// A [16][K] row-major, B [K][16] row-major, C0 [16][16]; D = C0 + A B accumulated 4 k at a time
__global__ void probe(const float *A, const float *B, const float *C0, float *D, int K)
{
    const int l = threadIdx.x, r = l & 15, kq = l >> 4;
    f32x4 acc;
    for (int i = 0; i < 4; ++i) acc[i] = C0[(4 * kq + i) * 16 + r];
    for (int s = 0; s < K / 4; ++s)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r * K + 4 * s + kq], B[(4 * s + kq) * 16 + r], acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[(4 * kq + i) * 16 + r] = acc[i];
}

int main()
{
    const int K = 64, trials = 2000;
    std::vector<float> a(16 * K), b(K * 16), c(256), d(256);
    float *dA, *dB, *dC, *dD;
    hipMalloc(&dA, a.size() * 4); hipMalloc(&dB, b.size() * 4); hipMalloc(&dC, 1024); hipMalloc(&dD, 1024);
    long bad_seq = 0, bad_pair = 0, n = 0;
    srand(7);
    for (int t = 0; t < trials; ++t) {
        const float scale = (t % 3 == 0) ? 1e-3f : (t % 3 == 1 ? 1.0f : 37.0f);
        for (auto &v : a) v = scale * ((rand() % 200001) - 100000) * 1e-5f;
        for (auto &v : b) v = ((rand() % 200001) - 100000) * 1e-5f;
        for (auto &v : c) v = ((rand() % 200001) - 100000) * 1e-5f;
        hipMemcpy(dA, a.data(), a.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, b.data(), b.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dC, c.data(), 1024, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, K);
        hipMemcpy(d.data(), dD, 1024, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                float s1 = c[i * 16 + j];
                for (int k = 0; k < K; ++k) s1 = fmaf(a[i * K + k], b[k * 16 + j], s1);
                // alternative: the four products of an instruction summed pairwise first
                float s2 = c[i * 16 + j];
                for (int s = 0; s < K / 4; ++s) {
                    float p = 0.f;
                    for (int q = 0; q < 4; ++q) p = fmaf(a[i * K + 4 * s + q], b[(4 * s + q) * 16 + j], p);
                    s2 += p;
                }
                uint32_t x, y, z;
                memcpy(&x, &d[i * 16 + j], 4); memcpy(&y, &s1, 4); memcpy(&z, &s2, 4);
                bad_seq += x != y; bad_pair += x != z; ++n;
            }
    }
    printf("v_mfma_f32_16x16x4_f32 vs fmaf chain in ascending k: %ld of %ld elements differ\n", bad_seq, n);
    printf("                       vs per-instruction partial sums: %ld of %ld elements differ\n", bad_pair, n);
    return bad_seq != 0;
}
