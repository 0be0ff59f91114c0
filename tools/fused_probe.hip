// Diagnostic build (never shipped): the product translation unit compiled with PW_STAMPS; prints the shader
// cycles wave 0 of workgroup 0 spends in each phase of pw_actor_fused_kernel and the launch time.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -I include tools/fused_probe.hip -o tools/fused_probe.bin
#define PW_STAMPS 1
#include "../multiagent_rl_amd/csrc/pworld.hip"
#include "../multiagent_rl_amd/csrc/pworld_policy.hip"
#include <vector>

int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 4096, N = 6, D = 16;
    float *X, *frag, *w1, *wih, *b1, *bih, *whf, *whr, *w2, *b2; int32_t *act;
    hipMalloc(&X, (size_t)B * N * D * 4); hipMalloc(&frag, pw_actor_front_pack_floats(D) * 4);
    hipMalloc(&w1, 64 * D * 4); hipMalloc(&wih, 256 * 64 * 4); hipMalloc(&b1, 256); hipMalloc(&bih, 1024);
    hipMalloc(&whf, 128 * 32 * 4); hipMalloc(&whr, 128 * 32 * 4); hipMalloc(&w2, 320 * 4); hipMalloc(&b2, 32);
    hipMalloc(&act, (size_t)B * N * 4);
    std::vector<float> h((size_t)B * N * D); for (auto &v : h) v = (rand() % 2001 - 1000) * 1e-3f;
    hipMemcpy(X, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    auto fill = [&](float *p, size_t n) { std::vector<float> t(n); for (auto &v : t) v = (rand() % 2001 - 1000) * 2e-4f; hipMemcpy(p, t.data(), n * 4, hipMemcpyHostToDevice); };
    fill(w1, 64 * D); fill(wih, 256 * 64); fill(b1, 64); fill(bih, 256); fill(whf, 4096); fill(whr, 4096); fill(w2, 320); fill(b2, 5);
    pw_actor_front_pack(w1, wih, D, frag, nullptr);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 10; ++i) pw_actor_fused(X, frag, b1, bih, whf, whr, w2, b2, 5, 0, B, N, D, 1, 1, i, nullptr, nullptr, nullptr, act, nullptr);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 100; ++i) pw_actor_fused(X, frag, b1, bih, whf, whr, w2, b2, 5, 0, B, N, D, 1, 1, i, nullptr, nullptr, nullptr, act, nullptr);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("B=%d: %.2f us per launch (%s)\n", B, ms * 10.f, pw_last_error());
    unsigned long long s[16];
    hipMemcpyFromSymbol(s, HIP_SYMBOL(g_pw_stamps), sizeof(s));
    const char *names[8] = {"initial fill + barrier", "stage 1 (X loads + MFMA + relu)", "weights fill + barrier (2 dirs)", "stage 2 MFMA -> Gs (2 dirs)",
                            "barrier after stage 2", "W_hh -> regs + recurrence (2 dirs)", "barrier after LSTM", "head + sampling"};
    double sum = 0; for (int i = 0; i < 8; ++i) sum += s[i];
    for (int i = 0; i < 8; ++i) printf("  %-40s %8llu cycles %5.1f%%\n", names[i], s[i], 100.0 * s[i] / sum);
    printf("  total %.0f cycles\n", sum);
    return 0;
}
