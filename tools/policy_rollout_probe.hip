// Diagnostic build (never shipped): pworld.hip compiled with PW_STAMPS; shader cycles of wave 0 / workgroup 0 per
// phase of pw_policy_rollout3_kernel (forms 1 / 2 were retired in round 4).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -I include tools/policy_rollout_probe.hip -o tools/policy_rollout_probe.bin
#define PW_STAMPS 1
#include "../multiagent_rl_amd/csrc/pworld.hip"
#include "../multiagent_rl_amd/csrc/pworld_policy.hip"
#include <vector>

int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 3 ? atoi(argv[3]) : 6, T = 100;   // argv: B, debug mask, N
    pw_config cfg;
    pw_config_default(&cfg, PW_SIMPLE_SPREAD, B, N, -1, 0);
    cfg.auto_reset = 1; cfg.max_episode_len = 25;
    pw_handle *h;
    if (pw_create(&cfg, &h)) { printf("create: %s\n", pw_last_error()); return 1; }
    void *state; hipMalloc(&state, pw_state_bytes(h)); hipMemset(state, 0, pw_state_bytes(h));
    pw_bind_state(h, state);
    const int D = pw_obs_dim(h);
    float *frag, *w1, *wih, *b1, *bih, *whf, *whr, *w2, *b2; int32_t *act;
    hipMalloc(&frag, pw_actor_front_pack_floats(D) * 4);
    hipMalloc(&w1, 64 * D * 4); hipMalloc(&wih, 256 * 64 * 4); hipMalloc(&b1, 256); hipMalloc(&bih, 1024);
    hipMalloc(&whf, 128 * 32 * 4); hipMalloc(&whr, 128 * 32 * 4); hipMalloc(&w2, 320 * 4); hipMalloc(&b2, 32);
    size_t BN = (size_t)B * N;
    hipMalloc(&act, T * BN * 4);
    auto fill = [&](float *p, size_t n) { std::vector<float> t(n); for (auto &v : t) v = (rand() % 2001 - 1000) * 2e-4f; hipMemcpy(p, t.data(), n * 4, hipMemcpyHostToDevice); };
    fill(w1, 64 * D); fill(wih, 256 * 64); fill(b1, 64); fill(bih, 256); fill(whf, 4096); fill(whr, 4096); fill(w2, 320); fill(b2, 5);
    pw_actor_front_pack(w1, wih, D, frag, nullptr);
    pw_step_io io = {};
    hipMalloc((void **)&io.obs, T * BN * D * 4); hipMalloc((void **)&io.final_obs, T * BN * D * 4);
    hipMalloc((void **)&io.rew, T * BN * 4); hipMalloc((void **)&io.rew_shared, (size_t)T * B * 4);
    hipMalloc((void **)&io.done, T * BN); hipMalloc((void **)&io.terminal, (size_t)T * B);
    pw_reset(h, nullptr, nullptr, nullptr);
    { int dbg = argc > 2 ? atoi(argv[2]) : 0; hipMemcpyToSymbol(HIP_SYMBOL(g_pw_debug), &dbg, sizeof(dbg)); printf("debug mask %d\n", dbg); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) if (pw_policy_rollout(h, frag, b1, bih, whf, whr, w2, b2, 1, 1, 0, nullptr, &io, act, T, nullptr, nullptr)) { printf("%s\n", pw_last_error()); return 1; }
    hipEventRecord(e0, 0);
    for (int i = 0; i < 5; ++i) pw_policy_rollout(h, frag, b1, bih, whf, whr, w2, b2, 1, 1, 100 * i, nullptr, &io, act, T, nullptr, nullptr);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("B=%d: %.2f us per step (%s)\n", B, ms * 1000.f / (5 * T), pw_last_error());
    unsigned long long s[16];
    hipMemcpyFromSymbol(s, HIP_SYMBOL(g_pw_stamps), sizeof(s));
    {
        const char *mn[8] = {"dense1 block", "  barrier", "  BiLSTM timestep loop (N barriers)", "  head (one barrier inside)", "  env step, part 1 (env waves) / noise", "  barrier", "  env step, tail (env waves)", "  -"};
        for (int wv = 0; wv < 2; ++wv) {
            double sm = 0; for (int i = 0; i < 8; ++i) sm += s[8 * wv + i];
            printf(" wave %d\n", 7 * wv);
            for (int i = 0; i < 8; ++i) printf("  %-44s %9.0f cycles/step %5.1f%%\n", mn[i], s[8 * wv + i] / (double)T, 100.0 * s[8 * wv + i] / sm);
            printf("  total %.0f cycles/step\n", sm / T);
        }
    }
    return 0;
}
