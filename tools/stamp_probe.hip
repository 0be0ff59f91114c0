// Diagnostic build (never shipped): the product translation unit compiled with PW_STAMPS, driven
// through its own C ABI, printing the share of shader cycles per segment of one wave's step loop.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -I include tools/stamp_probe.hip -o /tmp/stamp_probe
#define PW_STAMPS 1
#include "../multiagent_rl_amd/csrc/pworld.hip"
#include <vector>

int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 2 ? atoi(argv[2]) : 6, T = argc > 3 ? atoi(argv[3]) : 25, reps = argc > 3 ? 12 : 40;
    pw_config cfg;
    const bool tag = getenv("ST_TAG") != nullptr;  // simple_tag 4+2 (pass N = 6)
    pw_config_default(&cfg, tag ? PW_SIMPLE_TAG : PW_SIMPLE_SPREAD, B, N, tag ? 2 : -1, tag ? 4 : 0);
    cfg.auto_reset = 1;
    pw_handle *h;
    if (pw_create(&cfg, &h)) { printf("create: %s\n", pw_last_error()); return 1; }
    void *state; hipMalloc(&state, pw_state_bytes(h)); hipMemset(state, 0, pw_state_bytes(h));
    pw_bind_state(h, state);
    const int D = pw_obs_dim(h);
    size_t BN = (size_t)B * N;
    int32_t *act; hipMalloc(&act, (size_t)T * BN * 4);
    std::vector<int32_t> ha((size_t)T * BN); for (auto &a : ha) a = rand() % 5;
    hipMemcpy(act, ha.data(), (size_t)T * BN * 4, hipMemcpyHostToDevice);
    pw_step_io io = {};
    io.act_idx = act;
    hipMalloc((void **)&io.obs, (size_t)T * BN * D * 4); hipMalloc((void **)&io.final_obs, (size_t)T * BN * D * 4);
    hipMalloc((void **)&io.rew, (size_t)T * BN * 4); hipMalloc((void **)&io.rew_shared, (size_t)T * B * 4);
    hipMalloc((void **)&io.done, (size_t)T * BN); hipMalloc((void **)&io.terminal, (size_t)T * B);
    pw_reset(h, nullptr, nullptr, nullptr);
    unsigned long long tot[16] = {0};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double wall_ms = 0;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0, nullptr);
        if (pw_rollout(h, &io, T, nullptr)) { printf("rollout: %s\n", pw_last_error()); return 1; }
        hipEventRecord(e1, nullptr);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (r >= 5) wall_ms += ms;
        unsigned long long s[16];
        hipMemcpyFromSymbol(s, HIP_SYMBOL(g_pw_stamps), sizeof(s));
        if (r >= 5) for (int i = 0; i < 16; ++i) tot[i] += s[i];
    }
    if (tag) {
        const char *tn[12] = {"P: action decode", "P: near-pair force loops (agents, landmarks)", "P: integrate, speed clamp, publish", "P: barrier wait", "P: near-mask pass", "-", "-", "-",
                              "O / OA: barrier wait", "O / OA: masks + rewards (+ planes)", "O / OB: (planes,) reset check, obs rows", "OB: barrier wait (trio only)"};
        printf("simple_tag B=%d N=%d (duo kernel): cycles per step, workgroup 0; this stamped build runs %.3f us per step\n", B, N, wall_ms * 1e3 / ((reps - 5) * T));
        for (int i = 0; i < 12; ++i) if (tn[i][0] != '-') printf("  %-52s %7.0f cycles\n", tn[i], tot[i] / (double)((reps - 5) * T));
        return 0;
    }
    if (N == 6 && !getenv("PWORLD_NO_QUAD") && B <= 12288) {
        const char *qn[8] = {"P: pair phase (LDS read, near test, force, table write)", "P: row add + integrate + publish", "P: barrier wait", "P: wait for the step's action indices (vmcnt)",
                             "OA: barrier wait", "OA: masks + rewards + stores", "OB: barrier wait", "OB: observation rows"};
        printf("B=%d N=%d (quad kernel): cycles per step, workgroup 0; this stamped build runs %.3f us per step\n", B, N,
               wall_ms * 1e3 / ((reps - 5) * T));
        for (int i = 0; i < 8; ++i) printf("  %-58s %7.0f cycles\n", qn[i], tot[i] / (double)((reps - 5) * T));
        return 0;
    }
    const bool duo = !getenv("PWORLD_NO_DUO");
    const char *names_duo[8] = {"P: action decode + prefetch", "P: near-pair force loop", "P: integrate + publish slot",
                                "P: barrier wait (O behind?)", "P: near-mask pass", "-", "-", "-"};
    const char *names_stream[8] = {"action decode + prefetch issue", "near-pair force loop", "integrate + LDS exchange",
                            "partner pass (d2, masks, min)", "reward: sqrt + 12 shuffles", "stores rew/done/.. + reset check",
                            "obs row build + stores", "vmcnt(K) hint"};
    const char **names = duo ? names_duo : names_stream;
    double sum = 0; for (int i = 0; i < 8; ++i) sum += tot[i];
    printf("B=%d N=%d (%s kernel): cycles per step (wave 0, incl. ~40/stamp overhead): %.0f\n", B, N, duo ? "duo" : "stream", sum / ((reps - 5) * T));
    for (int i = 0; i < 8; ++i) printf("  %-36s %7.0f cycles  %5.1f%%\n", names[i], tot[i] / (double)((reps - 5) * T), 100.0 * tot[i] / sum);
    if (duo) {
        const char *on[4] = {"O: barrier wait (P behind)", "O: partner pass + reward + LDS exchanges",
                             "O: stores + reset check + obs rows", "O: vmcnt(K) hint"};
        double so = 0; for (int i = 8; i < 12; ++i) so += tot[i];
        printf("wave O: cycles per step %.0f\n", so / ((reps - 5) * T));
        for (int i = 0; i < 4; ++i) printf("  %-44s %7.0f cycles  %5.1f%%\n", on[i], tot[8 + i] / (double)((reps - 5) * T), 100.0 * tot[8 + i] / so);
    }
    return 0;
}
