// Diagnostic (never shipped): the product translation unit driven through its own C ABI, timing pw_rollout launches of
// T steps into a ring of output slots with HIP events -- an A/B harness for kernel variants (build it from two source
// trees, run both in one GPU session).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -I include tools/step_time.hip -o tools/step_time.bin
#include "../multiagent_rl_amd/csrc/pworld.hip"
#include <vector>

int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 2 ? atoi(argv[2]) : 6, T = argc > 3 ? atoi(argv[3]) : 1000;
    const int reps = argc > 4 ? atoi(argv[4]) : 12, slots = argc > 5 ? atoi(argv[5]) : 2;
    pw_config cfg;
    const bool tag = getenv("ST_TAG") != nullptr;
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
    std::vector<pw_step_io> io(slots);
    for (auto &s : io) {
        s = pw_step_io{};
        s.act_idx = act;
        hipMalloc((void **)&s.obs, (size_t)T * BN * D * 4); hipMalloc((void **)&s.final_obs, (size_t)T * BN * D * 4);
        hipMalloc((void **)&s.rew, (size_t)T * BN * 4); hipMalloc((void **)&s.rew_shared, (size_t)T * B * 4);
        hipMalloc((void **)&s.done, (size_t)T * BN); hipMalloc((void **)&s.terminal, (size_t)T * B);
    }
    pw_reset(h, nullptr, nullptr, nullptr);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 3; ++r) if (pw_rollout(h, &io[r % slots], T, nullptr)) { printf("rollout: %s\n", pw_last_error()); return 1; }
    hipDeviceSynchronize();
    hipEventRecord(e0, nullptr);
    for (int r = 0; r < reps; ++r) pw_rollout(h, &io[r % slots], T, nullptr);
    hipEventRecord(e1, nullptr);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s B=%d N=%d T=%d: %.4f us/step  %.3e env-steps/s  [%s]\n", argv[0], B, N, T, ms * 1e3 / ((double)reps * T),
           (double)B * reps * T / (ms * 1e-3), pw_rollout_kernel(h));
    return 0;
}
