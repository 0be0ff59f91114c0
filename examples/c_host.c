/* A plain-C host of libpworld.so: no Python, no PyTorch -- the C ABI of include/pworld.h is all a caller needs
 * (device memory comes from the HIP runtime's C API).  It runs the reference's rollout shape on the GPU:
 * B simple_spread worlds, reset, T steps with fixed pseudo-random action indices in ONE pw_rollout launch, and
 * prints every observation / reward of the last step so that tools/check_c_host.py can compare them, bit for
 * bit, with the CPU oracle.  Then the gather path from C (0.1.6): the chunk condensed into a STATE-ONLY wire block
 * (pw_state_wire_begin before the launch, pw_state_wire_finalize after it), appended to a STATE ring
 * (pw_replay_store.state_rows) and sampled back with pw_replay_gather -- the rebuilt rows must equal the rollout's own
 * rows bit for bit (the pre-reset rows where a step ended an episode).
 *
 *   gcc -std=c11 -Wall -Wextra -Werror -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/c_host.c \
 *       -L multiagent_rl_amd -lpworld -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/multiagent_rl_amd -o examples/c_host
 *   ./examples/c_host > gpurun_out/c_host.txt && python tools/check_c_host.py gpurun_out/c_host.txt
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pworld.h"

#define B 64
#define N 3
#define T 30
#define SEED 2024u

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define PW_OKAY(x) do { if ((x) != PW_OK) { fprintf(stderr, "%s: %s\n", #x, pw_last_error()); return 3; } } while (0)

int main(void)
{
    pw_config cfg;
    PW_OKAY(pw_config_default(&cfg, PW_SIMPLE_SPREAD, B, N, -1, 0));
    cfg.max_episode_len = 25;
    cfg.auto_reset = 1;
    cfg.seed = SEED;
    pw_handle *h = NULL;
    PW_OKAY(pw_create(&cfg, &h));
    const int D = pw_obs_dim(h);

    void *state = NULL;
    HIP_OK(hipMalloc(&state, pw_state_bytes(h)));
    HIP_OK(hipMemset(state, 0, pw_state_bytes(h)));
    PW_OKAY(pw_bind_state(h, state));

    /* actions: a fixed LCG so that the checker can regenerate them */
    static int32_t act[T * B * N];
    uint32_t s = 12345u;
    for (size_t i = 0; i < (size_t)T * B * N; ++i) {
        s = s * 1664525u + 1013904223u;
        act[i] = (int32_t)((s >> 16) % 5u);
    }
    pw_step_io io;
    memset(&io, 0, sizeof(io));
    int32_t *d_act;
    float *d_obs, *d_fin, *d_rew, *d_rs, *d_obs0;
    uint8_t *d_done, *d_term;
    HIP_OK(hipMalloc((void **)&d_act, sizeof(act)));
    HIP_OK(hipMemcpy(d_act, act, sizeof(act), hipMemcpyHostToDevice));
    HIP_OK(hipMalloc((void **)&d_obs0, (size_t)B * N * D * 4));
    HIP_OK(hipMalloc((void **)&d_obs, (size_t)T * B * N * D * 4));
    HIP_OK(hipMalloc((void **)&d_fin, (size_t)T * B * N * D * 4));
    HIP_OK(hipMalloc((void **)&d_rew, (size_t)T * B * N * 4));
    HIP_OK(hipMalloc((void **)&d_rs, (size_t)T * B * 4));
    HIP_OK(hipMalloc((void **)&d_done, (size_t)T * B * N));
    HIP_OK(hipMalloc((void **)&d_term, (size_t)T * B));
    io.act_idx = d_act; io.obs = d_obs; io.final_obs = d_fin; io.rew = d_rew; io.rew_shared = d_rs;
    io.done = d_done; io.terminal = d_term;

    PW_OKAY(pw_reset(h, NULL, d_obs0, NULL));       /* NULL stream = the default stream */
    /* the chunk's wire block: the rollout writes rew_shared straight into it */
    pw_state_wire w;
    PW_OKAY(pw_state_wire_layout(T, B, N, cfg.num_landmarks, cfg.max_episode_len, &w));
    unsigned char *block;
    HIP_OK(hipMalloc((void **)&block, w.total_bytes));
    HIP_OK(hipMemset(block, 0, w.total_bytes));
    PW_OKAY(pw_state_wire_begin(h, &w, block, NULL));
    io.rew_shared = (float *)(block + w.rew_shared);
    PW_OKAY(pw_rollout(h, &io, T, NULL));
    PW_OKAY(pw_state_wire_finalize(h, &w, block, d_obs, d_fin, d_term, d_act, NULL));
    HIP_OK(hipMemcpy(d_rs, block + w.rew_shared, (size_t)T * B * 4, hipMemcpyDeviceToDevice));
    /* the learner rank's side: a STATE ring, one append, one sampled batch = the transitions of step 24 (it ends the first
     * episode: next_obs is the PRE-reset row) and of the last step */
    const int L = cfg.num_landmarks, cap = T * B;
    pw_replay_store st;
    memset(&st, 0, sizeof(st));
    HIP_OK(hipMalloc((void **)&st.obs, (size_t)cap * N * 16));
    HIP_OK(hipMalloc((void **)&st.next_obs, (size_t)cap * N * 16));
    HIP_OK(hipMalloc((void **)&st.lm, (size_t)cap * L * 8));
    HIP_OK(hipMalloc((void **)&st.act, (size_t)cap * N));
    HIP_OK(hipMalloc((void **)&st.rew, (size_t)cap * 4));
    HIP_OK(hipMalloc((void **)&st.done, (size_t)cap * 4));
    st.capacity = cap; st.num_agents = N; st.obs_dim = D;
    st.state_rows = 1; st.num_landmarks = L; st.scenario = PW_SIMPLE_SPREAD;
    PW_OKAY(pw_replay_add_state_wire(&st, 0, &w, block, NULL));
    static int64_t idx[2 * B];
    for (int e = 0; e < B; ++e) { idx[e] = (int64_t)24 * B + e; idx[B + e] = (int64_t)(T - 1) * B + e; }
    int64_t *d_idx;
    float *g_obs, *g_next, *g_act, *g_rew, *g_done;
    HIP_OK(hipMalloc((void **)&d_idx, sizeof(idx)));
    HIP_OK(hipMemcpy(d_idx, idx, sizeof(idx), hipMemcpyHostToDevice));
    HIP_OK(hipMalloc((void **)&g_obs, (size_t)2 * B * N * D * 4));
    HIP_OK(hipMalloc((void **)&g_next, (size_t)2 * B * N * D * 4));
    HIP_OK(hipMalloc((void **)&g_act, (size_t)2 * B * N * 5 * 4));
    HIP_OK(hipMalloc((void **)&g_rew, (size_t)2 * B * 4));
    HIP_OK(hipMalloc((void **)&g_done, (size_t)2 * B * 4));
    PW_OKAY(pw_replay_gather(&st, d_idx, 2 * B, g_obs, g_act, g_rew, g_next, g_done, NULL));
    HIP_OK(hipDeviceSynchronize());
    static float h_next[2 * B * N * 16], h_want[2 * B * N * 16], h_obs[2 * B * N * 16], h_prev[2 * B * N * 16], h_grew[2 * B], h_wrew[2 * B];
    const size_t row_bytes = (size_t)B * N * D * 4;
    HIP_OK(hipMemcpy(h_next, g_next, 2 * row_bytes, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(h_obs, g_obs, 2 * row_bytes, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(h_want, d_fin + (size_t)24 * B * N * D, row_bytes, hipMemcpyDeviceToHost));            /* pre-reset rows of step 24 */
    HIP_OK(hipMemcpy((char *)h_want + row_bytes, d_obs + (size_t)(T - 1) * B * N * D, row_bytes, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(h_prev, d_obs + (size_t)23 * B * N * D, row_bytes, hipMemcpyDeviceToHost));             /* what the policy saw at 24 */
    HIP_OK(hipMemcpy((char *)h_prev + row_bytes, d_obs + (size_t)(T - 2) * B * N * D, row_bytes, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(h_grew, g_rew, sizeof(h_grew), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(h_wrew, d_rs + (size_t)24 * B, (size_t)B * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(h_wrew + B, d_rs + (size_t)(T - 1) * B, (size_t)B * 4, hipMemcpyDeviceToHost));
    const int ring_next_ok = memcmp(h_next, h_want, 2 * row_bytes) == 0, ring_obs_ok = memcmp(h_obs, h_prev, 2 * row_bytes) == 0;
    const int ring_rew_ok = memcmp(h_grew, h_wrew, sizeof(h_grew)) == 0;

    static float obs[B * N * 16], rew[B * N], rs[B];
    static uint8_t term[T * B];
    if (D > 16) return 4;
    HIP_OK(hipMemcpy(obs, d_obs + (size_t)(T - 1) * B * N * D, (size_t)B * N * D * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(rew, d_rew + (size_t)(T - 1) * B * N, (size_t)B * N * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(rs, d_rs + (size_t)(T - 1) * B, (size_t)B * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(term, d_term, (size_t)T * B, hipMemcpyDeviceToHost));

    printf("pworld %d B %d N %d D %d T %d seed %u bytes_per_env_step %zu\n", pw_version(), B, N, D, T, SEED,
           pw_algorithmic_bytes_per_env_step(h));
    int terms = 0;
    for (int i = 0; i < T * B; ++i) terms += term[i];
    printf("terminals %d\n", terms);
    printf("state_ring wire_bytes_per_env_step %.1f next_obs_equal %d obs_equal %d rew_equal %d\n", (double)w.total_bytes / (T * B), ring_next_ok,
           ring_obs_ok, ring_rew_ok);
    for (int i = 0; i < B * N * D; ++i) { uint32_t u; memcpy(&u, &obs[i], 4); printf("o %08x\n", u); }
    for (int i = 0; i < B * N; ++i) { uint32_t u; memcpy(&u, &rew[i], 4); printf("r %08x\n", u); }
    for (int i = 0; i < B; ++i) { uint32_t u; memcpy(&u, &rs[i], 4); printf("s %08x\n", u); }
    pw_destroy(h);
    return 0;
}
