/* A plain-C host of libpworld.so: no Python, no PyTorch -- the C ABI of include/pworld.h is all a caller needs
 * (device memory comes from the HIP runtime's C API).  It runs the reference's rollout shape on the GPU:
 * B simple_spread worlds, reset, T steps with fixed pseudo-random action indices in ONE pw_rollout launch, and
 * prints every observation / reward of the last step so that tools/check_c_host.py can compare them, bit for
 * bit, with the CPU oracle.
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
    PW_OKAY(pw_rollout(h, &io, T, NULL));
    HIP_OK(hipDeviceSynchronize());

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
    for (int i = 0; i < B * N * D; ++i) { uint32_t u; memcpy(&u, &obs[i], 4); printf("o %08x\n", u); }
    for (int i = 0; i < B * N; ++i) { uint32_t u; memcpy(&u, &rew[i], 4); printf("r %08x\n", u); }
    for (int i = 0; i < B; ++i) { uint32_t u; memcpy(&u, &rs[i], 4); printf("s %08x\n", u); }
    pw_destroy(h);
    return 0;
}
