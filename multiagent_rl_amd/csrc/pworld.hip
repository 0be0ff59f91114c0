// libpworld.so -- batched particle world for MI355X (gfx950, wave64).  This translation unit: the environment (every
// pw_step / pw_rollout kernel and the dispatcher), the replay ring and the wire blocks; pworld_policy.hip: the actor and
// the policy-in-the-loop rollouts.  Two units only so that they compile in parallel (and an env-kernel experiment
// rebuilds one of them).
//
// One fused kernel advances all B envs: _set_action -> apply_action_force ->
// apply_environment_force (pairwise get_collision_force) -> integrate_state ->
// per-agent observation / reward / done, optionally for T consecutive steps with
// the state held in registers + LDS (pw_rollout).  Entry points and the upstream
// functions they replace are declared in include/pworld.h.
//
// Mapping: one lane per (env, agent); a 64-lane wave holds EPW = floor(64 / N)
// whole envs, so every per-env exchange is a wave-local LDS broadcast or a
// ds_bpermute shuffle and no env ever straddles a wave.  A workgroup is ONE wave
// (64 threads): at B = 4096, N = 6 that is 410 independent workgroups over the
// 256 CUs, and cross-lane hand-offs need only an LDS wait (wave_lds_sync), never a barrier.
// State planes are SoA over the flattened [B x N] index g = env * N + agent, so a
// wave's loads/stores are one contiguous run of EPW*N floats per plane.
//
// Arithmetic is IEEE float32 in the upstream operation order, no FMA contraction,
// with the deterministic softplus/exp of include/pworld_math.h, so a CPU
// restatement reproduces every output bit (tests/ compare against oracle/).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <type_traits>

#include "pworld.h"
#include "pworld_math.h"

#include "pw_common.hpp"
#include "pw_kernels_spread.hpp"
#include "pw_kernels_spread_quad.hpp"
#include "pw_kernels_tag.hpp"
#include "pw_kernels_reference.hpp"
#include "pw_kernels_generic.hpp"
#include "pw_kernels_replay.hpp"

#include "pw_handle.hpp"

namespace {

int obs_dim_of(const pw_config &c)
{
    const int N = c.num_agents, L = c.num_landmarks;
    if (c.scenario == PW_SIMPLE_SPREAD) return c.obs_mode == PW_OBS_FULL ? 4 + 2 * L + 4 * (N - 1) : 4 + 2 * L;
    if (c.scenario == PW_SIMPLE_REFERENCE) return 2 + 2 * L + 3 + PW_DIM_C * (N - 1);
    if (c.scenario == PW_SIMPLE_SPEAKER_LISTENER) return 2 + 2 * L + 3;
    const int G = N - c.num_adversaries;
    return 4 + 2 * L + 2 * (N - 1) + 2 * (c.num_adversaries > 0 ? G : G - 1);
}

bool is_comm_scenario(int scenario) { return scenario == PW_SIMPLE_REFERENCE || scenario == PW_SIMPLE_SPEAKER_LISTENER; }
int dim_c_of(int scenario) { return scenario == PW_SIMPLE_REFERENCE ? PW_DIM_C : scenario == PW_SIMPLE_SPEAKER_LISTENER ? PW_SL_DIM_C : 0; }

template <typename F>
int dispatch(const pw_handle *h, F &&f)
{
    if (h->cfg.scenario == PW_SIMPLE_TAG) return f(std::integral_constant<int, PW_SIMPLE_TAG>(), std::integral_constant<int, PW_OBS_LOCAL>());
    if (h->cfg.obs_mode == PW_OBS_FULL) return f(std::integral_constant<int, PW_SIMPLE_SPREAD>(), std::integral_constant<int, PW_OBS_FULL>());
    return f(std::integral_constant<int, PW_SIMPLE_SPREAD>(), std::integral_constant<int, PW_OBS_LOCAL>());
}

// coll_thr2 = min{y : sqrtf(y) >= dist_min}: then sqrtf(d2) < dist_min <=> d2 < coll_thr2 (0 if unusable)
float exact_coll_thr2(float dmin)
{
    if (!(dmin > 0.0f) || !std::isfinite(dmin)) return 0.0f;
    float y = dmin * dmin;
    while (sqrtf(y) >= dmin) y = std::nextafterf(y, 0.0f);
    while (!(sqrtf(y) >= dmin)) y = std::nextafterf(y, INFINITY);
    return y;
}

// beyond sqrt(result) the softplus argument is <= -87.5 < -87 (pw_exp's exact-zero cut); 0 if unusable
float exact_near_thr2(float dmin, float contact_margin)
{
    const double near_r = (double)dmin + 88.5 * (double)contact_margin;
    float n2 = (float)(near_r * near_r * (1.0 + 1e-6));
    n2 = std::nextafterf(n2, INFINITY);
    const float x_at = -(sqrtf(n2) - dmin) / contact_margin;
    if (!(x_at <= -87.5f) || !std::isfinite(n2) || !(n2 > 0.0f)) return 0.0f;
    return n2;
}

// div_chain1 (pw_common.hpp) divides by the contact margin k with ONE Newton correction, which is the correctly rounded
// quotient if (a) the reciprocal the kernels use, y = fma(fma(-k, rcp, 1), rcp, rcp) with ANY rcp within 1 ulp of 1 / k,
// is the correctly rounded 1 / k, (b) k's significand is not all ones, (c) k is inside div_chain's operand range.  (a) is
// decided exactly: k and y are float32, so k * y - 1 is exact in double, and y = RN(1 / k) iff |1 - k y| <= k ulp(y) / 2
// (strictly below: no float is a midpoint of 1 / k for k not a power of two; a power of two has y exact).  The hardware's
// v_rcp_f32 is only specified to 1 ulp, so all three candidates y0 in {rcp - 1ulp, rcp, rcp + 1ulp} must refine to that y.
bool margin_one_correction(float k)
{
    if (!(k >= 9.094947017729282e-13f && k <= 1099511627776.0f)) return false;
    uint32_t kb;
    std::memcpy(&kb, &k, 4);
    if ((kb & 0x7FFFFFu) == 0x7FFFFFu) return false;
    const float y_rn = (float)(1.0 / (double)k);  // a candidate; verified below, not trusted
    for (int d = -1; d <= 1; ++d) {
        float y0 = y_rn;
        uint32_t yb;
        std::memcpy(&yb, &y0, 4);
        yb += (uint32_t)d;
        std::memcpy(&y0, &yb, 4);
        const float y = std::fmaf(std::fmaf(-k, y0, 1.0f), y0, y0);
        const double resid = std::fabs(std::fma((double)k, (double)y, -1.0));       // |k y - 1|, exact
        const double ulp_y = (double)std::nextafterf(y, INFINITY) - (double)y;
        if (!(resid < (double)k * ulp_y * 0.5) && !(resid == 0.0)) return false;
    }
    return true;
}

// Decide whether pw_tag_stream_kernel applies (both roles homogeneous) and derive its tables.
void setup_tag_path(pw_handle *h)
{
    const pw_config &c = h->cfg;
    const KParams &kp = h->kp;
    h->tag_fast = false;
    if (h->disp.force_generic) return;
    if (c.scenario != PW_SIMPLE_TAG || !c.landmark_collide || kp.N < 1) return;
    const int A = kp.A, N = kp.N;
    const int rep[2] = {A > 0 ? 0 : 0, A < N ? A : 0};  // representative agent of each class
    for (int i = 0; i < N; ++i) {
        const int r = rep[i >= A ? 1 : 0];
        if (kp.agent_size[i] != kp.agent_size[r] || kp.agent_sens[i] != kp.agent_sens[r] ||
            kp.agent_fscale[i] != kp.agent_fscale[r] || kp.agent_max_speed[i] != kp.agent_max_speed[r])
            return;
    }
    TagParams &t = h->tp;
    std::memset(&t, 0, sizeof(t));
    t.B = kp.B; t.N = N; t.L = kp.L; t.A = A; t.D = kp.D; t.epw = kp.epw;
    t.max_episode_len = kp.max_episode_len; t.auto_reset = kp.auto_reset;
    t.seed = kp.seed; t.env_id_base = kp.env_id_base;
    t.dt = kp.dt; t.damp = kp.damp; t.contact_force = kp.contact_force; t.contact_margin = kp.contact_margin;
    t.mass = kp.mass;
    float size[2];
    for (int cl = 0; cl < 2; ++cl) {
        size[cl] = kp.agent_size[rep[cl]];
        t.sens[cl] = kp.agent_sens[rep[cl]];
        t.fscale[cl] = kp.agent_fscale[rep[cl]];
        t.max_speed[cl] = kp.agent_max_speed[rep[cl]];
    }
    for (int ci = 0; ci < 2; ++ci) {
        for (int cj = 0; cj < 2; ++cj) {
            const volatile float dmin = size[ci] + size[cj];
            t.dist_min[ci][cj] = dmin;
            t.coll_thr2[ci][cj] = exact_coll_thr2(dmin);
            t.near_thr2[ci][cj] = exact_near_thr2(dmin, kp.contact_margin);
            if (t.coll_thr2[ci][cj] == 0.0f || t.near_thr2[ci][cj] == 0.0f) return;
        }
        const volatile float dl = size[ci] + kp.landmark_size;
        t.dist_min_lm[ci] = dl;
        t.near_thr2_lm[ci] = exact_near_thr2(dl, kp.contact_margin);
        if (t.near_thr2_lm[ci] == 0.0f) return;
    }
    h->tag_fast = true;
}

// Decide whether pw_spread_fast_kernel applies and derive its exact thresholds on the host.
void setup_fast_path(pw_handle *h)
{
    const pw_config &c = h->cfg;
    const KParams &kp = h->kp;
    h->fast = false;
    if (h->disp.force_generic) return;
    if (c.scenario != PW_SIMPLE_SPREAD || c.obs_mode != PW_OBS_LOCAL || c.landmark_collide) return;
    if (kp.L > kp.N) return;
    for (int i = 0; i < kp.N; ++i) {
        if (kp.agent_size[i] != kp.agent_size[0] || kp.agent_sens[i] != kp.agent_sens[0] ||
            kp.agent_fscale[i] != kp.agent_fscale[0] || kp.agent_max_speed[i] >= 0.0f)
            return;
    }
    FastConsts &fc = h->fc;
    fc.size = kp.agent_size[0];
    fc.sens = kp.agent_sens[0];
    fc.fscale = kp.agent_fscale[0];
    const volatile float dmin = fc.size + fc.size;  // float add, as the kernels do
    fc.dist_min = dmin;
    if (!(dmin > 0.0f) || !std::isfinite(dmin)) return;
    fc.coll_thr2 = exact_coll_thr2(dmin);
    fc.near_thr2 = exact_near_thr2(dmin, kp.contact_margin);
    if (fc.coll_thr2 == 0.0f || fc.near_thr2 == 0.0f) return;
    fc.k1 = margin_one_correction(kp.contact_margin) ? 1 : 0;
    h->fast = true;
}

#ifndef PW_TRIO_ROWS_LO
#define PW_TRIO_ROWS_LO 1      // workgroups: the row-wise three-wave form of N = 9 / 12 (measured below)
#define PW_TRIO_ROWS_HI 1280
#endif
#ifndef PW_N3_TRIO_HI
#define PW_N3_TRIO_HI 600   // workgroups (C5 N = 3 at B = 4096: 512 of 8 envs): measured B = 1024 ... 4096: -6.5 % step time
#endif

void dispatch_defaults(pw_dispatch *d)
{
    std::memset(d, 0, sizeof(*d));
    d->struct_size = sizeof(pw_dispatch);
    d->duo = d->quad = d->obs_block = d->trio = d->p_prio = -1;
}

// The PWORLD_* variables of the creating process, read ONCE per handle (pw_create): A/B runs of unmodified host programs.
void dispatch_from_environment(pw_dispatch *d, int *actor_bf16x3)
{
    dispatch_defaults(d);
    *actor_bf16x3 = 0;  // not a selection (it changes results): no environment variable reaches it, only pw_set_actor_precision
    if (std::getenv("PWORLD_FORCE_GENERIC")) d->force_generic = 1;
    if (std::getenv("PWORLD_NO_STREAM")) d->no_stream = 1;
    if (std::getenv("PWORLD_FORCE_DUO")) d->duo = 1;
    if (std::getenv("PWORLD_NO_DUO")) d->duo = 0;
    if (std::getenv("PWORLD_FORCE_QUAD")) d->quad = 1;
    if (std::getenv("PWORLD_NO_QUAD")) d->quad = 0;
    if (const char *e = std::getenv("PWORLD_OBS_BLOCK")) d->obs_block = std::atoi(e) != 0;
    if (const char *e = std::getenv("PWORLD_SPREAD_TRIO")) d->trio = std::atoi(e) != 0;
    if (const char *e = std::getenv("PWORLD_TAG_TRIO")) d->trio = std::atoi(e) != 0;
    if (const char *e = std::getenv("PWORLD_P_PRIO")) d->p_prio = std::atoi(e) & 0xFFFF;
    if (const char *e = std::getenv("PWORLD_EPW")) d->envs_per_wave = std::atoi(e) >= 1 ? std::atoi(e) : 0;
    if (std::getenv("PWORLD_POLICY_V3J")) d->policy_form = 4;
    if (std::getenv("PWORLD_POLICY_V3")) d->policy_form = 3;
}

// Everything the selection decides ahead of a launch: envs per wave and which specialised paths apply.
void apply_dispatch(pw_handle *h)
{
    KParams &kp = h->kp;
    kp.epw = kWave / kp.N;
    // Envs per wave.  Dense packing (64 / N) is not the fastest for the small N of the BASELINE configs: measured on
    // MI355X (tools/sweep.py), 48-lane waves beat 60/63-lane ones at N = 6 (8 vs 10 envs per wave: +3.5 %
    // at B = 4096, +7 % at 8192, +2-4 % up to 262144; simple_tag 4+2: +10-12 %) and N = 3 (16 vs 21: +14 % at
    // B = 65536) -- fewer near-pair iterations per wave-step (max over the wave's lanes) and workgroup counts that are
    // multiples of the 256 CUs -- while N = 9, 12, 24 are faster densely packed.
    if (kp.N == 6) kp.epw = 8;
    else if (kp.N == 3) kp.epw = 16;
    // Small batches are latency bound (one wave per SIMD, the chip not even full): spread the envs over about
    // 512 workgroups (2 waves each in the duo kernels = the 1024 SIMDs) instead of packing them densely.
    {
        const int spread = (kp.B + 511) / 512;
        if (spread < kp.epw) kp.epw = spread < 1 ? 1 : spread;
    }
    if (h->disp.envs_per_wave >= 1)  // experiments / tests: e.g. 64 = the dense packing of large batches
        kp.epw = h->disp.envs_per_wave < kWave / kp.N ? h->disp.envs_per_wave : kWave / kp.N;
    setup_fast_path(h);
    setup_tag_path(h);
}

// every env-step launch records which kernel it was (pw_rollout_kernel: bench.py and the profile tools name the
// dominant kernel from the dispatcher's own decision, not from a table kept beside it)
#define PW_LAUNCH(h, kern, ...)                \
    do {                                       \
        (h)->last_kernel = #kern;              \
        hipLaunchKernelGGL(kern, __VA_ARGS__); \
    } while (0)

int launch_rollout(pw_handle *h, const pw_step_io *io, int T, void *stream)
{
    if (int rc = check_ready(h)) return rc;
    if (!io) return fail(PW_EINVAL, "null pw_step_io");
    if (T < 1) return fail(PW_EINVAL, "num_steps must be >= 1");
    if ((io->act_idx == nullptr) == (io->act_vec == nullptr))
        return fail(PW_EINVAL, "exactly one of act_idx / act_vec must be given");
    if (io->act_comm && h->cfg.scenario != PW_SIMPLE_REFERENCE)
        return fail(PW_EINVAL, "act_comm only applies to simple_reference (simple_speaker_listener's speaker takes its symbol "
                               "from act_idx; the other scenarios' agents are silent)");
    if (io->obs && (reinterpret_cast<uintptr_t>(io->obs) & 15))
        return fail(PW_EINVAL, "obs must be 16-byte aligned");
    if (io->final_obs && (reinterpret_cast<uintptr_t>(io->final_obs) & 15))
        return fail(PW_EINVAL, "final_obs must be 16-byte aligned");
    const KParams &kp = h->kp;
    if (h->cfg.scenario == PW_SIMPLE_REFERENCE) {
        if (io->act_idx && !io->act_comm) return fail(PW_EINVAL, "simple_reference needs act_comm next to act_idx");
        PW_LAUNCH(h, (pw_reference_rollout_kernel<kDimC, false>), dim3((kp.B + 31) / 32), dim3(kWave), 0,
                           static_cast<hipStream_t>(stream), ref_params(h), *io, io->act_comm, T);
        PW_HIP_CHECK(hipGetLastError());
        return PW_OK;
    }
    if (h->cfg.scenario == PW_SIMPLE_SPEAKER_LISTENER) {
        PW_LAUNCH(h, (pw_reference_rollout_kernel<kDimCSL, true>), dim3((kp.B + 31) / 32), dim3(kWave), 0,
                           static_cast<hipStream_t>(stream), ref_params(h), *io, nullptr, T);
        PW_HIP_CHECK(hipGetLastError());
        return PW_OK;
    }
    const dim3 grid((kp.B + kp.epw - 1) / kp.epw), block(kWave);
    const size_t shmem = smem_bytes(kp);
    // max_episode_len = 1 with auto-reset: EVERY step ends an episode.  The multi-wave forms (duo / trio / quad) hand {pos, vel} from
    // the physics wave to the output waves through a 3- or 4-slot LDS ring and use TWO slots in a step that resets (pre-reset state,
    // post-reset state); with a reset in every step the physics wave, one step ahead, overwrites the pre-reset slot the output
    // wave is still composing final_obs from (found in round 5 by rebuilding simple_tag rows from the state: the other agents' entries of
    // the pre-reset row were post-reset values).  The one-wave forms keep the state in registers: they serve this degenerate setting.
    const bool every_step_resets = kp.auto_reset && kp.max_episode_len == 1;
    if (h->tag_fast && io->act_idx && io->obs && io->rew && io->rew_shared && io->done && io->terminal &&
        (size_t)kp.B * kp.N * kp.D * sizeof(float) < (1ull << 31) && !h->disp.no_stream) {
        const pw_dispatch &dp = h->disp;
        TagParams A = h->tp;
        A.pos_x = kp.pos_x; A.pos_y = kp.pos_y; A.vel_x = kp.vel_x; A.vel_y = kp.vel_y;
        A.lm_x = kp.lm_x; A.lm_y = kp.lm_y; A.ep_step = kp.ep_step; A.ep_count = kp.ep_count;
        A.act = io->act_idx; A.obs = io->obs; A.final_obs = io->final_obs; A.rew = io->rew;
        A.rew_shared = io->rew_shared; A.done = io->done; A.terminal = io->terminal; A.coll = io->coll;
        const bool wc = io->coll != nullptr;  // the optional collision-mask output: its own instantiations
        hipStream_t st = static_cast<hipStream_t>(stream);
        const size_t shm = 2 * kWave * sizeof(float2) + 3 * kWave * sizeof(float) + (size_t)kp.epw * kp.L * sizeof(float2);
        const bool um = kp.mass == 1.0f;
        // two waves per env group, as for simple_spread.  With the block-wise observation stores (below) the duo form
        // leads on every grid measured: B = 8192: 1.69 vs 2.41 us per step, B = 65536: 10.3 vs 19.0 (profiles/r2_tag_block.txt)
        const bool duo = !every_step_resets && (dp.duo < 0 ? grid.x <= 8192 : dp.duo != 0);
        size_t shm2 = ((3 * kWave * sizeof(float4) + 3 * kWave * sizeof(float) + 2 * (size_t)kp.epw * kp.L * sizeof(float2) + 15) &
                       ~(size_t)15) + kActRingBytes;  // ring, masks, rewards, both waves' landmarks | wave P's action ring
        // Block-wise observation stores of the duo kernel (rows staged in LDS): short rows only (LDS), chunks of 4 floats
        // when every wave's block starts and ends on 16 bytes, else of 2 (D is even).  pw_dispatch.obs_block overrides.
        A.obs_block = 0;
        // the physics wave first where it shares a SIMD with output waves: -2..-4 % step time on grids up to 2048 workgroups
        // (C3, N = 3 / 12, B = 16384), +2..4 % on the larger ones (profiles/r2_priority.txt)
        A.p_prio = dp.p_prio >= 0 ? dp.p_prio : (grid.x <= 2048 ? 1 : 0);
        if (duo && kp.D <= 32) {
            bool on = grid.x > 700;   // as for simple_spread (profiles/r2_obs_block_threshold.txt)
            if (dp.obs_block >= 0) on = dp.obs_block != 0;
            if (on) {
                const bool v4 = ((size_t)kp.B * kp.N * kp.D) % 4 == 0 && ((size_t)kp.epw * kp.N * kp.D) % 4 == 0 &&
                                (reinterpret_cast<uintptr_t>(io->obs) & 15) == 0 && kp.B % kp.epw == 0;
                A.obs_block = v4 ? 4 : 2;
                shm2 += (size_t)kWave * kp.D * sizeof(float);
            }
        }
        // Three waves per env group (the output wave split into a rewards wave and an observation wave) where the single
        // output wave is the step's critical path: measured crossover in profiles/r2_tag_block.txt.  pw_dispatch.trio overrides.
        bool trio = duo && grid.x >= 512 && grid.x <= 1280;
        if (dp.trio >= 0) trio = duo && dp.trio != 0;
        const dim3 block2((trio ? 3 : 2) * kWave);
#define PW_TAG_LAUNCH(n, a, l, c)                                                                              \
    do {                                                                                                       \
        if (duo && trio) {                                                                                     \
            if (um) PW_LAUNCH(h, (pw_tag_duo_kernel<n, a, l, true, c, true>), grid, block2, shm2, st, A, T);   \
            else PW_LAUNCH(h, (pw_tag_duo_kernel<n, a, l, false, c, true>), grid, block2, shm2, st, A, T);     \
        } else if (duo) {                                                                                      \
            if (um) PW_LAUNCH(h, (pw_tag_duo_kernel<n, a, l, true, c>), grid, block2, shm2, st, A, T);   \
            else PW_LAUNCH(h, (pw_tag_duo_kernel<n, a, l, false, c>), grid, block2, shm2, st, A, T);     \
        } else if (um) PW_LAUNCH(h, (pw_tag_stream_kernel<n, a, l, true, c>), grid, block, shm, st, A, T); \
        else PW_LAUNCH(h, (pw_tag_stream_kernel<n, a, l, false, c>), grid, block, shm, st, A, T);        \
    } while (0)
        if (wc) {
            if (kp.N == 6 && kp.A == 4 && kp.L == 2 && um) PW_TAG_LAUNCH(6, 4, 2, true);
            else PW_TAG_LAUNCH(0, -1, 0, true);
        } else if (kp.N == 6 && kp.A == 4 && kp.L == 2) PW_TAG_LAUNCH(6, 4, 2, false);   // BASELINE configs[2]
        else if (kp.N == 4 && kp.A == 3 && kp.L == 2) PW_TAG_LAUNCH(4, 3, 2, false);      // canonical upstream roster
        else PW_TAG_LAUNCH(0, -1, 0, false);
#undef PW_TAG_LAUNCH
        PW_HIP_CHECK(hipGetLastError());
        return PW_OK;
    }
    if (h->fast && io->act_idx && io->obs && io->rew && io->rew_shared && io->done && io->terminal &&
        (size_t)kp.B * kp.N * kp.D * sizeof(float) < (1ull << 31) && !h->disp.no_stream) {
        const pw_dispatch &dp = h->disp;
        StreamParams A;
        A.B = kp.B; A.N = kp.N; A.L = kp.L; A.epw = kp.epw;
        // issue priority per wave, 2 bits each: the physics wave first where it shares a SIMD with output waves: -2..-4 % step
        // time on grids up to 2048 workgroups (N = 3 / 12, B = 16384), +2..4 % on the larger ones (profiles/r2_priority.txt)
        A.p_prio = dp.p_prio >= 0 ? dp.p_prio : (grid.x <= 2048 ? 3 : 0);
        A.max_episode_len = kp.max_episode_len; A.auto_reset = kp.auto_reset;
        A.seed = kp.seed; A.env_id_base = kp.env_id_base;
        A.dt = kp.dt; A.damp = kp.damp; A.contact_force = kp.contact_force; A.contact_margin = kp.contact_margin;
        A.mass = kp.mass;
        A.dist_min = h->fc.dist_min; A.coll_thr2 = h->fc.coll_thr2; A.near_thr2 = h->fc.near_thr2;
        A.sens = h->fc.sens; A.fscale = h->fc.fscale;
        A.pos_x = kp.pos_x; A.pos_y = kp.pos_y; A.vel_x = kp.vel_x; A.vel_y = kp.vel_y;
        A.lm_x = kp.lm_x; A.lm_y = kp.lm_y; A.ep_step = kp.ep_step; A.ep_count = kp.ep_count;
        A.act = io->act_idx; A.obs = io->obs; A.final_obs = io->final_obs; A.rew = io->rew;
        A.rew_shared = io->rew_shared; A.done = io->done; A.terminal = io->terminal; A.coll = io->coll;
        hipStream_t st = static_cast<hipStream_t>(stream);
        const size_t shm = (size_t)(kWave + kp.epw * kp.L) * sizeof(float2) + kWave * sizeof(float4);
        const bool um = kp.mass == 1.0f;
        const bool wc = io->coll != nullptr;  // the optional collision-mask output: instantiated for N = 3, 6 and runtime N
        // Observation rows stored row-per-lane, or as one contiguous block per wave (stream_write_obs_block): the
        // block form is what the write path wants once the run is store-bound (B = 65536, N = 6: +58 %; N = 24 / 48 at
        // B = 4096: +59 % / +47 %), but it lengthens the output wave's step, which latency-bound small grids pay for
        // (C2: -5 %; N = 3, B = 16384: -38 %).  Measured crossover (profiles/r2_obs_block_threshold.txt): between 512
        // and 820 workgroups for N = 6 ... 24, above 1024 for N = 3 (8-byte chunks).  pw_dispatch.obs_block overrides.
        bool blk = kp.N == 3 ? grid.x > 3000 : grid.x > 700;
        if (dp.obs_block >= 0) blk = dp.obs_block != 0;
        const int key = kp.N != kp.L ? 0 : (wc && !(um && (kp.N == 3 || kp.N == 6))) ? 0 : kp.N;
        // N = L = 6 on small grids (BASELINE configs[1]): four cooperating waves per 8 envs, pair-parallel physics
        // (pw_kernels_spread_quad.hpp).  More total instructions than the duo form, shorter dependent chains: it pays
        // while the chip is latency bound.  pw_dispatch.quad overrides (tests, experiments).  With the collision-mask output
        // requested it is the COLL instantiation: the same kernel plus one mask store in its reward wave.
        {
            const unsigned qgrid = (unsigned)((kp.B + 7) / 8);
            const bool quad_ok = kp.N == 6 && kp.L == 6 && um;
            bool quad = quad_ok && qgrid <= 1536 && dp.duo != 0;  // B <= 12288: measured crossover (profiles/r2_sweeps_final.txt)
            if (dp.quad >= 0) quad = quad_ok && dp.quad != 0;
            if (every_step_resets) quad = false;
            if (quad) {
                const size_t qshm = 4 * kWave * sizeof(float4) + 2 * 4 * 6 * 6 * sizeof(float2) + 2 * kWave * sizeof(float) +
                                    8 * 6 * sizeof(float2) + 2 * kQuadActRingBytes + 2 * 8 * sizeof(float2);
                const bool k1 = h->fc.k1 != 0;  // the canonical margin 1e-3 qualifies
                if (wc && k1) PW_LAUNCH(h, (pw_spread_quad_kernel<true, true, true>), dim3(qgrid), dim3(4 * kWave), qshm, st, A, T);
                else if (wc) PW_LAUNCH(h, (pw_spread_quad_kernel<true, true>), dim3(qgrid), dim3(4 * kWave), qshm, st, A, T);
                else if (k1) PW_LAUNCH(h, (pw_spread_quad_kernel<true, false, true>), dim3(qgrid), dim3(4 * kWave), qshm, st, A, T);
                else PW_LAUNCH(h, (pw_spread_quad_kernel<true>), dim3(qgrid), dim3(4 * kWave), qshm, st, A, T);
                PW_HIP_CHECK(hipGetLastError());
                return PW_OK;
            }
        }
        // two cooperating waves per env group pay off while the chip is latency bound (few workgroups
        // per CU); once every SIMD holds several waves the single-wave kernel issues fewer instructions
        const bool duo = !every_step_resets && (dp.duo < 0 ? grid.x <= 8192 : dp.duo != 0);
        if (duo) {
            const size_t shm2 = 3 * kWave * sizeof(float4) + (size_t)kp.epw * kp.L * sizeof(float2) +
                                2 * kWave * sizeof(float) + 16 + kWave * sizeof(float4) + 16 * sizeof(float2) + kActRingBytes;   // .. utab + zero (padded), action ring
            // three waves per env group (the output wave split in two) where the single output wave is the step's critical
            // path: block-store instantiations on mid-size grids (measured: profiles/r2_trio.txt).  pw_dispatch.trio overrides.
            // (only the compile-time instantiations below have the three-wave form: a runtime-N launch stays two waves wide)
            const bool trio_inst = key == 6 || key == 9 || key == 12 || key == 24 || key == 48;
            bool trio = um && blk && !wc && key >= 12 && trio_inst && grid.x >= 512 && grid.x <= 1280;
            if (dp.trio >= 0) trio = um && blk && !wc && trio_inst && dp.trio != 0;
            // N = L = 3 with row-wise stores on grids that do not fill the chip (C5's N = 3 point: 256 workgroups): the single
            // output wave (rewards 730 + rows 560 busy cycles) is longer than the physics wave (1210), and there are idle SIMDs
            // for a third wave: its own three-wave instantiation (profiles/r3_n3_trio.txt).  pw_dispatch.trio overrides.
            const bool trio3 = key == 3 && um && !blk && !wc && (dp.trio >= 0 ? dp.trio != 0 : grid.x <= PW_N3_TRIO_HI);
            if (trio3) {
                if (dp.p_prio < 0) A.p_prio = 3;
                PW_LAUNCH(h, (pw_spread_duo_kernel<3, 3, true, false, false, true>), grid, dim3(3 * kWave), shm2, st, A, T);
                PW_HIP_CHECK(hipGetLastError());
                return PW_OK;
            }
            // N = L = 9 / 12 with ROW-wise stores (grids below the block-store crossover; N = 9 -- the reference's middle scalability
            // setting, main_scalability_1.py:30 -- never stores blocks: its 88-byte rows leave a wave's block off the 16-byte grid): the single
            // output wave (N = 9 at B = 4096, stamps: rewards 1490 + rows 2550 busy cycles against the physics wave's 2640) splits in two.
            // pw_dispatch.trio overrides.  (profiles/r5_trio_rows.txt)
            const bool trio_rows = (key == 9 || key == 12) && um && !blk && !wc &&
                                   (dp.trio >= 0 ? dp.trio != 0 : (grid.x >= PW_TRIO_ROWS_LO && grid.x <= PW_TRIO_ROWS_HI));
            if (trio_rows) {
                if (dp.p_prio < 0) A.p_prio = 3 | (3 << 4);
                if (key == 9) PW_LAUNCH(h, (pw_spread_duo_kernel<9, 9, true, false, false, true>), grid, dim3(3 * kWave), shm2, st, A, T);
                else PW_LAUNCH(h, (pw_spread_duo_kernel<12, 12, true, false, false, true>), grid, dim3(3 * kWave), shm2, st, A, T);
                PW_HIP_CHECK(hipGetLastError());
                return PW_OK;
            }
            if (trio && dp.p_prio < 0) A.p_prio = 3 | (3 << 4);  // three waves: physics and observation wave first (N = 12: -2.5 %)
            const dim3 block2((trio ? 3 : 2) * kWave);
            if (wc) {
                if (key == 3) PW_LAUNCH(h, (pw_spread_duo_kernel<3, 3, true, true>), grid, block2, shm2, st, A, T);
                else if (key == 6) PW_LAUNCH(h, (pw_spread_duo_kernel<6, 6, true, true>), grid, block2, shm2, st, A, T);
                else if (um) PW_LAUNCH(h, (pw_spread_duo_kernel<0, 0, true, true>), grid, block2, shm2, st, A, T);
                else PW_LAUNCH(h, (pw_spread_duo_kernel<0, 0, false, true>), grid, block2, shm2, st, A, T);
                PW_HIP_CHECK(hipGetLastError());
                return PW_OK;
            }
            switch (key) {
#define PW_DUO_CASE(n)                                                                                              \
    case n:                                                                                                         \
        if (um && blk && trio) PW_LAUNCH(h, (pw_spread_duo_kernel<n, n, true, false, true, true>), grid, block2, shm2, st, A, T); \
        else if (um && blk) PW_LAUNCH(h, (pw_spread_duo_kernel<n, n, true, false, true>), grid, block2, shm2, st, A, T); \
        else if (um) PW_LAUNCH(h, (pw_spread_duo_kernel<n, n, true>), grid, block2, shm2, st, A, T);          \
        else PW_LAUNCH(h, (pw_spread_duo_kernel<n, n, false>), grid, block2, shm2, st, A, T);                 \
        break;
                PW_DUO_CASE(3) PW_DUO_CASE(6) PW_DUO_CASE(9) PW_DUO_CASE(12) PW_DUO_CASE(24) PW_DUO_CASE(48)
#undef PW_DUO_CASE
            default:
                if (um) PW_LAUNCH(h, (pw_spread_duo_kernel<0, 0, true>), grid, block2, shm2, st, A, T);
                else PW_LAUNCH(h, (pw_spread_duo_kernel<0, 0, false>), grid, block2, shm2, st, A, T);
            }
            PW_HIP_CHECK(hipGetLastError());
            return PW_OK;
        }
        if (wc) {
            if (key == 3) PW_LAUNCH(h, (pw_spread_stream_kernel<3, 3, true, true>), grid, block, shm, st, A, T);
            else if (key == 6) PW_LAUNCH(h, (pw_spread_stream_kernel<6, 6, true, true>), grid, block, shm, st, A, T);
            else if (um) PW_LAUNCH(h, (pw_spread_stream_kernel<0, 0, true, true>), grid, block, shm, st, A, T);
            else PW_LAUNCH(h, (pw_spread_stream_kernel<0, 0, false, true>), grid, block, shm, st, A, T);
            PW_HIP_CHECK(hipGetLastError());
            return PW_OK;
        }
        switch (key) {
#define PW_STREAM_CASE(n)                                                                                           \
    case n:                                                                                                         \
        if (um && blk) PW_LAUNCH(h, (pw_spread_stream_kernel<n, n, true, false, true>), grid, block, shm, st, A, T); \
        else if (um) PW_LAUNCH(h, (pw_spread_stream_kernel<n, n, true>), grid, block, shm, st, A, T);         \
        else PW_LAUNCH(h, (pw_spread_stream_kernel<n, n, false>), grid, block, shm, st, A, T);                \
        break;
            PW_STREAM_CASE(3) PW_STREAM_CASE(6) PW_STREAM_CASE(9) PW_STREAM_CASE(12) PW_STREAM_CASE(24) PW_STREAM_CASE(48)
#undef PW_STREAM_CASE
        default:
            if (um) PW_LAUNCH(h, (pw_spread_stream_kernel<0, 0, true>), grid, block, shm, st, A, T);
            else PW_LAUNCH(h, (pw_spread_stream_kernel<0, 0, false>), grid, block, shm, st, A, T);
        }
        PW_HIP_CHECK(hipGetLastError());
        return PW_OK;
    }
    if (h->fast) {
        hipStream_t st = static_cast<hipStream_t>(stream);
        switch (kp.N) {
#define PW_FAST_CASE(n)                                                                                      \
    case n:                                                                                                  \
        PW_LAUNCH(h, (pw_spread_fast_kernel<n>), grid, block, shmem, st, kp, *io, T, h->fc);           \
        break;
            PW_FAST_CASE(3) PW_FAST_CASE(6) PW_FAST_CASE(9) PW_FAST_CASE(12)
#undef PW_FAST_CASE
        default:
            PW_LAUNCH(h, (pw_spread_fast_kernel<0>), grid, block, shmem, st, kp, *io, T, h->fc);
        }
        PW_HIP_CHECK(hipGetLastError());
        return PW_OK;
    }
    return dispatch(h, [&](auto scen, auto obs) {
        PW_LAUNCH(h, (pw_rollout_kernel<decltype(scen)::value, decltype(obs)::value>), grid, block, shmem,
                           static_cast<hipStream_t>(stream), kp, *io, T);
        PW_HIP_CHECK(hipGetLastError());
        return (int)PW_OK;
    });
}

int launch_aux(pw_handle *h, int mode, const uint8_t *env_mask, float *obs, float *rew, uint64_t *coll, void *stream)
{
    if (int rc = check_ready(h)) return rc;
    if (obs && (reinterpret_cast<uintptr_t>(obs) & 15)) return fail(PW_EINVAL, "obs must be 16-byte aligned");
    const KParams &kp = h->kp;
    if (is_comm_scenario(h->cfg.scenario)) {
        if (coll) return fail(PW_EINVAL, "the communication scenarios have no collisions");
        if (h->cfg.scenario == PW_SIMPLE_REFERENCE)
            hipLaunchKernelGGL((pw_reference_aux_kernel<kDimC, false>), dim3((kp.B + 31) / 32), dim3(kWave), 0,
                               static_cast<hipStream_t>(stream), ref_params(h), mode, env_mask, obs, rew);
        else
            hipLaunchKernelGGL((pw_reference_aux_kernel<kDimCSL, true>), dim3((kp.B + 31) / 32), dim3(kWave), 0,
                               static_cast<hipStream_t>(stream), ref_params(h), mode, env_mask, obs, rew);
        PW_HIP_CHECK(hipGetLastError());
        return PW_OK;
    }
    const dim3 grid((kp.B + kp.epw - 1) / kp.epw), block(kWave);
    const size_t shmem = smem_bytes(kp);
    return dispatch(h, [&](auto scen, auto om) {
        hipLaunchKernelGGL((pw_aux_kernel<decltype(scen)::value, decltype(om)::value>), grid, block, shmem,
                           static_cast<hipStream_t>(stream), kp, mode, env_mask, obs, rew, coll);
        PW_HIP_CHECK(hipGetLastError());
        return (int)PW_OK;
    });
}

}  // namespace

namespace {
thread_local std::string g_last_error;
}

extern "C" {

void pw_internal_set_error(const char *msg) { g_last_error = msg ? msg : ""; }

int pw_version(void) { return PW_VERSION; }

const char *pw_last_error(void) { return g_last_error.c_str(); }

int pw_config_default(pw_config *cfg, int scenario, int num_envs, int num_agents, int num_landmarks,
                      int num_adversaries)
{
    if (!cfg) return fail(PW_EINVAL, "null cfg");
    if (num_agents < 1 || num_agents > PW_MAX_AGENTS) return fail(PW_EINVAL, "num_agents out of range [1, 64]");
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->struct_size = sizeof(pw_config);
    cfg->scenario = scenario;
    cfg->num_envs = num_envs;
    cfg->num_agents = num_agents;
    cfg->obs_mode = PW_OBS_LOCAL;
    cfg->max_episode_len = 25;       // rls/arglist.py:5
    cfg->auto_reset = 0;
    cfg->force_discrete_action = 1;  // experiments/scenarios.py:191
    cfg->seed = 12345678;            // main.py:41
    cfg->dt = 0.1f;
    cfg->damping = 0.25f;
    cfg->contact_force = 100.0f;
    cfg->contact_margin = 1e-3f;
    cfg->default_sensitivity = 5.0f;
    cfg->mass = 1.0f;
    if (scenario == PW_SIMPLE_SPREAD) {
        cfg->num_landmarks = num_landmarks < 0 ? num_agents : num_landmarks;
        cfg->num_adversaries = 0;
        cfg->landmark_collide = 0;
        cfg->landmark_size = 0.05f;
        for (int i = 0; i < num_agents; ++i) {
            cfg->agent_size[i] = 0.15f;
            cfg->agent_accel[i] = -1.0f;
            cfg->agent_max_speed[i] = -1.0f;
        }
    } else if (scenario == PW_SIMPLE_TAG) {
        if (num_adversaries < 0 || num_adversaries > num_agents) return fail(PW_EINVAL, "num_adversaries out of range");
        cfg->num_landmarks = num_landmarks < 0 ? 2 : num_landmarks;
        cfg->num_adversaries = num_adversaries;
        cfg->landmark_collide = 1;
        cfg->landmark_size = 0.2f;
        for (int i = 0; i < num_agents; ++i) {
            const bool adv = i < num_adversaries;
            cfg->agent_size[i] = adv ? 0.075f : 0.05f;
            cfg->agent_accel[i] = adv ? 3.0f : 4.0f;
            cfg->agent_max_speed[i] = adv ? 1.0f : 1.3f;
        }
    } else if (scenario == PW_SIMPLE_REFERENCE) {
        cfg->num_agents = 2;  // upstream simple_reference.make_world: two agents, three landmarks
        cfg->num_landmarks = num_landmarks < 0 ? 3 : num_landmarks;
        cfg->num_adversaries = 0;
        cfg->landmark_collide = 0;
        cfg->landmark_size = 0.05f;
        for (int i = 0; i < 2; ++i) {
            cfg->agent_size[i] = 0.05f;
            cfg->agent_accel[i] = -1.0f;
            cfg->agent_max_speed[i] = -1.0f;
        }
    } else if (scenario == PW_SIMPLE_SPEAKER_LISTENER) {
        cfg->num_agents = 2;  // upstream simple_speaker_listener.make_world: speaker + listener, three landmarks
        cfg->num_landmarks = num_landmarks < 0 ? 3 : num_landmarks;
        cfg->num_adversaries = 0;
        cfg->landmark_collide = 0;
        cfg->landmark_size = 0.04f;
        for (int i = 0; i < 2; ++i) {
            cfg->agent_size[i] = 0.075f;
            cfg->agent_accel[i] = -1.0f;
            cfg->agent_max_speed[i] = -1.0f;
        }
    } else {
        return fail(PW_EINVAL, "unknown scenario");
    }
    return PW_OK;
}

int pw_create(const pw_config *cfg, pw_handle **out)
{
    if (!cfg || !out) return fail(PW_EINVAL, "null argument");
    if (cfg->struct_size != sizeof(pw_config)) return fail(PW_EINVAL, "pw_config.struct_size mismatch (ABI)");
    if (cfg->scenario != PW_SIMPLE_SPREAD && cfg->scenario != PW_SIMPLE_TAG && !is_comm_scenario(cfg->scenario))
        return fail(PW_EINVAL, "unknown scenario");
    if (is_comm_scenario(cfg->scenario) && (cfg->num_agents != 2 || cfg->num_landmarks < 1 || cfg->num_landmarks > 3))
        return fail(PW_EINVAL, "simple_reference / simple_speaker_listener are two agents and 1..3 landmarks");
    if (cfg->num_envs < 1) return fail(PW_EINVAL, "num_envs must be >= 1");
    if (cfg->num_agents < 1 || cfg->num_agents > PW_MAX_AGENTS) return fail(PW_EINVAL, "num_agents out of range [1, 64]");
    if (cfg->num_landmarks < 0 || cfg->num_landmarks > PW_MAX_LANDMARKS) return fail(PW_EINVAL, "num_landmarks out of range [0, 64]");
    if (cfg->scenario == PW_SIMPLE_TAG && (cfg->num_adversaries < 0 || cfg->num_adversaries > cfg->num_agents))
        return fail(PW_EINVAL, "num_adversaries out of range");
    if (cfg->scenario == PW_SIMPLE_TAG && cfg->num_adversaries == cfg->num_agents && cfg->num_agents > 0 &&
        obs_dim_of(*cfg) < 4)
        return fail(PW_EINVAL, "bad simple_tag roster");
    if (cfg->obs_mode != PW_OBS_LOCAL && cfg->obs_mode != PW_OBS_FULL) return fail(PW_EINVAL, "unknown obs_mode");
    if (!(cfg->contact_margin > 0.0f) || !(cfg->mass > 0.0f)) return fail(PW_EINVAL, "contact_margin and mass must be > 0");
    pw_handle *h = new (std::nothrow) pw_handle;
    if (!h) return fail(PW_ENOMEM, "out of host memory");
    h->cfg = *cfg;
    h->bound = false;
    h->last_kernel = nullptr;
    KParams &kp = h->kp;
    std::memset(&kp, 0, sizeof(kp));
    kp.B = cfg->num_envs; kp.N = cfg->num_agents; kp.L = cfg->num_landmarks;
    kp.A = cfg->scenario == PW_SIMPLE_TAG ? cfg->num_adversaries : 0;
    kp.D = obs_dim_of(*cfg);
    kp.max_episode_len = cfg->max_episode_len;
    kp.auto_reset = cfg->auto_reset;
    kp.force_discrete = cfg->force_discrete_action;
    kp.landmark_collide = cfg->landmark_collide;
    kp.seed = cfg->seed; kp.env_id_base = cfg->env_id_base;
    kp.dt = cfg->dt; kp.damp = 1.0f - cfg->damping;
    kp.contact_force = cfg->contact_force; kp.contact_margin = cfg->contact_margin;
    kp.mass = cfg->mass; kp.landmark_size = cfg->landmark_size;
    for (int i = 0; i < kp.N; ++i) {
        kp.agent_size[i] = cfg->agent_size[i];
        kp.agent_sens[i] = cfg->agent_accel[i] >= 0.0f ? cfg->agent_accel[i] : cfg->default_sensitivity;
        kp.agent_fscale[i] = cfg->action_force_uses_accel
                                 ? (cfg->agent_accel[i] >= 0.0f ? cfg->mass * cfg->agent_accel[i] : cfg->mass)
                                 : 1.0f;
        kp.agent_max_speed[i] = cfg->agent_max_speed[i];
    }
    dispatch_from_environment(&h->disp, &h->actor_bf16x3);
    apply_dispatch(h);
    const size_t BN = (size_t)kp.B * kp.N, BL = (size_t)kp.B * kp.L;
    pw_state_layout &lo = h->layout;
    size_t off = 0;
    lo.pos_x = off; off = align_up(off + BN * 4, 256);
    lo.pos_y = off; off = align_up(off + BN * 4, 256);
    lo.vel_x = off; off = align_up(off + BN * 4, 256);
    lo.vel_y = off; off = align_up(off + BN * 4, 256);
    lo.lm_x = off; off = align_up(off + BL * 4, 256);
    lo.lm_y = off; off = align_up(off + BL * 4, 256);
    lo.ep_step = off; off = align_up(off + (size_t)kp.B * 4, 256);
    lo.ep_count = off; off = align_up(off + (size_t)kp.B * 4, 256);
    lo.comm = lo.goal = 0;
    if (is_comm_scenario(cfg->scenario)) {
        lo.comm = off; off = align_up(off + BN * dim_c_of(cfg->scenario) * 4, 256);
        lo.goal = off; off = align_up(off + BN * 4, 256);
    }
    lo.total_bytes = off;
    *out = h;
    return PW_OK;
}

void pw_destroy(pw_handle *h) { delete h; }

int pw_obs_dim(const pw_handle *h) { return h ? h->kp.D : fail(PW_EINVAL, "null handle"); }

int pw_get_config(const pw_handle *h, pw_config *out)
{
    if (!h || !out) return fail(PW_EINVAL, "null argument");
    *out = h->cfg;
    return PW_OK;
}

int pw_margin_one_correction(float contact_margin) { return margin_one_correction(contact_margin) ? 1 : 0; }

int pw_dispatch_default(pw_dispatch *d)
{
    if (!d) return fail(PW_EINVAL, "null argument");
    dispatch_defaults(d);
    return PW_OK;
}

int pw_set_dispatch(pw_handle *h, const pw_dispatch *d)
{
    if (!h || !d) return fail(PW_EINVAL, "null argument");
    if (d->struct_size != sizeof(pw_dispatch)) return fail(PW_EINVAL, "pw_dispatch.struct_size mismatch (ABI)");
    if (d->duo < -1 || d->duo > 1 || d->quad < -1 || d->quad > 1 || d->obs_block < -1 || d->obs_block > 1 || d->trio < -1 ||
        d->trio > 1 || d->p_prio < -1 || d->p_prio > 0xFFFF || d->envs_per_wave < 0 || d->policy_form < 0 || d->policy_form > 4)
        return fail(PW_EINVAL, "pw_dispatch field out of range");
    h->disp = *d;
    h->disp.force_generic = d->force_generic != 0;
    h->disp.no_stream = d->no_stream != 0;
    apply_dispatch(h);
    return PW_OK;
}

int pw_set_actor_precision(pw_handle *h, int32_t mode)
{
    if (!h) return fail(PW_EINVAL, "null handle");
    if (mode != PW_ACTOR_F32 && mode != PW_ACTOR_BF16X3) return fail(PW_EINVAL, "mode: PW_ACTOR_F32 or PW_ACTOR_BF16X3");
    h->actor_bf16x3 = mode == PW_ACTOR_BF16X3;
    return PW_OK;
}

int pw_get_actor_precision(const pw_handle *h) { return h && h->actor_bf16x3 ? PW_ACTOR_BF16X3 : PW_ACTOR_F32; }

int pw_get_dispatch(const pw_handle *h, pw_dispatch *out)
{
    if (!h || !out) return fail(PW_EINVAL, "null argument");
    *out = h->disp;
    return PW_OK;
}

int pw_set_force_discrete_action(pw_handle *h, int on)
{
    if (!h) return fail(PW_EINVAL, "null handle");
    h->cfg.force_discrete_action = on ? 1 : 0;
    h->kp.force_discrete = on ? 1 : 0;
    return PW_OK;
}

int pw_get_state_layout(const pw_handle *h, pw_state_layout *out)
{
    if (!h || !out) return fail(PW_EINVAL, "null argument");
    *out = h->layout;
    return PW_OK;
}

size_t pw_state_bytes(const pw_handle *h) { return h ? h->layout.total_bytes : 0; }

int pw_bind_state(pw_handle *h, void *block)
{
    if (!h || !block) return fail(PW_EINVAL, "null argument");
    if (reinterpret_cast<uintptr_t>(block) & 255) return fail(PW_EINVAL, "state block must be 256-byte aligned");
    unsigned char *b = static_cast<unsigned char *>(block);
    KParams &kp = h->kp;
    kp.pos_x = reinterpret_cast<float *>(b + h->layout.pos_x);
    kp.pos_y = reinterpret_cast<float *>(b + h->layout.pos_y);
    kp.vel_x = reinterpret_cast<float *>(b + h->layout.vel_x);
    kp.vel_y = reinterpret_cast<float *>(b + h->layout.vel_y);
    kp.lm_x = reinterpret_cast<float *>(b + h->layout.lm_x);
    kp.lm_y = reinterpret_cast<float *>(b + h->layout.lm_y);
    kp.ep_step = reinterpret_cast<int32_t *>(b + h->layout.ep_step);
    kp.ep_count = reinterpret_cast<uint32_t *>(b + h->layout.ep_count);
    h->comm = reinterpret_cast<float *>(b + h->layout.comm);
    h->goal = reinterpret_cast<int32_t *>(b + h->layout.goal);
    h->bound = true;
    return PW_OK;
}

int pw_set_state(pw_handle *h, const float *pos, const float *vel, const float *lm, const int32_t *ep_step,
                 const uint32_t *ep_count, void *stream)
{
    if (int rc = check_ready(h)) return rc;
    const KParams &kp = h->kp;
    size_t n = (size_t)kp.B * (kp.N > kp.L ? kp.N : kp.L);
    if (n < (size_t)kp.B) n = kp.B;
    hipLaunchKernelGGL(pw_scatter_state_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), kp, pos, vel, lm, ep_step, ep_count);
    PW_HIP_CHECK(hipGetLastError());
    return PW_OK;
}

int pw_get_state(pw_handle *h, float *pos, float *vel, float *lm, int32_t *ep_step, uint32_t *ep_count, void *stream)
{
    if (int rc = check_ready(h)) return rc;
    const KParams &kp = h->kp;
    size_t n = (size_t)kp.B * (kp.N > kp.L ? kp.N : kp.L);
    if (n < (size_t)kp.B) n = kp.B;
    hipLaunchKernelGGL(pw_gather_state_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), kp, pos, vel, lm, ep_step, ep_count);
    PW_HIP_CHECK(hipGetLastError());
    return PW_OK;
}

int pw_set_comm_state(pw_handle *h, const float *comm, const int32_t *goal, void *stream)
{
    if (int rc = check_ready(h)) return rc;
    if (!is_comm_scenario(h->cfg.scenario)) return fail(PW_EINVAL, "this scenario has no communication state");
    const int dc = dim_c_of(h->cfg.scenario);
    const size_t n = (size_t)h->kp.B * 2 * dc;
    hipLaunchKernelGGL(pw_reference_state_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), ref_params(h), 1, dc, const_cast<float *>(comm),
                       const_cast<int32_t *>(goal));
    PW_HIP_CHECK(hipGetLastError());
    return PW_OK;
}

int pw_get_comm_state(pw_handle *h, float *comm, int32_t *goal, void *stream)
{
    if (int rc = check_ready(h)) return rc;
    if (!is_comm_scenario(h->cfg.scenario)) return fail(PW_EINVAL, "this scenario has no communication state");
    const int dc = dim_c_of(h->cfg.scenario);
    const size_t n = (size_t)h->kp.B * 2 * dc;
    hipLaunchKernelGGL(pw_reference_state_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), ref_params(h), 0, dc, comm, goal);
    PW_HIP_CHECK(hipGetLastError());
    return PW_OK;
}

int pw_reset(pw_handle *h, const uint8_t *env_mask, float *obs, void *stream)
{
    return launch_aux(h, 1 | (obs ? 2 : 0), env_mask, obs, nullptr, nullptr, stream);
}

int pw_observe(pw_handle *h, float *obs, void *stream)
{
    if (!obs) return fail(PW_EINVAL, "null obs");
    return launch_aux(h, 2, nullptr, obs, nullptr, nullptr, stream);
}

int pw_reward(pw_handle *h, float *rew, uint64_t *coll, void *stream)
{
    if (!rew && !coll) return fail(PW_EINVAL, "nothing to write");
    return launch_aux(h, 4, nullptr, nullptr, rew, coll, stream);
}

int pw_step(pw_handle *h, const pw_step_io *io, void *stream) { return launch_rollout(h, io, 1, stream); }

const char *pw_rollout_kernel(const pw_handle *h) { return h && h->last_kernel ? h->last_kernel : ""; }

int pw_rollout(pw_handle *h, const pw_step_io *io, int num_steps, void *stream)
{
    return launch_rollout(h, io, num_steps, stream);
}

size_t pw_algorithmic_bytes_per_env_step(const pw_handle *h)
{
    if (!h) return 0;
    const size_t N = h->kp.N, L = h->kp.L, D = h->kp.D;
    // read: state 16N + landmarks 8L + action 4N; write: state 16N + obs 4ND + reward 4N + done N
    return 16 * N + 8 * L + 4 * N + 16 * N + 4 * N * D + 4 * N + N;
}

int pw_counter_add(int64_t *counter, int64_t delta, int64_t modulo, void *stream)
{
    if (!counter) return fail(PW_EINVAL, "null counter");
    hipLaunchKernelGGL(pw_counter_add_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), counter, delta, modulo);
    PW_HIP_CHECK(hipGetLastError());
    return PW_OK;
}

int pw_replay_add(const pw_replay_store *st, int64_t start, const int64_t *start_dev, int32_t B, const float *obs,
                  const int32_t *act_idx, const float *rew_shared, const float *next_obs, const float *final_obs,
                  const uint8_t *terminal, const float *done, void *stream)
{
    if (!st || !obs || !act_idx || !rew_shared || !next_obs) return fail(PW_EINVAL, "null argument");
    if (st->state_rows) return fail(PW_EINVAL, "pw_replay_add: a STATE ring is filled by pw_replay_add_state_wire only (rows do not determine the landmarks)");
    if (st->capacity < 1 || B < 1 || B > st->capacity || start < 0) return fail(PW_EINVAL, "bad ring arguments");
    if (st->obs_dim < 2) return fail(PW_EINVAL, "obs_dim must be >= 2");
    if (st->act_heads < 0 || st->act_heads > 2 || (st->act_heads == 2 && (st->head_width[0] < 0 || st->head_width[1] < 1)))
        return fail(PW_EINVAL, "bad act_heads / head_width");
    const size_t total = (size_t)B * st->num_agents * st->obs_dim;
    size_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(pw_replay_add_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       *st, start, start_dev, B, obs, act_idx, rew_shared, next_obs, final_obs, terminal, done);
    PW_HIP_CHECK(hipGetLastError());
    return PW_OK;
}

int pw_replay_add_tail(const pw_replay_store *st, int64_t start, const int64_t *start_dev, int64_t *next_start_dev,
                       int32_t B, const float *obs, const int32_t *act_idx, const float *rew_shared,
                       const float *next_obs, const float *final_obs, const uint8_t *terminal, const float *done,
                       float *episode_return, double *finished_sum, int64_t *finished_count, int64_t *step_counter,
                       void *stream)
{
    if (!st || !obs || !act_idx || !rew_shared || !next_obs || !terminal || !episode_return || !finished_sum ||
        !finished_count)
        return fail(PW_EINVAL, "null argument");
    if (int rc = plain_ring_only(st, "pw_replay_add_tail")) return rc;
    if (st->capacity < 1 || B < 1 || B > st->capacity || start < 0) return fail(PW_EINVAL, "bad ring arguments");
    if (st->obs_dim < 5) return fail(PW_EINVAL, "obs_dim must be >= 5");
    if (start_dev && next_start_dev == start_dev)
        return fail(PW_EINVAL, "next_start_dev must not alias start_dev (every workgroup reads start_dev)");
    const size_t total = (size_t)B * st->num_agents * st->obs_dim;
    size_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    ReplayTail tl;
    tl.episode_return = episode_return; tl.finished_sum = finished_sum; tl.finished_count = finished_count;
    tl.next_start_dev = next_start_dev; tl.step_counter = step_counter;
    hipLaunchKernelGGL(pw_replay_add_tail_kernel, dim3((unsigned)blocks + 1), dim3(256), 0, static_cast<hipStream_t>(stream),
                       *st, start, start_dev, B, obs, act_idx, rew_shared, next_obs, final_obs, terminal, done, tl);
    PW_HIP_CHECK(hipGetLastError());
    return PW_OK;
}

int pw_replay_add_rollout(const pw_replay_store *st, int64_t start, int32_t B, int32_t T, const float *obs0,
                          const pw_step_io *io, const int32_t *act, float *episode_return, double *finished_sum,
                          int64_t *finished_count, void *scratch, void *stream)
{
    if (!st || !obs0 || !io || !act || !io->obs || !io->rew_shared || !io->terminal) return fail(PW_EINVAL, "null argument");
    if (st->per_agent) return fail(PW_EINVAL, "pw_replay_add_rollout: per-agent rings are served by pw_replay_add and pw_replay_gather only");
    if (st->state_rows) return fail(PW_EINVAL, "pw_replay_add_rollout: a STATE ring is filled by pw_replay_add_state_wire only");
    if (st->capacity < 1 || B < 1 || T < 1 || (int64_t)B * T > st->capacity || start < 0)
        return fail(PW_EINVAL, "bad ring arguments (the chunk must fit the ring)");
    if (st->obs_dim < 5) return fail(PW_EINVAL, "obs_dim must be >= 5");
    if (st->act_heads == 2 && st->num_agents * st->obs_dim < 2 * st->num_agents) return fail(PW_EINVAL, "two-head ring: rows too short");
    if (episode_return && (!finished_sum || !finished_count || !scratch))
        return fail(PW_EINVAL, "bookkeeping needs episode_return, finished_sum, finished_count and scratch");
    const size_t total = (size_t)T * B * st->num_agents * st->obs_dim;
    size_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    const unsigned stat_blocks = episode_return ? (unsigned)((B + 255) / 256) : 0;
    ReplayTail tl;
    tl.episode_return = episode_return; tl.finished_sum = finished_sum; tl.finished_count = finished_count;
    tl.next_start_dev = nullptr; tl.step_counter = nullptr;
    hipLaunchKernelGGL(pw_replay_add_rollout_kernel, dim3((unsigned)blocks + stat_blocks), dim3(256), 0,
                       static_cast<hipStream_t>(stream), *st, start, B, T, obs0, *io, act, tl, stat_blocks,
                       static_cast<unsigned long long *>(scratch));
    PW_HIP_CHECK(hipGetLastError());
    return PW_OK;
}

size_t pw_policy_rollout_scratch_bytes(const pw_handle *h)
{
    if (!h) return 0;
    return (size_t)(2 * (size_t)h->kp.B + 1) * 8;  // one partial (sum, count) per workgroup; a workgroup holds >= 1 env
}

size_t pw_replay_add_rollout_scratch_bytes(int32_t B) { return (size_t)(2 * ((B + 255) / 256) + 1) * 8; }

int pw_replay_gather(const pw_replay_store *st, const int64_t *idx, int32_t b, float *out_obs, float *out_act,
                     float *out_rew, float *out_next_obs, float *out_done, void *stream)
{
    if (!st || !idx) return fail(PW_EINVAL, "null argument");
    if (b < 1) return fail(PW_EINVAL, "batch must be >= 1");
    if (st->obs_dim < 2) return fail(PW_EINVAL, "obs_dim must be >= 2");
    if (st->act_heads < 0 || st->act_heads > 2 || (st->act_heads == 2 && (st->head_width[0] < 0 || st->head_width[1] < 1)))
        return fail(PW_EINVAL, "bad act_heads / head_width");
    if (st->state_rows) {  // STATE ring: the rows are rebuilt from the slot's states and landmarks
        if (int rc = state_ring_ok(st, "pw_replay_gather")) return rc;
        if ((reinterpret_cast<uintptr_t>(out_obs) | reinterpret_cast<uintptr_t>(out_next_obs)) & 7)
            return fail(PW_EINVAL, "pw_replay_gather (STATE ring): out_obs / out_next_obs must be 8-byte aligned");
        size_t sblocks = ((size_t)b * st->num_agents * (st->obs_dim / 2) + 255) / 256;
        if (sblocks > 8192) sblocks = 8192;
        hipLaunchKernelGGL(pw_replay_gather_state_kernel, dim3((unsigned)sblocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                           *st, idx, b, out_obs, out_act, out_rew, out_next_obs, out_done);
        PW_HIP_CHECK(hipGetLastError());
        return PW_OK;
    }
    const size_t total = (size_t)b * st->num_agents * st->obs_dim;
    size_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(pw_replay_gather_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       *st, idx, b, out_obs, out_act, out_rew, out_next_obs, out_done);
    PW_HIP_CHECK(hipGetLastError());
    return PW_OK;
}

int pw_pack_transitions(const pw_step_io *io, int32_t B, int32_t N, int32_t D, const int32_t *sel_t,
                        const int32_t *sel_e, int32_t R, float *rows, void *stream)
{
    if (!io || !sel_t || !sel_e || !rows) return fail(PW_EINVAL, "null argument");
    if (!io->obs || !io->act_idx || !io->rew_shared) return fail(PW_EINVAL, "chunk needs obs, act_idx and rew_shared");
    if (B < 1 || N < 1 || D < 1 || R < 1) return fail(PW_EINVAL, "bad sizes");
    const size_t total = (size_t)R * (2 * (size_t)N * D + N + 2);
    size_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pw_pack_transitions_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       *io, B, N, D, sel_t, sel_e, R, rows);
    PW_HIP_CHECK(hipGetLastError());
    return PW_OK;
}

int pw_replay_add_packed(const pw_replay_store *st, int64_t start, int32_t R, const float *rows, void *stream)
{
    if (!st || !rows) return fail(PW_EINVAL, "null argument");
    if (int rc = plain_ring_only(st, "pw_replay_add_packed")) return rc;
    if (st->capacity < 1 || R < 1 || R > st->capacity || start < 0) return fail(PW_EINVAL, "bad ring arguments");
    const size_t total = (size_t)R * (2 * (size_t)st->num_agents * st->obs_dim + st->num_agents + 2);
    size_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pw_replay_add_packed_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       *st, start, R, rows);
    PW_HIP_CHECK(hipGetLastError());
    return PW_OK;
}

int pw_exchange(const pw_replay_store *st, int64_t start, int32_t R_in, const float *rows_in, const pw_step_io *io,
                int32_t B, int32_t N, int32_t D, const int32_t *sel_t, const int32_t *sel_e, int32_t R_out,
                float *rows_out, void *stream)
{
    const bool ingest = st && rows_in && R_in > 0;
    const bool pack = io && rows_out && R_out > 0;
    if (int rc = plain_ring_only(st, "pw_exchange")) return rc;
    if (!ingest && !pack) return fail(PW_EINVAL, "nothing to do");
    if (ingest && (st->capacity < 1 || R_in > st->capacity || start < 0)) return fail(PW_EINVAL, "bad ring arguments");
    if (pack && (!sel_t || !sel_e || !io->obs || !io->act_idx || !io->rew_shared || B < 1 || N < 1 || D < 1))
        return fail(PW_EINVAL, "chunk needs obs, act_idx, rew_shared and a selection");
    if (ingest && pack && (st->num_agents != N || st->obs_dim != D)) return fail(PW_EINVAL, "row width mismatch");
    const int Nn = pack ? N : st->num_agents, Dd = pack ? D : st->obs_dim;
    const size_t W = 2 * (size_t)Nn * Dd + Nn + 2;
    auto blocks_for = [&](int R) { size_t b = ((size_t)R * W + 255) / 256; return (int)(b > 2048 ? 2048 : b); };
    const int nb_in = ingest ? blocks_for(R_in) : 0, nb_out = pack ? blocks_for(R_out) : 0;
    pw_replay_store dummy_st;
    std::memset(&dummy_st, 0, sizeof(dummy_st));
    pw_step_io dummy_io;
    std::memset(&dummy_io, 0, sizeof(dummy_io));
    hipLaunchKernelGGL(pw_exchange_kernel, dim3((unsigned)(nb_in + nb_out)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       ingest ? *st : dummy_st, start, ingest ? R_in : 0, rows_in, nb_in, pack ? *io : dummy_io, B, Nn, Dd,
                       sel_t, sel_e, pack ? R_out : 0, rows_out);
    PW_HIP_CHECK(hipGetLastError());
    return PW_OK;
}

namespace {
void fill_wire(int32_t T, int32_t B, int32_t N, int32_t D, int32_t F, pw_chunk_wire *out)
{
    std::memset(out, 0, sizeof(*out));
    out->T = T; out->B = B; out->N = N; out->D = D; out->F = F;
    const size_t row = (size_t)B * N * D * sizeof(float);
    size_t off = 0;
    auto plane = [&](size_t bytes) { const size_t o = off; off = align_up(off + bytes, 256); return o; };
    out->obs0 = plane(row);
    out->obs = plane((size_t)T * row);
    out->final_rows = plane((size_t)F * row);
    out->rew_shared = plane((size_t)T * B * sizeof(float));
    out->act = plane((size_t)T * B * N);
    out->fin_slot = plane((size_t)T * B);
    out->total_bytes = off;
}

int check_wire(const pw_chunk_wire *w, const void *wire)
{
    if (!w || !wire) return fail(PW_EINVAL, "null argument");
    if (w->T < 1 || w->B < 1 || w->N < 1 || w->D < 1 || w->F < 0 || w->F > 254) return fail(PW_EINVAL, "bad wire layout");
    pw_chunk_wire ref;  // offsets are not trusted blindly: they must be the ones pw_chunk_wire_layout produces
    fill_wire(w->T, w->B, w->N, w->D, w->F, &ref);
    if (ref.obs0 != w->obs0 || ref.obs != w->obs || ref.final_rows != w->final_rows || ref.rew_shared != w->rew_shared ||
        ref.act != w->act || ref.fin_slot != w->fin_slot || ref.total_bytes != w->total_bytes)
        return fail(PW_EINVAL, "wire layout was not produced by pw_chunk_wire_layout");
    if (reinterpret_cast<uintptr_t>(wire) & 255) return fail(PW_EINVAL, "wire block must be 256-byte aligned");
    return PW_OK;
}
}  // namespace

int pw_chunk_wire_layout(int32_t T, int32_t B, int32_t N, int32_t D, int32_t max_episode_len, pw_chunk_wire *out)
{
    if (!out) return fail(PW_EINVAL, "null argument");
    if (T < 1 || B < 1 || N < 1 || D < 1 || max_episode_len < 0) return fail(PW_EINVAL, "bad sizes");
    const int64_t F = max_episode_len > 0 ? ((int64_t)T + max_episode_len - 1) / max_episode_len : 0;
    if (F > 254) return fail(PW_EINVAL, "more than 254 episode ends per env and chunk: use shorter chunks");
    fill_wire(T, B, N, D, (int32_t)F, out);
    return PW_OK;
}

int pw_chunk_wire_finalize(const pw_chunk_wire *w, void *wire, const float *obs0, const float *final_obs,
                           const uint8_t *terminal, const int32_t *act, void *stream)
{
    if (int rc = check_wire(w, wire)) return rc;
    if (!obs0 || !terminal || !act) return fail(PW_EINVAL, "null argument");
    const size_t per_step = (size_t)w->B * w->N * w->D;
    const unsigned row_blocks = (unsigned)((per_step + 255) / 256);
    size_t act_blocks = ((size_t)w->T * w->B * w->N + 255) / 256;
    if (act_blocks > 2048) act_blocks = 2048;
    hipLaunchKernelGGL(pw_chunk_wire_finalize_kernel, dim3(row_blocks + (unsigned)act_blocks), dim3(256), 0,
                       static_cast<hipStream_t>(stream), *w, wire, obs0, final_obs, terminal, act, row_blocks);
    PW_HIP_CHECK(hipGetLastError());
    return PW_OK;
}

int pw_replay_add_wire(const pw_replay_store *st, int64_t start, const pw_chunk_wire *w, const void *wire, void *stream)
{
    if (!st) return fail(PW_EINVAL, "null argument");
    if (int rc = plain_ring_only(st, "pw_replay_add_wire")) return rc;
    if (int rc = check_wire(w, wire)) return rc;
    if (st->num_agents != w->N || st->obs_dim != w->D) return fail(PW_EINVAL, "ring / wire shape mismatch");
    if (st->capacity < 1 || (int64_t)w->T * w->B > st->capacity || start < 0)
        return fail(PW_EINVAL, "bad ring arguments (the chunk must fit the ring)");
    const int ND = w->N * w->D;
    const bool vec = ND % 4 == 0 && ((reinterpret_cast<uintptr_t>(st->obs) | reinterpret_cast<uintptr_t>(st->next_obs)) & 15) == 0;
    const size_t total = (size_t)w->T * w->B * (vec ? ND / 4 : ND);
    size_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    if (vec)
        hipLaunchKernelGGL(pw_replay_add_wire_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                           *st, start, *w, wire);
    else
        hipLaunchKernelGGL(pw_replay_add_wire_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                           *st, start, *w, wire);
    PW_HIP_CHECK(hipGetLastError());
    return PW_OK;
}

namespace {
int state_wire_row_dim(int scenario, int N, int L, int A)
{
    if (scenario == PW_SIMPLE_SPREAD) return 4 + 2 * L;
    if (scenario == PW_SIMPLE_TAG) return 4 + 2 * L + 2 * (N - 1) + 2 * (N - A);
    return -1;
}

void fill_state_wire(int32_t scenario, int32_t T, int32_t B, int32_t N, int32_t L, int32_t A, int32_t F, pw_state_wire *out)
{
    std::memset(out, 0, sizeof(*out));
    out->T = T; out->B = B; out->N = N; out->L = L; out->D = state_wire_row_dim(scenario, N, L, A); out->F = F;
    out->scenario = scenario; out->num_adversaries = A;
    const size_t st = (size_t)B * N * sizeof(float4);
    size_t off = 0;
    auto plane = [&](size_t bytes) { const size_t o = off; off = align_up(off + bytes, 256); return o; };
    out->state0 = plane(st);
    out->state = plane((size_t)T * st);
    out->final_state = plane((size_t)F * st);
    out->lm = plane((size_t)(F + 1) * B * L * sizeof(float2));
    out->ep0 = plane((size_t)B * sizeof(uint32_t));
    out->rew_shared = plane((size_t)T * B * sizeof(float));
    out->act = plane((size_t)T * B * N);
    out->epi = plane((size_t)T * B);
    out->total_bytes = off;
}

int check_state_wire(const pw_state_wire *w, const void *wire)
{
    if (!w || !wire) return fail(PW_EINVAL, "null argument");
    if (w->T < 1 || w->B < 1 || w->N < 1 || w->L < 0 || w->F < 0 || w->F > 126 || w->num_adversaries < 0 || w->num_adversaries > w->N ||
        w->D < 4 || w->D != state_wire_row_dim(w->scenario, w->N, w->L, w->num_adversaries))
        return fail(PW_EINVAL, "bad state-wire layout");
    pw_state_wire ref;  // offsets must be the ones pw_state_wire_layout produces
    fill_state_wire(w->scenario, w->T, w->B, w->N, w->L, w->num_adversaries, w->F, &ref);
    if (std::memcmp(&ref, w, sizeof(ref)) != 0) return fail(PW_EINVAL, "wire layout was not produced by pw_state_wire_layout");
    if (reinterpret_cast<uintptr_t>(wire) & 255) return fail(PW_EINVAL, "wire block must be 256-byte aligned");
    return PW_OK;
}

int state_wire_handle_ok(const pw_handle *h, const pw_state_wire *w)
{
    if (!h) return fail(PW_EINVAL, "null handle");
    const bool spread = h->cfg.scenario == PW_SIMPLE_SPREAD && h->cfg.obs_mode == PW_OBS_LOCAL;
    const bool tag = h->cfg.scenario == PW_SIMPLE_TAG;
    if (!(spread || tag) || h->kp.D != state_wire_row_dim(h->cfg.scenario, h->kp.N, h->kp.L, h->kp.A))
        return fail(PW_EINVAL, "state-only wire blocks serve simple_spread with the local observation and simple_tag (rows that are a "
                               "function of {vel, pos} and the landmarks); use pw_chunk_wire_* elsewhere");
    if (w && (w->scenario != h->cfg.scenario || w->B != h->kp.B || w->N != h->kp.N || w->L != h->kp.L ||
              (tag && w->num_adversaries != h->kp.A)))
        return fail(PW_EINVAL, "wire / handle shape mismatch");
    return PW_OK;
}
}  // namespace

int pw_state_wire_layout_scn(int32_t scenario, int32_t T, int32_t B, int32_t N, int32_t L, int32_t num_adversaries,
                             int32_t max_episode_len, pw_state_wire *out)
{
    if (!out) return fail(PW_EINVAL, "null argument");
    if (scenario != PW_SIMPLE_SPREAD && scenario != PW_SIMPLE_TAG)
        return fail(PW_EINVAL, "state-only wire blocks serve simple_spread (local observation) and simple_tag");
    if (scenario == PW_SIMPLE_SPREAD) num_adversaries = 0;
    if (T < 1 || B < 1 || N < 1 || L < 0 || max_episode_len < 0 || num_adversaries < 0 || num_adversaries > N) return fail(PW_EINVAL, "bad sizes");
    const int64_t F = max_episode_len > 0 ? ((int64_t)T + max_episode_len - 1) / max_episode_len : 0;
    if (F > 126) return fail(PW_EINVAL, "more than 126 episode ends per env and chunk: use shorter chunks");
    fill_state_wire(scenario, T, B, N, L, num_adversaries, (int32_t)F, out);
    return PW_OK;
}

int pw_state_wire_layout(int32_t T, int32_t B, int32_t N, int32_t L, int32_t max_episode_len, pw_state_wire *out)
{
    return pw_state_wire_layout_scn(PW_SIMPLE_SPREAD, T, B, N, L, 0, max_episode_len, out);
}

int pw_state_wire_begin(const pw_handle *h, const pw_state_wire *w, void *wire, void *stream)
{
    if (int rc = check_ready(h)) return rc;
    if (int rc = check_state_wire(w, wire)) return rc;
    if (int rc = state_wire_handle_ok(h, w)) return rc;
    const KParams &kp = h->kp;
    const size_t n = (size_t)w->B * (w->N > w->L ? w->N : w->L);
    hipLaunchKernelGGL(pw_state_wire_begin_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       *w, wire, kp.pos_x, kp.pos_y, kp.vel_x, kp.vel_y, kp.lm_x, kp.lm_y, kp.ep_count);
    PW_HIP_CHECK(hipGetLastError());
    return PW_OK;
}

int pw_state_wire_finalize(const pw_handle *h, const pw_state_wire *w, void *wire, const float *obs, const float *final_obs,
                           const uint8_t *terminal, const int32_t *act, void *stream)
{
    if (int rc = check_state_wire(w, wire)) return rc;
    if (int rc = state_wire_handle_ok(h, w)) return rc;
    if (!obs || !terminal || !act) return fail(PW_EINVAL, "null argument");
    if ((reinterpret_cast<uintptr_t>(obs) | reinterpret_cast<uintptr_t>(final_obs)) & 7)
        return fail(PW_EINVAL, "obs and final_obs must be 8-byte aligned");
    const size_t BN = (size_t)w->B * w->N, total = (size_t)w->T * BN;
    size_t copy_blocks = (total + 255) / 256;
    if (copy_blocks > 8192) copy_blocks = 8192;
    const unsigned env_blocks = (unsigned)((BN + 255) / 256);
    size_t act_blocks = (total + 255) / 256;
    if (act_blocks > 2048) act_blocks = 2048;
    // the landmarks an in-chunk reset drew: simple_spread U(-1, 1), simple_tag U(-0.9, 0.9) (upstream reset_world)
    const float lm_lo = w->scenario == PW_SIMPLE_TAG ? -0.9f : -1.0f, lm_hi = w->scenario == PW_SIMPLE_TAG ? 0.9f : 1.0f;
    hipLaunchKernelGGL(pw_state_wire_finalize_kernel, dim3((unsigned)(copy_blocks + env_blocks + act_blocks)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), *w, wire, obs, final_obs, terminal, act, (uint64_t)h->kp.seed,
                       (uint64_t)h->kp.env_id_base, lm_lo, lm_hi, (unsigned)copy_blocks, env_blocks);
    PW_HIP_CHECK(hipGetLastError());
    return PW_OK;
}

int pw_replay_add_state_wire(const pw_replay_store *st, int64_t start, const pw_state_wire *w, const void *wire, void *stream)
{
    if (!st) return fail(PW_EINVAL, "null argument");
    if (int rc = check_state_wire(w, wire)) return rc;
    if (st->num_agents != w->N || st->obs_dim != w->D) return fail(PW_EINVAL, "ring / wire shape mismatch");
    if (st->capacity < 1 || (int64_t)w->T * w->B > st->capacity || start < 0)
        return fail(PW_EINVAL, "bad ring arguments (the chunk must fit the ring)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (st->state_rows) {  // STATE ring: copy what the block carries; pw_replay_gather rebuilds the rows
        if (int rc = state_ring_ok(st, "pw_replay_add_state_wire")) return rc;
        if (st->scenario != w->scenario || st->num_landmarks != w->L || st->num_adversaries != w->num_adversaries)
            return fail(PW_EINVAL, "STATE ring / wire scenario mismatch");
        size_t blocks = ((size_t)w->T * w->B * w->N + 255) / 256;
        if (blocks > 16384) blocks = 16384;
        hipLaunchKernelGGL(pw_replay_add_state_wire_to_state_ring_kernel, dim3((unsigned)blocks), dim3(256), 0, s, *st, start, *w, wire);
        PW_HIP_CHECK(hipGetLastError());
        return PW_OK;
    }
    if (int rc = plain_ring_only(st, "pw_replay_add_state_wire")) return rc;
    const uintptr_t al = reinterpret_cast<uintptr_t>(st->obs) | reinterpret_cast<uintptr_t>(st->next_obs);
    if (al & 7) return fail(PW_EINVAL, "ring observation planes must be 8-byte aligned");
    if (w->scenario != PW_SIMPLE_SPREAD) {  // simple_tag: 8-byte units (other agents' states feed every row)
        size_t blocks = ((size_t)w->T * w->B * w->N * (w->D / 2) + 255) / 256;
        if (blocks > 16384) blocks = 16384;
        hipLaunchKernelGGL(pw_replay_add_state_wire_units_kernel, dim3((unsigned)blocks), dim3(256), 0, s, *st, start, *w, wire);
        PW_HIP_CHECK(hipGetLastError());
        return PW_OK;
    }
    const bool v4 = w->L % 2 == 0 && (al & 15) == 0;
    const size_t total = (size_t)w->T * w->B * w->N * (v4 ? w->D / 4 : w->D / 2);
    size_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    if (v4)
        hipLaunchKernelGGL(pw_replay_add_state_wire_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, s, *st, start, *w, wire);
    else
        hipLaunchKernelGGL(pw_replay_add_state_wire_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, s, *st, start, *w, wire);
    PW_HIP_CHECK(hipGetLastError());
    return PW_OK;
}

}  // extern "C"

namespace {
void fill_ref_wire(int32_t T, int32_t B, int32_t F, pw_ref_wire *out)
{
    std::memset(out, 0, sizeof(*out));
    out->T = T; out->B = B; out->F = F;
    const size_t hd = (size_t)B * kRefN * kRefHead * sizeof(float);
    size_t off = 0;
    auto plane = [&](size_t bytes) { const size_t o = off; off = align_up(off + bytes, 256); return o; };
    out->head0 = plane(hd);
    out->head = plane((size_t)T * hd);
    out->final_head = plane((size_t)F * hd);
    out->goal = plane((size_t)(F + 1) * B * kRefN);
    out->comm0 = plane((size_t)B * kRefN);
    out->rew_shared = plane((size_t)T * B * sizeof(float));
    out->act = plane((size_t)T * B * kRefN * 2);
    out->epi = plane((size_t)T * B);
    out->total_bytes = off;
}

int check_ref_wire(const pw_ref_wire *w, const void *wire)
{
    if (!w || !wire) return fail(PW_EINVAL, "null argument");
    if (w->T < 1 || w->B < 1 || w->F < 0 || w->F > 126) return fail(PW_EINVAL, "bad ref-wire layout");
    pw_ref_wire ref;
    fill_ref_wire(w->T, w->B, w->F, &ref);
    if (std::memcmp(&ref, w, sizeof(ref)) != 0) return fail(PW_EINVAL, "wire layout was not produced by pw_ref_wire_layout");
    if (reinterpret_cast<uintptr_t>(wire) & 255) return fail(PW_EINVAL, "wire block must be 256-byte aligned");
    return PW_OK;
}
}  // namespace

extern "C" {

int pw_ref_wire_layout(int32_t T, int32_t B, int32_t max_episode_len, pw_ref_wire *out)
{
    if (!out) return fail(PW_EINVAL, "null argument");
    if (T < 1 || B < 1 || max_episode_len < 0) return fail(PW_EINVAL, "bad sizes");
    const int64_t F = max_episode_len > 0 ? ((int64_t)T + max_episode_len - 1) / max_episode_len : 0;
    if (F > 126) return fail(PW_EINVAL, "more than 126 episode ends per env and chunk: use shorter chunks");
    fill_ref_wire(T, B, (int32_t)F, out);
    return PW_OK;
}

int pw_ref_wire_finalize(const pw_ref_wire *w, void *wire, const float *obs0, const float *obs, const float *final_obs,
                         const uint8_t *terminal, const int32_t *act, void *stream)
{
    if (int rc = check_ref_wire(w, wire)) return rc;
    if (!obs0 || !obs || !terminal || !act) return fail(PW_EINVAL, "null argument");
    if (w->F > 0 && !final_obs) return fail(PW_EINVAL, "final_obs is needed when episodes end inside the chunk");
    const size_t BN = (size_t)w->B * kRefN;
    size_t copy_blocks = ((size_t)w->T * BN * kRefHead + 255) / 256;
    if (copy_blocks > 8192) copy_blocks = 8192;
    const unsigned env_blocks = (unsigned)((BN + 255) / 256);
    size_t act_blocks = ((size_t)w->T * BN * 2 + 255) / 256;
    if (act_blocks > 2048) act_blocks = 2048;
    hipLaunchKernelGGL(pw_ref_wire_finalize_kernel, dim3((unsigned)(copy_blocks + env_blocks + act_blocks)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), *w, wire, obs0, obs, final_obs, terminal, act, (unsigned)copy_blocks, env_blocks);
    PW_HIP_CHECK(hipGetLastError());
    return PW_OK;
}

int pw_replay_add_ref_wire(const pw_replay_store *st, int64_t start, const pw_ref_wire *w, const void *wire, void *stream)
{
    if (!st) return fail(PW_EINVAL, "null argument");
    if (int rc = check_ref_wire(w, wire)) return rc;
    if (st->state_rows || st->per_agent || st->act_heads != 2 || st->head_width[1] != PW_DIM_C || (st->head_width[0] != 0 && st->head_width[0] != 5))
        return fail(PW_EINVAL, "pw_replay_add_ref_wire: the ring must be the two-head ring of simple_reference (act_heads = 2, head widths 5 | dim_c)");
    if (st->num_agents != kRefN || st->obs_dim != kRefD) return fail(PW_EINVAL, "ring / wire shape mismatch (simple_reference: N = 2, D = 21)");
    if (st->capacity < 1 || (int64_t)w->T * w->B > st->capacity || start < 0)
        return fail(PW_EINVAL, "bad ring arguments (the chunk must fit the ring)");
    size_t blocks = ((size_t)w->T * w->B * kRefN * kRefD + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(pw_replay_add_ref_wire_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), *st, start, *w, wire);
    PW_HIP_CHECK(hipGetLastError());
    return PW_OK;
}

}  // extern "C"
