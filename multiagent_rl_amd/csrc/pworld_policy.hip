// libpworld.so, second translation unit -- the action producer (rls/model/ac_network_multi_gumbel.py:24-67,
// ddpg_gumbel_fix.py:86-116) and the policy-in-the-loop rollouts: pw_dense / pw_actor_front / pw_bilstm_forward /
// pw_actor_head / pw_actor_fused, pw_policy_rollout (the env step inside it is pw_kernels_spread.hpp's /
// pw_kernels_tag.hpp's arithmetic), the rollout bookkeeping launches and the device-math test hook.
// Entry points are declared in include/pworld.h; the handle and the error text are shared with pworld.hip.
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
#include "pw_kernels_tag.hpp"
#include "pw_kernels_reference.hpp"
#include "pw_handle.hpp"
#include "pw_kernels_policy.hpp"
#include "pw_kernels_actor16.hpp"
#include "pw_kernels_policy2.hpp"
#include "pw_kernels_policy3.hpp"
#include "pw_kernels_policy3j.hpp"
#include "pw_kernels_policy_tag.hpp"
#include "pw_kernels_policy_ref.hpp"

extern "C" {

int pw_bilstm_forward(const float *G, const float *w_hh_fw, const float *w_hh_bw, int32_t B, int32_t N,
                      int32_t relu_out, float *H, void *stream)
{
    if (!G || !w_hh_fw || !w_hh_bw || !H) return fail(PW_EINVAL, "null argument");
    if (B < 1 || N < 1) return fail(PW_EINVAL, "bad sizes");
    if ((reinterpret_cast<uintptr_t>(w_hh_fw) | reinterpret_cast<uintptr_t>(w_hh_bw)) & 15)
        return fail(PW_EINVAL, "w_hh must be 16-byte aligned");
    const long seqs = 2L * B;
    hipLaunchKernelGGL(pw_bilstm_kernel, dim3((unsigned)((seqs + 7) / 8)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       G, w_hh_fw, w_hh_bw, B, N, relu_out, H);
    PW_HIP_CHECK(hipGetLastError());
    return PW_OK;
}

int pw_actor_head(const float *H, const float *w2, const float *b2, int64_t rows, uint64_t seed, uint64_t step,
                  const int64_t *step_dev, float *logits, int32_t *act, void *stream)
{
    if (!H || !w2 || !b2 || (!logits && !act)) return fail(PW_EINVAL, "null argument");
    if (rows < 1) return fail(PW_EINVAL, "bad sizes");
    if (reinterpret_cast<uintptr_t>(H) & 15) return fail(PW_EINVAL, "H must be 16-byte aligned");
    hipLaunchKernelGGL(pw_actor_head_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), H, w2, b2, (long)rows, seed, step, step_dev, logits, act);
    PW_HIP_CHECK(hipGetLastError());
    return PW_OK;
}

namespace {
int g_actor_bf16x3 = 0;  // process-wide, off unless pw_actor_set_bf16x3 turns it on (no environment reads)
}  // namespace

int pw_actor_set_bf16x3(int32_t on)
{
    const int prev = g_actor_bf16x3;
    g_actor_bf16x3 = on != 0;
    return prev;
}

int pw_actor_fused(const float *X, const float *frag, const float *b1, const float *b_ih, const float *w_hh_fw,
                   const float *w_hh_bw, const float *w2, const float *b2, int32_t n_out0, int32_t n_out1, int64_t B,
                   int32_t N, int32_t in_dim, int32_t relu_out, uint64_t seed, uint64_t step, const int64_t *step_dev,
                   float *H, float *logits, int32_t *act, void *stream)
{
    if (n_out0 < 1 || n_out1 < 0 || n_out0 + n_out1 > 16) return fail(PW_EINVAL, "head sizes: n_out0 >= 1, n_out0 + n_out1 <= 16");
    if (!X || !frag || !b1 || !b_ih || !w_hh_fw || !w_hh_bw || !w2 || !b2 || (!H && !logits && !act))
        return fail(PW_EINVAL, "null argument");
    if (B < 1 || N < 1 || N > 96 || B > (int64_t)0x7fffffff) return fail(PW_EINVAL, "N must be in [1, 96]");
    if (in_dim < 1 || in_dim > 64) return fail(PW_EINVAL, "in_dim must be in [1, 64]");
    if ((reinterpret_cast<uintptr_t>(frag) | reinterpret_cast<uintptr_t>(w_hh_fw) | reinterpret_cast<uintptr_t>(w_hh_bw)) & 15)
        return fail(PW_EINVAL, "frag and w_hh must be 16-byte aligned");
    ActorFusedArgs a;
    a.X = X; a.frag = frag; a.b1 = b1; a.bih = b_ih; a.whh_f = w_hh_fw; a.whh_r = w_hh_bw; a.w2 = w2; a.b2 = b2;
    a.B = (int)B; a.N = N; a.D = in_dim; a.relu_out = relu_out; a.n_out0 = n_out0; a.n_out1 = n_out1;
    a.E = 96 / N < 16 ? 96 / N : 16;
    a.seed = seed; a.step = step; a.step_dev = step_dev; a.H = H; a.logits = logits; a.act = act;
    a.bf16x3 = g_actor_bf16x3;  // honoured by the 16x16x4-core kernel (N <= 16) only
    const int S1C = (in_dim + 7) / 8, S1 = 4 * S1C;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // N <= 16: the BiLSTM on v_mfma_f32_16x16x4_f32 (pw_kernels_actor16.hpp), 16 environments per workgroup whatever N is; same
    // bits as the kernel below, which keeps the long sequences (its 96-row workgroups hold N <= 96)
    const size_t shm16 = actor16_lds_floats(N, 16 * N, S1) * sizeof(float);
    if (N <= 16 && shm16 <= 160 * 1024) {
        a.E = 16;
        const unsigned grid16 = (unsigned)((B + 15) / 16);
#define PW_FUSED16B(C, BF)                                                                                               \
    do {                                                                                                                 \
        static unsigned long long attr_set16 = 0; /* bit = device */                                                     \
        PW_LDS_OPTIN(&attr_set16, (pw_actor_fused16_kernel<C, BF>)); \
        hipLaunchKernelGGL((pw_actor_fused16_kernel<C, BF>), dim3(grid16), dim3(512), shm16, st, a);                     \
    } while (0)
#define PW_FUSED16(C) case C: if (a.bf16x3) PW_FUSED16B(C, true); else PW_FUSED16B(C, false); break;
        switch (S1C) {
            PW_FUSED16(1) PW_FUSED16(2) PW_FUSED16(3) PW_FUSED16(4) PW_FUSED16(5) PW_FUSED16(6) PW_FUSED16(7) PW_FUSED16(8)
        }
#undef PW_FUSED16
#undef PW_FUSED16B
        PW_HIP_CHECK(hipGetLastError());
        return PW_OK;
    }
    if (a.bf16x3) return fail(PW_EINVAL, "pw_actor_set_bf16x3 serves N <= 16 only");
    const size_t shm = actor_lds_bytes(S1);
    const unsigned grid = (unsigned)((B + a.E - 1) / a.E);
    static unsigned long long attr_set[9] = {};  // per kernel: bit = device
#define PW_FUSED(C)                                                                                                      \
    case C:                                                                                                              \
        PW_LDS_OPTIN(&attr_set[C], (pw_actor_fused_kernel<C>)); \
        hipLaunchKernelGGL(pw_actor_fused_kernel<C>, dim3(grid), dim3(512), shm, st, a);                                 \
        break;
    switch (S1C) {
        PW_FUSED(1) PW_FUSED(2) PW_FUSED(3) PW_FUSED(4) PW_FUSED(5) PW_FUSED(6) PW_FUSED(7) PW_FUSED(8)
    }
#undef PW_FUSED
    PW_HIP_CHECK(hipGetLastError());
    return PW_OK;
}

int pw_policy_rollout(pw_handle *h, const float *frag, const float *b1, const float *b_ih, const float *w_hh_fw,
                      const float *w_hh_bw, const float *w2, const float *b2, int32_t relu_out, uint64_t seed,
                      uint64_t step, const int64_t *step_dev, const pw_step_io *io, int32_t *act_out, int32_t num_steps,
                      const pw_rollout_sink *sink, void *stream)
{
    if (int rc = check_ready(h)) return rc;
    if (!frag || !b1 || !b_ih || !w_hh_fw || !w_hh_bw || !w2 || !b2 || !io) return fail(PW_EINVAL, "null argument");
    if (num_steps < 1) return fail(PW_EINVAL, "num_steps must be >= 1");
    const KParams &kp = h->kp;
    if (h->cfg.scenario == PW_SIMPLE_REFERENCE) {
        // the MultiDiscrete scenario: two-head actor [5 | PW_DIM_C] (w2 [15,64], b2 [15]), act_out [T,B,N,2]
        const bool rsink = sink && sink->ring;
        if (sink) {
            if (sink->ring && (sink->ring->act_heads != 2 || sink->ring->per_agent || sink->ring->head_width[1] != PW_DIM_C ||
                               (sink->ring->head_width[0] != 0 && sink->ring->head_width[0] != 5)))
                return fail(PW_EINVAL, "pw_policy_rollout sink on simple_reference: the ring must be the two-head ring (act_heads = 2, widths 5 | dim_c)");
            if (sink->ring && (sink->ring->num_agents != 2 || sink->ring->obs_dim != kp.D || sink->ring->capacity < 1 ||
                               sink->ring_start < 0 || sink->ring_start >= sink->ring->capacity || (int64_t)num_steps * kp.B > sink->ring->capacity))
                return fail(PW_EINVAL, "ring sink: shape mismatch or the chunk does not fit the ring");
            if (sink->episode_return && (!sink->finished_sum || !sink->finished_count || !sink->scratch))
                return fail(PW_EINVAL, "bookkeeping needs episode_return, finished_sum, finished_count and scratch");
        }
        if (io->act_idx || io->act_vec || io->act_comm || io->coll)
            return fail(PW_EINVAL, "pw_policy_rollout produces the actions itself (act_out) and has no coll output");
        if (!rsink && (!act_out || !io->obs || !io->rew || !io->rew_shared || !io->done || !io->terminal))
            return fail(PW_EINVAL, "without a ring sink, act_out and the obs, rew, rew_shared, done, terminal outputs are required");
        if ((reinterpret_cast<uintptr_t>(io->obs) | reinterpret_cast<uintptr_t>(io->final_obs) | reinterpret_cast<uintptr_t>(frag) |
             reinterpret_cast<uintptr_t>(w_hh_fw) | reinterpret_cast<uintptr_t>(w_hh_bw)) & 15)
            return fail(PW_EINVAL, "obs, final_obs, frag and w_hh must be 16-byte aligned");
        PolicyRolloutRefArgs R;
        std::memset(&R, 0, sizeof(R));
        ActorFusedArgs &ra = R.A;
        ra.frag = frag; ra.b1 = b1; ra.bih = b_ih; ra.whh_f = w_hh_fw; ra.whh_r = w_hh_bw; ra.w2 = w2; ra.b2 = b2;
        ra.B = kp.B; ra.N = 2; ra.D = kp.D; ra.relu_out = relu_out; ra.n_out0 = 5; ra.n_out1 = PW_DIM_C;
        ra.E = 16;
        ra.seed = seed; ra.step = step; ra.step_dev = step_dev;
        if (h->actor_bf16x3) return fail(PW_EINVAL, "PW_ACTOR_BF16X3 serves the simple_spread rollout (and pw_actor_fused) only");
        R.V = ref_params(h);
        R.T = num_steps; R.act_out = act_out;
        R.obs = io->obs; R.final_obs = io->final_obs; R.rew = io->rew; R.rew_shared = io->rew_shared;
        R.done = io->done; R.terminal = io->terminal;
        const int rS1C = (kp.D + 7) / 8;
        if (rS1C != 3) return fail(PW_EINVAL, "simple_reference one-launch rollout: the observation is 21 numbers (3 landmarks)");
        const size_t rshm = actor16_lds_floats(2, 32, 4 * rS1C) * sizeof(float) + (size_t)2 * kFusedRows * kp.D * sizeof(float) + 2 * kFusedRows * sizeof(int32_t) +
                            16 * (sizeof(double) + sizeof(int)) + (size_t)actor16_noise_floats(32, 5 + PW_DIM_C) * sizeof(float);
        if (rsink) { R.ring = *sink->ring; R.has_ring = 1; R.ring_start = sink->ring_start; }
        if (sink && sink->episode_return) {
            R.episode_return = sink->episode_return; R.finished_sum = sink->finished_sum;
            R.finished_count = sink->finished_count; R.scratch = static_cast<unsigned long long *>(sink->scratch);
        }
        if (sink) {
            static unsigned long long attr_sets = 0; /* bit = device */
            PW_LDS_OPTIN(&attr_sets, (pw_policy_rollout_ref_kernel<3, true>));
            hipLaunchKernelGGL((pw_policy_rollout_ref_kernel<3, true>), dim3((unsigned)((kp.B + 15) / 16)), dim3(512), rshm,
                               static_cast<hipStream_t>(stream), R);
        } else {
            static unsigned long long attr_set = 0; /* bit = device */
            PW_LDS_OPTIN(&attr_set, (pw_policy_rollout_ref_kernel<3>));
            hipLaunchKernelGGL((pw_policy_rollout_ref_kernel<3>), dim3((unsigned)((kp.B + 15) / 16)), dim3(512), rshm,
                               static_cast<hipStream_t>(stream), R);
        }
        PW_HIP_CHECK(hipGetLastError());
        h->last_kernel = "pw_policy_rollout_ref_kernel<3>";
        return PW_OK;
    }
    const bool tag = h->cfg.scenario == PW_SIMPLE_TAG && h->tag_fast;
    if (!h->fast && !tag)
        return fail(PW_EINVAL, "pw_policy_rollout serves the simple_spread fast-path configurations (local observation, "
                               "homogeneous agents, L <= N) and simple_tag with homogeneous roles");
    // observation rows longer than 64 numbers (simple_spread N = L > 30) are served by the just-in-time form alone (D <= 104)
    const bool wide = kp.D > 64;
    if (kp.N > 64 || kp.D > 104) return fail(PW_EINVAL, "N must be <= 64 and the observation length <= 104");
    if (io->act_idx || io->act_vec || io->act_comm || io->coll)
        return fail(PW_EINVAL, "pw_policy_rollout produces the actions itself (act_out) and has no coll output");
    const bool have_sink = sink && sink->ring;
    if (!have_sink && (!act_out || !io->obs || !io->rew || !io->rew_shared || !io->done || !io->terminal))
        return fail(PW_EINVAL, "without a ring sink, act_out and the obs, rew, rew_shared, done, terminal outputs are required");
    if (sink) {
        if (sink->ring && sink->ring->state_rows) {   // a STATE ring as the sink: {vel, pos} + the episode's landmarks instead of the two rows
            if (int rc = state_ring_ok(sink->ring, "pw_policy_rollout sink")) return rc;
            if (sink->ring->scenario != h->cfg.scenario || sink->ring->num_landmarks != kp.L || sink->ring->num_adversaries != (tag ? kp.A : 0))
                return fail(PW_EINVAL, "pw_policy_rollout sink: the STATE ring belongs to another scenario / shape");
        } else if (int rc = plain_ring_only(sink->ring, "pw_policy_rollout sink")) return rc;
        if (sink->ring && (sink->ring->num_agents != kp.N || sink->ring->obs_dim != kp.D || sink->ring->capacity < 1 ||
                           sink->ring_start < 0 || sink->ring_start >= sink->ring->capacity || (int64_t)num_steps * kp.B > sink->ring->capacity))
            return fail(PW_EINVAL, "ring sink: shape mismatch or the chunk does not fit the ring");
        if (sink->episode_return && (!sink->finished_sum || !sink->finished_count || !sink->scratch))
            return fail(PW_EINVAL, "bookkeeping needs episode_return, finished_sum, finished_count and scratch");
    }
    if ((reinterpret_cast<uintptr_t>(io->obs) | reinterpret_cast<uintptr_t>(io->final_obs) | reinterpret_cast<uintptr_t>(frag) |
         reinterpret_cast<uintptr_t>(w_hh_fw) | reinterpret_cast<uintptr_t>(w_hh_bw) |
         (have_sink ? reinterpret_cast<uintptr_t>(sink->ring->next_obs) | reinterpret_cast<uintptr_t>(sink->ring->obs) : 0)) & 15)
        return fail(PW_EINVAL, "obs, final_obs, frag, w_hh and the ring planes must be 16-byte aligned");
    if (tag) {
        PolicyRolloutTagArgs Q;
        std::memset(&Q, 0, sizeof(Q));
        ActorFusedArgs &qa = Q.A;
        qa.frag = frag; qa.b1 = b1; qa.bih = b_ih; qa.whh_f = w_hh_fw; qa.whh_r = w_hh_bw; qa.w2 = w2; qa.b2 = b2;
        qa.B = kp.B; qa.N = kp.N; qa.D = kp.D; qa.relu_out = relu_out; qa.n_out0 = 5; qa.n_out1 = 0;
        qa.E = 96 / kp.N < 16 ? 96 / kp.N : 16;
        qa.seed = seed; qa.step = step; qa.step_dev = step_dev;
        if (h->actor_bf16x3) return fail(PW_EINVAL, "PW_ACTOR_BF16X3 serves the simple_spread rollout (and pw_actor_fused) only");
        Q.V = h->tp;
        TagParams &tv = Q.V;
        tv.pos_x = kp.pos_x; tv.pos_y = kp.pos_y; tv.vel_x = kp.vel_x; tv.vel_y = kp.vel_y;
        tv.lm_x = kp.lm_x; tv.lm_y = kp.lm_y; tv.ep_step = kp.ep_step; tv.ep_count = kp.ep_count;
        tv.obs = io->obs; tv.final_obs = io->final_obs; tv.rew = io->rew; tv.rew_shared = io->rew_shared;
        tv.done = io->done; tv.terminal = io->terminal;
        Q.T = num_steps; Q.act_out = act_out;
        if (have_sink) { Q.ring = *sink->ring; Q.has_ring = 1; Q.ring_start = sink->ring_start; }
        if (sink && sink->episode_return) {
            Q.episode_return = sink->episode_return; Q.finished_sum = sink->finished_sum;
            Q.finished_count = sink->finished_count; Q.scratch = static_cast<unsigned long long *>(sink->scratch);
        }
        const int tS1C = (kp.D + 7) / 8;
        const size_t tshm = policy_tag_lds_bytes(4 * tS1C, kp.D, qa.E, kp.L, kp.N);
        if (tshm > 160 * 1024 || tS1C < 2 || tS1C > 6)
            return fail(PW_EINVAL, "simple_tag one-launch rollout: observation length must be in [9, 48] and fit the LDS");
        const unsigned tgrid = (unsigned)((kp.B + qa.E - 1) / qa.E);
        hipStream_t tst = static_cast<hipStream_t>(stream);
#define PW_TG3(C, SK)                                                                                                    \
    do {                                                                                                                 \
        static unsigned long long attr_set = 0; /* bit = device */                                            \
        PW_LDS_OPTIN(&attr_set, (pw_policy_rollout_tag_kernel<C, SK>)); \
        hipLaunchKernelGGL((pw_policy_rollout_tag_kernel<C, SK>), dim3(tgrid), dim3(512), tshm, tst, Q);                 \
    } while (0)
#define PW_TG(C) case C: if (sink) PW_TG3(C, true); else PW_TG3(C, false); break;
        switch (tS1C) { PW_TG(2) PW_TG(3) PW_TG(4) PW_TG(5) PW_TG(6) }
#undef PW_TG3
#undef PW_TG
        PW_HIP_CHECK(hipGetLastError());
        h->last_kernel = "pw_policy_rollout_tag_kernel";
        return PW_OK;
    }
    PolicyRolloutArgs P;
    std::memset(&P, 0, sizeof(P));
    ActorFusedArgs &a = P.A;
    a.frag = frag; a.b1 = b1; a.bih = b_ih; a.whh_f = w_hh_fw; a.whh_r = w_hh_bw; a.w2 = w2; a.b2 = b2;
    a.B = kp.B; a.N = kp.N; a.D = kp.D; a.relu_out = relu_out; a.n_out0 = 5; a.n_out1 = 0;
    a.E = 96 / kp.N < 16 ? 96 / kp.N : 16;
    a.seed = seed; a.step = step; a.step_dev = step_dev; a.bf16x3 = h->actor_bf16x3;
    StreamParams &A = P.V;
    A.B = kp.B; A.N = kp.N; A.L = kp.L; A.epw = kp.epw;
    A.max_episode_len = kp.max_episode_len; A.auto_reset = kp.auto_reset;
    A.seed = kp.seed; A.env_id_base = kp.env_id_base;
    A.dt = kp.dt; A.damp = kp.damp; A.contact_force = kp.contact_force; A.contact_margin = kp.contact_margin;
    A.mass = kp.mass;
    A.dist_min = h->fc.dist_min; A.coll_thr2 = h->fc.coll_thr2; A.near_thr2 = h->fc.near_thr2;
    A.sens = h->fc.sens; A.fscale = h->fc.fscale;
    A.pos_x = kp.pos_x; A.pos_y = kp.pos_y; A.vel_x = kp.vel_x; A.vel_y = kp.vel_y;
    A.lm_x = kp.lm_x; A.lm_y = kp.lm_y; A.ep_step = kp.ep_step; A.ep_count = kp.ep_count;
    A.obs = io->obs; A.final_obs = io->final_obs; A.rew = io->rew; A.rew_shared = io->rew_shared;
    A.done = io->done; A.terminal = io->terminal;
    P.T = num_steps; P.act_out = act_out;
    if (have_sink) { P.ring = *sink->ring; P.has_ring = 1; P.ring_start = sink->ring_start; }
    if (sink && sink->episode_return) {
        P.episode_return = sink->episode_return; P.finished_sum = sink->finished_sum;
        P.finished_count = sink->finished_count; P.scratch = static_cast<unsigned long long *>(sink->scratch);
    }
    const int S1C = (kp.D + 7) / 8, S1 = 4 * S1C;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // Kernel forms.  pw_policy_rollout3_kernel (pw_kernels_policy3.hpp): the whole BiLSTM on v_mfma_f32_16x16x4_f32, one timestep
    // per barrier, 16 environments per workgroup at any N; its dense1 output takes 4 KB of LDS per agent, so the environments per
    // workgroup shrink past N = 12.  pw_policy_rollout3j_kernel (pw_kernels_policy3j.hpp): the same with dense1 just in time.
    // (The first, phase-by-phase form and the second, role-specialised-waves form -- policy_form 1 / 2 -- were retired in 0.1.5:
    // nothing selected them.)  pw_dispatch.policy_form overrides the choice.
    const int form = h->disp.policy_form;
    if (form == 1 || form == 2)
        return fail(PW_EINVAL, "policy_form 1 / 2 (the first two rollout kernels) were retired in 0.1.5; use 0 (automatic), 3 or 4");
    int E3 = 0;
    for (int e = kp.B < 16 ? kp.B : 16; e >= 1; --e)  // 16 MFMA columns = 16 environments whatever N is (no 96-row limit here)
        if (roll3_lds_bytes(e, kp.N, kp.L, kp.D, S1) <= 160 * 1024) { E3 = e; break; }
    const int full = kp.B < 16 ? kp.B : 16;     // environments per workgroup that fill the 16 MFMA columns
    bool use_v3 = form == 3 ? E3 > 0 : form == 0 && E3 >= full;
    // Long agent axes: the third form with dense1 just in time and no observation rows in LDS (pw_kernels_policy3j.hpp) keeps 16
    // environments per workgroup up to N = 30 (8 beyond: rows up to 104 numbers), where the plain third form has to drop columns (N >= 13)
    // and the second form costs 400 us per step (N = 24).  us per step at B = 4096, forms 2 / 3 / 3j (profiles/r4_policy_forms.txt):
    // N = 12: 67 / 33.0 / 34.1; N = 14: 111 / 77 / 41.5; N = 16: 170 / 89 / 49.0; N = 20: 270 / 211 / 60; N = 24: 414 / 495 / 77.1;
    // N = 30: 494 / - / 100; N = 48: - / - / 342.  Automatic wherever the plain third form does not keep its 16 columns busy; policy_form 4 forces it
    // (tests run it at small N too).
    int E3j = 0, NPj = kp.N;
    bool halfj = S1C >= 9;   // the kernel's half-storage head (pw_kernels_policy3j.hpp): always beyond 64-number rows, by need at S1C = 7, 8
    if (!tag && kp.D == 4 + 2 * kp.L && kp.L <= kp.N && !a.bf16x3 && (form == 4 || (form == 0 && !use_v3))) {
        // 16 environments = the 16 MFMA columns.  Full head input (256 B per agent in LDS, the lanes' state in registers: one slot per
        // environment wave, i.e. at most 8 x (64 / N) environments) wherever that leaves room for all 16; otherwise -- rows of more than 48
        // numbers only: N >= 30 -- the half-storage head with environment slots (pw_kernels_policy3j.hpp; until round 5 such workgroups held 8).
        const int want = kp.B < 16 ? kp.B : 16;
        if (!halfj) {
            const int ecap = 8 * (kWave / kp.N) < want ? 8 * (kWave / kp.N) : want;
            for (int e = ecap; e >= 1; --e)
                if (roll3j_lds_bytes(e, kp.N, kp.L, false) <= 160 * 1024) { E3j = e; break; }
        }
        if (S1C >= 7 && E3j < want) {
            for (int e = want; e > E3j; --e)
                if (roll3j_lds_bytes(e, kp.N, kp.L, true) <= 160 * 1024) { E3j = e; halfj = true; break; }
        }
        // an odd LDS row stride (no bank conflicts between the sixteen sequences: pw_kernels_policy3j.hpp) wherever it costs no environment
        if (E3j > 0 && roll3j_lds_bytes(E3j, kp.N | 1, kp.L, halfj) <= 160 * 1024) NPj = kp.N | 1;
    }
    if (form == 4 && E3j == 0) return fail(PW_EINVAL, "policy_form 4 (just-in-time dense1) serves the local observation with L <= N");
    if (wide && (E3j == 0 || (form != 0 && form != 4)))
        return fail(PW_EINVAL, "observation rows longer than 64 numbers are served by the just-in-time rollout form only (policy_form 0 or 4)");
    if (wide) use_v3 = false;
    if (form == 0 && !use_v3 && E3j < (kp.B < 8 ? kp.B : 8) && E3 >= (kp.B < 8 ? kp.B : 8)) use_v3 = true;  // the round-3 rule
    if (!use_v3 && E3j > 0 && (form == 4 || wide || E3j >= (kp.B < 8 ? kp.B : 8))) {
        a.E = E3j;
        P.NP = NPj;
        const size_t shmj = roll3j_lds_bytes(E3j, NPj, kp.L, halfj);
        const unsigned gridj = (unsigned)((kp.B + E3j - 1) / E3j);
#define PW_R3J2(C, SK, HF)                                                                                               \
    do {                                                                                                                 \
        static unsigned long long attr_setj = 0; /* bit = device */                                                      \
        PW_LDS_OPTIN(&attr_setj, (pw_policy_rollout3j_kernel<C, SK, HF>));                                               \
        hipLaunchKernelGGL((pw_policy_rollout3j_kernel<C, SK, HF>), dim3(gridj), dim3(512), shmj, st, P);                \
    } while (0)
#define PW_R3J(C, HF) case C: if (sink) PW_R3J2(C, true, HF); else PW_R3J2(C, false, HF); break;
        if (halfj) switch (S1C) {
            PW_R3J(7, true) PW_R3J(8, true)
            PW_R3J(9, true) PW_R3J(10, true) PW_R3J(11, true) PW_R3J(12, true) PW_R3J(13, true)     // D = 65 .. 104 (N = L = 31 .. 50)
        } else switch (S1C) {
            PW_R3J(1, false) PW_R3J(2, false) PW_R3J(3, false) PW_R3J(4, false) PW_R3J(5, false) PW_R3J(6, false) PW_R3J(7, false) PW_R3J(8, false)
        }
#undef PW_R3J2
#undef PW_R3J
        PW_HIP_CHECK(hipGetLastError());
        h->last_kernel = "pw_policy_rollout3j_kernel";
        return PW_OK;
    }
    if (!use_v3)
        return fail(PW_EINVAL, a.bf16x3 ? "PW_ACTOR_BF16X3 needs the third kernel form (policy_form 0 or 3, N <= 12: 16 environments per workgroup with the observation rows in LDS)"
                                        : "the selected rollout form does not fit this configuration's observation rows in LDS (policy_form 0 chooses)");
    {
        a.E = E3;
        const size_t shm2 = roll3_lds_bytes(E3, kp.N, kp.L, kp.D, S1);
        const unsigned grid2 = (unsigned)((kp.B + E3 - 1) / E3);
#define PW_R24(C, NT, SK, BF)                                                                                            \
    do {                                                                                                                 \
        static unsigned long long attr_set3 = 0; /* bit = device */                                                      \
        PW_LDS_OPTIN(&attr_set3, (pw_policy_rollout3_kernel<C, NT, SK, BF>));                                            \
        hipLaunchKernelGGL((pw_policy_rollout3_kernel<C, NT, SK, BF>), dim3(grid2), dim3(512), shm2, st, P);             \
    } while (0)
#define PW_R23(C, NT, SK) do { if (a.bf16x3) PW_R24(C, NT, SK, true); else PW_R24(C, NT, SK, false); } while (0)
#define PW_R22(C, NT) do { if (sink) PW_R23(C, NT, true); else PW_R23(C, NT, false); } while (0)
#define PW_R2(C) case C: PW_R22(C, 0); break;
        if (kp.N == 6 && kp.L == 6) PW_R22(2, 6);        // BASELINE configs[1]: D = 16
        else switch (S1C) {
            PW_R2(1) PW_R2(2) PW_R2(3) PW_R2(4) PW_R2(5) PW_R2(6) PW_R2(7) PW_R2(8)
        }
#undef PW_R24
#undef PW_R23
#undef PW_R22
#undef PW_R2
        PW_HIP_CHECK(hipGetLastError());
        h->last_kernel = a.bf16x3 ? "pw_policy_rollout3_kernel<bf16x3>" : "pw_policy_rollout3_kernel";
        return PW_OK;
    }
}

int pw_rollout_tail(const float *rew_shared, const uint8_t *terminal, int32_t B, float *episode_return,
                    double *finished_sum, int64_t *finished_count, int64_t *counter0, int64_t delta0, int64_t modulo0,
                    int64_t *counter1, int64_t delta1, int64_t modulo1, void *stream)
{
    if (!rew_shared || !terminal || !episode_return || !finished_sum || !finished_count)
        return fail(PW_EINVAL, "null argument");
    if (B < 1) return fail(PW_EINVAL, "bad sizes");
    TailCounters tc;
    tc.c0 = counter0; tc.d0 = delta0; tc.m0 = modulo0;
    tc.c1 = counter1; tc.d1 = delta1; tc.m1 = modulo1;
    hipLaunchKernelGGL(pw_episode_stats_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), rew_shared,
                       terminal, B, episode_return, finished_sum, finished_count, tc);
    PW_HIP_CHECK(hipGetLastError());
    return PW_OK;
}

int pw_episode_stats(const float *rew_shared, const uint8_t *terminal, int32_t B, float *episode_return,
                     double *finished_sum, int64_t *finished_count, void *stream)
{
    return pw_rollout_tail(rew_shared, terminal, B, episode_return, finished_sum, finished_count, nullptr, 0, 0, nullptr, 0,
                           0, stream);
}

int pw_dense(const float *X, const float *W, const float *b, int64_t rows, int32_t in_dim, int32_t out_dim,
             int32_t relu, float *Y, void *stream)
{
    if (!X || !W || !b || !Y) return fail(PW_EINVAL, "null argument");
    if (rows < 1 || in_dim < 1 || in_dim > 64) return fail(PW_EINVAL, "in_dim must be in [1, 64]");
    if (out_dim < 64 || (out_dim & 63)) return fail(PW_EINVAL, "out_dim must be a positive multiple of 64");
    const int chunks = out_dim / 64;
    // ~2048 waves (two per SIMD) when there are enough rows, at least 8 rows per wave
    long rpw = (rows * chunks + 2047) / 2048;
    if (rpw < 8) rpw = 8;
    const long waves = ((rows + rpw - 1) / rpw) * chunks;
    const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
#define PW_DENSE_LAUNCH(k)                                                                                          \
    do {                                                                                                            \
        if (relu) hipLaunchKernelGGL((pw_dense_kernel<k, true>), grid, block, 0, st, X, W, b, (long)rows, in_dim,   \
                                     out_dim, (int)rpw, Y);                                                         \
        else hipLaunchKernelGGL((pw_dense_kernel<k, false>), grid, block, 0, st, X, W, b, (long)rows, in_dim,       \
                                out_dim, (int)rpw, Y);                                                              \
    } while (0)
    switch (in_dim) {
    case 10: PW_DENSE_LAUNCH(10); break;
    case 16: PW_DENSE_LAUNCH(16); break;
    case 22: PW_DENSE_LAUNCH(22); break;
    case 64: PW_DENSE_LAUNCH(64); break;
    default: PW_DENSE_LAUNCH(0);
    }
#undef PW_DENSE_LAUNCH
    PW_HIP_CHECK(hipGetLastError());
    return PW_OK;
}

int pw_debug_math(int32_t fn, const float *x, float aux, float *y, int64_t n, void *stream)
{
    if (!x || !y || n < 1 || fn < 0 || fn > 12) return fail(PW_EINVAL, "bad argument");
    hipLaunchKernelGGL(pw_debug_math_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), fn, x, aux, y, (long)n);
    PW_HIP_CHECK(hipGetLastError());
    return PW_OK;
}

size_t pw_actor_front_pack_floats(int32_t in_dim) { return actor_frag16_offset(((in_dim + 7) / 8) * 4) + (size_t)8 * 2 * 4 * 64 * 4; }

int pw_actor_front_pack(const float *w1, const float *w_ih, int32_t in_dim, float *frag, void *stream)
{
    if (!w1 || !w_ih || !frag) return fail(PW_EINVAL, "null argument");
    // in_dim <= 64: every consumer; 65 .. 104: the W1 image is wider (S1 = 4 * ceil(in_dim / 8) k steps) and only the just-in-time
    // rollout form (pw_policy_rollout3j_kernel: simple_spread up to N = L = 50) reads it
    if (in_dim < 1 || in_dim > 104) return fail(PW_EINVAL, "in_dim must be in [1, 104]");
    if ((reinterpret_cast<uintptr_t>(w_ih) | reinterpret_cast<uintptr_t>(frag)) & 15)
        return fail(PW_EINVAL, "w_ih and frag must be 16-byte aligned");
    const unsigned w1_threads = 2u * (unsigned)(((in_dim + 7) >> 3) * 4) * 64u;   // the W1 section; the other two take 4096 threads
    const unsigned pack_blocks = w1_threads > 4096u ? (w1_threads + 255u) / 256u : 16u;
    hipLaunchKernelGGL(pw_actor_front_pack_kernel, dim3(pack_blocks), dim3(256), 0, static_cast<hipStream_t>(stream), w1, w_ih, in_dim, frag);
    PW_HIP_CHECK(hipGetLastError());
    return PW_OK;
}

int pw_actor_front(const float *X, const float *frag, const float *b1, const float *b_ih, int64_t rows, int32_t in_dim,
                   float *G, void *stream)
{
    if (!X || !frag || !b1 || !b_ih || !G) return fail(PW_EINVAL, "null argument");
    if (rows < 1 || in_dim < 1 || in_dim > 64) return fail(PW_EINVAL, "in_dim must be in [1, 64]");
    if ((reinterpret_cast<uintptr_t>(frag) | reinterpret_cast<uintptr_t>(G)) & 15)
        return fail(PW_EINVAL, "frag and G must be 16-byte aligned");
    const int S1C = (in_dim + 7) / 8, S1 = 4 * S1C;
    const size_t shm = (size_t)8 * 2 * 4 * 64 * sizeof(float4) + (size_t)(2 * S1 * 64 + 64 + 256 + 4 * 32 * 33) * sizeof(float);
    const long tiles = (rows + 127) / 128;
    hipStream_t st = static_cast<hipStream_t>(stream);
    static unsigned long long attr_set[9] = {};  // per kernel: bit = device
#define PW_FRONT(C)                                                                                                      \
    case C:                                                                                                              \
        PW_LDS_OPTIN(&attr_set[C], (pw_actor_front_kernel<C>)); \
        hipLaunchKernelGGL(pw_actor_front_kernel<C>, dim3((unsigned)tiles), dim3(256), shm, st, X, frag, b1, b_ih,       \
                           (long)rows, in_dim, G);                                                                       \
        break;
    switch (S1C) {
        PW_FRONT(1) PW_FRONT(2) PW_FRONT(3) PW_FRONT(4) PW_FRONT(5) PW_FRONT(6) PW_FRONT(7) PW_FRONT(8)
    }
#undef PW_FRONT
    PW_HIP_CHECK(hipGetLastError());
    return PW_OK;
}

}  // extern "C"
