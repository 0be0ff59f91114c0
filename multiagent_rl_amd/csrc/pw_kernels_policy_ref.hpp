// pw_kernels_policy_ref.hpp -- part of libpworld.so (translation unit csrc/pworld_policy.hip).
// Policy-in-the-loop rollout as ONE launch for simple_reference, the MultiDiscrete scenario of the reference's sweep
// (main.py:24, :52-54: a two-head actor [5 movement | 10 communication logits]; experiments/run.py:39-41 concatenates
// the heads' one-hots into the env action).
#pragma once

namespace {

// ------------------------------------------------------------------------------------------
// T x (two-head actor forward + one Gumbel-argmax per head + environment step + auto-reset) per workgroup of 16
// environments (32 observation rows: N = 2), observations / sampled (movement, symbol) pairs / world state resident on the
// CU between steps -- the simple_reference counterpart of pw_policy_rollout_kernel.  The actor pass is actor16_forward
// (pw_kernels_actor16.hpp: the arithmetic and Philox keying of pw_actor_fused with n_out0 = 5, n_out1 = 10); the environment step is
// pw_reference_rollout_kernel's arithmetic on the workgroup's first wave (lane = (env, agent), the partner is lane ^ 1,
// every exchange a shuffle), so the results equal the loop "act = pw_actor_fused(obs); pw_step(act)" bit for bit.
// Outputs as pw_rollout's; act_out [T,B,N,2] int32 = (movement index, symbol index).
// ------------------------------------------------------------------------------------------
struct PolicyRolloutRefArgs {
    ActorFusedArgs A;   // weights, B, N = 2, D = 21, E, heads (5, dim_c), seed, step / step_dev (Philox step of the FIRST pass)
    RefParams V;        // world constants and state planes
    int T;
    int32_t *act_out;   // [T,B,N,2] (or NULL)
    float *obs, *final_obs, *rew, *rew_shared;
    uint8_t *done, *terminal;
    // optional direct sink (round 4, as the simple_spread / simple_tag rollouts): the transitions go straight into the TWO-HEAD replay
    // ring (act [cap,N,2] u8; slot (ring_start + t*B + env) % capacity) and the episode returns are kept here -- no second launch
    pw_replay_store ring;
    int has_ring;
    int64_t ring_start;
    float *episode_return;
    double *finished_sum;
    int64_t *finished_count;
    unsigned long long *scratch;
};

#ifdef PW_STAMPS
__device__ unsigned long long g_pw_ref_stamps[30];
#endif
template <int S1C, bool SINK = false>
__global__ void __launch_bounds__(512) pw_policy_rollout_ref_kernel(const PolicyRolloutRefArgs P)
{
    constexpr int DC = kDimC, N = 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const ActorFusedArgs &A = P.A;
    const RefParams &V = P.V;
    const Actor16Lds S = actor16_carve(reinterpret_cast<float *>(smem_raw), N, A.E * N, 4 * S1C);
    const int D = A.D;
    Actor16W W;  // the actor's weights: registers for the whole launch (pw_kernels_actor16.hpp)
    actor16_load<S1C>(A, S, W);
    Actor16D1<S1C> T1;  // dense1 as 16 x 16 tiles: with N = 2 the 32 x 32 form has two blocks for eight waves
    actor16_load_d1<S1C>(A, T1);
    // observation rows, TWO buffers of [96][D]: the policy reads buffer `cur`, the environment lanes publish the next rows into the other
    // one -- so that, with a ring sink, the IDLE waves can copy the rows the policy acted on into ring.obs during the environment step
    float *s_obs2 = reinterpret_cast<float *>(S.end);
    int32_t *s_act = reinterpret_cast<int32_t *>(s_obs2 + 2 * kFusedRows * D);  // [96][2] (movement, symbol)
    double *s_fs = reinterpret_cast<double *>(s_act + 2 * kFusedRows);     // [16] (+ [16] ints): finished-episode sums / counts (SINK)
    int *s_fc = reinterpret_cast<int *>(s_fs + 16);
    float *s_noise = reinterpret_cast<float *>(smem_raw) +                 // [32][4 blocks][4] Gumbel noise of the coming heads (drawn a step
                     (((int)(reinterpret_cast<float *>(s_fc + 16) - reinterpret_cast<float *>(smem_raw)) + 3) & ~3);   // ahead), 16-byte aligned

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long env0 = (long)blockIdx.x * A.E;
    const int envs_here = (int)((long)A.B - env0 < (long)A.E ? (long)A.B - env0 : (long)A.E);
    const int rows_here = envs_here * N;
    const long row_base = env0 * N;
    const size_t BN = (size_t)A.B * N;

    // ---- environment lanes: the LAST wave, lane = el * 2 + a (E <= 16 envs: 32 lanes) -- the tail of a step (reward, stores) then runs
    // beside the next actor pass's dense1 blocks and head tiles, which sit on waves 0 and 1 (actor16_forward's pre / mid windows)
    const bool env_wave = wave == 7;
    int el = lane >> 1;
    const int a = lane & 1;
    const bool live = env_wave && el < envs_here;
    if (!live) el = 0;                  // idle lanes shadow env 0 of the workgroup (same agent parity: shuffles stay paired)
    const int r = el * N + a;
    const int env = (int)(env0 + el);
    const size_t g = (size_t)env * N + a;
    RefLane<DC> s;
    int ep_step = 0;
    uint32_t ep_count = 0;
    float ep_ret = 0.f;
    double fin_sum = 0.0;
    int fin_cnt = 0;
    size_t slot = 0;
    if (env_wave) {
        if (SINK && P.episode_return && live && a == 0) ep_ret = P.episode_return[env];
        ref_load<DC>(V, env, a, s);
        ep_step = V.ep_step[env];
        ep_count = V.ep_count[env];
        float co[DC];
#pragma unroll
        for (int q = 0; q < DC; ++q) co[q] = __shfl_xor(s.c[q], 1, kWave);
        if (live) ref_write_obs<DC, false>(V, s, co, a, s_obs2 + r * D);
    }
    const uint64_t step0 = A.step_dev ? (uint64_t)*A.step_dev : A.step;
    actor16_draw_noise(A, s_noise, rows_here, row_base, step0, tid, 512);
    wg_lds_barrier();

    // the rest of a step once the agents are advanced and the next observation rows published: rewards + episode step count
    // (tail_compute), every global store (tail_stores); a wave with an episode ending in this step does everything at once
    int ai = 0, ci = 0, tail_t = 0, tail_stage = 0;
    float t_rw = 0.f, t_acc = 0.f;
    bool t_term = false;
    float co[DC];  // the other agent's symbol (this step's)
    auto tail_compute = [&]() {
        const float ox = __shfl_xor(s.px, 1, kWave), oy = __shfl_xor(s.py, 1, kWave);
        float glx, gly;
        ref_goal_landmark<DC>(s, s.goal, glx, gly);
        const float dx = ox - glx, dy = oy - gly;
        const float rw = -(dx * dx + dy * dy);
        const float r_other = __shfl_xor(rw, 1, kWave);
        t_rw = rw;
        t_acc = (0.0f + (a == 0 ? rw : r_other)) + (a == 0 ? r_other : rw);  // agent order
        ep_step += 1;
        t_term = V.max_episode_len > 0 && ep_step >= V.max_episode_len;
        if (SINK && live && a == 0 && P.episode_return) {  // run.py:55-65, per env
            const float rsum = ep_ret + t_acc;
            if (t_term) { fin_sum += (double)rsum; fin_cnt += 1; ep_ret = 0.0f; }
            else ep_ret = rsum;
        }
    };
    auto tail_stores = [&](const int t, const bool with_obs) {
        const size_t row = (size_t)t * BN + g;
        if (live) {
            if (P.act_out) { P.act_out[2 * row] = ai; P.act_out[2 * row + 1] = ci; }
            if (P.rew) P.rew[row] = t_rw;
            if (P.done) P.done[row] = 0;
            if (a == 0) {
                if (P.rew_shared) P.rew_shared[(size_t)t * A.B + env] = t_acc;
                if (P.terminal) P.terminal[(size_t)t * A.B + env] = t_term ? 1 : 0;
            }
            if (with_obs && P.obs) ref_write_obs<DC, false>(V, s, co, a, P.obs + row * D);
            if (SINK && P.has_ring) {  // next_obs is the PRE-reset observation (run.py:52 vs :60)
                ref_write_obs<DC, false>(V, s, co, a, P.ring.next_obs + (slot * N + a) * D);
                if (a == 0) { P.ring.rew[slot] = t_acc; P.ring.done[slot] = 0.0f; }
            }
        }
    };
    auto pre_hook = [&]() { if (tail_stage == 1) { tail_compute(); tail_stage = 2; } };
    auto mid_hook = [&]() { if (tail_stage == 2) { tail_stores(tail_t, true); tail_stage = 0; } };

#ifdef PW_STAMPS
    unsigned long long rs[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, r0_ = 0, r1_ = 0;
    PW_R2_START;
#define PW_REF_STAMP_ARGS , rs, &r0_
#else
#define PW_REF_STAMP_ARGS
#endif
    for (int t = 0; t < P.T; ++t) {
        float *s_obs = s_obs2 + (t & 1) * (kFusedRows * D);          // what the policy acts on in this step
        float *s_next = s_obs2 + ((t + 1) & 1) * (kFusedRows * D);   // where the environment lanes publish the next rows
        // ---- policy: observation rows (LDS) -> one sampled index per head and row (LDS)
        actor16_forward<S1C, false>(A, S, W, s_obs, D, rows_here, envs_here, row_base, step0 + (uint64_t)t, nullptr, s_act, pre_hook,
                                    mid_hook, s_noise, &T1 PW_REF_STAMP_ARGS);  // a barrier at its end
        // the noise of the NEXT step's heads: by the seven waves that wait for the environment wave
        if (t + 1 < P.T && !env_wave) actor16_draw_noise(A, s_noise, rows_here, row_base, step0 + (uint64_t)(t + 1), tid, 7 * kWave);
        if (SINK && P.has_ring && !env_wave) {
            // the observations the policy acted on -> ring.obs, by the seven waves that would otherwise wait for the environment wave
            for (int idx = tid; idx < rows_here * D; idx += 448) {
                const int rr = idx / D, c = idx - rr * D;
                const size_t sl = ring_slot(P.ring_start, t, A.B, env0 + (rr >> 1), P.ring.capacity);
                P.ring.obs[(sl * N + (rr & 1)) * D + c] = s_obs[idx];
            }
        }
        PW_R2_STAMP(7);
        // ---- environment step (pw_reference_rollout_kernel's arithmetic, index actions)
        if (env_wave) {
            const size_t row = (size_t)t * BN + g;
            ai = s_act[2 * r]; ci = s_act[2 * r + 1];
            if (SINK && P.has_ring) {  // the pair the policy sampled -> ring.act (the observation rows: the idle waves, above)
                slot = ring_slot(P.ring_start, t, A.B, (long)env, P.ring.capacity);
                if (live) {
                    P.ring.act[(slot * N + a) * 2] = (uint8_t)ai;
                    P.ring.act[(slot * N + a) * 2 + 1] = (uint8_t)ci;
                }
            }
            const float a1 = ai == 1, a2 = ai == 2, a3 = ai == 3, a4 = ai == 4;
            float cn[DC];
#pragma unroll
            for (int q = 0; q < DC; ++q) cn[q] = q == ci ? 1.0f : 0.0f;
            {
                float ux = 0.0f + (a1 - a2), uy = 0.0f + (a3 - a4);
                ux *= V.sens; uy *= V.sens;
                const float fx = ux + 0.0f, fy = uy + 0.0f;
                s.vx = s.vx * V.damp; s.vy = s.vy * V.damp;
                s.vx = s.vx + (fx / V.mass) * V.dt;
                s.vy = s.vy + (fy / V.mass) * V.dt;
                s.px = s.px + s.vx * V.dt;
                s.py = s.py + s.vy * V.dt;
            }
#pragma unroll
            for (int q = 0; q < DC; ++q) s.c[q] = cn[q] + 0.0f;
#pragma unroll
            for (int q = 0; q < DC; ++q) co[q] = __shfl_xor(s.c[q], 1, kWave);
            const bool ends = V.auto_reset && V.max_episode_len > 0 && ep_step + 1 >= V.max_episode_len;
            if (__any(ends)) {
                tail_compute();
                tail_stores(t, false);
                if (t_term && V.auto_reset) {
                    if (live && P.final_obs) ref_write_obs<DC, false>(V, s, co, a, P.final_obs + row * D);
                    ep_count += 1;
                    ep_step = 0;
                    ref_reset<DC, false>(V, V.env_id_base + (uint64_t)env, ep_count, a, s);
#pragma unroll
                    for (int q = 0; q < DC; ++q) co[q] = 0.f;  // the other agent reset too
                }
                if (live) {
                    if (P.obs) ref_write_obs<DC, false>(V, s, co, a, P.obs + row * D);
                    ref_write_obs<DC, false>(V, s, co, a, s_next + r * D);
                }
            } else {
                if (live) ref_write_obs<DC, false>(V, s, co, a, s_next + r * D);
                tail_stage = 1;
                tail_t = t;
            }
        }
        PW_R2_STAMP(8);
        wg_lds_barrier();
        PW_R2_STAMP(9);
    }
#ifdef PW_STAMPS
    if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 7 || wave == 3))
        for (int i_ = 0; i_ < 10; ++i_) g_pw_ref_stamps[(wave == 0 ? 0 : wave == 7 ? 1 : 2) * 10 + i_] = rs[i_];
#endif
    if (tail_stage == 1) tail_compute();
    if (tail_stage != 0) tail_stores(tail_t, true);
    if (live) {
        ref_store<DC>(V, env, a, s);
        if (a == 0) {
            V.ep_step[env] = ep_step; V.ep_count[env] = ep_count;
            if (SINK && P.episode_return) P.episode_return[env] = ep_ret;
        }
    }
    if (SINK && P.episode_return) {  // finished-episode statistics: per-workgroup partials, the last workgroup adds them up in order
        wg_lds_barrier();
        if (live && a == 0) { s_fs[el] = fin_sum; s_fc[el] = fin_cnt; }
        wg_lds_barrier();
        rollout_finish_stats(envs_here, s_fs, s_fc, P.scratch, P.finished_sum, P.finished_count, smem_raw);
    }
}

}  // namespace
