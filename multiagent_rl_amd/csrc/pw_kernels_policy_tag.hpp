// pw_kernels_policy_tag.hpp -- part of libpworld.so (one translation unit: csrc/pworld.hip includes it).
// Policy-in-the-loop rollout for simple_tag (BASELINE configs[2]): pw_policy_rollout_kernel's structure (actor pass
// of the whole workgroup, then the first waves advance the environments) with pw_tag_stream_kernel's arithmetic.
#pragma once

namespace {

struct PolicyRolloutTagArgs {
    ActorFusedArgs A;   // weights, B, N, D, E, heads, seed, step / step_dev (Philox step of the FIRST pass)
    TagParams V;        // world constants (2x2 class tables), state planes, outputs (V.act unused)
    int T;
    int32_t *act_out;   // [T,B,N] sampled action indices (or NULL)
    pw_replay_store ring;
    int has_ring;
    int64_t ring_start;
    float *episode_return;
    double *finished_sum;
    int64_t *finished_count;
    unsigned long long *scratch;
};

__host__ __device__ inline size_t policy_tag_lds_bytes(int S1, int D, int E, int L, int N)
{
    return actor16_lds_floats(N, E * N, S1) * sizeof(float) + (size_t)kFusedRows * D * sizeof(float) + kFusedRows * sizeof(int32_t) +
           4 * kWave * sizeof(float2) + (size_t)E * L * sizeof(float2) + 3 * 2 * kWave * sizeof(float) +
           16 * (sizeof(double) + sizeof(int)) + (size_t)actor16_noise_floats(kFusedRows, 5) * sizeof(float);
}

// One 5-logit head per agent: every agent of simple_tag takes the same five movement actions; the observation rows
// of the good agents are zero-padded to the adversaries' width D (as every simple_tag kernel here writes them).
#ifdef PW_STAMPS
__device__ unsigned long long g_pw_tag_stamps[30];
#endif
template <int S1C, bool SINK>
__global__ void __launch_bounds__(512) pw_policy_rollout_tag_kernel(const PolicyRolloutTagArgs P)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const ActorFusedArgs &A = P.A;
    const TagParams &V = P.V;
    const int N = A.N, L = V.L, D = A.D, NA = V.A;
    const Actor16Lds S = actor16_carve(reinterpret_cast<float *>(smem_raw), N, A.E * N, 4 * S1C);
    Actor16W W;  // the actor's weights: registers for the whole launch (pw_kernels_actor16.hpp)
    actor16_load<S1C>(A, S, W);
    float *s_obs = reinterpret_cast<float *>(S.end);                       // [96][D] observation rows
    int32_t *s_act = reinterpret_cast<int32_t *>(s_obs + kFusedRows * D);  // [96]
    float2 *s_posb = reinterpret_cast<float2 *>(s_act + kFusedRows);       // [2 env waves][64]
    float2 *s_velb = s_posb + 2 * kWave;                                   // [2][64]
    float2 *s_lmb = s_velb + 2 * kWave;                                    // [E * L]
    uint32_t *s_mlob = reinterpret_cast<uint32_t *>(s_lmb + A.E * L);      // [2][64] collision masks, low / high words
    uint32_t *s_mhib = s_mlob + 2 * kWave;                                 // (unused since the masks are 32 bits wide; keeps the carve-up)
    float *s_rewb = reinterpret_cast<float *>(s_mhib + 2 * kWave);         // [2][64]
    double *s_fs = reinterpret_cast<double *>(s_rewb + 2 * kWave);         // [16] (+ [16] ints)
    int *s_fc = reinterpret_cast<int *>(s_fs + 16);
    float *s_noise = reinterpret_cast<float *>(smem_raw) +                 // [96][2 blocks][4] Gumbel noise of the coming head (drawn a step
                     (((int)(reinterpret_cast<float *>(s_fc + 16) - reinterpret_cast<float *>(smem_raw)) + 3) & ~3);   // ahead), 16-byte aligned

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long env0 = (long)blockIdx.x * A.E;
    const int envs_here = (int)((long)A.B - env0 < (long)A.E ? (long)A.B - env0 : (long)A.E);
    const int rows_here = envs_here * N;
    const long row_base = env0 * N;
    const size_t BN = (size_t)A.B * N;

    const int epw_max = A.E < kWave / N ? A.E : kWave / N;
    const int waves_full = (A.E + epw_max - 1) / epw_max;
    const int epw = (A.E + waves_full - 1) / waves_full;
    const int n_env_waves = (envs_here + epw - 1) / epw;
    // the LAST waves: the tail of an environment step (partner pass, rewards, stores) then runs beside the next actor pass's
    // dense1 blocks and head tiles, which are dealt from wave 0 up (actor16_forward's pre / mid windows)
    const int ewi = wave - (8 - n_env_waves);
    const bool env_wave = ewi >= 0;
    int e_loc = lane / N, a = lane - e_loc * N;
    int el = ewi * epw + e_loc;
    const bool live = env_wave && e_loc < epw && el < envs_here;
    if (!live) { e_loc = 0; a = 0; el = env_wave ? ewi * epw : 0; }  // idle lanes shadow lane 0, store nothing
    const int base = e_loc * N, me = base + a, r_lane = el * N + a, r = r_lane;
    const long env = env0 + el;
    const uint32_t g_lane = (uint32_t)env * (uint32_t)N + (uint32_t)a, g = g_lane;
    const int ew = env_wave ? ewi : 0;
    float2 *s_pos = s_posb + ew * kWave, *s_vel = s_velb + ew * kWave;
    uint32_t *s_mlo = s_mlob + ew * kWave;
    float *s_rew = s_rewb + ew * kWave;
    const float2 *pp = s_pos + base, *vv = s_vel + base;
    float2 *lmv = s_lmb + el * L;
    const int cls = a >= NA ? 1 : 0;
    // Register diet (this kernel sat at the 256-register cap of a 512-thread workgroup and spilled 4 .. 10 VGPRs to scratch): the
    // masks are 32 bits wide (the host admits D = 4 + 2L + 2(N - 1) + 2G <= 48 here: N <= 22, L <= 21), the finished-episode sums live
    // in LDS (touched once per episode), the ring slot and the Philox env id are recomputed where they are used.
    const uint32_t adv_bits = (1u << NA) - 1u;
    const float my_sens = V.sens[cls], my_fscale = V.fscale[cls], my_maxspeed = V.max_speed[cls];
    const float dmin_adv = V.dist_min[cls][0], dmin_good = V.dist_min[cls][1], dmin_lm = V.dist_min_lm[cls];
    const float cthr_adv = V.coll_thr2[cls][0], cthr_good = V.coll_thr2[cls][1];
    const float nthr_adv = V.near_thr2[cls][0], nthr_good = V.near_thr2[cls][1], nthr_lm = V.near_thr2_lm[cls];

    float px = 0.f, py = 0.f, vx = 0.f, vy = 0.f;
    int ep_step = 0;
    uint32_t ep_count = 0;
    uint32_t coll = 0, near_a = 0, near_l = 0;
    float ep_ret = 0.f;
    // BASELINE configs[2]'s roster (4 adversaries + 2 good agents, 2 landmarks) takes the unrolled row writer / partner loops: with
    // runtime bounds every LDS read of a loop is a round trip of its own (~120 exposed cycles each, ~10 per row)
    const bool c3 = N == 6 && NA == 4 && L == 2;
    auto write_row = [&](float *dst) {
        if (c3) tag_write_obs<6, 4, 2>(dst, N, NA, L, D, a, lmv, pp, vv, px, py, vx, vy);
        else tag_write_obs<0, -1, 0>(dst, N, NA, L, D, a, lmv, pp, vv, px, py, vx, vy);
    };
    auto partner_pass = [&]() {
        coll = 0; near_a = 0; near_l = 0;
        auto agent = [&](const float2 q, const int j) {
            const float dx = q.x - px, dy = q.y - py;
            const float d2 = dx * dx + dy * dy;
            const bool jg = j >= NA;
            if (d2 < (jg ? cthr_good : cthr_adv)) coll |= 1u << j;
            if (bits_near(d2, jg ? nthr_good : nthr_adv)) near_a |= 1u << j;
        };
        auto landmark = [&](const float2 q, const int l) {
            const float dx = q.x - px, dy = q.y - py;
            if (bits_near(dx * dx + dy * dy, nthr_lm)) near_l |= 1u << l;
        };
        if (c3) {  // all eight LDS reads in flight
            const float2 q0 = pp[0], q1 = pp[1], q2 = pp[2], q3 = pp[3], q4 = pp[4], q5 = pp[5], l0 = lmv[0], l1 = lmv[1];
            agent(q0, 0); agent(q1, 1); agent(q2, 2); agent(q3, 3); agent(q4, 4); agent(q5, 5);
            landmark(l0, 0); landmark(l1, 1);
        } else {
            for (int j = 0; j < N; ++j) agent(pp[j], j);
            for (int l = 0; l < L; ++l) landmark(lmv[l], l);
        }
        near_a &= ~(1u << a);
    };
    if (SINK && tid < 16) { s_fs[tid] = 0.0; s_fc[tid] = 0; }   // per-env finished-episode (sum, count) of this launch
    if (env_wave) {
        if (SINK && P.episode_return && live && a == 0) ep_ret = P.episode_return[env];
        px = V.pos_x[g]; py = V.pos_y[g]; vx = V.vel_x[g]; vy = V.vel_y[g];
        ep_step = V.ep_step[env];
        ep_count = V.ep_count[env];
        if (live)
            for (int l = a; l < L; l += N) lmv[l] = make_float2(V.lm_x[(size_t)env * L + l], V.lm_y[(size_t)env * L + l]);
        if (live) { s_pos[me] = make_float2(px, py); s_vel[me] = make_float2(vx, vy); }
        wave_lds_sync();
        partner_pass();
        if (live) write_row(s_obs + r * D);
    }
    const float k = V.contact_margin, cf = V.contact_force, dt = V.dt, damp = V.damp, mass = V.mass;
    const uint64_t step0 = A.step_dev ? (uint64_t)*A.step_dev : A.step;
    // the Gumbel noise of step t + 1 is drawn by the waves without environment duty while the environment waves advance step t (by
    // everybody, first, when every wave has environment duty)
    const int noise_thr = n_env_waves < 8 ? (8 - n_env_waves) * kWave : 512;
    actor16_draw_noise(A, s_noise, rows_here, row_base, step0, tid, 512);
    wg_lds_barrier();

    // The rest of an environment step once the agents are advanced and the next observation rows published, in two pieces:
    //   tail_compute  partner pass on the new positions (the next step's forces need its masks), rewards, episode bookkeeping
    //   tail_stores   every global store of the step
    // run by the environment waves inside the NEXT actor pass (before its dense1 blocks / before its head tiles).  A wave with an
    // episode ending in this step runs both, the reset and the post-reset partner pass before it publishes the rows.
    int ai = 0, tail_t = 0, tail_stage = 0;  // tail_stage: 0 nothing pending, 1 tail_compute pending, 2 tail_stores pending
    float t_rw = 0.f, t_acc = 0.f;
    bool t_term = false;
    auto tail_compute = [&]() {
        partner_pass();
        // simple_tag.reward
        if (live) s_mlo[me] = coll;
        wave_lds_sync();
        float rw = 0.0f;
        if (cls) {
            for (int q = 0; q < NA; ++q)
                if ((coll >> q) & 1) rw -= 10.0f;
            rw -= tag_bound(fabsf(px));
            rw -= tag_bound(fabsf(py));
        } else if (c3) {  // both good agents' masks in flight (N = 6: the low words hold everything)
            const uint32_t m4 = s_mlo[base + 4], m5 = s_mlo[base + 5];
            rw += 10.0f * (float)__builtin_popcount(m4 & adv_bits);
            rw += 10.0f * (float)__builtin_popcount(m5 & adv_bits);
        } else {
            for (int gj = NA; gj < N; ++gj) rw += 10.0f * (float)__builtin_popcount(s_mlo[base + gj] & adv_bits);
        }
        if (live) s_rew[me] = rw;
        wave_lds_sync();
        float acc = 0.0f;
        if (c3) {  // the six rewards in flight, added in agent order
            const float r0 = s_rew[base], r1 = s_rew[base + 1], r2 = s_rew[base + 2], r3 = s_rew[base + 3], r4 = s_rew[base + 4],
                        r5 = s_rew[base + 5];
            acc += r0; acc += r1; acc += r2; acc += r3; acc += r4; acc += r5;
        } else {
            for (int i = 0; i < N; ++i) acc += s_rew[base + i];
        }
        ep_step += 1;
        t_term = V.max_episode_len > 0 && ep_step >= V.max_episode_len;
        t_rw = rw;
        t_acc = acc;
        if (SINK && live && a == 0 && P.episode_return) {
            const float rsum = ep_ret + acc;
            if (t_term) { s_fs[el] += (double)rsum; s_fc[el] += 1; ep_ret = 0.0f; }   // once per episode: LDS, not registers
            else ep_ret = rsum;
        }
    };
    // `opaque`: the lane's launch-invariant indices (g, r) pass through an empty asm wherever they feed an output address.  Without it
    // the compiler computes every `pointer + g` of the step's stores once per launch and keeps the 64-bit results in registers for
    // the whole kernel -- that is what pushed this kernel over the 256-register cap and into scratch memory; a few address
    // additions per step are cheaper than three scratch reloads.
    auto opaque = [](const uint32_t v) { uint32_t x = v; asm volatile("" : "+v"(x)); return x; };
    auto tail_stores = [&](const int t, const bool with_obs) {  // with_obs: V.obs too (no reset in between: the same row)
        const size_t tBN = (size_t)t * BN;
        const uint32_t g = opaque(g_lane);
        const int r = (int)opaque((uint32_t)r_lane);
        if (live) {
            if (P.act_out) P.act_out[tBN + g] = ai;
            if (V.rew) V.rew[tBN + g] = t_rw;
            if (V.done) V.done[tBN + g] = 0;
            if (a == 0) {
                if (V.rew_shared) V.rew_shared[(size_t)t * A.B + env] = t_acc;
                if (V.terminal) V.terminal[(size_t)t * A.B + env] = t_term ? 1 : 0;
            }
            // with_obs: no reset in between, so next_obs and obs are both the row this lane published in LDS -- copied, not rebuilt
            const float2 *row = reinterpret_cast<const float2 *>(s_obs + r * D);
            if (SINK && P.has_ring) {  // next_obs is the PRE-reset observation (run.py:52 vs :60)
                const size_t slot = ring_slot(P.ring_start, t, A.B, (long)env, P.ring.capacity);
                float *dst = P.ring.next_obs + (slot * N + a) * D;
                if (P.ring.state_rows) sink_state_next(P.ring, slot, N, a, px, py, vx, vy);
                else if (with_obs) for (int c = 0; c < D / 2; ++c) reinterpret_cast<float2 *>(dst)[c] = row[c];
                else write_row(dst);
                if (a == 0) { P.ring.rew[slot] = t_acc; P.ring.done[slot] = 0.0f; }
            }
            if (with_obs && V.obs) {
                float2 *dst = reinterpret_cast<float2 *>(V.obs + (tBN + g) * D);
                for (int c = 0; c < D / 2; ++c) dst[c] = row[c];
            }
        }
    };
    auto pre_hook = [&]() { if (tail_stage == 1) { tail_compute(); tail_stage = 2; } };
    auto mid_hook = [&]() { if (tail_stage == 2) { tail_stores(tail_t, true); tail_stage = 0; } };

#ifdef PW_STAMPS
    unsigned long long rs[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, r0_ = 0, r1_ = 0;
    PW_R2_START;
#define PW_TAG_STAMP_ARGS , rs, &r0_
#else
#define PW_TAG_STAMP_ARGS
#endif
    for (int t = 0; t < P.T; ++t) {
        actor16_forward<S1C, false>(A, S, W, s_obs, D, rows_here, envs_here, row_base, step0 + (uint64_t)t, nullptr, s_act, pre_hook,
                                    mid_hook, s_noise, nullptr PW_TAG_STAMP_ARGS);  // a barrier at its end
        if (t + 1 < P.T && tid < noise_thr) actor16_draw_noise(A, s_noise, rows_here, row_base, step0 + (uint64_t)(t + 1), tid, noise_thr);
        PW_R2_STAMP(7);
        if (env_wave) {
            const size_t tBN = (size_t)t * BN;
            const uint32_t g = opaque(g_lane);
            const int r = (int)opaque((uint32_t)r_lane);
            ai = s_act[r];
            if (SINK && P.has_ring) {  // the observation the policy acted on (still in LDS) -> ring.obs
                const size_t slot = ring_slot(P.ring_start, t, A.B, (long)env, P.ring.capacity);
                if (live) {
                    if (P.ring.state_rows) {
                        sink_state_obs(P.ring, slot, N, a, L, lmv, px, py, vx, vy);
                    } else {
                        const float2 *src = reinterpret_cast<const float2 *>(s_obs + r * D);
                        float2 *dst = reinterpret_cast<float2 *>(P.ring.obs + (slot * N + a) * D);
                        for (int c = 0; c < D / 2; ++c) dst[c] = src[c];
                    }
                    P.ring.act[slot * N + a] = (uint8_t)ai;
                }
            }
            float ux = 0.0f + ((ai == 1 ? 1.0f : 0.0f) - (ai == 2 ? 1.0f : 0.0f));
            float uy = 0.0f + ((ai == 3 ? 1.0f : 0.0f) - (ai == 4 ? 1.0f : 0.0f));
            ux *= my_sens; uy *= my_sens;
            if (my_fscale != 1.0f) { ux = my_fscale * ux; uy = my_fscale * uy; }
            float fx = ux + 0.0f, fy = uy + 0.0f;
            // U5: near agents (ascending j), then near landmarks (ascending l): upstream's entity order
            for (uint32_t m = live ? near_a : 0; m; m &= m - 1) {
                const int j = __builtin_ctz(m);
                const float2 q = pp[j];
                collision_force<true>(px, py, q.x, q.y, j >= NA ? dmin_good : dmin_adv, k, cf, fx, fy);
            }
            for (uint32_t m = live ? near_l : 0; m; m &= m - 1) {
                const float2 q = lmv[__builtin_ctz(m)];
                collision_force<true>(px, py, q.x, q.y, dmin_lm, k, cf, fx, fy);
            }
            // U6 with the max_speed clamp
            vx = vx * damp; vy = vy * damp;
            vx = vx + (fx / mass) * dt;
            vy = vy + (fy / mass) * dt;
            if (my_maxspeed >= 0.0f) {
                const float speed = sqrtf(vx * vx + vy * vy);
                if (speed > my_maxspeed) {
                    vx = vx / speed * my_maxspeed;
                    vy = vy / speed * my_maxspeed;
                }
            }
            px = px + vx * dt;
            py = py + vy * dt;
            wave_lds_sync();
            if (live) { s_pos[me] = make_float2(px, py); s_vel[me] = make_float2(vx, vy); }
            wave_lds_sync();
            const bool ends = V.auto_reset && V.max_episode_len > 0 && ep_step + 1 >= V.max_episode_len;
            if (__any(ends)) {
                tail_compute();
                tail_stores(t, false);
                const bool term = t_term;
                if (term && V.auto_reset) {  // same for every lane of an env
                    if (live && V.final_obs)
                        write_row(V.final_obs + (tBN + g) * D);
                    wave_lds_sync();
                    ep_count += 1;
                    ep_step = 0;
                    const uint64_t env_id = V.env_id_base + (uint64_t)env;
                    pw_reset_xy(V.seed, env_id, ep_count, (uint32_t)a, -1.0f, 1.0f, &px, &py);
                    vx = 0.f; vy = 0.f;
                    if (live)
                        for (int l = a; l < L; l += N) {
                            float x, y;
                            pw_reset_xy(V.seed, env_id, ep_count, (uint32_t)(N + l), -0.9f, 0.9f, &x, &y);
                            lmv[l] = make_float2(x, y);
                        }
                    if (live) { s_pos[me] = make_float2(px, py); s_vel[me] = make_float2(0.f, 0.f); }
                }
                wave_lds_sync();
                if (__any(term && V.auto_reset)) partner_pass();
                if (live) {
                    if (V.obs) write_row(V.obs + (tBN + g) * D);
                    write_row(s_obs + r * D);
                }
            } else {
                if (live) write_row(s_obs + r * D);
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
        for (int i_ = 0; i_ < 10; ++i_) g_pw_tag_stamps[(wave == 0 ? 0 : wave == 7 ? 1 : 2) * 10 + i_] = rs[i_];
#endif
    if (tail_stage == 1) tail_compute();
    if (tail_stage != 0) tail_stores(tail_t, true);

    if (live) {
        V.pos_x[g] = px; V.pos_y[g] = py;
        V.vel_x[g] = vx; V.vel_y[g] = vy;
        for (int l = a; l < L; l += N) {
            const float2 q = lmv[l];
            V.lm_x[(size_t)env * L + l] = q.x;
            V.lm_y[(size_t)env * L + l] = q.y;
        }
        if (a == 0) {
            V.ep_step[env] = ep_step;
            V.ep_count[env] = ep_count;
            if (SINK && P.episode_return) P.episode_return[env] = ep_ret;
        }
    }
    if (SINK && P.episode_return) {
        wg_lds_barrier();
        rollout_finish_stats(envs_here, s_fs, s_fc, P.scratch, P.finished_sum, P.finished_count, smem_raw);
    }
}

}  // namespace
