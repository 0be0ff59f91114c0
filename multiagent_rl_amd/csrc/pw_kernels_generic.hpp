// pw_kernels_generic.hpp -- part of libpworld.so (one translation unit: csrc/pworld.hip includes it).
// Generic kernels: pw_rollout_kernel (any scenario), pw_aux_kernel (reset/observe/reward), AoS<->SoA state copies.
#pragma once

namespace {

// ------------------------------------------------------------------------------------------
// MultiAgentEnv.step for T consecutive steps.
// ------------------------------------------------------------------------------------------
template <int SCEN, int OBS>
__global__ void __launch_bounds__(kWave) pw_rollout_kernel(const KParams P, const pw_step_io io, const int T)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const Smem S = carve(P, smem_raw);
    const Lane ln = make_lane(P);
    const int N = P.N, L = P.L, D = P.D;
    const size_t BN = (size_t)P.B * N;

    float px = 0.f, py = 0.f, vx = 0.f, vy = 0.f;
    int ep_step = 0;
    uint32_t ep_count = 0;
    float my_size = 0.f, my_sens = 0.f, my_fscale = 1.f, my_maxspeed = -1.f;
    if (ln.valid) {
        px = P.pos_x[ln.g]; py = P.pos_y[ln.g];
        vx = P.vel_x[ln.g]; vy = P.vel_y[ln.g];
        ep_step = P.ep_step[ln.env];
        ep_count = P.ep_count[ln.env];
        my_size = P.agent_size[ln.a];
        my_sens = P.agent_sens[ln.a];
        my_fscale = P.agent_fscale[ln.a];
        my_maxspeed = P.agent_max_speed[ln.a];
        for (int l = ln.a; l < L; l += N)
            S.lm[ln.e_local * L + l] = make_float2(P.lm_x[(size_t)ln.env * L + l], P.lm_y[(size_t)ln.env * L + l]);
        S.pos[threadIdx.x] = make_float2(px, py);
        if (SCEN == PW_SIMPLE_TAG) S.vel[threadIdx.x] = make_float2(vx, vy);
    }
    wave_lds_sync();

    const float k = P.contact_margin, cf = P.contact_force, dt = P.dt, damp = P.damp, mass = P.mass;
    const float near_margin = 88.5f * k;

    for (int t = 0; t < T; ++t) {
        const size_t row = (size_t)t * BN + ln.g;  // flattened [t, env, agent]
        // ---- U2 _set_action + U4 apply_action_force
        float fx = 0.f, fy = 0.f;
        if (ln.valid) {
            float ux, uy;
            if (io.act_idx) {
                const int a = io.act_idx[row];
                ux = 0.0f + ((a == 1 ? 1.0f : 0.0f) - (a == 2 ? 1.0f : 0.0f));
                uy = 0.0f + ((a == 3 ? 1.0f : 0.0f) - (a == 4 ? 1.0f : 0.0f));
            } else {
                const float *av = io.act_vec + row * 5;
                float a0 = av[0], a1 = av[1], a2 = av[2], a3 = av[3], a4 = av[4];
                if (P.force_discrete) {  // np.argmax: first maximum wins
                    int d = 0;
                    float best = a0;
                    if (a1 > best) { best = a1; d = 1; }
                    if (a2 > best) { best = a2; d = 2; }
                    if (a3 > best) { best = a3; d = 3; }
                    if (a4 > best) { best = a4; d = 4; }
                    a1 = d == 1; a2 = d == 2; a3 = d == 3; a4 = d == 4;
                }
                ux = 0.0f + (a1 - a2);
                uy = 0.0f + (a3 - a4);
            }
            ux *= my_sens; uy *= my_sens;
            if (my_fscale != 1.0f) { ux = my_fscale * ux; uy = my_fscale * uy; }
            fx = ux + 0.0f; fy = uy + 0.0f;
            // ---- U5 apply_environment_force: entities j ascending (agents, then landmarks).
            // First a cheap pass marks the partners whose force can be non-zero (beyond
            // dist_min + 88.5 k the softplus is exactly 0, see the fast path's note 1), then only
            // those are evaluated -- in the same ascending order, so the sums keep their bits.
            const float2 *pp = S.pos + ln.base;
            const float2 *lm = S.lm + ln.e_local * L;
            uint64_t near_a = 0, near_l = 0;
            for (int j = 0; j < N; ++j) {
                const float2 q = pp[j];
                const float dx = px - q.x, dy = py - q.y;
                if (j != ln.a && !provably_far(dx * dx + dy * dy, my_size + P.agent_size[j], near_margin))
                    near_a |= 1ull << j;
            }
            if (P.landmark_collide) {
                for (int l = 0; l < L; ++l) {
                    const float2 q = lm[l];
                    const float dx = px - q.x, dy = py - q.y;
                    if (!provably_far(dx * dx + dy * dy, my_size + P.landmark_size, near_margin)) near_l |= 1ull << l;
                }
            }
            for (uint64_t m = near_a; m; m &= m - 1) {
                const int j = __builtin_ctzll(m);
                const float2 q = pp[j];
                // dist_min = size_a + size_b is commutative, so either pair order gives the same bits
                collision_force(px, py, q.x, q.y, my_size + P.agent_size[j], k, cf, fx, fy);
            }
            for (uint64_t m = near_l; m; m &= m - 1) {
                const float2 q = lm[__builtin_ctzll(m)];
                collision_force(px, py, q.x, q.y, my_size + P.landmark_size, k, cf, fx, fy);
            }
            // ---- U6 integrate_state
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
        }
        wave_lds_sync();  // every lane has read the old positions
        if (ln.valid) {
            S.pos[threadIdx.x] = make_float2(px, py);
            if (SCEN == PW_SIMPLE_TAG) S.vel[threadIdx.x] = make_float2(vx, vy);
        }
        wave_lds_sync();

        // ---- reward / masks from the new state
        uint64_t mask = 0;
        float r = reward_and_mask<SCEN>(P, ln, px, py, my_size, S.pos, S.lm, S.red, mask);
        if (ln.valid) {
            if (io.rew) io.rew[row] = r;
            if (io.done) io.done[row] = 0;
            if (io.coll) io.coll[row] = mask;
        }
        if (io.rew_shared) {  // np.sum(rew_n), run.py:46, in agent order
            float acc = 0.0f;
            for (int i = 0; i < N; ++i) acc += __shfl(r, ln.base + i, kWave);
            if (ln.valid && ln.a == 0) io.rew_shared[(size_t)t * P.B + ln.env] = acc;
        }
        // ---- terminal rule (run.py:48-50) and auto-reset (run.py:59-60)
        ep_step += 1;
        const bool term = P.max_episode_len > 0 && ep_step >= P.max_episode_len;
        if (ln.valid && ln.a == 0 && io.terminal) io.terminal[(size_t)t * P.B + ln.env] = term ? 1 : 0;
        const bool do_reset = ln.valid && term && P.auto_reset;
        if (__any(do_reset)) {
            if (do_reset && io.final_obs)
                write_obs<SCEN, OBS>(P, ln, io.final_obs + row * D, px, py, vx, vy, S.pos, S.vel, S.lm);
            wave_lds_sync();
            if (do_reset) {
                ep_count += 1;
                ep_step = 0;
                reset_lane(P, ln, ep_count, SCEN, px, py, S.lm);
                vx = 0.f; vy = 0.f;
                S.pos[threadIdx.x] = make_float2(px, py);
                if (SCEN == PW_SIMPLE_TAG) S.vel[threadIdx.x] = make_float2(0.f, 0.f);
            }
            wave_lds_sync();
        }
        if (ln.valid && io.obs)
            write_obs<SCEN, OBS>(P, ln, io.obs + row * D, px, py, vx, vy, S.pos, S.vel, S.lm);
    }

    if (ln.valid) {
        P.pos_x[ln.g] = px; P.pos_y[ln.g] = py;
        P.vel_x[ln.g] = vx; P.vel_y[ln.g] = vy;
        for (int l = ln.a; l < L; l += N) {
            const float2 q = S.lm[ln.e_local * L + l];
            P.lm_x[(size_t)ln.env * L + l] = q.x;
            P.lm_y[(size_t)ln.env * L + l] = q.y;
        }
        if (ln.a == 0) {
            P.ep_step[ln.env] = ep_step;
            P.ep_count[ln.env] = ep_count;
        }
    }
}

// ------------------------------------------------------------------------------------------
// reset / observe / reward from the stored state (no physics).  mode bit 0: reset masked envs,
// bit 1: write obs, bit 2: write reward/coll.
// ------------------------------------------------------------------------------------------
template <int SCEN, int OBS>
__global__ void __launch_bounds__(kWave) pw_aux_kernel(const KParams P, const int mode, const uint8_t *env_mask,
                                                       float *obs, float *rew, uint64_t *coll)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const Smem S = carve(P, smem_raw);
    const Lane ln = make_lane(P);
    const int N = P.N, L = P.L;
    float px = 0.f, py = 0.f, vx = 0.f, vy = 0.f, my_size = 0.f;
    if (ln.valid) {
        my_size = P.agent_size[ln.a];
        const bool rs = (mode & 1) && (!env_mask || env_mask[ln.env]);
        if (rs) {
            const uint32_t ep = P.ep_count[ln.env] + 1;
            reset_lane(P, ln, ep, SCEN, px, py, S.lm);
        } else {
            px = P.pos_x[ln.g]; py = P.pos_y[ln.g];
            vx = P.vel_x[ln.g]; vy = P.vel_y[ln.g];
            for (int l = ln.a; l < L; l += N)
                S.lm[ln.e_local * L + l] = make_float2(P.lm_x[(size_t)ln.env * L + l], P.lm_y[(size_t)ln.env * L + l]);
        }
        S.pos[threadIdx.x] = make_float2(px, py);
        if (SCEN == PW_SIMPLE_TAG) S.vel[threadIdx.x] = make_float2(vx, vy);
    }
    wave_lds_sync();
    if (ln.valid && (mode & 1) && (!env_mask || env_mask[ln.env])) {
        P.pos_x[ln.g] = px; P.pos_y[ln.g] = py;
        P.vel_x[ln.g] = 0.f; P.vel_y[ln.g] = 0.f;
        for (int l = ln.a; l < L; l += N) {
            const float2 q = S.lm[ln.e_local * L + l];
            P.lm_x[(size_t)ln.env * L + l] = q.x;
            P.lm_y[(size_t)ln.env * L + l] = q.y;
        }
    }
    if (mode & 4) {
        uint64_t mask = 0;
        const float r = reward_and_mask<SCEN>(P, ln, px, py, my_size, S.pos, S.lm, S.red, mask);
        if (ln.valid) {
            if (rew) rew[ln.g] = r;
            if (coll) coll[ln.g] = mask;
        }
    }
    if ((mode & 2) && ln.valid && obs)
        write_obs<SCEN, OBS>(P, ln, obs + ln.g * P.D, px, py, vx, vy, S.pos, S.vel, S.lm);
    // counters last: every lane of the env has read ep_count above (same wave, program order)
    wave_lds_sync();
    if (ln.valid && (mode & 1) && ln.a == 0 && (!env_mask || env_mask[ln.env])) {
        P.ep_count[ln.env] += 1;
        P.ep_step[ln.env] = 0;
    }
}

// AoS [B,N,2] <-> SoA planes
__global__ void pw_scatter_state_kernel(const KParams P, const float *pos, const float *vel, const float *lm,
                                        const int32_t *ep_step, const uint32_t *ep_count)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t BN = (size_t)P.B * P.N, BL = (size_t)P.B * P.L;
    if (i < BN) {
        if (pos) { P.pos_x[i] = pos[2 * i]; P.pos_y[i] = pos[2 * i + 1]; }
        if (vel) { P.vel_x[i] = vel[2 * i]; P.vel_y[i] = vel[2 * i + 1]; }
    }
    if (i < BL && lm) { P.lm_x[i] = lm[2 * i]; P.lm_y[i] = lm[2 * i + 1]; }
    if (i < (size_t)P.B) {
        P.ep_step[i] = ep_step ? ep_step[i] : 0;
        P.ep_count[i] = ep_count ? ep_count[i] : 0;
    }
}

__global__ void pw_gather_state_kernel(const KParams P, float *pos, float *vel, float *lm,
                                       int32_t *ep_step, uint32_t *ep_count)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t BN = (size_t)P.B * P.N, BL = (size_t)P.B * P.L;
    if (i < BN) {
        if (pos) { pos[2 * i] = P.pos_x[i]; pos[2 * i + 1] = P.pos_y[i]; }
        if (vel) { vel[2 * i] = P.vel_x[i]; vel[2 * i + 1] = P.vel_y[i]; }
    }
    if (i < BL && lm) { lm[2 * i] = P.lm_x[i]; lm[2 * i + 1] = P.lm_y[i]; }
    if (i < (size_t)P.B) {
        if (ep_step) ep_step[i] = P.ep_step[i];
        if (ep_count) ep_count[i] = P.ep_count[i];
    }
}

}  // namespace
