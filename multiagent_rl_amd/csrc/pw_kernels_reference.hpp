// pw_kernels_reference.hpp -- part of libpworld.so (one translation unit: csrc/pworld.hip includes it).
// simple_reference and simple_speaker_listener: the communication scenarios of the reference's sweep
// (main.py:24).
#pragma once

namespace {

// ------------------------------------------------------------------------------------------
// simple_reference: N = 2 agents that move AND speak (world.dim_c = 10 symbols), L <= 3 landmarks,
// nobody collides.  Each agent knows a goal landmark (goal_b) that the OTHER agent should reach;
// reward_i = -|p_other - p_goal_b(i)|^2.  Upstream pieces replaced here:
//   MultiAgentEnv._set_action (MultiDiscrete split: 5 movement + 10 communication entries, the
//   movement part arg-maxed under force_discrete_action), World.step (action force, damped Euler,
//   update_agent_state: state.c = action.c), scenario reward / observation
//   ([p_vel] + landmark_rel + goal_b.color + the other agent's c -- experiments/scenarios.py:23-42).
// Lane = (env, agent): 32 envs per wave; the partner is lane ^ 1, so every exchange is a shuffle.
// Extra state planes: comm [B*N*dim_c] f32 (state.c), goal [B*N] i32 (landmark index of goal_b).
//
// simple_speaker_listener (template flag SL, dim_c = 3) shares the structure: agent 0 is the speaker (never
// moves: apply_action_force / integrate_state skip it; its Discrete(3) action becomes state.c), agent 1 the
// silent listener (moves; state.c = 0).  Both are rewarded with -|p_listener - p_goal|^2 where the goal is the
// speaker's goal_b; the observation is the one experiments/scenarios.py:45-64 patches in -- [p_vel] +
// landmark_rel + (goal_b.color for the speaker, zeros for the listener), 11 numbers for both agents, the spoken
// symbol NOT included.  act_idx = (symbol, movement); act_vec rows are 5 wide (the speaker's 3 first).
// ------------------------------------------------------------------------------------------
constexpr int kDimC = 10, kDimCSL = 3;

struct RefParams {
    int B, L, D, max_episode_len, auto_reset, force_discrete;
    uint64_t seed, env_id_base;
    float dt, damp, mass, sens;
    float *pos_x, *pos_y, *vel_x, *vel_y, *lm_x, *lm_y, *comm;
    int32_t *goal, *ep_step;
    uint32_t *ep_count;
};

template <int DC>
struct RefLane {
    float px, py, vx, vy, c[DC], lmx[3], lmy[3];
    int goal;
};

// The goal landmark of a lane.  All six coordinates are read first and the choice is made between VALUES: written as "start with
// landmark 0, overwrite if goal == 1 / 2" the compiler turns the conditional loads into one load at a run-time index, and a struct
// that is indexed at run time lives in scratch memory (88 B of private segment in every kernel that used it).
template <int DC>
__device__ __forceinline__ void ref_goal_landmark(const RefLane<DC> &s, const int gsel, float &glx, float &gly)
{
    const float x0 = s.lmx[0], x1 = s.lmx[1], x2 = s.lmx[2], y0 = s.lmy[0], y1 = s.lmy[1], y2 = s.lmy[2];
    glx = gsel == 2 ? x2 : gsel == 1 ? x1 : x0;
    gly = gsel == 2 ? y2 : gsel == 1 ? y1 : y0;
}

template <int DC, bool SL>
__device__ __forceinline__ void ref_reset(const RefParams &P, uint64_t env_id, uint32_t episode, int a, RefLane<DC> &s)
{
    uint32_t r[4];
    pw_philox4x32_10((uint32_t)a, episode, (uint32_t)env_id, (uint32_t)(env_id >> 32), (uint32_t)P.seed,
                     (uint32_t)(P.seed >> 32), r);
    s.px = 2.0f * ((float)(r[0] >> 8) * 5.9604644775390625e-8f) + -1.0f;
    s.py = 2.0f * ((float)(r[1] >> 8) * 5.9604644775390625e-8f) + -1.0f;
    s.goal = (SL && a == 1) ? 0 : (int)(r[2] % (uint32_t)P.L);  // the listener has no goal_b
    s.vx = 0.f; s.vy = 0.f;
#pragma unroll
    for (int q = 0; q < DC; ++q) s.c[q] = 0.f;
#pragma unroll
    for (int l = 0; l < 3; ++l)
        if (l < P.L) pw_reset_xy(P.seed, env_id, episode, (uint32_t)(2 + l), -1.0f, 1.0f, &s.lmx[l], &s.lmy[l]);
}

template <int DC, bool SL>
__device__ __forceinline__ void ref_write_obs(const RefParams &P, const RefLane<DC> &s, const float *c_other, const int a,
                                              float *__restrict__ o)
{
    int k = 0;
    o[k++] = s.vx;
    o[k++] = s.vy;
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        if (l < P.L) {
            o[k++] = s.lmx[l] - s.px;
            o[k++] = s.lmy[l] - s.py;
        }
    }
    if (SL) {  // speaker: colour of its goal landmark (0.65 on its own channel); listener: goal_b is None -> zeros
#pragma unroll
        for (int q = 0; q < 3; ++q) o[k++] = a == 0 ? (q == s.goal ? 0.65f : 0.15f) : 0.0f;
        return;
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) o[k++] = q == s.goal ? 0.75f : 0.25f;  // landmark colours (0.75 on its own channel)
#pragma unroll
    for (int q = 0; q < DC; ++q) o[k++] = c_other[q];
}

template <int DC>
__device__ __forceinline__ void ref_load(const RefParams &P, int env, int a, RefLane<DC> &s)
{
    const size_t g = (size_t)env * 2 + a;
    s.px = P.pos_x[g]; s.py = P.pos_y[g]; s.vx = P.vel_x[g]; s.vy = P.vel_y[g];
    s.goal = P.goal[g];
#pragma unroll
    for (int q = 0; q < DC; ++q) s.c[q] = P.comm[g * DC + q];
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        s.lmx[l] = l < P.L ? P.lm_x[(size_t)env * P.L + l] : 0.f;
        s.lmy[l] = l < P.L ? P.lm_y[(size_t)env * P.L + l] : 0.f;
    }
}

template <int DC>
__device__ __forceinline__ void ref_store(const RefParams &P, int env, int a, const RefLane<DC> &s)
{
    const size_t g = (size_t)env * 2 + a;
    P.pos_x[g] = s.px; P.pos_y[g] = s.py; P.vel_x[g] = s.vx; P.vel_y[g] = s.vy;
    P.goal[g] = s.goal;
#pragma unroll
    for (int q = 0; q < DC; ++q) P.comm[g * DC + q] = s.c[q];
    if (a == 0) {
#pragma unroll
        for (int l = 0; l < 3; ++l)
            if (l < P.L) { P.lm_x[(size_t)env * P.L + l] = s.lmx[l]; P.lm_y[(size_t)env * P.L + l] = s.lmy[l]; }
    }
}

template <int DC, bool SL>
__global__ void __launch_bounds__(kWave) pw_reference_rollout_kernel(const RefParams P, const pw_step_io io,
                                                                     const int32_t *act_comm, const int T)
{
    const int lane = threadIdx.x, a = lane & 1;
    int env = blockIdx.x * 32 + (lane >> 1);
    const bool valid = env < P.B;
    if (!valid) env = 0;
    const size_t g = (size_t)env * 2 + a, BN = (size_t)P.B * 2;
    const bool moves = !(SL && a == 0), speaks = !(SL && a == 1);
    constexpr int AW = SL ? 5 : 5 + DC;  // act_vec row width
    RefLane<DC> s;
    ref_load<DC>(P, env, a, s);
    int ep_step = P.ep_step[env];
    uint32_t ep_count = P.ep_count[env];
    for (int t = 0; t < T; ++t) {
        const size_t row = (size_t)t * BN + g;
        // ---- _set_action: per-agent action space (MultiDiscrete split [5 | 10], or Discrete(3) / Discrete(5))
        float a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, cn[DC];
        if (io.act_idx) {
            const int ai = io.act_idx[row], ci = SL ? ai : act_comm[row];
            if (moves) { a1 = ai == 1; a2 = ai == 2; a3 = ai == 3; a4 = ai == 4; }
#pragma unroll
            for (int q = 0; q < DC; ++q) cn[q] = (speaks && q == ci) ? 1.0f : 0.0f;
        } else {
            const float *av = io.act_vec + row * AW;
            if (moves) {
                float a0 = av[0];
                a1 = av[1]; a2 = av[2]; a3 = av[3]; a4 = av[4];
                if (P.force_discrete) {
                    int d = 0;
                    float best = a0;
                    if (a1 > best) { best = a1; d = 1; }
                    if (a2 > best) { best = a2; d = 2; }
                    if (a3 > best) { best = a3; d = 3; }
                    if (a4 > best) { best = a4; d = 4; }
                    a1 = d == 1; a2 = d == 2; a3 = d == 3; a4 = d == 4;
                }
            }
#pragma unroll
            for (int q = 0; q < DC; ++q) cn[q] = speaks ? av[(SL ? 0 : 5) + q] : 0.0f;
        }
        // ---- World.step: nobody collides; damped semi-implicit Euler for movable agents; update_agent_state
        if (moves) {
            float ux = 0.0f + (a1 - a2), uy = 0.0f + (a3 - a4);
            ux *= P.sens; uy *= P.sens;
            const float fx = ux + 0.0f, fy = uy + 0.0f;
            s.vx = s.vx * P.damp; s.vy = s.vy * P.damp;
            s.vx = s.vx + (fx / P.mass) * P.dt;
            s.vy = s.vy + (fy / P.mass) * P.dt;
            s.px = s.px + s.vx * P.dt;
            s.py = s.py + s.vy * P.dt;
        }
#pragma unroll
        for (int q = 0; q < DC; ++q) s.c[q] = speaks ? cn[q] + 0.0f : 0.0f;
        // ---- the other agent, by shuffle
        const float ox = __shfl_xor(s.px, 1, kWave), oy = __shfl_xor(s.py, 1, kWave);
        float co[DC];
        if (!SL) {
#pragma unroll
            for (int q = 0; q < DC; ++q) co[q] = __shfl_xor(s.c[q], 1, kWave);
        }
        // ---- reward: -|p_other - p_goal_b|^2 (speaker_listener: -|p_listener - p_goal_b(speaker)|^2 for both)
        const int og = __shfl_xor(s.goal, 1, kWave);
        const int gsel = (SL && a == 1) ? og : s.goal;
        float glx, gly;
        ref_goal_landmark<DC>(s, gsel, glx, gly);
        const float tx = (SL && a == 1) ? s.px : ox, ty = (SL && a == 1) ? s.py : oy;
        const float dx = tx - glx, dy = ty - gly;
        const float r = -(dx * dx + dy * dy);
        const float r_other = __shfl_xor(r, 1, kWave);
        const float acc = (0.0f + (a == 0 ? r : r_other)) + (a == 0 ? r_other : r);  // agent order
        if (valid) {
            if (io.rew) io.rew[row] = r;
            if (io.done) io.done[row] = 0;
            if (io.coll) io.coll[row] = 0;
            if (io.rew_shared && a == 0) io.rew_shared[(size_t)t * P.B + env] = acc;
        }
        ep_step += 1;
        const bool term = P.max_episode_len > 0 && ep_step >= P.max_episode_len;
        if (valid && a == 0 && io.terminal) io.terminal[(size_t)t * P.B + env] = term ? 1 : 0;
        if (term && P.auto_reset) {
            if (valid && io.final_obs) ref_write_obs<DC, SL>(P, s, co, a, io.final_obs + row * P.D);
            ep_count += 1;
            ep_step = 0;
            ref_reset<DC, SL>(P, P.env_id_base + (uint64_t)env, ep_count, a, s);
            if (!SL) {
#pragma unroll
                for (int q = 0; q < DC; ++q) co[q] = 0.f;  // the other agent reset too
            }
        }
        if (valid && io.obs) ref_write_obs<DC, SL>(P, s, co, a, io.obs + row * P.D);
    }
    if (valid) {
        ref_store<DC>(P, env, a, s);
        if (a == 0) { P.ep_step[env] = ep_step; P.ep_count[env] = ep_count; }
    }
}

// mode bit 0: reset masked envs, bit 1: write obs, bit 2: write reward
template <int DC, bool SL>
__global__ void __launch_bounds__(kWave) pw_reference_aux_kernel(const RefParams P, const int mode, const uint8_t *env_mask,
                                                                 float *obs, float *rew)
{
    const int lane = threadIdx.x, a = lane & 1;
    int env = blockIdx.x * 32 + (lane >> 1);
    const bool valid = env < P.B;
    if (!valid) env = 0;
    RefLane<DC> s;
    const bool rs = (mode & 1) && (!env_mask || env_mask[env]);
    if (rs) ref_reset<DC, SL>(P, P.env_id_base + (uint64_t)env, P.ep_count[env] + 1, a, s);
    else ref_load<DC>(P, env, a, s);
    float co[DC];
#pragma unroll
    for (int q = 0; q < DC; ++q) co[q] = __shfl_xor(s.c[q], 1, kWave);
    const float ox = __shfl_xor(s.px, 1, kWave), oy = __shfl_xor(s.py, 1, kWave);
    const int og = __shfl_xor(s.goal, 1, kWave);
    if ((mode & 4) && valid && rew) {
        const int gsel = (SL && a == 1) ? og : s.goal;
        float glx, gly;
        ref_goal_landmark<DC>(s, gsel, glx, gly);
        const float tx = (SL && a == 1) ? s.px : ox, ty = (SL && a == 1) ? s.py : oy;
        const float dx = tx - glx, dy = ty - gly;
        rew[(size_t)env * 2 + a] = -(dx * dx + dy * dy);
    }
    if ((mode & 2) && valid && obs) ref_write_obs<DC, SL>(P, s, co, a, obs + ((size_t)env * 2 + a) * P.D);
    if (rs && valid) {
        ref_store<DC>(P, env, a, s);
        // both lanes of the env have read ep_count above (same wave, program order); one of them advances it
        if (a == 0) { P.ep_count[env] += 1; P.ep_step[env] = 0; }
    }
}

// comm / goal planes <-> caller arrays ([B,N,dim_c] f32, [B,N] i32)
__global__ void pw_reference_state_kernel(const RefParams P, const int set, const int dim_c, float *comm, int32_t *goal)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t BN = (size_t)P.B * 2;
    if (i < BN * dim_c && comm) {
        if (set) P.comm[i] = comm[i]; else comm[i] = P.comm[i];
    }
    if (i < BN && goal) {
        if (set) P.goal[i] = goal[i]; else goal[i] = P.goal[i];
    }
}

}  // namespace
