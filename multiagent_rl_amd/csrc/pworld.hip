// libpworld.so -- batched particle world for MI355X (gfx950, wave64).
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

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string &msg)
{
    g_last_error = msg;
    return code;
}

#define PW_HIP_CHECK(expr)                                                             \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess)                                                          \
            return fail(PW_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e));   \
    } while (0)

constexpr int kWave = 64;

// Every workgroup is ONE wave, and a wave's LDS instructions execute in issue order, so the
// only thing a write -> cross-lane read hand-off through LDS needs is (a) that the compiler
// keeps the program order of the accesses and (b) that the data has landed before it is
// consumed.  Unlike __syncthreads() this does NOT drain vmcnt: the step's global stores
// (16 B x 4 per lane of observations) stay in flight across steps.
__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Everything a kernel needs, passed by value in the kernarg segment.
struct KParams {
    int B, N, L, A, D;
    int epw;          // envs per wave
    int max_episode_len, auto_reset, force_discrete, landmark_collide;
    uint64_t seed, env_id_base;
    float dt, damp, contact_force, contact_margin, mass, landmark_size;
    float *pos_x, *pos_y, *vel_x, *vel_y, *lm_x, *lm_y;
    int32_t *ep_step;
    uint32_t *ep_count;
    float agent_size[PW_MAX_AGENTS];
    float agent_sens[PW_MAX_AGENTS];      // accel if set else default_sensitivity (_set_action)
    float agent_fscale[PW_MAX_AGENTS];    // 1, or mass*accel with the fork knob (apply_action_force)
    float agent_max_speed[PW_MAX_AGENTS]; // < 0: None
};

// Per-lane view of "its" env inside the wave.
struct Lane {
    int e_local, a, base;  // env slot in wave, agent index, first lane of the env
    int env;               // local env index
    size_t g;              // env * N + a
    bool valid;
};

__device__ __forceinline__ Lane make_lane(const KParams &P)
{
    Lane ln;
    const int lane = threadIdx.x;
    ln.e_local = lane / P.N;
    ln.a = lane - ln.e_local * P.N;
    ln.base = ln.e_local * P.N;
    ln.env = blockIdx.x * P.epw + ln.e_local;
    ln.valid = ln.e_local < P.epw && ln.env < P.B;
    if (!ln.valid) {  // idle lanes alias env slot 0 for reads; they never write
        ln.e_local = 0; ln.a = 0; ln.base = 0; ln.env = 0;
    }
    ln.g = (size_t)ln.env * P.N + ln.a;
    return ln;
}

// Correctly rounded sqrtf for the hot loops.  The compiler's expansion of sqrtf spends half of its
// ~22 instructions on scaling subnormal-range inputs and on the 0 / inf / NaN pass-through.  For
// x in [2^-90, 2^90) neither is needed: v_sqrt_f32 is within 1 ulp, and two fused residual tests
// pick between {s-1ulp, s, s+1ulp} -- the same correction step the compiler emits.  Anything
// outside that range (never reached from finite, non-coincident states) takes the general sqrtf.
__device__ __forceinline__ float sqrt_rn_fast(float x)
{
    if (__builtin_expect(!(x >= 8.077935669463161e-28f && x < 1.2379400392853803e+27f), 0)) return sqrtf(x);
    float s = __builtin_amdgcn_sqrtf(x);
    const float s_dn = __uint_as_float(__float_as_uint(s) - 1u);
    const float s_up = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_dn = __builtin_fmaf(-s_dn, s, x);
    const float r_up = __builtin_fmaf(-s_up, s, x);
    s = r_dn <= 0.0f ? s_dn : s;
    s = r_up > 0.0f ? s_up : s;
    return s;
}

// pw_softplus with the exp argument known to be <= 0: same operations as include/pworld_math.h
// (so the same bits), but branch-free: the polynomial runs on a clamped argument and the
// exact-zero cut / NaN pass-through are selects.
__device__ __forceinline__ float softplus_branchless(float x)
{
    const float ax = x < 0.0f ? -x : x;
    const float m = x > 0.0f ? x : 0.0f;
    const float t0 = -ax;                       // <= 0, or NaN
    const float tc = t0 > -87.0f ? t0 : -86.0f;  // keep the exponent arithmetic in range when cut
    float t = tc * 1.44269504088896341f;
    float n = floorf(t + 0.5f);
    float r = tc - n * 0.693359375f;
    r = r - n * -2.12194440054690583e-4f;
    float p = 1.98412698412698413e-4f;
    p = p * r + 1.38888888888888894e-3f;
    p = p * r + 8.33333333333333322e-3f;
    p = p * r + 4.16666666666666644e-2f;
    p = p * r + 1.66666666666666657e-1f;
    p = p * r + 0.5f;
    p = p * r + 1.0f;
    p = p * r + 1.0f;
    const int32_t e = (int32_t)n + 127;
    float ex = p * __uint_as_float((uint32_t)e << 23);
    ex = t0 > -87.0f ? ex : (t0 != t0 ? t0 : 0.0f);  // pw_exp: x <= -87 -> +0, NaN -> NaN
    return m + pw_log1p01(ex);
}

// get_collision_force seen from entity i against entity j: force on i.
// delta = p_i - p_j; dist = sqrt(sum(delta^2)); pen = logaddexp(0, -(dist - dist_min)/k) * k;
// force = contact_force * delta / dist * pen.  (The force on the pair's second entity is
// the exact negation, which is what this evaluates to from that entity's side.)
template <bool FAST = false>
__device__ __forceinline__ void collision_force(float px, float py, float qx, float qy, float dist_min,
                                                float k, float cf, float &fx, float &fy)
{
    const float dx = px - qx, dy = py - qy;
    const float d2 = dx * dx + dy * dy;
    const float dist = FAST ? sqrt_rn_fast(d2) : sqrtf(d2);
    const float xarg = -(dist - dist_min) / k;
    const float pen = (FAST ? softplus_branchless(xarg) : pw_softplus(xarg)) * k;
    const float Fx = cf * dx / dist * pen;
    const float Fy = cf * dy / dist * pen;
    fx = Fx + fx;
    fy = Fy + fy;
}

// True only if the pair force is exactly +-0: d2 >= (dist_min + margin)^2 (1 + 1e-6) with margin = 88.5 k
// puts the softplus argument below -88.4 < -87, pw_exp's exact-zero cut; float rounding in this test is
// ~1e-7 relative against a slack of 1.4 k.  NaN and +inf are never "far" (they must propagate).
__device__ __forceinline__ bool provably_far(float d2, float dist_min, float margin)
{
    const float r = dist_min + margin;
    return d2 >= r * r * 1.000001f && d2 <= 3.402823466e+38f;
}

__device__ __forceinline__ float tag_bound(float x)
{
    if (x < 0.9f) return 0.0f;
    if (x < 1.0f) return (x - 0.9f) * 10.0f;
    const float b = pw_exp(2.0f * x - 2.0f);
    return b < 10.0f ? b : 10.0f;
}

// scenario.observation for lane's agent -> row o[0..D).  LDS holds the current
// positions (and velocities for simple_tag) of the wave's envs.
template <int SCEN, int OBS>
__device__ __forceinline__ void write_obs(const KParams &P, const Lane &ln, float *__restrict__ o,
                                          float px, float py, float vx, float vy,
                                          const float2 *s_pos, const float2 *s_vel, const float2 *s_lm)
{
    const int N = P.N, L = P.L;
    const float2 *lm = s_lm + ln.e_local * L;
    if (SCEN == PW_SIMPLE_SPREAD && OBS == PW_OBS_LOCAL && (L & 1) == 0) {
        // D = 4 + 2L is a multiple of 4: 16-byte row stores (experiments/scenarios.py:6-20 layout)
        float4 *o4 = reinterpret_cast<float4 *>(o);
        o4[0] = make_float4(vx, vy, px, py);
        for (int c = 0; c < L / 2; ++c) {
            const float2 l0 = lm[2 * c], l1 = lm[2 * c + 1];
            o4[1 + c] = make_float4(l0.x - px, l0.y - py, l1.x - px, l1.y - py);
        }
        return;
    }
    // every observation component is an (x, y) pair and D is even: 8-byte stores, half the store count
    float2 *o2 = reinterpret_cast<float2 *>(o);
    int k = 0;
    o2[k++] = make_float2(vx, vy);
    o2[k++] = make_float2(px, py);
    for (int l = 0; l < L; ++l) {
        const float2 q = lm[l];
        o2[k++] = make_float2(q.x - px, q.y - py);
    }
    if (SCEN == PW_SIMPLE_TAG || OBS == PW_OBS_FULL) {
        const float2 *pp = s_pos + ln.base;
        for (int j = 0; j < N; ++j) {
            if (j == ln.a) continue;
            const float2 q = pp[j];
            o2[k++] = make_float2(q.x - px, q.y - py);
        }
        if (SCEN == PW_SIMPLE_TAG) {
            const float2 *vv = s_vel + ln.base;
            for (int j = P.A; j < N; ++j) {  // velocities of the OTHER good agents
                if (j == ln.a) continue;
                o2[k++] = vv[j];
            }
        } else {
            for (int j = 0; j < N - 1; ++j) o2[k++] = make_float2(0.0f, 0.0f);  // comm of silent agents
        }
    }
    while (2 * k < P.D) o2[k++] = make_float2(0.0f, 0.0f);
}

// scenario.reward + is_collision mask for the lane's agent from the positions in LDS.
// s_red: EPW*L floats of wave-private scratch (per-landmark min distance).
template <int SCEN>
__device__ __forceinline__ float reward_and_mask(const KParams &P, const Lane &ln, float px, float py,
                                                 float my_size, const float2 *s_pos, const float2 *s_lm,
                                                 float *s_red, uint64_t &mask_out)
{
    const int N = P.N, L = P.L;
    const float2 *pp = s_pos + ln.base;
    uint64_t m = 0;
    for (int j = 0; j < N; ++j) {
        const float2 q = pp[j];
        const float dx = q.x - px, dy = q.y - py;
        const float d = sqrtf(dx * dx + dy * dy);
        if (d < P.agent_size[j] + my_size) m |= 1ull << j;
    }
    mask_out = m;
    float r = 0.0f;
    if (SCEN == PW_SIMPLE_SPREAD) {
        // shared term: -sum_l min_a |p_a - p_l|.  Lane a owns landmarks a, a+N, ...; sqrt is
        // monotone and correctly rounded, so min over distances == sqrt(min over squares).
        const float2 *lm = s_lm + ln.e_local * L;
        float own = 0.0f;
        for (int l = ln.a; l < L; l += N) {
            const float2 q = lm[l];
            float best = 0.0f;
            for (int a = 0; a < N; ++a) {
                const float2 p = pp[a];
                const float dx = p.x - q.x, dy = p.y - q.y;
                const float d2 = dx * dx + dy * dy;
                best = (a == 0 || d2 < best) ? d2 : best;
            }
            own = sqrtf(best);
            if (L > N && ln.valid) s_red[ln.e_local * L + l] = own;
        }
        if (L > N) {
            wave_lds_sync();
            for (int l = 0; l < L; ++l) r -= s_red[ln.e_local * L + l];
        } else {
            // per-env ordered reduction over the env's lanes by wave shuffle (ds_bpermute)
            for (int l = 0; l < L; ++l) r -= __shfl(own, ln.base + l, kWave);
        }
        for (int a = 0; a < N; ++a)
            if ((m >> a) & 1) r -= 1.0f;  // includes a == agent, as upstream
    } else {
        const int A = P.A;
        if (ln.a >= A) {
            for (int a = 0; a < A; ++a)
                if ((m >> a) & 1) r -= 10.0f;
            r -= tag_bound(fabsf(px));
            r -= tag_bound(fabsf(py));
        }
        // adversaries: +10 per colliding (good, adversary) pair; bit a of good lane g's mask
        const uint32_t mlo = (uint32_t)m, mhi = (uint32_t)(m >> 32);
        float radv = 0.0f;
        for (int g = A; g < N; ++g) {
            const uint64_t mg = ((uint64_t)(uint32_t)__shfl((int)mhi, ln.base + g, kWave) << 32) |
                                (uint32_t)__shfl((int)mlo, ln.base + g, kWave);
            for (int a = 0; a < A; ++a)
                if ((mg >> a) & 1) radv += 10.0f;
        }
        if (ln.a < A) r = radv;
    }
    return r;
}

__device__ __forceinline__ void reset_lane(const KParams &P, const Lane &ln, uint32_t episode, int SCEN,
                                           float &px, float &py, float2 *s_lm)
{
    const uint64_t env_id = P.env_id_base + (uint64_t)ln.env;
    pw_reset_xy(P.seed, env_id, episode, (uint32_t)ln.a, -1.0f, 1.0f, &px, &py);
    const float lo = SCEN == PW_SIMPLE_TAG ? -0.9f : -1.0f;
    for (int l = ln.a; l < P.L; l += P.N) {
        float x, y;
        pw_reset_xy(P.seed, env_id, episode, (uint32_t)(P.N + l), lo, -lo, &x, &y);
        s_lm[ln.e_local * P.L + l] = make_float2(x, y);
    }
}

// LDS carve-up of one (single-wave) workgroup
struct Smem {
    float2 *pos, *vel, *lm;
    float *red;
};
__device__ __forceinline__ Smem carve(const KParams &P, unsigned char *raw)
{
    Smem s;
    const int nl = P.epw * P.N, ll = P.epw * P.L;
    s.pos = reinterpret_cast<float2 *>(raw);
    s.vel = s.pos + nl;
    s.lm = s.vel + nl;
    s.red = reinterpret_cast<float *>(s.lm + ll);
    return s;
}
size_t smem_bytes(const KParams &P)
{
    return (size_t)P.epw * (2 * P.N + P.L) * sizeof(float2) + (size_t)P.epw * P.L * sizeof(float);
}


// ------------------------------------------------------------------------------------------
// Fast path: simple_spread, local observation, homogeneous agents (one size, no max_speed),
// landmarks that do not collide, L <= N.  Same arithmetic, same bits, fewer instructions:
//  * far pairs are skipped: beyond dist_min + 88 k the softplus is EXACTLY 0 (pw_exp underflow
//    cut), the force term is +-0 and adding it never changes the accumulator (which cannot be -0);
//  * is_collision needs no sqrt: sqrt is monotone and correctly rounded, so
//    sqrt(d2) < dist_min  <=>  d2 < coll_thr2 with coll_thr2 = min{y : sqrtf(y) >= dist_min},
//    found on the host;
//  * one pass over the env's positions in LDS after integration yields the collision mask of
//    step t, the near-pair mask of step t+1 and the owned landmark's min distance;
//  * NT > 0 fixes N at compile time (loops unrolled); the next step's action is prefetched.
// ------------------------------------------------------------------------------------------
struct FastConsts {
    float dist_min, coll_thr2, near_thr2, sens, fscale, size;
};

template <int NT>
__device__ __forceinline__ void partner_pass(const int N, const Lane &ln, const float2 *pp, float px, float py,
                                             bool own_lm, float olx, float oly, const FastConsts &C,
                                             uint64_t &coll, uint64_t &near, float &best)
{
    coll = 0; near = 0; best = 0.0f;
#pragma unroll(NT > 0 ? NT : 1)
    for (int j = 0; j < (NT ? NT : N); ++j) {
        const float2 q = pp[j];
        const float dx = q.x - px, dy = q.y - py;
        const float d2 = dx * dx + dy * dy;  // (q - p)^2 == (p - q)^2 bit for bit
        if (d2 < C.coll_thr2) coll |= 1ull << j;
        const bool far = d2 >= C.near_thr2 && d2 <= 3.402823466e+38f;  // NaN / inf stay "near"
        if (!far && j != ln.a) near |= 1ull << j;
        const float ex = q.x - olx, ey = q.y - oly;
        const float e2 = ex * ex + ey * ey;
        best = (j == 0 || e2 < best) ? e2 : best;
    }
    (void)own_lm;
}

template <int NT>
__global__ void __launch_bounds__(kWave) pw_spread_fast_kernel(const KParams P, const pw_step_io io, const int T,
                                                               const FastConsts C)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const Smem S = carve(P, smem_raw);
    const Lane ln = make_lane(P);
    const int N = NT ? NT : P.N, L = P.L, D = P.D;
    const size_t BN = (size_t)P.B * N;
    const float2 *pp = S.pos + ln.base;
    float2 *lmv = S.lm + ln.e_local * L;

    float px = 0.f, py = 0.f, vx = 0.f, vy = 0.f, olx = 0.f, oly = 0.f;
    int ep_step = 0;
    uint32_t ep_count = 0;
    const bool own_lm = ln.a < L;
    if (ln.valid) {
        px = P.pos_x[ln.g]; py = P.pos_y[ln.g];
        vx = P.vel_x[ln.g]; vy = P.vel_y[ln.g];
        ep_step = P.ep_step[ln.env];
        ep_count = P.ep_count[ln.env];
        if (own_lm) {
            olx = P.lm_x[(size_t)ln.env * L + ln.a];
            oly = P.lm_y[(size_t)ln.env * L + ln.a];
            lmv[ln.a] = make_float2(olx, oly);
        }
        S.pos[threadIdx.x] = make_float2(px, py);
    }
    wave_lds_sync();
    uint64_t coll, near;
    float best;
    partner_pass<NT>(N, ln, pp, px, py, own_lm, olx, oly, C, coll, near, best);

    const float k = P.contact_margin, cf = P.contact_force, dt = P.dt, damp = P.damp, mass = P.mass;
    int act_next = 0;
    if (ln.valid && io.act_idx) act_next = io.act_idx[ln.g];

    for (int t = 0; t < T; ++t) {
        const size_t row = (size_t)t * BN + ln.g;
        // ---- U2 + U4
        float ux, uy;
        if (io.act_idx) {
            const int a = act_next;
            if (t + 1 < T && ln.valid) act_next = io.act_idx[row + BN];  // prefetch step t+1
            ux = 0.0f + ((a == 1 ? 1.0f : 0.0f) - (a == 2 ? 1.0f : 0.0f));
            uy = 0.0f + ((a == 3 ? 1.0f : 0.0f) - (a == 4 ? 1.0f : 0.0f));
        } else {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
            if (ln.valid) {
                const float *av = io.act_vec + row * 5;
                a0 = av[0]; a1 = av[1]; a2 = av[2]; a3 = av[3]; a4 = av[4];
            }
            if (P.force_discrete) {
                int d = 0;
                float bst = a0;
                if (a1 > bst) { bst = a1; d = 1; }
                if (a2 > bst) { bst = a2; d = 2; }
                if (a3 > bst) { bst = a3; d = 3; }
                if (a4 > bst) { bst = a4; d = 4; }
                a1 = d == 1; a2 = d == 2; a3 = d == 3; a4 = d == 4;
            }
            ux = 0.0f + (a1 - a2);
            uy = 0.0f + (a3 - a4);
        }
        ux *= C.sens; uy *= C.sens;
        if (C.fscale != 1.0f) { ux = C.fscale * ux; uy = C.fscale * uy; }
        float fx = ux + 0.0f, fy = uy + 0.0f;
        // ---- U5: only partners whose force can be non-zero, ascending j
        for (uint64_t m = ln.valid ? near : 0; m; m &= m - 1) {
            const int j = __builtin_ctzll(m);
            const float2 q = pp[j];
            collision_force(px, py, q.x, q.y, C.dist_min, k, cf, fx, fy);
        }
        // ---- U6
        vx = vx * damp; vy = vy * damp;
        vx = vx + (fx / mass) * dt;
        vy = vy + (fy / mass) * dt;
        px = px + vx * dt;
        py = py + vy * dt;
        wave_lds_sync();
        if (ln.valid) S.pos[threadIdx.x] = make_float2(px, py);
        wave_lds_sync();

        partner_pass<NT>(N, ln, pp, px, py, own_lm, olx, oly, C, coll, near, best);
        // ---- simple_spread.reward
        const float own = sqrtf(best);
        float r = 0.0f;
        for (int l = 0; l < L; ++l) r -= __shfl(own, ln.base + l, kWave);
#pragma unroll(NT > 0 ? NT : 1)
        for (int a = 0; a < (NT ? NT : N); ++a)
            if ((coll >> a) & 1) r -= 1.0f;
        if (ln.valid) {
            if (io.rew) io.rew[row] = r;
            if (io.done) io.done[row] = 0;
            if (io.coll) io.coll[row] = coll;
        }
        if (io.rew_shared) {
            float acc = 0.0f;
#pragma unroll(NT > 0 ? NT : 1)
            for (int i = 0; i < (NT ? NT : N); ++i) acc += __shfl(r, ln.base + i, kWave);
            if (ln.valid && ln.a == 0) io.rew_shared[(size_t)t * P.B + ln.env] = acc;
        }
        ep_step += 1;
        const bool term = P.max_episode_len > 0 && ep_step >= P.max_episode_len;
        if (ln.valid && ln.a == 0 && io.terminal) io.terminal[(size_t)t * P.B + ln.env] = term ? 1 : 0;
        const bool do_reset = ln.valid && term && P.auto_reset;
        if (__any(do_reset)) {
            if (do_reset && io.final_obs)
                write_obs<PW_SIMPLE_SPREAD, PW_OBS_LOCAL>(P, ln, io.final_obs + row * D, px, py, vx, vy, S.pos, S.vel, S.lm);
            wave_lds_sync();
            if (do_reset) {
                ep_count += 1;
                ep_step = 0;
                reset_lane(P, ln, ep_count, PW_SIMPLE_SPREAD, px, py, S.lm);
                vx = 0.f; vy = 0.f;
                S.pos[threadIdx.x] = make_float2(px, py);
            }
            wave_lds_sync();
            if (own_lm) { const float2 q = lmv[ln.a]; olx = q.x; oly = q.y; }
            partner_pass<NT>(N, ln, pp, px, py, own_lm, olx, oly, C, coll, near, best);
        }
        if (ln.valid && io.obs)
            write_obs<PW_SIMPLE_SPREAD, PW_OBS_LOCAL>(P, ln, io.obs + row * D, px, py, vx, vy, S.pos, S.vel, S.lm);
    }

    if (ln.valid) {
        P.pos_x[ln.g] = px; P.pos_y[ln.g] = py;
        P.vel_x[ln.g] = vx; P.vel_y[ln.g] = vy;
        if (own_lm) {
            P.lm_x[(size_t)ln.env * L + ln.a] = olx;
            P.lm_y[(size_t)ln.env * L + ln.a] = oly;
        }
        if (ln.a == 0) {
            P.ep_step[ln.env] = ep_step;
            P.ep_count[ln.env] = ep_count;
        }
    }
}


// ------------------------------------------------------------------------------------------
// Streaming variant of the fast path: the same arithmetic as pw_spread_fast_kernel, laid out
// so the memory pipeline never stalls the step loop.
//  * gfx950 counts loads AND stores in one in-order vmcnt.  The next step's action is loaded
//    at the top of a step, before that step's stores; the wait for it is exact only if the
//    compiler knows how many stores follow, so every store here is unconditional: outputs
//    are all present (checked on the host), per-env values are stored by every lane of the
//    env (same address, same value), and idle lanes SHADOW lane 0 of their wave -- same
//    loads, same arithmetic, same stores -- instead of being branched around.
//  * NT / LT fix N and L at compile time; collision / near masks are 32-bit when N <= 32.
// ------------------------------------------------------------------------------------------
// Diagnostic build only (tools/stamp_probe.hip defines PW_STAMPS): per-segment shader-cycle sums of
// workgroup 0, written to a buffer nothing else reads.  The product build has no stamps.
#ifdef PW_STAMPS
__device__ unsigned long long g_pw_stamps[16];
#define PW_STAMP_DECL unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev = 0, st_now = 0; (void)st_now
#define PW_STAMP_START asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev)::"memory")
#define PW_STAMP(i)                                                                      \
    do {                                                                                 \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_now)::"memory");   \
        st_acc[i] += st_now - st_prev;                                                   \
        st_prev = st_now;                                                                \
    } while (0)
#define PW_STAMP_FLUSH                                                                   \
    do {                                                                                 \
        if (blockIdx.x == 0 && threadIdx.x == 0)                                         \
            for (int i_ = 0; i_ < 8; ++i_) g_pw_stamps[i_] = st_acc[i_];                 \
    } while (0)
#else
#define PW_STAMP_DECL
#define PW_STAMP_START
#define PW_STAMP(i)
#define PW_STAMP_FLUSH
#endif

// Near-pair force accumulation in ascending partner order.  The partner position for the NEXT
// iteration is fetched from LDS before the current force is evaluated, so its ~100-cycle latency
// hides behind the ~400 cycles of IEEE sqrt / divisions / softplus of the current pair.
template <typename MaskT, typename PosT>
__device__ __forceinline__ void near_force_loop(MaskT m, const PosT *pp, float px, float py, float dist_min, float k,
                                                float cf, float &fx, float &fy)
{
    if (!m) return;
    int j = sizeof(MaskT) == 4 ? __builtin_ctz((uint32_t)m) : __builtin_ctzll((uint64_t)m);
    float2 q = *reinterpret_cast<const float2 *>(pp + j);
    for (;;) {
        m &= m - 1;
        float2 qn = q;
        if (m) {
            j = sizeof(MaskT) == 4 ? __builtin_ctz((uint32_t)m) : __builtin_ctzll((uint64_t)m);
            qn = *reinterpret_cast<const float2 *>(pp + j);
        }
        collision_force<true>(px, py, q.x, q.y, dist_min, k, cf, fx, fy);
        if (!m) break;
        q = qn;
    }
}

// x / mass; the division is the identity when mass == 1 (IEEE: x / 1.0f == x for every x), which
// the host knows at launch (UNIT_MASS) -- a runtime select would still pay for the division.
template <bool UNIT_MASS>
__device__ __forceinline__ float div_mass(float x, float mass) { return UNIT_MASS ? x : x / mass; }

struct StreamParams {
    int B, N, L, epw, max_episode_len, auto_reset;
    uint64_t seed, env_id_base;
    float dt, damp, contact_force, contact_margin, mass;
    float dist_min, coll_thr2, near_thr2, sens, fscale;
    float *pos_x, *pos_y, *vel_x, *vel_y, *lm_x, *lm_y;
    int32_t *ep_step;
    uint32_t *ep_count;
    const int32_t *act;
    float *obs, *final_obs, *rew, *rew_shared;
    uint8_t *done, *terminal;
};

template <int LT>
__device__ __forceinline__ void stream_write_obs(float *__restrict__ o, const int L, const float2 *lm, float px,
                                                 float py, float vx, float vy)
{
    if ((LT ? LT : L) % 2 == 0) {
        float4 *o4 = reinterpret_cast<float4 *>(o);
        o4[0] = make_float4(vx, vy, px, py);
#pragma unroll(LT > 0 ? LT / 2 : 1)
        for (int c = 0; c < (LT ? LT : L) / 2; ++c) {
            const float2 l0 = lm[2 * c], l1 = lm[2 * c + 1];
            o4[1 + c] = make_float4(l0.x - px, l0.y - py, l1.x - px, l1.y - py);
        }
    } else {
        float2 *o2 = reinterpret_cast<float2 *>(o);
        o2[0] = make_float2(vx, vy);
        o2[1] = make_float2(px, py);
#pragma unroll(LT > 0 ? LT : 1)
        for (int l = 0; l < (LT ? LT : L); ++l) {
            const float2 q = lm[l];
            o2[2 + l] = make_float2(q.x - px, q.y - py);
        }
    }
}

template <int NT, typename MaskT>
__device__ __forceinline__ void stream_partner_pass(const int N, const int a, const float2 *pp, float px, float py,
                                                    float olx, float oly, float coll_thr2, float near_thr2,
                                                    MaskT &coll, MaskT &near, float &best)
{
    coll = 0; near = 0; best = 0.0f;
#pragma unroll(NT > 0 ? NT : 1)
    for (int j = 0; j < (NT ? NT : N); ++j) {
        const float2 q = pp[j];
        const float dx = q.x - px, dy = q.y - py;
        const float d2 = dx * dx + dy * dy;
        if (d2 < coll_thr2) coll |= (MaskT)1 << j;
        const bool far = d2 >= near_thr2 && d2 <= 3.402823466e+38f;
        if (!far && j != a) near |= (MaskT)1 << j;
        const float ex = q.x - olx, ey = q.y - oly;
        const float e2 = ex * ex + ey * ey;
        best = (j == 0 || e2 < best) ? e2 : best;
    }
}

template <int NT, int LT, bool UNIT_MASS>
__global__ void __launch_bounds__(kWave) pw_spread_stream_kernel(const StreamParams A, const int T)
{
    using MaskT = typename std::conditional<(NT > 0 && NT <= 32), uint32_t, uint64_t>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int N = NT ? NT : A.N, L = LT ? LT : A.L, D = 4 + 2 * L;
    float2 *s_pos = reinterpret_cast<float2 *>(smem_raw);  // [64]
    float2 *s_lm = s_pos + kWave;                          // [epw * L]

    int e_local = (int)threadIdx.x / N;
    int a = (int)threadIdx.x - e_local * N;
    int env = blockIdx.x * A.epw + e_local;
    if (e_local >= A.epw || env >= A.B) {  // idle lane: shadow lane 0 (env slot 0, agent 0)
        e_local = 0; a = 0; env = blockIdx.x * A.epw;
    }
    const int base = e_local * N;
    const uint32_t g = (uint32_t)env * (uint32_t)N + (uint32_t)a;
    const size_t BN = (size_t)A.B * N;
    const float2 *pp = s_pos + base;
    float2 *lmv = s_lm + e_local * L;
    const int la = a < L ? a : 0;  // the landmark this lane "owns" (lanes a >= L duplicate 0, unused)

    float px = A.pos_x[g], py = A.pos_y[g], vx = A.vel_x[g], vy = A.vel_y[g];
    int ep_step = A.ep_step[env];
    uint32_t ep_count = A.ep_count[env];
    float olx = 0.f, oly = 0.f;
    if (L > 0) {
        olx = A.lm_x[(size_t)env * L + la];
        oly = A.lm_y[(size_t)env * L + la];
        lmv[la] = make_float2(olx, oly);
    }
    s_pos[base + a] = make_float2(px, py);
    wave_lds_sync();
    MaskT coll, near;
    float best;
    stream_partner_pass<NT, MaskT>(N, a, pp, px, py, olx, oly, A.coll_thr2, A.near_thr2, coll, near, best);

    const float k = A.contact_margin, cf = A.contact_force, dt = A.dt, damp = A.damp, mass = A.mass;
    int act_next = A.act[g];
    // Vector-memory ops issued per step AFTER the action prefetch: rew, done, rew_shared, terminal
    // + the observation row.  An explicit vmcnt(K) at the end of the step tells the compiler's
    // waitcnt pass that the prefetched load has retired while the K stores stay in flight (it is
    // a hint only: the compiler still inserts any wait it cannot prove redundant).
    constexpr int kStoresPerStep = LT > 0 ? 4 + (LT % 2 == 0 ? 1 + LT / 2 : 2 + LT) : 0;
    constexpr int kVm = kStoresPerStep < 63 ? kStoresPerStep : 63;
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): enter the loop with nothing pending
    PW_STAMP_DECL;

    for (int t = 0; t < T; ++t) {
        PW_STAMP_START;
        const size_t tBN = (size_t)t * BN;
        // ---- U2 + U4 (action index path); prefetch the next step's action before any store
        const int ai = act_next;
        {
            const int tn = t + 1 < T ? t + 1 : t;
            act_next = A.act[(size_t)tn * BN + g];
        }
        float ux = 0.0f + ((ai == 1 ? 1.0f : 0.0f) - (ai == 2 ? 1.0f : 0.0f));
        float uy = 0.0f + ((ai == 3 ? 1.0f : 0.0f) - (ai == 4 ? 1.0f : 0.0f));
        ux *= A.sens; uy *= A.sens;
        if (A.fscale != 1.0f) { ux = A.fscale * ux; uy = A.fscale * uy; }
        float fx = ux + 0.0f, fy = uy + 0.0f;
        PW_STAMP(0);
        // ---- U5
        near_force_loop<MaskT, float2>(near, pp, px, py, A.dist_min, k, cf, fx, fy);
        PW_STAMP(1);
        // ---- U6
        vx = vx * damp; vy = vy * damp;
        vx = vx + div_mass<UNIT_MASS>(fx, mass) * dt;
        vy = vy + div_mass<UNIT_MASS>(fy, mass) * dt;
        px = px + vx * dt;
        py = py + vy * dt;
        wave_lds_sync();
        s_pos[base + a] = make_float2(px, py);
        wave_lds_sync();
        PW_STAMP(2);

        stream_partner_pass<NT, MaskT>(N, a, pp, px, py, olx, oly, A.coll_thr2, A.near_thr2, coll, near, best);
        PW_STAMP(3);
        // ---- simple_spread.reward
        const float own = sqrtf(best);
        float r = 0.0f;
#pragma unroll(LT > 0 ? LT : 1)
        for (int l = 0; l < (LT ? LT : L); ++l) r -= __shfl(own, base + l, kWave);
#pragma unroll(NT > 0 ? NT : 1)
        for (int j = 0; j < (NT ? NT : N); ++j)
            if ((coll >> j) & 1) r -= 1.0f;
        float acc = 0.0f;
#pragma unroll(NT > 0 ? NT : 1)
        for (int i = 0; i < (NT ? NT : N); ++i) acc += __shfl(r, base + i, kWave);
        PW_STAMP(4);
        A.rew[tBN + g] = r;
        A.done[tBN + g] = 0;
        A.rew_shared[(size_t)t * A.B + env] = acc;
        ep_step += 1;
        const bool term = A.max_episode_len > 0 && ep_step >= A.max_episode_len;
        A.terminal[(size_t)t * A.B + env] = term ? 1 : 0;
        if (term && A.auto_reset) {  // same for every lane of an env; rare (1 step in max_episode_len)
            if (A.final_obs) stream_write_obs<LT>(A.final_obs + (tBN + g) * D, L, lmv, px, py, vx, vy);
            wave_lds_sync();
            ep_count += 1;
            ep_step = 0;
            const uint64_t env_id = A.env_id_base + (uint64_t)env;
            pw_reset_xy(A.seed, env_id, ep_count, (uint32_t)a, -1.0f, 1.0f, &px, &py);
            vx = 0.f; vy = 0.f;
            if (L > 0) {
                pw_reset_xy(A.seed, env_id, ep_count, (uint32_t)(N + la), -1.0f, 1.0f, &olx, &oly);
                lmv[la] = make_float2(olx, oly);
            }
            s_pos[base + a] = make_float2(px, py);
        }
        // (lanes whose env did not reset wait here for the ones that did: one wave, reconverged)
        wave_lds_sync();
        if (A.auto_reset && __any(term))
            stream_partner_pass<NT, MaskT>(N, a, pp, px, py, olx, oly, A.coll_thr2, A.near_thr2, coll, near, best);
        PW_STAMP(5);
        stream_write_obs<LT>(A.obs + (tBN + g) * D, L, lmv, px, py, vx, vy);
        PW_STAMP(6);
        if (LT > 0) __builtin_amdgcn_s_waitcnt((kVm & 0xF) | 0x0F70 | ((kVm >> 4) << 14));  // vmcnt(kVm)
        PW_STAMP(7);
    }
    PW_STAMP_FLUSH;

    A.pos_x[g] = px; A.pos_y[g] = py;
    A.vel_x[g] = vx; A.vel_y[g] = vy;
    if (L > 0) {
        A.lm_x[(size_t)env * L + la] = olx;
        A.lm_y[(size_t)env * L + la] = oly;
    }
    A.ep_step[env] = ep_step;
    A.ep_count[env] = ep_count;
}


// ------------------------------------------------------------------------------------------
// Duo variant of the streaming path: the per-wave instruction stream is the critical path at
// small B (a lone wave issues one VALU op per ~5 cycles and 384 waves cannot fill 1024 SIMDs),
// so the step is split over TWO cooperating waves of one workgroup:
//   wave P (physics): action -> near-pair collision forces -> integrate -> publish
//                     {pos, vel} of step t+1 into an LDS ring slot -> near mask for step t+1
//   wave O (outputs): one step behind: collision mask, landmark min-distances, reward, shared
//                     reward, done/terminal, observation rows, every global store
// One s_barrier per step hands a ring slot from P to O.  The ring has 3 slots: a step that
// auto-resets publishes the pre-reset state (O needs it for reward / final_obs) AND the
// post-reset state (both waves continue from it), so slot indices are per-env values.
// Both waves evaluate the Philox reset for the entities they own (agents: both; landmarks: O).
// Arithmetic and results are identical to the other kernels (same bit-exact tests).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void duo_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NT, typename MaskT>
__device__ __forceinline__ MaskT duo_near_pass(const int N, const int a, const float4 *slot, float px, float py,
                                               float near_thr2)
{
    // far <=> near_thr2 <= d2 < +inf.  d2 is a sum of squares (never -0), so on the raw bits this is one
    // unsigned range test; NaN (either sign) and +inf fall outside the range and stay "near".
    const uint32_t lo = __float_as_uint(near_thr2), span = 0x7F800000u - lo;
    MaskT near = 0;
#pragma unroll(NT > 0 ? NT : 1)
    for (int j = 0; j < (NT ? NT : N); ++j) {
        const float2 q = *reinterpret_cast<const float2 *>(slot + j);
        const float dx = q.x - px, dy = q.y - py;
        const float d2 = dx * dx + dy * dy;
        if (__float_as_uint(d2) - lo >= span) near |= (MaskT)1 << j;
    }
    return near & ~((MaskT)1 << a);
}

template <int NT, int LT, bool UNIT_MASS>
__global__ void __launch_bounds__(2 * kWave) pw_spread_duo_kernel(const StreamParams A, const int T)
{
    using MaskT = typename std::conditional<(NT > 0 && NT <= 32), uint32_t, uint64_t>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int N = NT ? NT : A.N, L = LT ? LT : A.L, D = 4 + 2 * L;
    float4 *s_ring = reinterpret_cast<float4 *>(smem_raw);            // [3][64] {px, py, vx, vy}
    float2 *s_lm = reinterpret_cast<float2 *>(s_ring + 3 * kWave);    // [epw * L]      (wave O only)
    float *s_min = reinterpret_cast<float *>(s_lm + A.epw * L);       // [64] per-landmark min dist (O)
    float *s_rew = s_min + kWave;                                     // [64] per-agent reward      (O)

    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int lane = (int)threadIdx.x & 63;
    int e_local = lane / N;
    int a = lane - e_local * N;
    int env = blockIdx.x * A.epw + e_local;
    if (e_local >= A.epw || env >= A.B) {  // idle lane: shadow lane 0
        e_local = 0; a = 0; env = blockIdx.x * A.epw;
    }
    const int base = e_local * N, me = base + a;
    const uint32_t g = (uint32_t)env * (uint32_t)N + (uint32_t)a;
    const size_t BN = (size_t)A.B * N;
    int ep_step = A.ep_step[env];
    uint32_t ep_count = A.ep_count[env];
    const uint64_t env_id = A.env_id_base + (uint64_t)env;
    int cur = 0;  // ring slot holding this env's current state

    if (wave == 0) {
        // ================================ wave P: physics ================================
        float px = A.pos_x[g], py = A.pos_y[g], vx = A.vel_x[g], vy = A.vel_y[g];
        s_ring[me] = make_float4(px, py, vx, vy);
        wave_lds_sync();
        MaskT near = duo_near_pass<NT, MaskT>(N, a, s_ring + base, px, py, A.near_thr2);
        const float k = A.contact_margin, cf = A.contact_force, dt = A.dt, damp = A.damp, mass = A.mass;
        int act_next = A.act[g];
        PW_STAMP_DECL;
        for (int t = 0; t < T; ++t) {
            PW_STAMP_START;
            const int ai = act_next;
            {
                const int tn = t + 1 < T ? t + 1 : t;
                act_next = A.act[(size_t)tn * BN + g];
            }
            float ux = 0.0f + ((ai == 1 ? 1.0f : 0.0f) - (ai == 2 ? 1.0f : 0.0f));
            float uy = 0.0f + ((ai == 3 ? 1.0f : 0.0f) - (ai == 4 ? 1.0f : 0.0f));
            ux *= A.sens; uy *= A.sens;
            if (A.fscale != 1.0f) { ux = A.fscale * ux; uy = A.fscale * uy; }
            float fx = ux + 0.0f, fy = uy + 0.0f;
            const float4 *pp = s_ring + cur * kWave + base;
            PW_STAMP(0);
            near_force_loop<MaskT, float4>(near, pp, px, py, A.dist_min, k, cf, fx, fy);
            PW_STAMP(1);
            vx = vx * damp; vy = vy * damp;
            vx = vx + div_mass<UNIT_MASS>(fx, mass) * dt;
            vy = vy + div_mass<UNIT_MASS>(fy, mass) * dt;
            px = px + vx * dt;
            py = py + vy * dt;
            int nxt = cur + 1; nxt = nxt == 3 ? 0 : nxt;
            s_ring[nxt * kWave + me] = make_float4(px, py, vx, vy);
            ep_step += 1;
            if (A.auto_reset && A.max_episode_len > 0 && ep_step >= A.max_episode_len) {
                ep_count += 1;
                ep_step = 0;
                pw_reset_xy(A.seed, env_id, ep_count, (uint32_t)a, -1.0f, 1.0f, &px, &py);
                vx = 0.f; vy = 0.f;
                nxt = nxt + 1; nxt = nxt == 3 ? 0 : nxt;
                s_ring[nxt * kWave + me] = make_float4(px, py, 0.f, 0.f);
            }
            cur = nxt;
            PW_STAMP(2);
            duo_barrier();  // slot(s) published; O has finished with the slot P overwrites next
            PW_STAMP(3);
            near = duo_near_pass<NT, MaskT>(N, a, s_ring + cur * kWave + base, px, py, A.near_thr2);
            PW_STAMP(4);
        }
        PW_STAMP_FLUSH;
        A.pos_x[g] = px; A.pos_y[g] = py;
        A.vel_x[g] = vx; A.vel_y[g] = vy;
        A.ep_step[env] = ep_step;
        A.ep_count[env] = ep_count;
    } else {
        // ================================ wave O: outputs ================================
        float2 *lmv = s_lm + e_local * L;
        const int la = a < L ? a : 0;
        float olx = 0.f, oly = 0.f;
        if (L > 0) {
            olx = A.lm_x[(size_t)env * L + la];
            oly = A.lm_y[(size_t)env * L + la];
            lmv[la] = make_float2(olx, oly);
        }
        constexpr int kStoresPerStep = LT > 0 ? 4 + (LT % 2 == 0 ? 1 + LT / 2 : 2 + LT) : 0;
        constexpr int kVm = kStoresPerStep < 63 ? kStoresPerStep : 63;
        PW_STAMP_DECL;
        for (int t = 0; t < T; ++t) {
            const size_t tBN = (size_t)t * BN;
            PW_STAMP_START;
            duo_barrier();
            PW_STAMP(0);
            int nxt = cur + 1; nxt = nxt == 3 ? 0 : nxt;
            const float4 *slot = s_ring + nxt * kWave + base;
            const float4 mine = slot[a];
            float px = mine.x, py = mine.y, vx = mine.z, vy = mine.w;
            MaskT coll = 0;
            float best = 0.0f;
#pragma unroll(NT > 0 ? NT : 1)
            for (int j = 0; j < (NT ? NT : N); ++j) {
                const float2 q = *reinterpret_cast<const float2 *>(slot + j);
                const float dx = q.x - px, dy = q.y - py;
                const float d2 = dx * dx + dy * dy;
                if (d2 < A.coll_thr2) coll |= (MaskT)1 << j;
                const float ex = q.x - olx, ey = q.y - oly;
                const float e2 = ex * ex + ey * ey;
                best = (j == 0 || e2 < best) ? e2 : best;
            }
            s_min[me] = sqrtf(best);
            wave_lds_sync();
            float r = 0.0f;
#pragma unroll(LT > 0 ? LT : 1)
            for (int l = 0; l < (LT ? LT : L); ++l) r -= s_min[base + l];
#pragma unroll(NT > 0 ? NT : 1)
            for (int j = 0; j < (NT ? NT : N); ++j)
                if ((coll >> j) & 1) r -= 1.0f;
            s_rew[me] = r;
            wave_lds_sync();
            float acc = 0.0f;
#pragma unroll(NT > 0 ? NT : 1)
            for (int i = 0; i < (NT ? NT : N); ++i) acc += s_rew[base + i];
            PW_STAMP(1);
            A.rew[tBN + g] = r;
            A.done[tBN + g] = 0;
            A.rew_shared[(size_t)t * A.B + env] = acc;
            ep_step += 1;
            const bool term = A.max_episode_len > 0 && ep_step >= A.max_episode_len;
            A.terminal[(size_t)t * A.B + env] = term ? 1 : 0;
            if (term && A.auto_reset) {
                if (A.final_obs) stream_write_obs<LT>(A.final_obs + (tBN + g) * D, L, lmv, px, py, vx, vy);
                wave_lds_sync();
                ep_count += 1;
                ep_step = 0;
                if (L > 0) {
                    pw_reset_xy(A.seed, env_id, ep_count, (uint32_t)(N + la), -1.0f, 1.0f, &olx, &oly);
                    lmv[la] = make_float2(olx, oly);
                }
                nxt = nxt + 1; nxt = nxt == 3 ? 0 : nxt;
                const float4 fresh = s_ring[nxt * kWave + me];  // post-reset state published by P
                px = fresh.x; py = fresh.y; vx = fresh.z; vy = fresh.w;
            }
            cur = nxt;
            wave_lds_sync();
            stream_write_obs<LT>(A.obs + (tBN + g) * D, L, lmv, px, py, vx, vy);
            PW_STAMP(2);
            if (LT > 0) __builtin_amdgcn_s_waitcnt((kVm & 0xF) | 0x0F70 | ((kVm >> 4) << 14));
            PW_STAMP(3);
        }
#ifdef PW_STAMPS
        if (blockIdx.x == 0 && lane == 0)
            for (int i_ = 0; i_ < 8; ++i_) g_pw_stamps[8 + i_] = st_acc[i_];
#endif
        if (L > 0) {
            A.lm_x[(size_t)env * L + la] = olx;
            A.lm_y[(size_t)env * L + la] = oly;
        }
    }
}

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

// ------------------------------------------------------------------------------------------
// device replay ring (rls/replay_buffer.py ReplayBuffer.add / _encode_sample)
// ------------------------------------------------------------------------------------------
// hipGraph support: a captured launch freezes by-value arguments, so the two values that change from
// step to step (ring position, Philox step) can also be read from device memory and advanced by a
// one-thread kernel that is part of the same graph.
__global__ void pw_counter_add_kernel(int64_t *counter, const int64_t delta, const int64_t modulo)
{
    int64_t v = *counter + delta;
    if (modulo > 0) v %= modulo;
    *counter = v;
}

__global__ void pw_replay_add_kernel(const pw_replay_store st, int64_t start, const int64_t *start_dev, const int B,
                                     const float *obs, const int32_t *act_idx, const float *rew_shared,
                                     const float *next_obs, const float *final_obs, const uint8_t *terminal,
                                     const float *done)
{
    const int ND = st.num_agents * st.obs_dim, N = st.num_agents;
    const size_t total = (size_t)B * ND;
    if (start_dev) start = *start_dev;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i / ND, c = i - e * ND;
        const size_t slot = (size_t)((start + (int64_t)e) % st.capacity);
        st.obs[slot * ND + c] = obs[i];
        const bool fin = final_obs && terminal && terminal[e];
        st.next_obs[slot * ND + c] = fin ? final_obs[i] : next_obs[i];
        if (c < (size_t)N) st.act[slot * N + c] = (uint8_t)act_idx[e * N + c];
        if (c == 0) {
            st.rew[slot] = rew_shared[e];
            st.done[slot] = done ? done[e] : 0.0f;
        }
    }
}

__global__ void pw_replay_gather_kernel(const pw_replay_store st, const int64_t *idx, const int b,
                                        float *out_obs, float *out_act, float *out_rew, float *out_next_obs,
                                        float *out_done)
{
    const int ND = st.num_agents * st.obs_dim, N = st.num_agents;
    const size_t total = (size_t)b * ND;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i / ND, c = i - e * ND;
        const size_t slot = (size_t)idx[e];
        if (out_obs) out_obs[i] = st.obs[slot * ND + c];
        if (out_next_obs) out_next_obs[i] = st.next_obs[slot * ND + c];
        if (out_act && c < (size_t)N * 5) {  // ND >= 5N always (obs_dim >= 6)
            const size_t ag = c / 5, kk = c - ag * 5;
            out_act[e * N * 5 + c] = st.act[slot * N + ag] == kk ? 1.0f : 0.0f;
        }
        if (c == 0) {
            if (out_rew) out_rew[e] = st.rew[slot];
            if (out_done) out_done[e] = st.done[slot];
        }
    }
}

// Transition rows for the multi-GPU exchange: [obs ND | next_obs ND | act N | rew | done], f32.
// Row r is transition (t, e) = (sel_t[r], sel_e[r]) of a rollout chunk, t >= 1: the observation the
// policy acted on is obs[t-1], the stored next observation is the PRE-reset one (run.py:52 vs :60).
__global__ void pw_pack_transitions_kernel(const pw_step_io io, const int B, const int N, const int D,
                                           const int32_t *sel_t, const int32_t *sel_e, const int R, float *rows)
{
    const int ND = N * D, W = 2 * ND + N + 2;
    const size_t total = (size_t)R * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / W), c = (int)(i - (size_t)r * W);
        const int t = sel_t[r], e = sel_e[r];
        const size_t te = (size_t)t * B + e;
        float v;
        if (c < ND) {
            v = io.obs[((size_t)(t - 1) * B + e) * ND + c];
        } else if (c < 2 * ND) {
            const bool fin = io.final_obs && io.terminal && io.terminal[te];
            v = (fin ? io.final_obs : io.obs)[te * ND + (c - ND)];
        } else if (c < 2 * ND + N) {
            v = (float)io.act_idx[te * N + (c - 2 * ND)];
        } else if (c == 2 * ND + N) {
            v = io.rew_shared[te];
        } else {
            v = 0.0f;  // done: upstream done_callback is None
        }
        rows[i] = v;
    }
}

__global__ void pw_replay_add_packed_kernel(const pw_replay_store st, const int64_t start, const int R,
                                            const float *rows)
{
    const int N = st.num_agents, ND = N * st.obs_dim, W = 2 * ND + N + 2;
    const size_t total = (size_t)R * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / W), c = (int)(i - (size_t)r * W);
        const size_t slot = (size_t)((start + r) % st.capacity);
        const float v = rows[i];
        if (c < ND) st.obs[slot * ND + c] = v;
        else if (c < 2 * ND) st.next_obs[slot * ND + (c - ND)] = v;
        else if (c < 2 * ND + N) st.act[slot * N + (c - 2 * ND)] = (uint8_t)v;
        else if (c == 2 * ND + N) st.rew[slot] = v;
        else st.done[slot] = v;
    }
}

// One launch per exchange: blocks [0, nb_in) append the rows received by the PREVIOUS collective to the
// ring, blocks [nb_in, ...) pack this chunk's sampled transitions for the NEXT one (two tiny dependent
// launches would cost more in launch gaps than in work).
__global__ void pw_exchange_kernel(const pw_replay_store st, const int64_t start, const int R_in, const float *rows_in,
                                   const int nb_in, const pw_step_io io, const int B, const int N, const int D,
                                   const int32_t *sel_t, const int32_t *sel_e, const int R_out, float *rows_out)
{
    const int ND = N * D, W = 2 * ND + N + 2;
    if ((int)blockIdx.x < nb_in) {
        const size_t total = (size_t)R_in * W;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)nb_in * blockDim.x) {
            const int r = (int)(i / W), c = (int)(i - (size_t)r * W);
            const size_t slot = (size_t)((start + r) % st.capacity);
            const float v = rows_in[i];
            if (c < ND) st.obs[slot * ND + c] = v;
            else if (c < 2 * ND) st.next_obs[slot * ND + (c - ND)] = v;
            else if (c < 2 * ND + N) st.act[slot * N + (c - 2 * ND)] = (uint8_t)v;
            else if (c == 2 * ND + N) st.rew[slot] = v;
            else st.done[slot] = v;
        }
        return;
    }
    const int nb_out = gridDim.x - nb_in;
    const size_t total = (size_t)R_out * W;
    for (size_t i = (size_t)(blockIdx.x - nb_in) * blockDim.x + threadIdx.x; i < total; i += (size_t)nb_out * blockDim.x) {
        const int r = (int)(i / W), c = (int)(i - (size_t)r * W);
        const int t = sel_t[r], e = sel_e[r];
        const size_t te = (size_t)t * B + e;
        float v;
        if (c < ND) {
            v = io.obs[((size_t)(t - 1) * B + e) * ND + c];
        } else if (c < 2 * ND) {
            const bool fin = io.final_obs && io.terminal && io.terminal[te];
            v = (fin ? io.final_obs : io.obs)[te * ND + (c - ND)];
        } else if (c < 2 * ND + N) {
            v = (float)io.act_idx[te * N + (c - 2 * ND)];
        } else if (c == 2 * ND + N) {
            v = io.rew_shared[te];
        } else {
            v = 0.0f;
        }
        rows_out[i] = v;
    }
}

// ------------------------------------------------------------------------------------------
// Policy forward (rls/model/ac_network_multi_gumbel.py:24-67) pieces that MIOpen serves badly:
// its RNN path issues ~45 tiny kernels for a length-6 sequence (260-400 us per batched step at
// B = 4096, 500x the environment step).  The two dense input GEMMs stay in rocBLAS (they are real
// GEMMs: [B*N, 64] x [64, 256]); the recurrence and the output head + Gumbel sampling are fused here.
//
// pw_bilstm_kernel: lane = (env, direction, hidden unit j), 32 lanes per sequence.  A lane keeps the
// four W_hh rows of its unit (i, f, g, o gates; 128 weights) in VGPRs for the whole kernel, so a
// recurrence step is 128 FMAs + 5 activations per lane; h is exchanged through wave-private LDS
// (one write, eight broadcast ds_read_b128).  G holds x*W_ih^T + b_ih + b_hh for every timestep.
// ------------------------------------------------------------------------------------------
// v_exp_f32 / v_rcp_f32 (1 ulp): the policy net is ordinary float32 inference, not part of the
// bit-exact environment contract; tests compare against PyTorch's float32 LSTM with a 2e-5 bound.
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x)
{
    const float e = __expf(-2.0f * fabsf(x));  // in (0, 1]: no overflow
    const float t = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
    return copysignf(t, x);
}

__global__ void __launch_bounds__(256) pw_bilstm_kernel(const float *__restrict__ G, const float *__restrict__ w_fw,
                                                        const float *__restrict__ w_bw, const int B, const int N,
                                                        const int relu_out, float *__restrict__ H)
{
    // W_hh of both directions staged once per workgroup (32 KB), laid out [dir][gate][k/4][unit] as
    // float4 so that the 32 lanes of a sequence read consecutive 16-B slots (conflict-free); the lanes
    // then keep their 128 weights in VGPRs.  Global weight traffic: 32 KB per workgroup instead of
    // 512 B per lane (4x less), read with fully coalesced float4 loads.
    __shared__ float4 s_w[2 * 4 * 8 * 32];
    __shared__ __attribute__((aligned(16))) float s_h[256];  // [8 sequences per workgroup][32]
    for (int f = threadIdx.x; f < 2048; f += 256) {
        const int d = f >> 10, r = f & 1023, row = r >> 3, q = r & 7, gate = row >> 5, unit = row & 31;
        s_w[((d * 4 + gate) * 8 + q) * 32 + unit] = reinterpret_cast<const float4 *>(d ? w_bw : w_fw)[r];
    }
    __syncthreads();
    const int j = threadIdx.x & 31, grp = threadIdx.x >> 5;   // hidden unit, sequence slot in the workgroup
    const long seq = (long)blockIdx.x * 8 + grp;              // sequence id = env * 2 + dir
    const bool valid = seq < 2L * B;
    const long env = valid ? seq >> 1 : 0;
    const int dir = valid ? (int)(seq & 1) : 0;
    float wi[32], wf[32], wg[32], wo[32];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float4 a = s_w[((dir * 4 + 0) * 8 + q) * 32 + j];
        const float4 b = s_w[((dir * 4 + 1) * 8 + q) * 32 + j];
        const float4 c = s_w[((dir * 4 + 2) * 8 + q) * 32 + j];
        const float4 d = s_w[((dir * 4 + 3) * 8 + q) * 32 + j];
        wi[4 * q] = a.x; wi[4 * q + 1] = a.y; wi[4 * q + 2] = a.z; wi[4 * q + 3] = a.w;
        wf[4 * q] = b.x; wf[4 * q + 1] = b.y; wf[4 * q + 2] = b.z; wf[4 * q + 3] = b.w;
        wg[4 * q] = c.x; wg[4 * q + 1] = c.y; wg[4 * q + 2] = c.z; wg[4 * q + 3] = c.w;
        wo[4 * q] = d.x; wo[4 * q + 1] = d.y; wo[4 * q + 2] = d.z; wo[4 * q + 3] = d.w;
    }
    float h = 0.0f, c = 0.0f;
    float *hs = s_h + grp * 32;
    const float *g0 = G + (((size_t)env * N + (dir ? N - 1 : 0)) * 2 + dir) * 128;
    float ni = g0[j], nf = g0[32 + j], ng = g0[64 + j], no = g0[96 + j];
    for (int s = 0; s < N; ++s) {
        const int t = dir ? N - 1 - s : s;
        float ai = ni, af = nf, ag = ng, ao = no;
        if (s + 1 < N) {  // prefetch the next timestep's pre-activations under this step's FMAs
            const float *g = G + (((size_t)env * N + (dir ? t - 1 : t + 1)) * 2 + dir) * 128;
            ni = g[j]; nf = g[32 + j]; ng = g[64 + j]; no = g[96 + j];
        }
        hs[j] = h;
        wave_lds_sync();  // a sequence's 32 lanes sit in one wave
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 hv = reinterpret_cast<const float4 *>(hs)[q];
            ai = __builtin_fmaf(wi[4 * q], hv.x, ai); af = __builtin_fmaf(wf[4 * q], hv.x, af);
            ag = __builtin_fmaf(wg[4 * q], hv.x, ag); ao = __builtin_fmaf(wo[4 * q], hv.x, ao);
            ai = __builtin_fmaf(wi[4 * q + 1], hv.y, ai); af = __builtin_fmaf(wf[4 * q + 1], hv.y, af);
            ag = __builtin_fmaf(wg[4 * q + 1], hv.y, ag); ao = __builtin_fmaf(wo[4 * q + 1], hv.y, ao);
            ai = __builtin_fmaf(wi[4 * q + 2], hv.z, ai); af = __builtin_fmaf(wf[4 * q + 2], hv.z, af);
            ag = __builtin_fmaf(wg[4 * q + 2], hv.z, ag); ao = __builtin_fmaf(wo[4 * q + 2], hv.z, ao);
            ai = __builtin_fmaf(wi[4 * q + 3], hv.w, ai); af = __builtin_fmaf(wf[4 * q + 3], hv.w, af);
            ag = __builtin_fmaf(wg[4 * q + 3], hv.w, ag); ao = __builtin_fmaf(wo[4 * q + 3], hv.w, ao);
        }
        wave_lds_sync();  // all reads of h done before the next step overwrites it
        c = fast_sigmoid(af) * c + fast_sigmoid(ai) * fast_tanh(ag);
        h = fast_sigmoid(ao) * fast_tanh(c);
        if (valid) H[((size_t)env * N + t) * 64 + dir * 32 + j] = relu_out ? fmaxf(h, 0.0f) : h;
    }
}

// Output head: logits = H * W2^T + b2 (64 -> 5) for one (env, agent) row per lane, then the hard
// Gumbel-softmax sample of ddpg_gumbel_fix.py:109-116 as argmax(logits + g), g = -log(-log(u)),
// u from Philox4x32-10 keyed (seed; step, row) -- the action stays an int32 index in HBM.
__global__ void __launch_bounds__(256) pw_actor_head_kernel(const float *__restrict__ H, const float *__restrict__ w2,
                                                            const float *__restrict__ b2, const long rows,
                                                            const uint64_t seed, uint64_t step,
                                                            const int64_t *__restrict__ step_dev,
                                                            float *__restrict__ logits, int32_t *__restrict__ act)
{
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    if (step_dev) step = (uint64_t)*step_dev;
    float acc[5];
#pragma unroll
    for (int o = 0; o < 5; ++o) acc[o] = b2[o];
    const float4 *h4 = reinterpret_cast<const float4 *>(H + (size_t)r * 64);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const float4 hv = h4[q];
#pragma unroll
        for (int o = 0; o < 5; ++o) {  // w2 indices are uniform: scalar loads
            acc[o] = __builtin_fmaf(w2[o * 64 + 4 * q], hv.x, acc[o]);
            acc[o] = __builtin_fmaf(w2[o * 64 + 4 * q + 1], hv.y, acc[o]);
            acc[o] = __builtin_fmaf(w2[o * 64 + 4 * q + 2], hv.z, acc[o]);
            acc[o] = __builtin_fmaf(w2[o * 64 + 4 * q + 3], hv.w, acc[o]);
        }
    }
    if (logits) {
#pragma unroll
        for (int o = 0; o < 5; ++o) logits[(size_t)r * 5 + o] = acc[o];
    }
    if (act) {
        uint32_t u[8];
        pw_philox4x32_10((uint32_t)r, (uint32_t)((uint64_t)r >> 32), (uint32_t)step, (uint32_t)(step >> 32),
                         (uint32_t)seed, (uint32_t)(seed >> 32), u);
        pw_philox4x32_10((uint32_t)r, (uint32_t)((uint64_t)r >> 32) | 0x80000000u, (uint32_t)step, (uint32_t)(step >> 32),
                         (uint32_t)seed, (uint32_t)(seed >> 32), u + 4);
        int best = 0;
        float bv = 0.0f;
#pragma unroll
        for (int o = 0; o < 5; ++o) {
            const float uo = ((float)(u[o] >> 8) + 0.5f) * 5.9604644775390625e-8f;  // (0, 1)
            const float v = acc[o] - __logf(-__logf(uo));
            if (o == 0 || v > bv) { bv = v; best = o; }
        }
        act[r] = best;
    }
}

// Episode bookkeeping of the rollout loop (experiments/run.py:55-65, vectorised): return += shared
// reward; on terminal the return is added to (sum, count) and cleared.  ONE workgroup with a
// fixed-order tree reduction, so the statistics are bit-reproducible (no float atomics).
__global__ void __launch_bounds__(1024) pw_episode_stats_kernel(const float *rew_shared, const uint8_t *terminal,
                                                                const int B, float *episode_return,
                                                                double *finished_sum, int64_t *finished_count)
{
    __shared__ double s_sum[1024];
    __shared__ int s_cnt[1024];
    double acc = 0.0;
    int cnt = 0;
    for (int e = threadIdx.x; e < B; e += 1024) {
        const float r = episode_return[e] + rew_shared[e];
        if (terminal[e]) { acc += (double)r; cnt += 1; episode_return[e] = 0.0f; }
        else episode_return[e] = r;
    }
    s_sum[threadIdx.x] = acc;
    s_cnt[threadIdx.x] = cnt;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            s_sum[threadIdx.x] += s_sum[threadIdx.x + w];
            s_cnt[threadIdx.x] += s_cnt[threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *finished_sum += s_sum[0];
        *finished_count += s_cnt[0];
    }
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

struct pw_handle {
    pw_config cfg;
    KParams kp;
    pw_state_layout layout;
    bool bound;
    bool fast;      // pw_spread_fast_kernel applies
    FastConsts fc;
};

namespace {

int obs_dim_of(const pw_config &c)
{
    const int N = c.num_agents, L = c.num_landmarks;
    if (c.scenario == PW_SIMPLE_SPREAD) return c.obs_mode == PW_OBS_FULL ? 4 + 2 * L + 4 * (N - 1) : 4 + 2 * L;
    const int G = N - c.num_adversaries;
    return 4 + 2 * L + 2 * (N - 1) + 2 * (c.num_adversaries > 0 ? G : G - 1);
}

template <typename F>
int dispatch(const pw_handle *h, F &&f)
{
    if (h->cfg.scenario == PW_SIMPLE_TAG) return f(std::integral_constant<int, PW_SIMPLE_TAG>(), std::integral_constant<int, PW_OBS_LOCAL>());
    if (h->cfg.obs_mode == PW_OBS_FULL) return f(std::integral_constant<int, PW_SIMPLE_SPREAD>(), std::integral_constant<int, PW_OBS_FULL>());
    return f(std::integral_constant<int, PW_SIMPLE_SPREAD>(), std::integral_constant<int, PW_OBS_LOCAL>());
}

// Decide whether pw_spread_fast_kernel applies and derive its exact thresholds on the host.
void setup_fast_path(pw_handle *h)
{
    const pw_config &c = h->cfg;
    const KParams &kp = h->kp;
    h->fast = false;
    if (std::getenv("PWORLD_FORCE_GENERIC")) return;
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
    // coll_thr2 = min{y : sqrtf(y) >= dist_min}: then sqrtf(d2) < dist_min <=> d2 < coll_thr2
    float y = dmin * dmin;
    while (sqrtf(y) >= dmin) y = std::nextafterf(y, 0.0f);
    while (!(sqrtf(y) >= dmin)) y = std::nextafterf(y, INFINITY);
    fc.coll_thr2 = y;
    // beyond near_r the softplus argument is <= -88 < -87 (pw_exp's exact-zero cut), with margin
    const double near_r = (double)dmin + 88.5 * (double)kp.contact_margin;
    float n2 = (float)(near_r * near_r * (1.0 + 1e-6));
    n2 = std::nextafterf(n2, INFINITY);
    const float dist_at = sqrtf(n2);
    const float x_at = -(dist_at - dmin) / kp.contact_margin;
    if (!(x_at <= -87.5f) || !std::isfinite(n2)) return;
    fc.near_thr2 = n2;
    h->fast = true;
}

int check_ready(const pw_handle *h)
{
    if (!h) return fail(PW_EINVAL, "null handle");
    if (!h->bound) return fail(PW_ESTATE, "state block not bound: call pw_bind_state first");
    return PW_OK;
}

int launch_rollout(pw_handle *h, const pw_step_io *io, int T, void *stream)
{
    if (int rc = check_ready(h)) return rc;
    if (!io) return fail(PW_EINVAL, "null pw_step_io");
    if (T < 1) return fail(PW_EINVAL, "num_steps must be >= 1");
    if ((io->act_idx == nullptr) == (io->act_vec == nullptr))
        return fail(PW_EINVAL, "exactly one of act_idx / act_vec must be given");
    if (io->obs && (reinterpret_cast<uintptr_t>(io->obs) & 15))
        return fail(PW_EINVAL, "obs must be 16-byte aligned");
    if (io->final_obs && (reinterpret_cast<uintptr_t>(io->final_obs) & 15))
        return fail(PW_EINVAL, "final_obs must be 16-byte aligned");
    const KParams &kp = h->kp;
    const dim3 grid((kp.B + kp.epw - 1) / kp.epw), block(kWave);
    const size_t shmem = smem_bytes(kp);
    if (h->fast && io->act_idx && io->obs && io->rew && io->rew_shared && io->done && io->terminal && !io->coll &&
        (size_t)kp.B * kp.N * kp.D * sizeof(float) < (1ull << 31) && !std::getenv("PWORLD_NO_STREAM")) {
        StreamParams A;
        A.B = kp.B; A.N = kp.N; A.L = kp.L; A.epw = kp.epw;
        A.max_episode_len = kp.max_episode_len; A.auto_reset = kp.auto_reset;
        A.seed = kp.seed; A.env_id_base = kp.env_id_base;
        A.dt = kp.dt; A.damp = kp.damp; A.contact_force = kp.contact_force; A.contact_margin = kp.contact_margin;
        A.mass = kp.mass;
        A.dist_min = h->fc.dist_min; A.coll_thr2 = h->fc.coll_thr2; A.near_thr2 = h->fc.near_thr2;
        A.sens = h->fc.sens; A.fscale = h->fc.fscale;
        A.pos_x = kp.pos_x; A.pos_y = kp.pos_y; A.vel_x = kp.vel_x; A.vel_y = kp.vel_y;
        A.lm_x = kp.lm_x; A.lm_y = kp.lm_y; A.ep_step = kp.ep_step; A.ep_count = kp.ep_count;
        A.act = io->act_idx; A.obs = io->obs; A.final_obs = io->final_obs; A.rew = io->rew;
        A.rew_shared = io->rew_shared; A.done = io->done; A.terminal = io->terminal;
        hipStream_t st = static_cast<hipStream_t>(stream);
        const size_t shm = (size_t)(kWave + kp.epw * kp.L) * sizeof(float2);
        const int key = kp.N == kp.L ? kp.N : 0;
        const bool um = kp.mass == 1.0f;
        // two cooperating waves per env group pay off while the chip is latency bound (few workgroups
        // per CU); once every SIMD holds several waves the single-wave kernel issues fewer instructions
        const bool duo = grid.x <= 8192 && !std::getenv("PWORLD_NO_DUO");
        if (duo || std::getenv("PWORLD_FORCE_DUO")) {
            const size_t shm2 = 3 * kWave * sizeof(float4) + (size_t)kp.epw * kp.L * sizeof(float2) +
                                2 * kWave * sizeof(float);
            const dim3 block2(2 * kWave);
            switch (key) {
#define PW_DUO_CASE(n)                                                                                       \
    case n:                                                                                                  \
        if (um) hipLaunchKernelGGL((pw_spread_duo_kernel<n, n, true>), grid, block2, shm2, st, A, T);        \
        else hipLaunchKernelGGL((pw_spread_duo_kernel<n, n, false>), grid, block2, shm2, st, A, T);          \
        break;
                PW_DUO_CASE(3) PW_DUO_CASE(6) PW_DUO_CASE(9) PW_DUO_CASE(12)
#undef PW_DUO_CASE
            default:
                if (um) hipLaunchKernelGGL((pw_spread_duo_kernel<0, 0, true>), grid, block2, shm2, st, A, T);
                else hipLaunchKernelGGL((pw_spread_duo_kernel<0, 0, false>), grid, block2, shm2, st, A, T);
            }
            PW_HIP_CHECK(hipGetLastError());
            return PW_OK;
        }
        switch (key) {
#define PW_STREAM_CASE(n)                                                                                    \
    case n:                                                                                                  \
        if (um) hipLaunchKernelGGL((pw_spread_stream_kernel<n, n, true>), grid, block, shm, st, A, T);       \
        else hipLaunchKernelGGL((pw_spread_stream_kernel<n, n, false>), grid, block, shm, st, A, T);         \
        break;
            PW_STREAM_CASE(3) PW_STREAM_CASE(6) PW_STREAM_CASE(9) PW_STREAM_CASE(12)
#undef PW_STREAM_CASE
        default:
            if (um) hipLaunchKernelGGL((pw_spread_stream_kernel<0, 0, true>), grid, block, shm, st, A, T);
            else hipLaunchKernelGGL((pw_spread_stream_kernel<0, 0, false>), grid, block, shm, st, A, T);
        }
        PW_HIP_CHECK(hipGetLastError());
        return PW_OK;
    }
    if (h->fast) {
        hipStream_t st = static_cast<hipStream_t>(stream);
        switch (kp.N) {
#define PW_FAST_CASE(n)                                                                                      \
    case n:                                                                                                  \
        hipLaunchKernelGGL((pw_spread_fast_kernel<n>), grid, block, shmem, st, kp, *io, T, h->fc);           \
        break;
            PW_FAST_CASE(3) PW_FAST_CASE(6) PW_FAST_CASE(9) PW_FAST_CASE(12)
#undef PW_FAST_CASE
        default:
            hipLaunchKernelGGL((pw_spread_fast_kernel<0>), grid, block, shmem, st, kp, *io, T, h->fc);
        }
        PW_HIP_CHECK(hipGetLastError());
        return PW_OK;
    }
    return dispatch(h, [&](auto scen, auto obs) {
        hipLaunchKernelGGL((pw_rollout_kernel<decltype(scen)::value, decltype(obs)::value>), grid, block, shmem,
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

extern "C" {

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
    } else {
        return fail(PW_EINVAL, "unknown scenario");
    }
    return PW_OK;
}

int pw_create(const pw_config *cfg, pw_handle **out)
{
    if (!cfg || !out) return fail(PW_EINVAL, "null argument");
    if (cfg->struct_size != sizeof(pw_config)) return fail(PW_EINVAL, "pw_config.struct_size mismatch (ABI)");
    if (cfg->scenario != PW_SIMPLE_SPREAD && cfg->scenario != PW_SIMPLE_TAG) return fail(PW_EINVAL, "unknown scenario");
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
    KParams &kp = h->kp;
    std::memset(&kp, 0, sizeof(kp));
    kp.B = cfg->num_envs; kp.N = cfg->num_agents; kp.L = cfg->num_landmarks;
    kp.A = cfg->scenario == PW_SIMPLE_TAG ? cfg->num_adversaries : 0;
    kp.D = obs_dim_of(*cfg);
    kp.epw = kWave / kp.N;
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
    setup_fast_path(h);
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
    if (st->capacity < 1 || B < 1 || B > st->capacity || start < 0) return fail(PW_EINVAL, "bad ring arguments");
    if (st->obs_dim < 5) return fail(PW_EINVAL, "obs_dim must be >= 5");
    const size_t total = (size_t)B * st->num_agents * st->obs_dim;
    size_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(pw_replay_add_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       *st, start, start_dev, B, obs, act_idx, rew_shared, next_obs, final_obs, terminal, done);
    PW_HIP_CHECK(hipGetLastError());
    return PW_OK;
}

int pw_replay_gather(const pw_replay_store *st, const int64_t *idx, int32_t b, float *out_obs, float *out_act,
                     float *out_rew, float *out_next_obs, float *out_done, void *stream)
{
    if (!st || !idx) return fail(PW_EINVAL, "null argument");
    if (b < 1) return fail(PW_EINVAL, "batch must be >= 1");
    if (st->obs_dim < 5) return fail(PW_EINVAL, "obs_dim must be >= 5");
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

int pw_episode_stats(const float *rew_shared, const uint8_t *terminal, int32_t B, float *episode_return,
                     double *finished_sum, int64_t *finished_count, void *stream)
{
    if (!rew_shared || !terminal || !episode_return || !finished_sum || !finished_count)
        return fail(PW_EINVAL, "null argument");
    if (B < 1) return fail(PW_EINVAL, "bad sizes");
    hipLaunchKernelGGL(pw_episode_stats_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), rew_shared,
                       terminal, B, episode_return, finished_sum, finished_count);
    PW_HIP_CHECK(hipGetLastError());
    return PW_OK;
}

}  // extern "C"
