// pw_common.hpp -- part of libpworld.so (one translation unit: csrc/pworld.hip includes it).
// Error plumbing, kernel parameter blocks, lane mapping, the shared device functions (collision force, observation, reward/masks, reset).
#pragma once

// libpworld.so is two translation units (pworld.hip: environment, replay ring, wire blocks; pworld_policy.hip: actor and
// policy-in-the-loop rollouts).  The thread-local error text lives in pworld.hip; both reach it through this hook.
extern "C" __attribute__((visibility("hidden"))) void pw_internal_set_error(const char *msg);

namespace {

int fail(int code, const std::string &msg)
{
    pw_internal_set_error(msg.c_str());
    return code;
}

#define PW_HIP_CHECK(expr)                                                             \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess)                                                          \
            return fail(PW_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e));   \
    } while (0)

constexpr int kWave = 64;

}  // namespace
// Timing-only overlays (wrong results by design) are NOT product source: tools/experiments/pw_experiments.hpp, reachable only from
// a tools/ build that passes -DPW_EXPERIMENTS -I tools/experiments.  build_native.py never does; the product sees the defaults below.
#ifdef PW_EXPERIMENTS
#include "pw_experiments.hpp"
#endif
#ifndef PW_NEAR_MASK_HOOK
#define PW_NEAR_MASK_HOOK(m) do { } while (0)
#endif
namespace {

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
// ~22 instructions on scaling subnormal-range inputs and on the 0 / inf / NaN pass-through, and corrects the hardware
// estimate with two residual tests and two selects.  For x in [2^-90, 2^90) neither wrapper is needed, and ONE fused
// correction is exact: g = v_sqrt_f32(x) is within 1 ulp, h = 0.5 * v_rsq_f32(x) is computed beside it, and
//   s = RN(g + (x - g^2) * h)          (x - g^2 exact in the fma)
// is the correctly rounded root -- Markstein's final step for sqrt -- with two dependent operations behind the hardware
// estimate instead of five.  It is not taken on trust: tests/test_gpu_parity.py compares it with sqrtf over EVERY float32
// of the range (1.5 x 10^9 arguments, tools/sqrt_probe.py: 0 mismatches on gfx950), and sqrt_rn_core_tests, the two-test
// form it replaced, stays behind pw_debug_math for that comparison.  Anything outside the range (never reached from finite,
// non-coincident states) takes the general sqrtf.
__device__ __forceinline__ float sqrt_rn_core(float x)  // x in [2^-90, 2^90)
{
    const float g = __builtin_amdgcn_sqrtf(x);
    const float h = 0.5f * __builtin_amdgcn_rsqf(x);
    return __builtin_fmaf(__builtin_fmaf(-g, g, x), h, g);
}
__device__ __forceinline__ float sqrt_rn_core_tests(float x)  // the round-1/2 form: v_sqrt_f32 + two residual tests
{
    float s = __builtin_amdgcn_sqrtf(x);
    const float s_dn = __uint_as_float(__float_as_uint(s) - 1u);
    const float s_up = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_dn = __builtin_fmaf(-s_dn, s, x);
    const float r_up = __builtin_fmaf(-s_up, s, x);
    s = r_dn <= 0.0f ? s_dn : s;
    s = r_up > 0.0f ? s_up : s;
    return s;
}

__device__ __forceinline__ float sqrt_rn_fast(float x)
{
    if (__builtin_expect(!(x >= 8.077935669463161e-28f && x < 1.2379400392853803e+27f), 0)) return sqrtf(x);
    return sqrt_rn_core(x);
}

// IEEE float32 division without its scaling wrapper.  The compiler expands a / b into
//   v_div_scale x2, v_rcp, y = fma(fma(-b, rcp, 1), rcp, rcp), q0 = a*y, r0 = fma(-b, q0, a), q1 = fma(r0, y, q0),
//   r1 = fma(-b, q1, a), v_div_fmas (= fma(r1, y, q1)), v_div_fixup
// where v_div_scale / v_div_fmas only act when an operand or the quotient is within ~2^-100 / 2^96 of the
// float32 range limits, and v_div_fixup only on zeros, infinities and NaNs.  Away from those cases the quotient
// IS the bare FMA chain below -- bit for bit, it is the same instruction sequence -- so callers that know their
// operand ranges run the chain directly: 5 instructions per quotient (packed: 5 for two quotients) plus 3 per
// distinct divisor, instead of 11-12 per division.  Required: b in [2^-45, 2^45], a in [2^-69, 2^52] in
// magnitude (then neither the quotient nor any residual leaves the normal range), or a == +-0 where the sign
// of the zero result does not matter to the caller.  tests: pw_debug_math fn 6 / 7 against IEEE division.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float div_refined_rcp(float b)
{
    const float y0 = __builtin_amdgcn_rcpf(b);
    return __builtin_fmaf(__builtin_fmaf(-b, y0, 1.0f), y0, y0);
}
__device__ __forceinline__ float div_chain(float a, float b, float y)
{
    const float q0 = a * y;
    const float q1 = __builtin_fmaf(__builtin_fmaf(-b, q0, a), y, q0);
    return __builtin_fmaf(__builtin_fmaf(-b, q1, a), y, q1);
}
// a / b with ONE correction step: q0 = RN(a y), q1 = RN(q0 + (a - b q0) y).  With y = RN(1 / b) EXACTLY (the correctly
// rounded reciprocal) and q0 within one ulp of a / b, q1 is the correctly rounded quotient (Markstein, "Computation of
// elementary functions on the IBM RISC System/6000 processor", 1990, Theorem 4; the exception -- a significand of b of
// all ones -- is excluded by the caller).  The caller checks both conditions on the host for the one divisor this is
// used with, the wave-uniform contact margin (pworld.hip margin_one_correction); the same operand ranges as div_chain
// apply.  tests: pw_debug_math fn 10 against IEEE division over EVERY float32 significand, each exponent of the range.
__device__ __forceinline__ float div_chain1(float a, float b, float y)
{
    const float q0 = a * y;
    return __builtin_fmaf(__builtin_fmaf(-b, q0, a), y, q0);
}
__device__ __forceinline__ f32x2 div_chain2(f32x2 a, float b, float y)
{
    const f32x2 nb = {-b, -b}, yy = {y, y};
    const f32x2 q0 = a * yy;
    const f32x2 q1 = __builtin_elementwise_fma(__builtin_elementwise_fma(nb, q0, a), yy, q0);
    return __builtin_elementwise_fma(__builtin_elementwise_fma(nb, q1, a), yy, q1);
}

// ---- streaming stores -----------------------------------------------------------------------------------------------
// The rollout's outputs are written once and not read again by the GPU for a long time (the consumer is the learner, a
// chunk later): stored with the non-temporal hint they do not displace each other in L2 / MALL on their way out.
// Measured (tools/step_time.hip A/B, profiles/r2_nt_stores.txt): the block-wise observation stores with the hint: B = 65536,
// N = 6: +6.5 %, N = 24 at B = 4096: +6 %, C2: +1 %.  The row-per-lane observation stores write a quarter line each and
// stay plain.
#ifndef PW_HAVE_EXP_NT_STORE   // (a -DPW_EXPERIMENTS timing build may have replaced them: tools/experiments/pw_experiments.hpp)
template <typename T>
__device__ __forceinline__ void nt_store(T *p, const T v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void nt_store(float4 *p, const float4 v)
{
    typedef float nt4 __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(nt4{v.x, v.y, v.z, v.w}, reinterpret_cast<nt4 *>(p));
}
__device__ __forceinline__ void nt_store(float2 *p, const float2 v)
{
    typedef float nt2 __attribute__((ext_vector_type(2)));
    __builtin_nontemporal_store(nt2{v.x, v.y}, reinterpret_cast<nt2 *>(p));
}
#endif

// The per-agent / per-env planes take the hint only at N <= 6 (kNtPlanes in the spread kernels): C2 +2 %, B = 65536 +4 % on
// top of the block stores, but N = 12: -2 %, N = 24: -3 %.

// ---- action indices fetched AHEAD by LDS-direct loads ---------------------------------------------------------------
// A physics wave reads one action index per lane and step from HBM.  Issued one step ahead into a register, that load
// costs a full chip 250-400 cycles of every step (stamps: profiles/r2_action_prefetch.txt): HBM latency under a
// 3-4 TB/s write stream exceeds a step.  Fetching further ahead into registers does not work -- rotating pending
// registers through an unrolled loop makes the compiler copy them at the back edge, and a copy of a pending load waits
// for it -- so the loads write LDS directly (global_load_lds_dword: destination = M0 + 4 * lane, no register), four steps
// ahead, into a four-slot ring of the wave.  The compiler does not count these loads; the wave must have NO other vector
// memory operation in its step loop (the physics waves of the duo / quad kernels have none: the output waves do the
// stores), must have consumed every counted load before the first fetch, and writes its waits out: vmcnt(3) at the
// top of step t leaves the fetches of steps t+1 .. t+3 in flight.
__device__ __forceinline__ void act_fetch_issue(const int32_t *src_lane, const uint32_t lds_slot)
{
    unsigned keep;
    // every earlier LDS read of this wave has returned (the slot's previous content was read four steps ago, but the
    // order is cheap to guarantee); M0 is the compiler's: saved and restored inside the statement
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src_lane), "s"(lds_slot) : "memory");
}
__device__ __forceinline__ void act_fetch_wait3() { asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); }
__device__ __forceinline__ void act_fetch_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
constexpr int kActRingBytes = 4 * kWave * (int)sizeof(int32_t);  // per physics wave

// pw_softplus (include/pworld_math.h, contract revision 3): the same operations, so the same bits, but branch-free --
// the polynomial runs on a clamped argument and the exact-zero cut / NaN pass-through ride on the power-of-two scale
// factor, which is computed beside the polynomial, not after it.  The dependent chain from x to the result is 15
// operations: clamp, fma, floor, 2 reduction fmas, 3 Estrin levels, scale, 3 Estrin levels of Q, the final fma (revision
// 2: 42, a third of them the division inside log1p).
__device__ __forceinline__ float softplus_branchless(float x)
{
    const float m = x > 0.0f ? x : 0.0f;
    const float t0 = -__builtin_fabsf(x);        // <= 0, or NaN; a source modifier of the clamp below, not an instruction
    // keep the exponent arithmetic in range when cut (and for NaN): max(t0, -87) IS "t0 > -87 ? t0 : -87" for every input
    // (v_max_f32 returns the other operand for a NaN; -0 > -87), one instruction instead of compare + select on the chain
    const float tc = __builtin_fmaxf(t0, -87.0f);
    const float n = floorf(__builtin_fmaf(tc, 1.44269504088896341f, 0.5f));
    float r = __builtin_fmaf(n, -0.693359375f, tc);
    r = __builtin_fmaf(n, 2.12194440054690583e-4f, r);
    // pw_exp: x <= -87 -> +0, NaN -> NaN: as the scale factor (p is finite and positive on the clamped argument)
    const int32_t e = (int32_t)n + 127;
    float scale = __uint_as_float((uint32_t)e << 23);
    scale = t0 > -87.0f ? scale : (t0 != t0 ? t0 : 0.0f);
    const float r2 = r * r, r4 = r2 * r2;
    const float p01 = 1.0f + r;
    const float p23 = __builtin_fmaf(1.66666666666666657e-1f, r, 0.5f);
    const float p45 = __builtin_fmaf(8.33333333333333322e-3f, r, 4.16666666666666644e-2f);
    const float p67 = __builtin_fmaf(1.98412698412698413e-4f, r, 1.38888888888888894e-3f);
    const float p = __builtin_fmaf(__builtin_fmaf(p67, r2, p45), r4, __builtin_fmaf(p23, r2, p01));
    const float t = p * scale;
    // pw_log1p01's Q and the fused last step
    const float t2 = t * t, t4 = t2 * t2;
    const float q01 = __builtin_fmaf(PW_LQ1, t, PW_LQ0);
    const float q23 = __builtin_fmaf(PW_LQ3, t, PW_LQ2);
    const float q45 = __builtin_fmaf(PW_LQ5, t, PW_LQ4);
    const float q67 = __builtin_fmaf(PW_LQ7, t, PW_LQ6);
    const float q = __builtin_fmaf(__builtin_fmaf(q67, t2, q45), t4, __builtin_fmaf(q23, t2, q01));
    return __builtin_fmaf(t2, q, t + m);
}

// get_collision_force seen from entity i against entity j: force on i.
// delta = p_i - p_j; dist = sqrt(sum(delta^2)); pen = logaddexp(0, -(dist - dist_min)/k) * k;
// force = contact_force * delta / dist * pen.  (The force on the pair's second entity is
// the exact negation, which is what this evaluates to from that entity's side.)
// FAST: the same IEEE results through sqrt_rn_core / div_chain when every operand is provably in their range
// (one range test per pair); anything else -- coincident or astronomically distant entities, a coordinate
// difference below 2^-60, an exotic contact margin, NaN / inf -- takes the general expressions.
// collision_force_pair: the pair force (Fx, Fy) on the first entity itself; collision_force adds it to the accumulator
// (F + f, the upstream order of operands).  The force on the second entity is the exact negation: every operation
// below is odd in delta.
template <bool FAST = false, bool K1 = false>
__device__ __forceinline__ void collision_force_pair(float px, float py, float qx, float qy, float dist_min,
                                                     float k, float cf, float &Fx, float &Fy)
{
    const float dx = px - qx, dy = py - qy;
    const float d2 = dx * dx + dy * dy;
    if (FAST) {
        const f32x2 a = {cf * dx, cf * dy};
        // one range test per pair, evaluated without short-circuit branches: d2 in [2^-90, 2^90) on the raw bits
        // (d2 is a sum of squares: never negative; NaN fails), both numerators >= 2^-60 in magnitude (they are
        // < 2^52 because |delta| < 2^45 and contact_force <= 128), and the wave-uniform contact parameters
        const bool uni = (k >= 9.094947017729282e-13f) & (k <= 1099511627776.0f) & (cf >= 9.5367431640625e-07f) & (cf <= 128.0f);
        const bool c1 = (__float_as_uint(d2) - 0x12800000u) < (0x6C800000u - 0x12800000u);
        const bool c2 = fminf(fabsf(a.x), fabsf(a.y)) >= 8.673617379884035e-19f;
        const bool in_range = c1 & c2 & uni;
        if (__builtin_expect(in_range, 1)) {
            const float dist = sqrt_rn_core(d2);
            const float xarg = K1 ? div_chain1(-(dist - dist_min), k, div_refined_rcp(k))
                                  : div_chain(-(dist - dist_min), k, div_refined_rcp(k));
            const float pen = softplus_branchless(xarg) * k;
            const f32x2 F = div_chain2(a, dist, div_refined_rcp(dist)) * f32x2{pen, pen};
            Fx = F.x;
            Fy = F.y;
            return;
        }
    }
    const float dist = FAST ? sqrt_rn_fast(d2) : sqrtf(d2);
    const float xarg = -(dist - dist_min) / k;
    const float pen = (FAST ? softplus_branchless(xarg) : pw_softplus(xarg)) * k;
    Fx = cf * dx / dist * pen;
    Fy = cf * dy / dist * pen;
}

template <bool FAST = false>
__device__ __forceinline__ void collision_force(float px, float py, float qx, float qy, float dist_min,
                                                float k, float cf, float &fx, float &fy)
{
    float Fx, Fy;
    collision_force_pair<FAST>(px, py, qx, qy, dist_min, k, cf, Fx, Fy);
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


size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace
