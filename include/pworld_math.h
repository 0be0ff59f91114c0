/* pworld_math.h -- the deterministic float32 primitives libpworld's kernels use.
 *
 * Published so that a CPU implementation can reproduce the device results bit
 * for bit: only IEEE + - * / (round-to-nearest-even), fmaf (the correctly rounded
 * fused multiply-add of C99: one rounding), floorf, int<->float conversion and
 * bit casts; compile WITHOUT implicit FMA contraction (-ffp-contract=off: a fused
 * operation happens exactly where fmaf is written, nowhere else) and with correctly
 * rounded division.
 *
 * They replace, inside upstream get_collision_force / simple_tag's bound():
 *   np.logaddexp(0, x)  -> pw_softplus(x)
 *   np.exp(x)           -> pw_exp(x)
 *
 * Revision 3 of the contract (revision 1: separate * and +; revision 2: fmaf steps,
 * Horner polynomials, log1p through t / (2 + t)).  The batched env step is bound by
 * the LENGTH of the dependent instruction chain of one contact pair, not by the
 * number of instructions, so revision 3 states the same functions as short chains:
 *   - polynomials in Estrin form (independent pairs, then powers of two of the
 *     argument): 3 dependent fmaf levels instead of 7;
 *   - log1p(t) on [0, 1] as t + t^2 Q(t), Q a degree-7 minimax polynomial of
 *     (log1p(t) - t) / t^2 -- no division (revision 2's t / (2 + t) was a 10-deep
 *     chain on its own), and the leading term t is exact, so the result is rounded
 *     once at the end;
 *   - pw_softplus adds max(x, 0) into that last fused step.
 * A contact pair's softplus is 15 dependent operations deep (revision 2: 42).
 * Accuracy against float64 libm, measured by oracle/math_accuracy.c over every
 * float32 argument of the ranges used: see the table that tool prints
 * (profiles/r2_math_accuracy.txt): pw_exp <= 1.8e-7 relative (1.5 ulp), pw_log1p01
 * <= 1.7e-7, pw_softplus <= 2.6e-7 (revision 2 claimed 3e-7).
 */
#ifndef PWORLD_MATH_H
#define PWORLD_MATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define PW_HD __host__ __device__ __forceinline__
#else
#include <math.h>
#include <string.h>
#define PW_HD static inline
#endif

/* Q's coefficients (float32 values, written with 9 significant digits) */
#define PW_LQ0 (-4.999969006e-01f)
#define PW_LQ1 (3.332236707e-01f)
#define PW_LQ2 (-2.486616373e-01f)
#define PW_LQ3 (1.919044554e-01f)
#define PW_LQ4 (-1.383424997e-01f)
#define PW_LQ5 (8.017139137e-02f)
#define PW_LQ6 (-3.066807054e-02f)
#define PW_LQ7 (5.516789388e-03f)

PW_HD float pw_bits_to_float(uint32_t u)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f;
    memcpy(&f, &u, 4);
    return f;
#endif
}

/* exp(x). x <= -87 (or NaN < test fails) -> +0 exactly, so no subnormal is ever
 * produced; x >= 88 -> 2^127. Cody-Waite reduction by ln2, degree-7 Taylor in
 * Estrin form. */
PW_HD float pw_exp(float x)
{
    if (!(x > -87.0f)) return x != x ? x : 0.0f;
    if (x >= 88.0f) return pw_bits_to_float(0x7f000000u);
    float n = floorf(fmaf(x, 1.44269504088896341f, 0.5f));
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440054690583e-4f, r);
    float r2 = r * r;
    float r4 = r2 * r2;
    float p01 = 1.0f + r;
    float p23 = fmaf(1.66666666666666657e-1f, r, 0.5f);
    float p45 = fmaf(8.33333333333333322e-3f, r, 4.16666666666666644e-2f);
    float p67 = fmaf(1.98412698412698413e-4f, r, 1.38888888888888894e-3f);
    float lo = fmaf(p23, r2, p01);
    float hi = fmaf(p67, r2, p45);
    float p = fmaf(hi, r4, lo);
    int32_t e = (int32_t)n + 127;
    return p * pw_bits_to_float((uint32_t)e << 23);
}

/* Q(t) ~ (log1p(t) - t) / t^2 on [0, 1] (degree-7 minimax, weight t), Estrin form. */
PW_HD float pw_log1p01_q(float t)
{
    float t2 = t * t;
    float t4 = t2 * t2;
    float q01 = fmaf(PW_LQ1, t, PW_LQ0);
    float q23 = fmaf(PW_LQ3, t, PW_LQ2);
    float q45 = fmaf(PW_LQ5, t, PW_LQ4);
    float q67 = fmaf(PW_LQ7, t, PW_LQ6);
    float lo = fmaf(q23, t2, q01);
    float hi = fmaf(q67, t2, q45);
    return fmaf(hi, t4, lo);
}

/* log1p(t) for t in [0, 1]: t + t^2 Q(t).  log1p(0) = +0 exactly. */
PW_HD float pw_log1p01(float t)
{
    return fmaf(t * t, pw_log1p01_q(t), t);
}

/* logaddexp(0, x) = max(x, 0) + log1p(exp(-|x|)), the max added inside the last fused
 * step: fmaf(t^2, Q(t), t + max(x, 0)) with t = exp(-|x|); exactly +0 for x <= -87,
 * exactly x for x >= 87. */
PW_HD float pw_softplus(float x)
{
    float m = x > 0.0f ? x : 0.0f;
    float t = pw_exp(-fabsf(x));
    return fmaf(t * t, pw_log1p01_q(t), t + m);
}

/* Philox4x32-10 (Salmon, Moraes, Dror, Shaw; SC'11). */
PW_HD void pw_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                            uint32_t k0, uint32_t k1, uint32_t out[4])
{
    for (int i = 0; i < 10; ++i) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* initial (x, y) of one entity of one env for one episode: uniform on [lo, hi) */
PW_HD void pw_reset_xy(uint64_t seed, uint64_t env_id, uint32_t episode, uint32_t entity,
                       float lo, float hi, float *x, float *y)
{
    uint32_t r[4];
    pw_philox4x32_10(entity, episode, (uint32_t)env_id, (uint32_t)(env_id >> 32),
                     (uint32_t)seed, (uint32_t)(seed >> 32), r);
    float span = hi - lo;
    float u0 = (float)(r[0] >> 8) * 5.9604644775390625e-8f;
    float u1 = (float)(r[1] >> 8) * 5.9604644775390625e-8f;
    *x = span * u0 + lo;
    *y = span * u1 + lo;
}

#endif /* PWORLD_MATH_H */
