/* pworld_math.h -- the deterministic float32 primitives libpworld's kernels use.
 *
 * Published so that a CPU implementation can reproduce the device results bit
 * for bit: only IEEE + - * / (round-to-nearest-even), fmaf (the correctly rounded
 * fused multiply-add of C99: one rounding), floorf, int<->float conversion and
 * bit casts; compile WITHOUT implicit FMA contraction (-ffp-contract=off: a fused
 * operation happens exactly where fmaf is written, nowhere else) and with correctly
 * rounded division.  (Revision 2 of the contract: the polynomial and range-reduction
 * steps of pw_exp / pw_log1p01 are fmaf steps -- 16 instructions fewer per contact
 * pair on the GPU, a little more accurate; revision 1 used separate * and +.)
 *
 * They replace, inside upstream get_collision_force / simple_tag's bound():
 *   np.logaddexp(0, x)  -> pw_softplus(x)
 *   np.exp(x)           -> pw_exp(x)
 * Accuracy: <= 3e-7 relative against float64 libm over the ranges used.
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
 * produced; x >= 88 -> 2^127. Cody-Waite reduction by ln2, degree-7 Taylor. */
PW_HD float pw_exp(float x)
{
    if (!(x > -87.0f)) return x != x ? x : 0.0f;
    if (x >= 88.0f) return pw_bits_to_float(0x7f000000u);
    float n = floorf(fmaf(x, 1.44269504088896341f, 0.5f));
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440054690583e-4f, r);
    float p = 1.98412698412698413e-4f;
    p = fmaf(p, r, 1.38888888888888894e-3f);
    p = fmaf(p, r, 8.33333333333333322e-3f);
    p = fmaf(p, r, 4.16666666666666644e-2f);
    p = fmaf(p, r, 1.66666666666666657e-1f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    int32_t e = (int32_t)n + 127;
    return p * pw_bits_to_float((uint32_t)e << 23);
}

/* log1p(t) for t in [0, 1]: 2 atanh(t / (2 + t)), 8-term odd series. */
PW_HD float pw_log1p01(float t)
{
    float s = t / (2.0f + t);
    float z = s * s;
    float q = 6.66666666666666657e-2f;
    q = fmaf(q, z, 7.69230769230769273e-2f);
    q = fmaf(q, z, 9.09090909090909116e-2f);
    q = fmaf(q, z, 1.11111111111111105e-1f);
    q = fmaf(q, z, 1.42857142857142849e-1f);
    q = fmaf(q, z, 0.2f);
    q = fmaf(q, z, 3.33333333333333315e-1f);
    q = fmaf(q, z, 1.0f);
    return 2.0f * s * q;
}

/* logaddexp(0, x) = max(x, 0) + log1p(exp(-|x|)); exactly 0 for x <= -87. */
PW_HD float pw_softplus(float x)
{
    float ax = x < 0.0f ? -x : x;
    float m = x > 0.0f ? x : 0.0f;
    return m + pw_log1p01(pw_exp(-ax));
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
