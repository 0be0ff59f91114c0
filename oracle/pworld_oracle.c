/* CPU ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the batched particle world (simple_spread,
 * simple_tag, simple_reference, simple_speaker_listener) in two precisions:
 *   *_f64  NumPy-float64 semantics of the canonical upstream `multiagent`
 *          package the reference imports (experiments/scenarios.py:2-3) --
 *          libm exp/log1p, i.e. np.logaddexp's stable form.
 *   *_f32  the SAME operation order in IEEE float32 with a deterministic,
 *          libm-free softplus/exp (only + - * / sqrt and explicit fmaf steps -- include/pworld_math.h
 *          revision 3 --, no implicit FMA contraction), so
 *          the HIP kernels can be compared BIT FOR BIT, integer collision
 *          masks included.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  Nothing under multiagent_rl_amd/ links or calls it.
 *
 * PARITY UNPINNED: the reference holds no golden vectors for this path and the
 * arithmetic lives in an un-vendored, un-pinned third-party package
 * (SURVEY.md section 0, 8(c)).  This file is validated against
 * oracle/particle_oracle.py (the upstream-structured scalar NumPy
 * restatement) and hand-derived known-answer tests (tests/test_oracle_kat.py).
 *
 * Build (see oracle/Makefile): gcc -O2 -ffp-contract=off -fno-fast-math
 *                              -fPIC -shared -o libpworld_oracle.so pworld_oracle.c -lm
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include <string.h>

#define PO_EXPORT __attribute__((visibility("default")))
#define PO_MAX_AGENTS 64
#define PO_MAX_ENTITIES 192
enum { PO_SIMPLE_SPREAD = 0, PO_SIMPLE_TAG = 1, PO_SIMPLE_REFERENCE = 2, PO_SIMPLE_SPEAKER_LISTENER = 3 };
#define PO_DIM_C 10   /* simple_reference: world.dim_c */
#define PO_SL_DIM_C 3 /* simple_speaker_listener: world.dim_c */
enum { PO_OBS_LOCAL = 0, PO_OBS_FULL = 1 };

typedef struct po_config {
    int32_t scenario;
    int32_t num_agents;
    int32_t num_landmarks;
    int32_t num_adversaries;       /* simple_tag: agents [0, A) are adversaries */
    int32_t obs_mode;
    int32_t max_episode_len;       /* rls/arglist.py:5 -> 25; 0 = never terminal */
    int32_t auto_reset;
    int32_t force_discrete_action; /* experiments/scenarios.py:191 */
    int32_t landmark_collide;
    int32_t action_force_uses_accel; /* fork knob, canonical 0 */
    uint64_t seed;
    uint64_t env_id_base;
    double dt, damping, contact_force, contact_margin, default_sensitivity, mass, landmark_size;
    double agent_size[PO_MAX_AGENTS];
    double agent_accel[PO_MAX_AGENTS];     /* < 0 : None */
    double agent_max_speed[PO_MAX_AGENTS]; /* < 0 : None */
} po_config;

PO_EXPORT int po_config_size(void) { return (int)sizeof(po_config); }

PO_EXPORT int po_obs_dim(const po_config *c)
{
    const int N = c->num_agents, L = c->num_landmarks;
    if (c->scenario == PO_SIMPLE_SPREAD)
        return c->obs_mode == PO_OBS_FULL ? 4 + 2 * L + 4 * (N - 1) : 4 + 2 * L;
    if (c->scenario == PO_SIMPLE_REFERENCE) return 2 + 2 * L + 3 + PO_DIM_C * (N - 1);
    if (c->scenario == PO_SIMPLE_SPEAKER_LISTENER) return 2 + 2 * L + 3; /* experiments/scenarios.py:45-64 */
    /* simple_tag: adversary rows are the widest (see all G good velocities) */
    const int G = N - c->num_adversaries;
    return 4 + 2 * L + 2 * (N - 1) + 2 * (c->num_adversaries > 0 ? G : G - 1);
}

/* ---- deterministic float32 math (restated, not shared, from include/pworld_math.h) ---- */
static inline float po_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* exp(x), x in (-87, 88); x <= -87 -> 0 exactly; x >= 88 -> 2^127.  Contract revision 3: degree-7 Taylor in
 * Estrin form (pairs, then r^2, r^4) */
PO_EXPORT float po_exp_det_f32(float x)
{
    if (!(x > -87.0f)) return x != x ? x : 0.0f;
    if (x >= 88.0f) return po_u2f(0x7f000000u);
    float n = floorf(fmaf(x, 1.44269504088896341f, 0.5f));
    float r = fmaf(n, -0.693359375f, x);             /* ln2 hi (exact in 9 bits) */
    r = fmaf(n, 2.12194440054690583e-4f, r);         /* ln2 lo */
    const float r2 = r * r, r4 = r2 * r2;
    const float a = 1.0f + r;
    const float b = fmaf(1.66666666666666657e-1f, r, 0.5f);                     /* 1/2 + r/6 */
    const float c = fmaf(8.33333333333333322e-3f, r, 4.16666666666666644e-2f);  /* 1/24 + r/120 */
    const float d = fmaf(1.98412698412698413e-4f, r, 1.38888888888888894e-3f);  /* 1/720 + r/5040 */
    const float p = fmaf(fmaf(d, r2, c), r4, fmaf(b, r2, a));
    int32_t e = (int32_t)n + 127;
    return p * po_u2f((uint32_t)e << 23);
}

/* Q(t) ~ (log1p(t) - t) / t^2 on [0, 1]: degree-7 minimax, Estrin form */
static inline float po_log1p_q_f32(float t)
{
    const float t2 = t * t, t4 = t2 * t2;
    const float a = fmaf(3.332236707e-01f, t, -4.999969006e-01f);
    const float b = fmaf(1.919044554e-01f, t, -2.486616373e-01f);
    const float c = fmaf(8.017139137e-02f, t, -1.383424997e-01f);
    const float d = fmaf(5.516789388e-03f, t, -3.066807054e-02f);
    return fmaf(fmaf(d, t2, c), t4, fmaf(b, t2, a));
}

/* log1p(t), t in [0, 1]: t + t^2 Q(t) */
PO_EXPORT float po_log1p_det_f32(float t)
{
    return fmaf(t * t, po_log1p_q_f32(t), t);
}

/* logaddexp(0, x) = max(x, 0) + log1p(exp(-|x|)); the max goes into the last fused step */
PO_EXPORT float po_softplus_det_f32(float x)
{
    float m = x > 0.0f ? x : 0.0f;
    const float t = po_exp_det_f32(-fabsf(x));
    return fmaf(t * t, po_log1p_q_f32(t), t + m);
}

/* vectorised forms for the device-math bit tests: fn 0/4 sqrtf, 1/2 softplus, 3 exp, 5 x / aux */
PO_EXPORT void po_math_v(int fn, const float *x, float aux, float *y, long n)
{
    for (long i = 0; i < n; ++i) {
        const float v = x[i];
        y[i] = (fn == 0 || fn == 4) ? sqrtf(v) : (fn == 1 || fn == 2) ? po_softplus_det_f32(v)
               : fn == 3 ? po_exp_det_f32(v) : v / aux;
    }
}

/* ---- Philox4x32-10 (Salmon et al. 2011), reset RNG of the batched env ---- */
PO_EXPORT void po_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int i = 0; i < 10; ++i) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

PO_EXPORT void po_philox_xy(uint64_t seed, uint64_t env_id, uint32_t episode, uint32_t entity,
                            float lo, float hi, float *x, float *y)
{
    uint32_t ctr[4] = {entity, episode, (uint32_t)env_id, (uint32_t)(env_id >> 32)};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)}, r[4];
    po_philox4x32_10(ctr, key, r);
    float span = hi - lo;
    float u0 = (float)(r[0] >> 8) * 5.9604644775390625e-8f; /* 2^-24 */
    float u1 = (float)(r[1] >> 8) * 5.9604644775390625e-8f;
    *x = span * u0 + lo;
    *y = span * u1 + lo;
}

/* simple_reference: goal_b landmark of an agent = third Philox word of the agent's reset draw, mod L */
PO_EXPORT int32_t po_philox_goal(uint64_t seed, uint64_t env_id, uint32_t episode, uint32_t entity, int32_t L)
{
    uint32_t ctr[4] = {entity, episode, (uint32_t)env_id, (uint32_t)(env_id >> 32)};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)}, r[4];
    po_philox4x32_10(ctr, key, r);
    return (int32_t)(r[2] % (uint32_t)L);
}

static const float PO_LM_COLOR[3][3] = {{0.75f, 0.25f, 0.25f}, {0.25f, 0.75f, 0.25f}, {0.25f, 0.25f, 0.75f}};
static const float PO_SL_COLOR[3][3] = {{0.65f, 0.15f, 0.15f}, {0.15f, 0.65f, 0.15f}, {0.15f, 0.15f, 0.65f}};

PO_EXPORT int po_dim_c(const po_config *c)
{
    return c->scenario == PO_SIMPLE_REFERENCE ? PO_DIM_C : c->scenario == PO_SIMPLE_SPEAKER_LISTENER ? PO_SL_DIM_C : 0;
}

#define PO_CAT_(a, b) a##b
#define PO_CAT(a, b) PO_CAT_(a, b)

#define REAL float
#define PO_IS_F32 1
#define SUFFIX(x) PO_CAT(x, _f32)
#include "pworld_oracle_impl.h"
#undef REAL
#undef PO_IS_F32
#undef SUFFIX

#define REAL double
#define PO_IS_F32 0
#define SUFFIX(x) PO_CAT(x, _f64)
#include "pworld_oracle_impl.h"
#undef REAL
#undef PO_IS_F32
#undef SUFFIX
