// pw_kernels_spread.hpp -- part of libpworld.so (one translation unit: csrc/pworld.hip includes it).
// simple_spread fast paths: pw_spread_fast_kernel, pw_spread_stream_kernel, pw_spread_duo_kernel.
#pragma once

namespace {

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
    int k1;  // the contact margin qualifies for the one-correction division (margin_one_correction)
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
// sampled once, at the first step: afterwards every segment runs from the previous stamp, so the segments of one step add
// up to the whole loop body
#define PW_STAMP_START                                                                                  \
    do {                                                                                                \
        if (st_prev == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev)::"memory"); \
    } while (0)
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
    PW_NEAR_MASK_HOOK(m);
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
    int p_prio;  // duo kernel: issue priority per wave, 2 bits each (wave 0 = physics in bits 0-1, ...); set by the launch
    uint64_t seed, env_id_base;
    float dt, damp, contact_force, contact_margin, mass;
    float dist_min, coll_thr2, near_thr2, sens, fscale;
    float *pos_x, *pos_y, *vel_x, *vel_y, *lm_x, *lm_y;
    int32_t *ep_step;
    uint32_t *ep_count;
    const int32_t *act;
    float *obs, *final_obs, *rew, *rew_shared;
    uint8_t *done, *terminal;
    uint64_t *coll;  // [T,B,N] collision masks; written only by the COLL instantiations
};

// A store into one of the per-agent / per-env output planes: with the non-temporal hint where the enclosing kernel's
// kNtPlanes says so (pw_common.hpp, nt_store).  A macro, not a function template: routed through a function the plain
// branch cost the N = 48 kernel 71 VGPRs (168 -> 239) and a third of its occupancy.
#define PW_PLANE_STORE(lvalue, value)                     \
    do {                                                  \
        if constexpr (kNtPlanes) nt_store(&(lvalue), value); \
        else (lvalue) = (value);                          \
    } while (0)

template <int LT>
__device__ __forceinline__ void stream_write_obs(float *__restrict__ o, const int L, const float2 *lm, float px,
                                                 float py, float vx, float vy)
{
    if ((LT ? LT : L) % 2 == 0) {
        float4 *o4 = reinterpret_cast<float4 *>(o);
        o4[0] = make_float4(vx, vy, px, py);
        if (LT > 0) {
#pragma unroll
            for (int c = 0; c < LT / 2; ++c) {
                const float2 l0 = lm[2 * c], l1 = lm[2 * c + 1];
                o4[1 + c] = make_float4(l0.x - px, l0.y - py, l1.x - px, l1.y - py);
            }
        } else {  // runtime L: four landmark reads in flight per round
            int c = 0;
            for (; c + 2 <= L / 2; c += 2) {
                const float2 l0 = lm[2 * c], l1 = lm[2 * c + 1], l2 = lm[2 * c + 2], l3 = lm[2 * c + 3];
                o4[1 + c] = make_float4(l0.x - px, l0.y - py, l1.x - px, l1.y - py);
                o4[2 + c] = make_float4(l2.x - px, l2.y - py, l3.x - px, l3.y - py);
            }
            for (; c < L / 2; ++c) {
                const float2 l0 = lm[2 * c], l1 = lm[2 * c + 1];
                o4[1 + c] = make_float4(l0.x - px, l0.y - py, l1.x - px, l1.y - py);
            }
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

// Observation rows of ALL the wave's envs as ONE contiguous block.  A wave's lanes own consecutive rows
// (row = first row of the workgroup + lane), so its rows x D floats are contiguous in the obs plane; instead of
// every lane storing its own row (16 B per lane at a stride of D floats: each store instruction touches `rows`
// different cache lines with a quarter line each, and the write path is what bounds the large-B / large-N regimes,
// profiles/r2_regimes.txt), lane l of store instruction s writes chunk q = l + 64 s of the block: 64 lanes x 16 B =
// 1 KiB contiguous per instruction.  Chunk q belongs to row r = q / CH, column group c = q % CH (CH = D / 4):
// c = 0 is {vel, pos}, c >= 1 the two landmarks 2c-2, 2c-1 relative to the row's agent.  s_row[r] = {px, py, vx, vy}
// of the wave's row r (every lane publishes its own first), s_lm the wave's landmarks [e_local * L + l].
// Odd L: the same with 8-byte chunks (CH = D / 2: vel | pos | one landmark each).  The values are the ones
// stream_write_obs computes (same subtractions), only the lane that stores them differs.
template <int NT, int LT>
__device__ __forceinline__ void stream_write_obs_block(float *__restrict__ blk, const int rows, const int lane,
                                                       const float4 *s_row, const float2 *s_lm)
{
    static_assert(NT > 0 && LT > 0, "compile-time N and L only");
    constexpr int D = 4 + 2 * LT;
    constexpr bool WIDE = LT % 2 == 0;           // 16-byte chunks; odd L: 8-byte chunks
    constexpr int CH = WIDE ? D / 4 : D / 2;     // chunks per row
    constexpr int UNR = CH < 4 ? CH : 4;         // chunks in flight per lane: their LDS reads share one wait
    constexpr int DR = 64 / CH, DC = 64 % CH;    // chunk q + 64 is DR rows further and DC column groups to the right
    const int total = rows * CH;
    // (row, column group) of this lane's chunk, advanced incrementally: no division inside the loop, and nothing
    // for the compiler to hoist into dozens of loop-invariant registers (which cost the N = 48 kernel its occupancy)
    int r = lane / CH, c = lane - r * CH;
#pragma unroll 1
    for (int q0 = lane; q0 - lane < total; q0 += 64 * UNR) {  // wave-uniform trip count
        float4 st[UNR], lm[UNR];
        int cc[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int rr = r < rows ? r : 0;  // lanes past the block read row 0 (they store nothing)
            cc[u] = c;
            st[u] = s_row[rr];
            const int e = rr / NT;
            if (WIDE) {
                lm[u] = *reinterpret_cast<const float4 *>(s_lm + e * LT + (c > 0 ? 2 * c - 2 : 0));
            } else {
                const float2 l1 = s_lm[e * LT + (c > 1 ? c - 2 : 0)];
                lm[u] = make_float4(l1.x, l1.y, 0.0f, 0.0f);
            }
            r += DR; c += DC;
            if (c >= CH) { c -= CH; r += 1; }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int q = q0 + 64 * u;
            if (WIDE) {
                const float4 o = cc[u] == 0 ? make_float4(st[u].z, st[u].w, st[u].x, st[u].y)
                                            : make_float4(lm[u].x - st[u].x, lm[u].y - st[u].y, lm[u].z - st[u].x, lm[u].w - st[u].y);
                if (q < total) nt_store(reinterpret_cast<float4 *>(blk) + q, o);
            } else {
                const float2 o = cc[u] == 0 ? make_float2(st[u].z, st[u].w)
                                 : cc[u] == 1 ? make_float2(st[u].x, st[u].y)
                                              : make_float2(lm[u].x - st[u].x, lm[u].y - st[u].y);
                if (q < total) nt_store(reinterpret_cast<float2 *>(blk) + q, o);
            }
        }
    }
}

// Store instructions per step of stream_write_obs_block at the default packing of large batches (pw_create:
// 8 envs per wave at N = 6, 16 at N = 3, 64 / N otherwise); 0 when N / L are runtime values.
template <int NT, int LT>
constexpr int obs_block_stores()
{
    if (NT <= 0 || LT <= 0) return 0;
    const int rows = (NT == 6 ? 8 : NT == 3 ? 16 : 64 / NT) * NT;
    const int ch = LT % 2 == 0 ? (4 + 2 * LT) / 4 : (4 + 2 * LT) / 2;
    return (rows * ch + 63) / 64;
}

// "Planned" block-wise observation stores (duo kernel, even L, rows up to 64 floats... CH * rows <= 640 chunks): chunk n = lane + 64 i of the
// wave's rows x CH grid has a fixed (row, column group) for the whole launch, so the two LDS addresses it is composed from
// are computed ONCE per launch and kept packed in one register per chunk: A = 16 bytes -- the row's {vel, pos} (c = 0) or the
// env's landmarks 2c - 2, 2c - 1 (c >= 1) --, B = 8 bytes -- a zero that lives in LDS (c = 0: x - 0 is x, for every x) or
// the row's position.  A step is then, per chunk: two LDS reads, two packed subtractions (A.xy - B, A.zw - B: the
// subtractions stream_write_obs does), one store of 1 KiB contiguous per wave -- ~7 instructions instead of the ~18 of the
// incremental (row, column) walk of stream_write_obs_block.  Needs the rows staged as {vx, vy, px, py}.
#ifndef PW_PLAN_MAX
#define PW_PLAN_MAX 20   // registers for the plan: N = L = 6 / 12 / 24 / 48 need 4 / 7 / 10 / 19 (N = 48: 128 -> 168 VGPRs, three instead of four waves per SIMD, still -12 %)
#endif
template <int NT, int LT>
constexpr int obs_plan_iters()
{
    if (NT <= 0 || LT <= 0 || LT % 2) return 0;
    return ((64 / NT) * NT * ((4 + 2 * LT) / 4) + 63) / 64;
}
template <int NT, int LT, int NIT>
__device__ __forceinline__ void obs_plan_build(uint32_t (&plan)[NIT], const int rows, const int lane, const unsigned char *smem,
                                               const float4 *s_row, const float2 *s_lm, const float2 *s_zero)
{
    constexpr int CH = (4 + 2 * LT) / 4;
    const int total = rows * CH;
    const uint32_t zoff = (uint32_t)(reinterpret_cast<const unsigned char *>(s_zero) - smem);
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int n = lane + 64 * i, nn = n < total ? n : 0;
        const int r = nn / CH, c = nn - r * CH, e = r / NT;
        const uint32_t self = (uint32_t)(reinterpret_cast<const unsigned char *>(s_row + r) - smem);   // {vx, vy, px, py}
        const uint32_t offA = c == 0 ? self : (uint32_t)(reinterpret_cast<const unsigned char *>(s_lm + e * LT + 2 * c - 2) - smem);
        const uint32_t offB = c == 0 ? zoff : self + 8;
        plan[i] = offA | (offB << 16);
    }
}
template <int NT, int LT, int NIT>
__device__ __forceinline__ void obs_plan_store(float *__restrict__ blk, const int rows, const int lane, const uint32_t (&plan)[NIT],
                                               const unsigned char *smem)
{
    constexpr int CH = (4 + 2 * LT) / 4, UN = NIT < 4 ? NIT : 4;
    const int total = rows * CH;
    float4 *const out = reinterpret_cast<float4 *>(blk) + lane;
#pragma unroll
    for (int i0 = 0; i0 < NIT; i0 += UN) {
        float4 va[UN];
        float2 vb[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (i0 + u < NIT) {
                va[u] = *reinterpret_cast<const float4 *>(smem + (plan[i0 + u] & 0xFFFFu));
                vb[u] = *reinterpret_cast<const float2 *>(smem + (plan[i0 + u] >> 16));
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (i0 + u < NIT && lane + 64 * (i0 + u) < total)
                nt_store(out + 64 * (i0 + u), make_float4(va[u].x - vb[u].x, va[u].y - vb[u].y, va[u].z - vb[u].x, va[u].w - vb[u].y));
        }
    }
}

template <int NT, typename MaskT>
__device__ __forceinline__ void stream_partner_pass(const int N, const int a, const float2 *pp, float px, float py,
                                                    float olx, float oly, float coll_thr2, float near_thr2,
                                                    MaskT &coll, MaskT &near, float &best)
{
    coll = 0; near = 0; best = 0.0f;
    auto one = [&](const float2 q, const int j) {
        const float dx = q.x - px, dy = q.y - py;
        const float d2 = dx * dx + dy * dy;
        if (d2 < coll_thr2) coll |= (MaskT)1 << j;
        const bool far = d2 >= near_thr2 && d2 <= 3.402823466e+38f;
        if (!far && j != a) near |= (MaskT)1 << j;
        const float ex = q.x - olx, ey = q.y - oly;
        const float e2 = ex * ex + ey * ey;
        best = (j == 0 || e2 < best) ? e2 : best;
    };
    if (NT > 0) {
#pragma unroll
        for (int j = 0; j < NT; ++j) one(pp[j], j);
    } else {
        // runtime N: four LDS reads in flight per round (one read, wait, compute per partner cost ~120 cycles each -- 3 k cycles of a
        // step at N = 24 on waves nothing hides); the partners are still visited in ascending order: the same bits
        int j = 0;
        for (; j + 4 <= N; j += 4) {
            const float2 q0 = pp[j], q1 = pp[j + 1], q2 = pp[j + 2], q3 = pp[j + 3];
            one(q0, j); one(q1, j + 1); one(q2, j + 2); one(q3, j + 3);
        }
        for (; j < N; ++j) one(pp[j], j);
    }
}

template <int NT, int LT, bool UNIT_MASS, bool COLL = false, bool BLOCK = false>
__global__ void __launch_bounds__(kWave) pw_spread_stream_kernel(const StreamParams A, const int T)
{
    using MaskT = typename std::conditional<(NT > 0 && NT <= 32), uint32_t, uint64_t>::type;
    constexpr bool kNtPlanes = NT > 0 && NT <= 6;  // per-agent / per-env planes with the non-temporal hint (PW_PLANE_STORE; pw_common.hpp nt_store)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int N = NT ? NT : A.N, L = LT ? LT : A.L, D = 4 + 2 * L;
    float2 *s_pos = reinterpret_cast<float2 *>(smem_raw);  // [64]
    float4 *s_row = reinterpret_cast<float4 *>(s_pos + kWave);  // [64] {pos, vel} per row, for the block-wise obs stores
    float2 *s_lm = reinterpret_cast<float2 *>(s_row + kWave);   // [epw * L]
    const int rows_here = (A.B - (int)blockIdx.x * A.epw < A.epw ? A.B - (int)blockIdx.x * A.epw : A.epw) * N;

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
    constexpr int kStoresPerStep = LT > 0 ? 4 + (COLL ? 1 : 0) + (BLOCK ? obs_block_stores<NT, LT>() : (LT % 2 == 0 ? 1 + LT / 2 : 2 + LT)) : 0;
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
        if (NT > 12 || NT == 0) {  // as in the duo kernel's reward half: a loop up to the wave's largest collision count
            const int cnt = sizeof(MaskT) == 4 ? __builtin_popcount((uint32_t)coll) : __builtin_popcountll((uint64_t)coll);
            for (int k2 = 0; __any(k2 < cnt); ++k2) r -= k2 < cnt ? 1.0f : 0.0f;
        } else {
#pragma unroll(NT > 0 ? NT : 1)
            for (int j = 0; j < (NT ? NT : N); ++j)
                if ((coll >> j) & 1) r -= 1.0f;
        }
        float acc = 0.0f;
#pragma unroll(NT > 0 ? NT : 1)
        for (int i = 0; i < (NT ? NT : N); ++i) acc += __shfl(r, base + i, kWave);
        PW_STAMP(4);
        PW_PLANE_STORE(A.rew[tBN + g], r);
        PW_PLANE_STORE(A.done[tBN + g], (uint8_t)0);
        if (COLL) { PW_PLANE_STORE(A.coll[tBN + g], (uint64_t)coll); }  // is_collision bits of the state this step produced (pre-reset)
        PW_PLANE_STORE(A.rew_shared[(size_t)t * A.B + env], acc);
        ep_step += 1;
        const bool term = A.max_episode_len > 0 && ep_step >= A.max_episode_len;
        PW_PLANE_STORE(A.terminal[(size_t)t * A.B + env], (uint8_t)(term ? 1 : 0));
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
        if (BLOCK) s_row[base + a] = make_float4(px, py, vx, vy);
        wave_lds_sync();
        if (A.auto_reset && __any(term))
            stream_partner_pass<NT, MaskT>(N, a, pp, px, py, olx, oly, A.coll_thr2, A.near_thr2, coll, near, best);
        PW_STAMP(5);
        if constexpr (BLOCK)
            stream_write_obs_block<NT, LT>(A.obs + (tBN + (size_t)blockIdx.x * A.epw * N) * D, rows_here, (int)threadIdx.x,
                                           s_row, s_lm);
        else
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

template <int NT, int LT, bool UNIT_MASS, bool COLL = false, bool BLOCK = false, bool TRIO = false>
__global__ void __launch_bounds__((TRIO ? 3 : 2) * kWave) pw_spread_duo_kernel(const StreamParams A, const int T)
{
    using MaskT = typename std::conditional<(NT > 0 && NT <= 32), uint32_t, uint64_t>::type;
    constexpr bool kNtPlanes = NT > 0 && NT <= 6;  // per-agent / per-env planes with the non-temporal hint (PW_PLANE_STORE; pw_common.hpp nt_store)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int N = NT ? NT : A.N, L = LT ? LT : A.L, D = 4 + 2 * L;
    float4 *s_ring = reinterpret_cast<float4 *>(smem_raw);            // [3][64] {px, py, vx, vy}
    float2 *s_lm = reinterpret_cast<float2 *>(s_ring + 3 * kWave);    // [epw * L]      (wave O only)
    float *s_min = reinterpret_cast<float *>(s_lm + A.epw * L);       // [64] per-landmark min dist (O)
    float *s_rew = s_min + kWave;                                     // [64] per-agent reward      (O)
    // [64] {pos, vel} of every row after the step (post-reset where the env reset), for O's block-wise obs stores;
    // 16-byte aligned: 3*64*16 + epw*L*8 + 512 bytes precede it and epw*L*8 is a multiple of 16 when L is even
    // (odd L: the block writer reads it as float4 too, so pad)
    float4 *s_row = reinterpret_cast<float4 *>(smem_raw + ((3 * kWave * sizeof(float4) + (size_t)A.epw * L * sizeof(float2) +
                                                            2 * kWave * sizeof(float) + 15) & ~(size_t)15));
    float2 *s_utab = reinterpret_cast<float2 *>(s_row + kWave);      // [8] action force per index (wave P)
    float2 *s_zero = s_utab + 8;                                     // {0, 0} (planned observation stores)
    // Three-wave form at N = 12 (C5's N = 12 point: 820 workgroups at B = 4096): wave P's action indices arrive four steps ahead by
    // LDS-direct loads, as in pw_spread_quad_kernel / pw_tag_duo_kernel (pw_common.hpp act_fetch_issue).  With the output wave split
    // in two, P is the step's longest wave there and its one-step-ahead register load cost it 800 of its 3230 cycles per step (stamps,
    // round 5: HBM latency under the output waves' write stream exceeds a step): +2.8 % env-steps/s at B = 4096, unchanged at B = 2048 /
    // 6144 (profiles/r5_n12_action_ring.txt).  N = 24 in this form loses 1.4 % (B = 2048), and the two-wave form keeps the register load
    // everywhere: its output wave is the longer one, and P waits for the index instead of waiting at the barrier (profiles/r3_tag_prefetch.txt).
    constexpr bool kActRing = TRIO && NT == 12;
    int32_t *s_actr = reinterpret_cast<int32_t *>(s_utab + 16);     // [4][64] (kActRing; 16-byte aligned)
    constexpr int kPlanIters = obs_plan_iters<NT, LT>();
    constexpr bool kPlan = BLOCK && kPlanIters >= 1 && kPlanIters <= PW_PLAN_MAX;

    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int lane = (int)threadIdx.x & 63;
    const int rows_here = (A.B - (int)blockIdx.x * A.epw < A.epw ? A.B - (int)blockIdx.x * A.epw : A.epw) * N;
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
        // issue priorities (A.p_prio: 2 bits per wave, set by the launch): the physics wave first where it shares a SIMD
        // with output waves of another workgroup
        switch (A.p_prio & 3) { case 1: __builtin_amdgcn_s_setprio(1); break; case 2: __builtin_amdgcn_s_setprio(2); break;
                                case 3: __builtin_amdgcn_s_setprio(3); break; default: break; }
        float px = A.pos_x[g], py = A.pos_y[g], vx = A.vel_x[g], vy = A.vel_y[g];
        s_ring[me] = make_float4(px, py, vx, vy);
        wave_lds_sync();
        // U2 + U4 as a table: the action force of an index is one of five constants, computed here once with the step's
        // own expressions (so the bits are the step's), entry 5 = any other index (no force); a step reads ONE entry
        if (lane < 6) {
            const int ai = lane;
            float ux = 0.0f + ((ai == 1 ? 1.0f : 0.0f) - (ai == 2 ? 1.0f : 0.0f));
            float uy = 0.0f + ((ai == 3 ? 1.0f : 0.0f) - (ai == 4 ? 1.0f : 0.0f));
            ux *= A.sens; uy *= A.sens;
            if (A.fscale != 1.0f) { ux = A.fscale * ux; uy = A.fscale * uy; }
            s_utab[lane] = make_float2(ux + 0.0f, uy + 0.0f);
        }
        wave_lds_sync();
        MaskT near = duo_near_pass<NT, MaskT>(N, a, s_ring + base, px, py, A.near_thr2);
        const float k = A.contact_margin, cf = A.contact_force, dt = A.dt, damp = A.damp, mass = A.mass;
        int act_next = 0;
        const int32_t *act_g = A.act + g;
        const uint32_t act_lds = kActRing ? __builtin_amdgcn_readfirstlane((uint32_t)reinterpret_cast<uintptr_t>(s_actr)) : 0u;
        auto fetch_act = [&](int t) {  // indices of step t (clamped: the tail re-fetches the last step) -> slot t & 3
            act_fetch_issue(act_g + (size_t)(t < T ? t : T - 1) * BN, act_lds + (uint32_t)(t & 3) * (kWave * 4));
        };
        if constexpr (kActRing) {
            // every load the compiler counts is consumed before the first uncounted one is issued (pw_spread_quad_kernel)
            asm volatile("" :: "v"(ep_step), "v"(ep_count), "v"(px), "v"(py), "v"(vx), "v"(vy), "v"(near) : "memory");
            fetch_act(0); fetch_act(1); fetch_act(2); fetch_act(3);
        } else {
            act_next = A.act[g];
        }
        PW_STAMP_DECL;
        for (int t = 0; t < T; ++t) {
            PW_STAMP_START;
            uint32_t ai;
            if constexpr (kActRing) {
                act_fetch_wait3();  // step t's indices are in LDS (the later fetches stay in flight)
                ai = (uint32_t)s_actr[(t & 3) * kWave + lane];
                fetch_act(t + 4);   // into the slot just read
            } else {
                ai = (uint32_t)act_next;
                const int tn = t + 1 < T ? t + 1 : t;
                act_next = A.act[(size_t)tn * BN + g];
            }
            const float2 u0 = s_utab[ai < 5u ? ai : 5u];  // {u_x + 0, u_y + 0}: the accumulators' starting values
            float fx = u0.x, fy = u0.y;
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
        if constexpr (kActRing) act_fetch_drain();  // the tail's fetches have landed before the wave ends
        A.pos_x[g] = px; A.pos_y[g] = py;
        A.vel_x[g] = vx; A.vel_y[g] = vy;
        A.ep_step[env] = ep_step;
        A.ep_count[env] = ep_count;
    } else {
        // ================================ wave O: outputs ================================
        // One wave does both halves below -- or, TRIO, two do one each: wave 1 the collision masks, landmark minima,
        // rewards and the per-agent / per-env planes, wave 2 the observation rows.  They share nothing but the ring slot
        // they read; each follows the slot sequence, the episode clock and its landmark itself (only wave 2 keeps the
        // landmarks in LDS, for the rows).  Used where the single output wave is the step's critical path (mid-size grids).
        const bool do_rew = !TRIO || wave == 1, do_obs = !TRIO || wave == 2;  // compile-time true in the two-wave form
        switch ((A.p_prio >> (2 * wave)) & 3) { case 1: __builtin_amdgcn_s_setprio(1); break; case 2: __builtin_amdgcn_s_setprio(2); break;
                                                case 3: __builtin_amdgcn_s_setprio(3); break; default: break; }
        float2 *lmv = s_lm + e_local * L;
        const int la = a < L ? a : 0;
        float olx = 0.f, oly = 0.f;
        if (L > 0) {
            olx = A.lm_x[(size_t)env * L + la];
            oly = A.lm_y[(size_t)env * L + la];
            if (do_obs) lmv[la] = make_float2(olx, oly);
        }
        constexpr int kStoresPerStep = LT > 0 ? 4 + (COLL ? 1 : 0) + (BLOCK ? obs_block_stores<NT, LT>() : (LT % 2 == 0 ? 1 + LT / 2 : 2 + LT)) : 0;
        constexpr int kVm = kStoresPerStep < 63 ? kStoresPerStep : 63;
        uint32_t plan[kPlan ? kPlanIters : 1];
        if constexpr (kPlan) {
            if (do_obs) {
                if (lane == 0) *s_zero = make_float2(0.0f, 0.0f);
                obs_plan_build<NT, LT, kPlanIters>(plan, rows_here, lane, smem_raw, s_row, s_lm, s_zero);
            }
        }
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
            if (do_rew) {
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
                // "rew -= 1" once per colliding agent (itself included): the subtrahends are all 1.0, so only their NUMBER
                // matters.  Few agents collide at a time: for large N a loop up to the wave's largest count (a handful of
                // iterations) replaces N bit tests -- N = 48: ~12 instead of 144 instructions per step, on grids that are
                // VALU-issue bound (profiles/r3_n3_trio.txt); small N keeps the unrolled selects (a compare feeding a scalar
                // branch per iteration costs more than six of them)
                if (NT > 12 || NT == 0) {   // measured: N = 48 -13.6 %, N = 24 -3.3 % step time; N = 12 +2 % (stays unrolled)
                    const int cnt = sizeof(MaskT) == 4 ? __builtin_popcount((uint32_t)coll) : __builtin_popcountll((uint64_t)coll);
                    for (int k2 = 0; __any(k2 < cnt); ++k2) r -= k2 < cnt ? 1.0f : 0.0f;
                } else {
#pragma unroll(NT > 0 ? NT : 1)
                    for (int j = 0; j < (NT ? NT : N); ++j)
                        if ((coll >> j) & 1) r -= 1.0f;
                }
                s_rew[me] = r;
                wave_lds_sync();
                float acc = 0.0f;
#pragma unroll(NT > 0 ? NT : 1)
                for (int i = 0; i < (NT ? NT : N); ++i) acc += s_rew[base + i];
                PW_STAMP(1);
                PW_PLANE_STORE(A.rew[tBN + g], r);
                if (COLL) { PW_PLANE_STORE(A.coll[tBN + g], (uint64_t)coll); }
                PW_PLANE_STORE(A.rew_shared[(size_t)t * A.B + env], acc);
            }
            ep_step += 1;
            const bool term = A.max_episode_len > 0 && ep_step >= A.max_episode_len;
            // the two constant-ish planes: in the three-wave form by the observation wave, which has the slack (N = 12, stamps: 1370 busy
            // cycles of a 3200-cycle step against the reward wave's 2640, the longest wave once the physics wave stopped waiting for its indices)
            if (TRIO ? do_obs : do_rew) {
                PW_PLANE_STORE(A.done[tBN + g], (uint8_t)0);
                PW_PLANE_STORE(A.terminal[(size_t)t * A.B + env], (uint8_t)(term ? 1 : 0));
            }
            if (term && A.auto_reset) {
                if (do_obs && A.final_obs) stream_write_obs<LT>(A.final_obs + (tBN + g) * D, L, lmv, px, py, vx, vy);
                wave_lds_sync();
                ep_count += 1;
                ep_step = 0;
                if (L > 0) {
                    pw_reset_xy(A.seed, env_id, ep_count, (uint32_t)(N + la), -1.0f, 1.0f, &olx, &oly);
                    if (do_obs) lmv[la] = make_float2(olx, oly);
                }
                nxt = nxt + 1; nxt = nxt == 3 ? 0 : nxt;
                const float4 fresh = s_ring[nxt * kWave + me];  // post-reset state published by P
                px = fresh.x; py = fresh.y; vx = fresh.z; vy = fresh.w;
            }
            cur = nxt;
            if (do_obs) {
                if (BLOCK) s_row[me] = kPlan ? make_float4(vx, vy, px, py) : make_float4(px, py, vx, vy);
                wave_lds_sync();
                if constexpr (kPlan)
                    obs_plan_store<NT, LT, kPlanIters>(A.obs + (tBN + (size_t)blockIdx.x * A.epw * N) * D, rows_here, lane, plan, smem_raw);
                else if constexpr (BLOCK)
                    stream_write_obs_block<NT, LT>(A.obs + (tBN + (size_t)blockIdx.x * A.epw * N) * D, rows_here, lane, s_row, s_lm);
                else
                    stream_write_obs<LT>(A.obs + (tBN + g) * D, L, lmv, px, py, vx, vy);
            }
            PW_STAMP(2);
            if (LT > 0 && !TRIO) __builtin_amdgcn_s_waitcnt((kVm & 0xF) | 0x0F70 | ((kVm >> 4) << 14));
            PW_STAMP(3);
        }
#ifdef PW_STAMPS
        if (blockIdx.x == 0 && lane == 0)
            for (int i_ = 0; i_ < 8; ++i_) g_pw_stamps[8 + i_] = st_acc[i_];
#endif
        if (L > 0 && do_obs) {
            A.lm_x[(size_t)env * L + la] = olx;
            A.lm_y[(size_t)env * L + la] = oly;
        }
    }
}

}  // namespace
