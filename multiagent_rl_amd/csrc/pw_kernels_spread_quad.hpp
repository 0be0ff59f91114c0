// pw_kernels_spread_quad.hpp -- part of libpworld.so (one translation unit: csrc/pworld.hip includes it).
// simple_spread N = L = 6 on small grids (the latency-bound regime of BASELINE configs[1]): four cooperating waves.
#pragma once

namespace {

// ------------------------------------------------------------------------------------------
// At B = 4096 a step of pw_spread_duo_kernel is ONE wave's dependent chain (wave P: ~1930 cycles, a third of them
// issue slots; profiles/r2_c2_b4096_duo_summary.json): a lane owns an agent, loops over its near partners one after
// the other (~600 cycles of dependent float32 work per partner), then a second pass over all partners finds the next
// step's near set; the output wave O needs ~1500 cycles for its own chain of LDS exchanges.  The chip has four times
// more lanes than this batch has agents, so this kernel spends lanes to shorten both chains:
//   waves P0, P1  physics of 4 envs each, PAIR-parallel: lane = one of the env's 15 unordered agent pairs (60 lanes).
//                 Every pair is tested and, if near, evaluated ONCE, all pairs at the same time -- no per-lane partner
//                 loop, no separate near-set pass; the pair's force goes to the first agent, its exact negation (IEEE:
//                 delta, quotients and products only change sign) to the second, through a [agent][partner] table in
//                 LDS (wave shuffles were measured: slower); the env's 6 agent lanes then add their row in ascending partner order (the upstream
//                 accumulation order, so the bits do not change; far pairs contribute +-0, which cannot change an
//                 accumulator that is never -0), integrate and publish {pos, vel} into the ring.
//   wave OA       8 envs, one step behind (as the duo kernel's O): collision masks, landmark minima, rewards, shared
//                 reward, done / terminal stores.
//   wave OB       8 envs, one step behind: observation rows, stored as one contiguous block per step
//                 (stream_write_obs_block), and the pre-reset rows at episode ends.
// One s_barrier per step.  The ring slot sequence is WORKGROUP-uniform: in a step in which ANY of the 8 envs resets,
// every env publishes a pre-reset and a post-reset slot (identical for the envs that do not reset), so no wave needs
// another env's slot index; each wave follows all 8 episode clocks itself.  Envs out of step with each other can make
// that happen in consecutive steps, so the ring has 4 slots: the two the output waves are reading (one step behind)
// and the two the physics waves may be writing.
// Per-step overheads are kept off the physics waves' instruction stream: the episode clocks are offsets behind ONE scalar
// compare per step (the vector work runs in reset steps only), the action force is a six-entry table lookup, the action
// indices arrive four steps ahead by LDS-direct loads (pw_common.hpp act_fetch_issue), the pair operands of the next step
// are read back before the barrier.  Measured state (profiles/r2_action_prefetch.txt, r2_c2_b4096_quad_summary.json):
// 0.62 us per step at B = 4096, the three kinds of waves within 10 % of each other, 0.53 us for a lone workgroup.
// Arithmetic and results are identical to the other simple_spread kernels (same bit-exact tests, `quad` path).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void quad_pair_of(const int q, int &i, int &j)
{
    // unordered pairs of 6 agents in lexicographic order: (0,1) (0,2) ... (0,5) (1,2) ... (4,5)
    i = q < 5 ? 0 : q < 9 ? 1 : q < 12 ? 2 : q < 14 ? 3 : 4;
    j = q - (i == 0 ? 0 : i == 1 ? 5 : i == 2 ? 9 : i == 3 ? 12 : 14) + i + 1;
}

// COLL: wave OA also stores the step's collision masks (pw_step_io.coll) -- it holds the six threshold tests anyway, so
// the instantiation differs by one mask accumulation and one 8-byte store per lane and step; every other output is
// bit-identical to the plain form (tests: the `quad+coll` path, and the bench-path test at C2 full size).
// K1: the division by the contact margin inside the force runs as ONE Newton correction (pw_common.hpp div_chain1); the
// host sets it only for margins whose refined reciprocal is the correctly rounded one (pworld.hip margin_one_correction).
#ifndef PW_QUAD_BARRIER
#define PW_QUAD_BARRIER(t) duo_barrier()   // one workgroup meeting per step
#endif
#ifndef PW_QUAD_ACT_AHEAD
#define PW_QUAD_ACT_AHEAD 4   // steps the physics waves' action indices are fetched ahead (a power of two; LDS ring slots per wave)
#endif
constexpr int kQuadActAhead = PW_QUAD_ACT_AHEAD;
constexpr int kQuadActRingBytes = kQuadActAhead * kWave * (int)sizeof(int32_t);  // per physics wave
template <bool UNIT_MASS, bool COLL = false, bool K1 = false>
__global__ void __launch_bounds__(4 * kWave) pw_spread_quad_kernel(const StreamParams A, const int T)
{
    constexpr int N = 6, L = 6, D = 16, P2 = 15, EPP = 4, EPW = 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float4 *s_ring = reinterpret_cast<float4 *>(smem_raw);                 // [4][64] {px, py, vx, vy}, index e_local * 6 + a
    float2 *s_ftab = reinterpret_cast<float2 *>(s_ring + 4 * kWave);       // [2 P waves][24 agents][6 partners]
    float *s_min = reinterpret_cast<float *>(s_ftab + 2 * EPP * N * N);    // [64] OA
    float *s_rew = s_min + kWave;                                          // [64] OA
    float2 *s_lmB = reinterpret_cast<float2 *>(s_rew + kWave);             // [8 * 6] OB (16-byte aligned)
    int32_t *s_act = reinterpret_cast<int32_t *>(s_lmB + EPW * L);         // [2 P waves][4 steps][64] action indices
    float2 *s_utab = reinterpret_cast<float2 *>(s_act + 2 * kQuadActAhead * kWave);    // [2 P waves][8] action force per index

    // Roles by wave index, swapped in every other batch of 256 workgroups: the hardware places a workgroup's waves 0..3 on
    // the CU's SIMDs in order, and with two workgroups per CU (B = 4096: 512 workgroups on 256 CUs, workgroup j and j + 256
    // on one CU) equal roles would share a SIMD -- the two physics waves, the step's critical path, competing for issue
    // while two output waves with hundreds of cycles of slack share another.  Swapped, every SIMD holds one physics and one
    // output wave: -2.2 % step time (A/B, profiles/r3_quad_waves.txt).
    const int wave = __builtin_amdgcn_readfirstlane((((int)threadIdx.x >> 6) + 2 * (((int)blockIdx.x >> 8) & 1)) & 3);
    const int lane = (int)threadIdx.x & 63;
    const int env0 = (int)blockIdx.x * EPW;
    const int envs_here = A.B - env0 < EPW ? A.B - env0 : EPW;
    const size_t BN = (size_t)A.B * N;

    // Episode clocks.  Every wave follows the clocks of all 8 envs (lane e < 8 <-> env e), but not by counting: a clock
    // is kept as an offset -- clock before step t = t + off -- so nothing has to be incremented, and the workgroup-uniform
    // question "does any env reset in this step?" is one scalar compare with t_reset, the first step in which one does;
    // the vector work (which envs, the new offsets, the next t_reset) runs only in those steps.
    const bool resets = A.auto_reset && A.max_episode_len > 0;
    int offs_all = A.ep_step[env0 + (lane < envs_here ? lane : 0)];
    auto next_reset_step = [&](int t_from) -> int {  // first step >= t_from in which some env's clock reaches max_episode_len
        if (!resets) return 0x7fffffff;
        int rem = A.max_episode_len - 1 - (t_from + offs_all);
        rem = rem < 0 ? 0 : rem;
        int m = 0x7fffffff;
        for (int e = 0; e < envs_here; ++e) {  // 8 scalar reads: runs once per episode end
            const int r = __builtin_amdgcn_readlane(rem, e);
            m = r < m ? r : m;
        }
        return t_from + m;
    };
    int t_reset = next_reset_step(0);
    auto reset_step_update = [&](int t) {  // call in a step with t == t_reset, once: the clocks that reached the end restart
        if (t + 1 + offs_all >= A.max_episode_len) offs_all = -(t + 1);
        t_reset = next_reset_step(t + 1);
    };
    int cur = 0;  // ring slot of the current state: workgroup-uniform

    // Measured and NOT kept (round 3, profiles/r3_quad_waves.txt): LDS step counters instead of the per-step s_barrier (the
    // physics waves polling-free, the output waves polling with s_sleep).  With the output waves shortened the barrier costs
    // a physics wave little beyond the LDS wait it needs anyway (its next operands), and the counters' own reads and writes
    // cost more: 0.656 instead of 0.615 us per step.
    if (wave < 2) {
        // ================================ waves P0, P1: pair-parallel physics ================================
        // the physics waves are the step's critical path, the output waves have slack: where one of each shares a SIMD
        // (two workgroups per CU at B = 4096) the arbiter should serve the physics wave first
        __builtin_amdgcn_s_setprio(3);
        const int e0 = wave * EPP;
        const int nv = envs_here - e0 < 0 ? 0 : envs_here - e0 < EPP ? envs_here - e0 : EPP;  // envs of this wave
        if (nv == 0) {  // nothing to advance: keep the workgroup's barriers company
            for (int t = 0; t < T; ++t) PW_QUAD_BARRIER(t);
            return;
        }
        // agent lanes (lane < nv * 6; the others shadow lane 0)
        const int la = lane < nv * N ? lane : 0;
        const int e4 = la / N, a = la - e4 * N;
        const int me = (e0 + e4) * N + a;                       // ring index
        const int env = env0 + e0 + e4;
        const uint32_t g = (uint32_t)env * (uint32_t)N + (uint32_t)a;
        const uint64_t env_id = A.env_id_base + (uint64_t)env;
        // pair lanes (lane < nv * 15; the others shadow pair 0)
        const int lp = lane < nv * P2 ? lane : 0;
        const int pe = lp / P2, pq = lp - pe * P2;
        int pi, pj;
        quad_pair_of(pq, pi, pj);
        const int ri = (e0 + pe) * N + pi, rj = (e0 + pe) * N + pj;
        // force table [agent][partner slot]: a row holds the agent's 5 partners in ascending order (slot = partner index,
        // minus one behind the agent's own index), padded to 6 entries so that rows stay 16-byte aligned
        float2 *ftab = s_ftab + wave * (EPP * N * N);
        float2 *f_ij = ftab + (pe * N + pi) * N + (pj - 1), *f_ji = ftab + (pe * N + pj) * N + pi;   // pi < pj
        const float2 *row = ftab + (e4 * N + a) * N;

        float px = A.pos_x[g], py = A.pos_y[g], vx = A.vel_x[g], vy = A.vel_y[g];
        int ep_off = A.ep_step[env];  // clock before step t = t + ep_off
        uint32_t ep_count = A.ep_count[env];
        s_ring[me] = make_float4(px, py, vx, vy);
        // U2 + U4 as a table: the action force of an index is one of five constants, computed here once with the step's
        // own expressions (so the bits are the step's), entry 5 = any other index (no force); a step reads ONE entry
        float2 *utab = s_utab + wave * 8;
        if (lane < 6) {
            const int ai = lane;
            float ux = 0.0f + ((ai == 1 ? 1.0f : 0.0f) - (ai == 2 ? 1.0f : 0.0f));
            float uy = 0.0f + ((ai == 3 ? 1.0f : 0.0f) - (ai == 4 ? 1.0f : 0.0f));
            ux *= A.sens; uy *= A.sens;
            if (A.fscale != 1.0f) { ux = A.fscale * ux; uy = A.fscale * uy; }
            utab[lane] = make_float2(ux + 0.0f, uy + 0.0f);
        }
        wave_lds_sync();
        const float k = A.contact_margin, cf = A.contact_force, dt = A.dt, damp = A.damp, mass = A.mass;
        const uint32_t near_lo = __float_as_uint(A.near_thr2), near_span = 0x7F800000u - near_lo;
        // Action indices: fetched four steps ahead by LDS-direct loads (pw_common.hpp, act_fetch_issue: under a full chip
        // a load issued one step ahead cost 385 cycles of every step)
        const int32_t *act_g = A.act + g;
        int32_t *act_ring = s_act + wave * (kQuadActAhead * kWave);
        const uint32_t act_lds = __builtin_amdgcn_readfirstlane((uint32_t)reinterpret_cast<uintptr_t>(act_ring));
        auto fetch_act = [&](int t) {  // indices of step t (clamped: the tail re-fetches the last step) -> slot t & 3
            act_fetch_issue(act_g + (size_t)(t < T ? t : T - 1) * BN, act_lds + (uint32_t)(t & (kQuadActAhead - 1)) * (kWave * 4));
        };
        // every load the compiler counts is consumed before the first uncounted one is issued: a counted wait inside the
        // loop (for a value first used there) would be short by the fetches in flight, i.e. drain them
        asm volatile("" :: "v"(ep_off), "v"(ep_count), "v"(offs_all), "v"(px), "v"(py), "v"(vx), "v"(vy), "s"(t_reset));
        for (int t0 = 0; t0 < kQuadActAhead; ++t0) fetch_act(t0);
        // the pair lanes' operands of the coming step are fetched right after the publish, before the barrier (this
        // wave only reads its own envs' entries, and a wave's LDS operations execute in issue order): the read's latency
        // hides behind the barrier
        float2 qi = *reinterpret_cast<const float2 *>(s_ring + ri);
        float2 qj = *reinterpret_cast<const float2 *>(s_ring + rj);
        PW_STAMP_DECL;
        for (int t = 0; t < T; ++t) {
            PW_STAMP_START;
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kQuadActAhead - 1) : "memory");  // step t's indices are in LDS (the later fetches stay in flight)
            PW_STAMP(3);
            const uint32_t ai = (uint32_t)act_ring[(t & (kQuadActAhead - 1)) * kWave + lane];
            fetch_act(t + kQuadActAhead);  // into the slot just read (its wait for the read above is the one the table lookup needs anyway)
            const float2 u0 = utab[ai < 5u ? ai : 5u];  // {u_x + 0, u_y + 0}: the accumulators' starting values
            const bool two_slots = t == t_reset;        // workgroup-uniform: some env of the workgroup resets in this step
            // ---- pair phase: every unordered pair of the wave's envs at once
            float Fx = 0.0f, Fy = 0.0f;
            {
                // the near test on (p_i - p_j)^2, the force's own delta, so that the two share their first operations (the other
                // kernels test (p_j - p_i)^2: the same value -- a difference and its negation have the same square): -1.7 % step time
                const float dx = qi.x - qj.x, dy = qi.y - qj.y;
                const float d2 = dx * dx + dy * dy;
                if (__float_as_uint(d2) - near_lo >= near_span)  // not provably far (NaN / inf included)
                    collision_force_pair<true, K1>(qi.x, qi.y, qj.x, qj.y, A.dist_min, k, cf, Fx, Fy);
            }
            *f_ij = make_float2(Fx, Fy);
            *f_ji = make_float2(-Fx, -Fy);
            PW_STAMP(0);
            float fx = u0.x, fy = u0.y;
            asm volatile("" ::: "memory");  // program-order point between table writes and row reads
            // ---- U5: the agent's row, ascending partner order
#pragma unroll
            for (int j = 0; j < N - 1; ++j) {
                const float2 F = row[j];
                fx = F.x + fx;
                fy = F.y + fy;
            }
            // ---- U6
            vx = vx * damp; vy = vy * damp;
            vx = vx + div_mass<UNIT_MASS>(fx, mass) * dt;
            vy = vy + div_mass<UNIT_MASS>(fy, mass) * dt;
            px = px + vx * dt;
            py = py + vy * dt;
            int nxt = (cur + 1) & 3;
            s_ring[nxt * kWave + me] = make_float4(px, py, vx, vy);
            if (two_slots) {  // rare (once per episode): the envs at their episode's end restart, EVERY env publishes a second slot
                if (t + 1 + ep_off >= A.max_episode_len) {
                    ep_count += 1;
                    ep_off = -(t + 1);
                    pw_reset_xy(A.seed, env_id, ep_count, (uint32_t)a, -1.0f, 1.0f, &px, &py);
                    vx = 0.f; vy = 0.f;
                }
                reset_step_update(t);
                nxt = (nxt + 1) & 3;
                s_ring[nxt * kWave + me] = make_float4(px, py, vx, vy);
            }
            cur = nxt;
            asm volatile("" ::: "memory");
            qi = *reinterpret_cast<const float2 *>(s_ring + cur * kWave + ri);
            qj = *reinterpret_cast<const float2 *>(s_ring + cur * kWave + rj);
            PW_STAMP(1);
            PW_QUAD_BARRIER(t);
            PW_STAMP(2);
        }
        act_fetch_drain();  // the tail's fetches have landed before the wave ends
#ifdef PW_STAMPS
        if (blockIdx.x == 0 && wave == 0 && lane == 0)
            for (int i_ = 0; i_ < 4; ++i_) g_pw_stamps[i_] = st_acc[i_];
#endif
        A.pos_x[g] = px; A.pos_y[g] = py;
        A.vel_x[g] = vx; A.vel_y[g] = vy;
        A.ep_step[env] = T + ep_off;
        A.ep_count[env] = ep_count;
        return;
    }

    // ================================ waves OA, OB: outputs of 8 envs, one step behind ================================
    int e_local = lane / N;
    int a = lane - e_local * N;
    if (e_local >= envs_here) { e_local = 0; a = 0; }  // idle lane: shadow lane 0
    const int env = env0 + e_local;
    const int base = e_local * N, me = base + a;
    const uint32_t g = (uint32_t)env * (uint32_t)N + (uint32_t)a;
    const uint64_t env_id = A.env_id_base + (uint64_t)env;
    int ep_off = A.ep_step[env];  // clock before step t = t + ep_off
    uint32_t ep_count = A.ep_count[env];
    float olx = A.lm_x[(size_t)env * L + a], oly = A.lm_y[(size_t)env * L + a];  // the landmark this lane owns

    if (wave == 2) {
        // (round 2 ran this wave at priority 3 too, when it was as long as the physics waves; software-pipelined it has
        // several hundred cycles of slack per step and yields to them)
        // ---------------- OA: masks, rewards, small stores ----------------
        // A step's work has two halves: A(t) -- read the published slot, the six collision tests, the owned landmark's
        // minimum over the agents, its sqrt -- and B(t) -- the ordered per-env sums through two rounds of wave shuffles,
        // the stores.  Each half is a chain of long-latency operations (LDS round trips, shuffles, sqrt) with little to
        // issue in between, and B(t) needs nothing but A(t)'s three registers: so the loop is software-pipelined, iteration
        // t running B(t - 1) interleaved with A(t) (the shuffles of one fly while the arithmetic of the other issues).
        // Same operations on the same values: the bits do not change.
        float own_p = 0.0f;   // A(t - 1): sqrt of the owned landmark's minimum squared distance
        int cnt_p = 0;        //           number of agents within the collision threshold (self included)
        uint32_t cmask_p = 0; //           their mask (COLL)
        // the pieces, in the order an iteration interleaves them
        float2 mine, q6[N];
        auto a_reads = [&](const int nxt) __attribute__((always_inline)) {
            const float4 *slot = s_ring + nxt * kWave + base;
            mine = *reinterpret_cast<const float2 *>(slot + a);
#pragma unroll
            for (int j = 0; j < N; ++j) q6[j] = *reinterpret_cast<const float2 *>(slot + j);
        };
        float best, e0;
        int cnt;
        uint32_t cmask;
        auto a_dist = [&]() __attribute__((always_inline)) {
            const float px = mine.x, py = mine.y;
            float e2[N];
            cnt = 0; cmask = 0;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const float2 q = q6[j];
                const float dx = q.x - px, dy = q.y - py;
                const float d2 = dx * dx + dy * dy;
                cnt += d2 < A.coll_thr2 ? 1 : 0;
                if (COLL) cmask |= d2 < A.coll_thr2 ? 1u << j : 0u;
                const float ex = q.x - olx, ey = q.y - oly;
                e2[j] = ex * ex + ey * ey;
            }
            // min() keeps its first argument unless a later one is smaller: a NaN in front stays, NaNs behind are
            // skipped -- i.e. the NaN-ignoring minimum of all six unless the first one is NaN (e2 is never -0): a tree
            best = __builtin_fminf(__builtin_fminf(__builtin_fminf(e2[0], e2[1]), __builtin_fminf(e2[2], e2[3])),
                                   __builtin_fminf(e2[4], e2[5]));
            e0 = e2[0];
        };
        auto a_sqrt = [&]() __attribute__((always_inline)) -> float {
            const float b = e0 != e0 ? e0 : best;
            return sqrtf(b);  // branch-free expansion: the guarded fast form (a vector compare feeding exec) measured 8 % slower on this chain
        };
        auto a_clock = [&](const int t, int nxt) __attribute__((always_inline)) {  // episode clocks; the slot A(t + 1) reads
            if (t == t_reset) {  // workgroup-uniform, once per episode
                if (t + 1 + ep_off >= A.max_episode_len) {
                    ep_count += 1;
                    ep_off = -(t + 1);
                    pw_reset_xy(A.seed, env_id, ep_count, (uint32_t)(N + a), -1.0f, 1.0f, &olx, &oly);
                }
                reset_step_update(t);
                nxt = (nxt + 1) & 3;
            }
            cur = nxt;
        };
        float sh_own[L], sh_r[N], r;
        auto b_shfl_own = [&]() __attribute__((always_inline)) {  // per-env reductions by wave shuffle (ds_bpermute: one trip each, no LDS write -> wait -> read)
#pragma unroll
            for (int l = 0; l < L; ++l) sh_own[l] = __shfl(own_p, base + l, kWave);
        };
        auto b_reward = [&]() __attribute__((always_inline)) {
            r = 0.0f;
#pragma unroll
            for (int l = 0; l < L; ++l) r -= sh_own[l];
            // "rew -= 1" once per colliding agent (itself included): the subtrahends are all the same, so only their
            // number matters; r - 0 is r, so the six steps are selects of the subtrahend, not branches (a loop up to the
            // wave's largest count was measured: a vector compare feeding a scalar branch per iteration costs more)
#pragma unroll
            for (int k2 = 0; k2 < N; ++k2) r -= k2 < cnt_p ? 1.0f : 0.0f;
#pragma unroll
            for (int i = 0; i < N; ++i) sh_r[i] = __shfl(r, base + i, kWave);
        };
        auto b_store = [&](const int tb) __attribute__((always_inline)) {
            const size_t tBN = (size_t)tb * BN;
            float acc = 0.0f;
#pragma unroll
            for (int i = 0; i < N; ++i) acc += sh_r[i];
            nt_store(A.rew + tBN + g, r);
            nt_store(A.rew_shared + (size_t)tb * A.B + env, acc);   // (done / terminal: wave OB, which has the slack)
            if (COLL) nt_store(A.coll + tBN + g, (uint64_t)cmask_p);  // is_collision bits of the state step tb produced (pre-reset)
        };
        PW_STAMP_DECL;
        // prologue: A(0)
        {
            PW_STAMP_START;
            PW_QUAD_BARRIER(0);
            PW_STAMP(0);
            const int nxt = (cur + 1) & 3;
            a_reads(nxt);
            a_dist();
            own_p = a_sqrt(); cnt_p = cnt; cmask_p = cmask;
            a_clock(0, nxt);
            PW_STAMP(1);
        }
        for (int t = 1; t < T; ++t) {  // A(t) interleaved with B(t - 1): no branch but the (rare) reset step's
            PW_STAMP_START;
            PW_QUAD_BARRIER(t);
            PW_STAMP(0);
            const int nxt = (cur + 1) & 3;
            a_reads(nxt);
            b_shfl_own();
            a_dist();
            b_reward();
            const float own = a_sqrt();
            b_store(t - 1);
            own_p = own; cnt_p = cnt; cmask_p = cmask;
            a_clock(t, nxt);
            PW_STAMP(1);
        }
        b_shfl_own();  // epilogue: B(T - 1)
        b_reward();
        b_store(T - 1);
#ifdef PW_STAMPS
        if (blockIdx.x == 0 && lane == 0)
            for (int i_ = 0; i_ < 2; ++i_) g_pw_stamps[4 + i_] = st_acc[i_];
#endif
        return;
    }

    // ---------------- OB: observation rows ----------------
    // The wave's 48 rows are contiguous in the obs plane and are stored as ONE block: 192 chunks of 16 bytes, chunk
    // q = row * 4 + column group, lane l stores chunks l, l + 64, l + 128 -- 1 KiB contiguous per store instruction.
    // With D = 16 a lane's column group c = l & 3 never changes and its rows are l / 4, + 16, + 32: c = 0 is the row's
    // {vel, pos}, c >= 1 the landmarks 2c - 2, 2c - 1 relative to the row's agent (stream_write_obs's subtractions, only
    // the storing lane differs).  The rows' {pos, vel} are read STRAIGHT from the ring slot (its index is the row index),
    // and a lane's three landmark pairs live in registers for the whole episode (they change in reset steps only): a
    // step is three 16-byte LDS reads, the subtractions and three stores.
    const int rows_here = envs_here * N;
    const int cgrp = lane & 3;
    int row_u[3];
    bool ok_u[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int r = (lane >> 2) + 16 * u;
        ok_u[u] = r < rows_here;
        row_u[u] = ok_u[u] ? r : 0;  // chunks past the block read row 0 (and store nothing)
    }
    float4 lm_u[3];
    auto load_landmarks = [&]() {  // the two landmarks of this lane's column group, per row's env
#pragma unroll
        for (int u = 0; u < 3; ++u)
            lm_u[u] = *reinterpret_cast<const float4 *>(s_lmB + (row_u[u] / N) * L + (cgrp > 0 ? 2 * cgrp - 2 : 0));
    };
    s_lmB[me] = make_float2(olx, oly);
    wave_lds_sync();
    load_landmarks();
    float4 *const blk4 = reinterpret_cast<float4 *>(A.obs + ((size_t)env0 * N) * D) + lane;
    PW_STAMP_DECL;
    for (int t = 0; t < T; ++t) {
        const size_t tBN = (size_t)t * BN;
        PW_STAMP_START;
        nt_store(A.done + tBN + g, (uint8_t)0);
        PW_QUAD_BARRIER(t);
        PW_STAMP(0);
        int nxt = (cur + 1) & 3;
        nt_store(A.terminal + (size_t)t * A.B + env, (uint8_t)(A.max_episode_len > 0 && t + 1 + ep_off >= A.max_episode_len ? 1 : 0));
        if (t == t_reset) {  // workgroup-uniform, once per episode
            const bool rst = t + 1 + ep_off >= A.max_episode_len;
            if (rst) {
                if (A.final_obs) {
                    const float4 st = s_ring[nxt * kWave + me];  // this lane's own (env, agent) row, pre-reset
                    stream_write_obs<L>(A.final_obs + (tBN + g) * D, L, s_lmB + base, st.x, st.y, st.z, st.w);
                }
                ep_count += 1;
                ep_off = -(t + 1);
                pw_reset_xy(A.seed, env_id, ep_count, (uint32_t)(N + a), -1.0f, 1.0f, &olx, &oly);
            }
            reset_step_update(t);
            wave_lds_sync();     // the pre-reset rows have read the old landmarks
            if (rst) s_lmB[me] = make_float2(olx, oly);
            wave_lds_sync();
            load_landmarks();
            nxt = (nxt + 1) & 3;  // the post-reset slot (the same state for envs that did not reset)
        }
        cur = nxt;
        const float4 *slot = s_ring + nxt * kWave;
        float4 *const out4 = blk4 + tBN * (D / 4);
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const float4 st = slot[row_u[u]];
            const float4 o = cgrp == 0 ? make_float4(st.z, st.w, st.x, st.y)
                                       : make_float4(lm_u[u].x - st.x, lm_u[u].y - st.y, lm_u[u].z - st.x, lm_u[u].w - st.y);
            if (ok_u[u]) nt_store(out4 + 64 * u, o);
        }
        PW_STAMP(1);
    }
#ifdef PW_STAMPS
    if (blockIdx.x == 0 && lane == 0)
        for (int i_ = 0; i_ < 2; ++i_) g_pw_stamps[6 + i_] = st_acc[i_];
#endif
    A.lm_x[(size_t)env * L + a] = olx;
    A.lm_y[(size_t)env * L + a] = oly;
}

}  // namespace
