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
// Arithmetic and results are identical to the other simple_spread kernels (same bit-exact tests, `quad` path).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void quad_pair_of(const int q, int &i, int &j)
{
    // unordered pairs of 6 agents in lexicographic order: (0,1) (0,2) ... (0,5) (1,2) ... (4,5)
    i = q < 5 ? 0 : q < 9 ? 1 : q < 12 ? 2 : q < 14 ? 3 : 4;
    j = q - (i == 0 ? 0 : i == 1 ? 5 : i == 2 ? 9 : i == 3 ? 12 : 14) + i + 1;
}

template <bool UNIT_MASS>
__global__ void __launch_bounds__(4 * kWave) pw_spread_quad_kernel(const StreamParams A, const int T)
{
    constexpr int N = 6, L = 6, D = 16, P2 = 15, EPP = 4, EPW = 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float4 *s_ring = reinterpret_cast<float4 *>(smem_raw);                 // [4][64] {px, py, vx, vy}, index e_local * 6 + a
    float2 *s_ftab = reinterpret_cast<float2 *>(s_ring + 4 * kWave);       // [2 P waves][24 agents][6 partners]
    float *s_min = reinterpret_cast<float *>(s_ftab + 2 * EPP * N * N);    // [64] OA
    float *s_rew = s_min + kWave;                                          // [64] OA
    float2 *s_lmB = reinterpret_cast<float2 *>(s_rew + kWave);             // [8 * 6] OB (16-byte aligned)
    float4 *s_rowB = reinterpret_cast<float4 *>(s_lmB + EPW * L);          // [64] OB: {pos, vel} of every row
    int32_t *s_act = reinterpret_cast<int32_t *>(s_rowB + kWave);          // [2 P waves][4 steps][64] action indices

    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int lane = (int)threadIdx.x & 63;
    const int env0 = (int)blockIdx.x * EPW;
    const int envs_here = A.B - env0 < EPW ? A.B - env0 : EPW;
    const size_t BN = (size_t)A.B * N;

    // every wave follows the episode clocks of all 8 envs (lane e < 8 <-> env e): is there a reset in this step?
    int eps_all = A.ep_step[env0 + (lane < envs_here ? lane : 0)];
    auto any_reset_step = [&]() -> bool {
        const int e1 = eps_all + 1;
        const bool rst = A.auto_reset && A.max_episode_len > 0 && e1 >= A.max_episode_len;
        eps_all = rst ? 0 : e1;
        return __any(rst && lane < envs_here);
    };
    int cur = 0;  // ring slot of the current state: workgroup-uniform

    if (wave < 2) {
        // ================================ waves P0, P1: pair-parallel physics ================================
        const int e0 = wave * EPP;
        const int nv = envs_here - e0 < 0 ? 0 : envs_here - e0 < EPP ? envs_here - e0 : EPP;  // envs of this wave
        if (nv == 0) {  // nothing to advance: keep the workgroup's barriers company
            for (int t = 0; t < T; ++t) duo_barrier();
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
        float2 *ftab = s_ftab + wave * (EPP * N * N);
        float2 *f_ij = ftab + (pe * N + pi) * N + pj, *f_ji = ftab + (pe * N + pj) * N + pi;
        const float2 *row = ftab + (e4 * N + a) * N;

        float px = A.pos_x[g], py = A.pos_y[g], vx = A.vel_x[g], vy = A.vel_y[g];
        int ep_step = A.ep_step[env];
        uint32_t ep_count = A.ep_count[env];
        s_ring[me] = make_float4(px, py, vx, vy);
        // the table's diagonal is never written by a pair lane: +0 there lets the agent lanes add the whole row (an
        // accumulator that is never -0 is unchanged by + 0) instead of selecting around their own index
        ftab[(e4 * N + a) * N + a] = make_float2(0.0f, 0.0f);
        wave_lds_sync();
        const float k = A.contact_margin, cf = A.contact_force, dt = A.dt, damp = A.damp, mass = A.mass;
        const uint32_t near_lo = __float_as_uint(A.near_thr2), near_span = 0x7F800000u - near_lo;
        // Action indices: fetched four steps ahead by LDS-direct loads (pw_common.hpp, act_fetch_issue: under a full chip
        // a load issued one step ahead cost 385 cycles of every step)
        const int32_t *act_g = A.act + g;
        int32_t *act_ring = s_act + wave * (4 * kWave);
        const uint32_t act_lds = __builtin_amdgcn_readfirstlane((uint32_t)reinterpret_cast<uintptr_t>(act_ring));
        auto fetch_act = [&](int t) {  // indices of step t (clamped: the tail re-fetches the last step) -> slot t & 3
            act_fetch_issue(act_g + (size_t)(t < T ? t : T - 1) * BN, act_lds + (uint32_t)(t & 3) * (kWave * 4));
        };
        // every load the compiler counts is consumed before the first uncounted one is issued: a counted wait inside the
        // loop (for a value first used there) would be short by the fetches in flight, i.e. drain them
        asm volatile("" :: "v"(ep_step), "v"(ep_count), "v"(eps_all), "v"(px), "v"(py), "v"(vx), "v"(vy));
        fetch_act(0); fetch_act(1); fetch_act(2); fetch_act(3);
        // the pair lanes' operands of the coming step are fetched right after the publish, before the barrier (this
        // wave only reads its own envs' entries, and a wave's LDS operations execute in issue order): the read's latency
        // hides behind the barrier
        float2 qi = *reinterpret_cast<const float2 *>(s_ring + ri);
        float2 qj = *reinterpret_cast<const float2 *>(s_ring + rj);
        PW_STAMP_DECL;
        for (int t = 0; t < T; ++t) {
            PW_STAMP_START;
            act_fetch_wait3();  // step t's indices are in LDS
            PW_STAMP(3);
            const int ai = act_ring[(t & 3) * kWave + lane];
            const bool two_slots = any_reset_step();  // workgroup-uniform; early: it does not depend on the physics
            // ---- pair phase: every unordered pair of the wave's envs at once
            float Fx = 0.0f, Fy = 0.0f;
            {
                const float dx = qj.x - qi.x, dy = qj.y - qi.y;   // the near test of the other kernels: (q - p)^2
                const float d2 = dx * dx + dy * dy;
                if (__float_as_uint(d2) - near_lo >= near_span)  // not provably far (NaN / inf included)
                    collision_force_pair<true>(qi.x, qi.y, qj.x, qj.y, A.dist_min, k, cf, Fx, Fy);
            }
            *f_ij = make_float2(Fx, Fy);
            *f_ji = make_float2(-Fx, -Fy);
            PW_STAMP(0);
            // ---- U2 + U4 of the agent lanes, meanwhile
            float ux = 0.0f + ((ai == 1 ? 1.0f : 0.0f) - (ai == 2 ? 1.0f : 0.0f));
            float uy = 0.0f + ((ai == 3 ? 1.0f : 0.0f) - (ai == 4 ? 1.0f : 0.0f));
            ux *= A.sens; uy *= A.sens;
            if (A.fscale != 1.0f) { ux = A.fscale * ux; uy = A.fscale * uy; }
            float fx = ux + 0.0f, fy = uy + 0.0f;
            fetch_act(t + 4);  // into the slot just read; also the program-order point between table writes and row reads
            // ---- U5: the agent's row, ascending partner order
#pragma unroll
            for (int j = 0; j < N; ++j) {
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
            ep_step += 1;
            if (A.auto_reset && A.max_episode_len > 0 && ep_step >= A.max_episode_len) {
                ep_count += 1;
                ep_step = 0;
                pw_reset_xy(A.seed, env_id, ep_count, (uint32_t)a, -1.0f, 1.0f, &px, &py);
                vx = 0.f; vy = 0.f;
            }
            if (two_slots) {  // every env publishes a second slot
                nxt = (nxt + 1) & 3;
                s_ring[nxt * kWave + me] = make_float4(px, py, vx, vy);
            }
            cur = nxt;
            asm volatile("" ::: "memory");
            qi = *reinterpret_cast<const float2 *>(s_ring + cur * kWave + ri);
            qj = *reinterpret_cast<const float2 *>(s_ring + cur * kWave + rj);
            PW_STAMP(1);
            duo_barrier();
            PW_STAMP(2);
        }
        act_fetch_drain();  // the tail's fetches have landed before the wave ends
#ifdef PW_STAMPS
        if (blockIdx.x == 0 && wave == 0 && lane == 0)
            for (int i_ = 0; i_ < 4; ++i_) g_pw_stamps[i_] = st_acc[i_];
#endif
        A.pos_x[g] = px; A.pos_y[g] = py;
        A.vel_x[g] = vx; A.vel_y[g] = vy;
        A.ep_step[env] = ep_step;
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
    int ep_step = A.ep_step[env];
    uint32_t ep_count = A.ep_count[env];
    float olx = A.lm_x[(size_t)env * L + a], oly = A.lm_y[(size_t)env * L + a];  // the landmark this lane owns

    if (wave == 2) {
        // ---------------- OA: masks, rewards, small stores ----------------
        PW_STAMP_DECL;
        for (int t = 0; t < T; ++t) {
            const size_t tBN = (size_t)t * BN;
            PW_STAMP_START;
            duo_barrier();
            PW_STAMP(0);
            int nxt = (cur + 1) & 3;
            const float4 *slot = s_ring + nxt * kWave + base;
            const float2 mine = *reinterpret_cast<const float2 *>(slot + a);
            const float px = mine.x, py = mine.y;
            uint32_t coll = 0;
            float best = 0.0f;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const float2 q = *reinterpret_cast<const float2 *>(slot + j);
                const float dx = q.x - px, dy = q.y - py;
                const float d2 = dx * dx + dy * dy;
                if (d2 < A.coll_thr2) coll |= 1u << j;
                const float ex = q.x - olx, ey = q.y - oly;
                const float e2 = ex * ex + ey * ey;
                best = (j == 0 || e2 < best) ? e2 : best;
            }
            // per-env reductions by wave shuffle (ds_bpermute: one trip each, no LDS write -> wait -> read)
            const float own = sqrtf(best);
            float r = 0.0f;
#pragma unroll
            for (int l = 0; l < L; ++l) r -= __shfl(own, base + l, kWave);
#pragma unroll
            for (int j = 0; j < N; ++j)
                if ((coll >> j) & 1) r -= 1.0f;
            float acc = 0.0f;
#pragma unroll
            for (int i = 0; i < N; ++i) acc += __shfl(r, base + i, kWave);
            A.rew[tBN + g] = r;
            A.done[tBN + g] = 0;
            A.rew_shared[(size_t)t * A.B + env] = acc;
            ep_step += 1;
            const bool term = A.max_episode_len > 0 && ep_step >= A.max_episode_len;
            A.terminal[(size_t)t * A.B + env] = term ? 1 : 0;
            if (term && A.auto_reset) {
                ep_count += 1;
                ep_step = 0;
                pw_reset_xy(A.seed, env_id, ep_count, (uint32_t)(N + a), -1.0f, 1.0f, &olx, &oly);
            }
            if (any_reset_step()) { nxt = (nxt + 1) & 3; }
            cur = nxt;
            PW_STAMP(1);
        }
#ifdef PW_STAMPS
        if (blockIdx.x == 0 && lane == 0)
            for (int i_ = 0; i_ < 2; ++i_) g_pw_stamps[4 + i_] = st_acc[i_];
#endif
        return;
    }

    // ---------------- OB: observation rows ----------------
    // The wave's 48 rows are contiguous in the obs plane: stored as ONE block (stream_write_obs_block: 1 KiB per store
    // instruction instead of a quarter cache line per lane and instruction) -- at 5e9 env-steps/s the write path matters
    // at this batch size too.
    const int rows_here = envs_here * N;
    s_lmB[me] = make_float2(olx, oly);
    wave_lds_sync();
    PW_STAMP_DECL;
    for (int t = 0; t < T; ++t) {
        const size_t tBN = (size_t)t * BN;
        PW_STAMP_START;
        duo_barrier();
        PW_STAMP(0);
        int nxt = (cur + 1) & 3;
        float4 st = s_ring[nxt * kWave + me];
        ep_step += 1;
        const bool rst = A.auto_reset && A.max_episode_len > 0 && ep_step >= A.max_episode_len;
        if (rst) {
            if (A.final_obs) stream_write_obs<L>(A.final_obs + (tBN + g) * D, L, s_lmB + base, st.x, st.y, st.z, st.w);
            ep_count += 1;
            ep_step = 0;
            pw_reset_xy(A.seed, env_id, ep_count, (uint32_t)(N + a), -1.0f, 1.0f, &olx, &oly);
        }
        if (any_reset_step()) {  // workgroup-uniform
            wave_lds_sync();     // the pre-reset rows have read the old landmarks
            if (rst) s_lmB[me] = make_float2(olx, oly);
            nxt = (nxt + 1) & 3;
            st = s_ring[nxt * kWave + me];  // post-reset state (the same state for envs that did not reset)
        }
        cur = nxt;
        s_rowB[me] = st;
        wave_lds_sync();
        stream_write_obs_block<N, L>(A.obs + (tBN + (size_t)env0 * N) * D, rows_here, lane, s_rowB, s_lmB);
        wave_lds_sync();  // the block's LDS reads are done before s_rowB / s_lmB change again
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
