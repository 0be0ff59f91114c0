// pw_kernels_policy3.hpp -- part of libpworld.so (translation unit csrc/pworld_policy.hip includes it).
// Policy-in-the-loop rollout, third form: the whole BiLSTM -- input projection AND recurrence -- on the matrix cores, one
// timestep at a time, every weight resident in registers, no G tile in LDS.
#pragma once

namespace {

// ------------------------------------------------------------------------------------------
// What the stamps of the second form say (profiles/r3_policy_phases.txt): the f32 MFMA and the vector ALU of a SIMD share
// their multipliers, so "matrix waves" and "LSTM waves" take turns anyway, and while one role works the other's four waves
// idle (18 k of a step's 47 k cycles); and the recurrence runs at the packed-FMA rate a single wave can issue
// (5.5 cycles per v_pk_fma_f32, 23 MAC/clk/SIMD) where the matrix pipe does 32.
//
// Here a workgroup still owns 16 environments, but its eight waves are identical: wave w = (direction w / 4, hidden
// quarter w % 4) owns 8 hidden units x 4 gates = 32 gate units as two 16-row MFMA tiles and runs, per LSTM timestep,
//     acc  = W_ih(tile) * x1(ts)        16 x v_mfma_f32_16x16x4_f32 per tile  (k in the order of the other forms)
//     acc += b_ih + b_hh
//     acc  = W_hh(tile) * h(ts - 1) + acc  8 x v_mfma_f32_16x16x4_f32 per tile (k ascending)
//     cell update in the lane, h -> LDS (exchange + head input), one workgroup barrier per timestep
// with the 16 sequences of the workgroup as the MFMA's 16 columns.  Tile rows are ordered (unit, gate): accumulator
// register i of lane (column n, row group rg) is gate i of unit 4 * tile + rg -- all four gates of a cell in one lane, no
// cross-lane traffic.  A fragments: 32 (W_ih) + 16 (W_hh) VGPRs per lane for the whole launch; nothing but h (8 KB, double
// buffered) and the dense1 output (4 KB per timestep) passes through LDS.  The input projection of timestep ts + 1 is
// issued before the barrier of timestep ts, so the matrix pipe has work while the workgroup meets.
//   dense1 + ReLU: as 32 x 32 blocks on v_mfma_f32_32x32x2_f32 (one block per wave, once per workgroup), scattered to LDS
//   directly in the B-fragment order of the timestep loop.
//   head: logits^T = W2 * relu(h)^T on the same 16x16x4 tiles (W2 as A fragments, C-in = b2), minus the Gumbel noise, arg-max in
//   the lanes.  Gumbel noise: one Philox block per four logits, drawn for the NEXT step by the waves without environment duty
//   while the environment waves advance the agents.
//   environment step: on the LAST waves (lane = (env, agent), pw_spread_stream_kernel's arithmetic), in three pieces: advance the
//   agents and publish the next observation rows (all the other waves wait for); partner pass + episode step count beside the next
//   step's dense1 blocks; rewards and every global store beside its head tiles (both on the first waves).  A wave with an episode
//   ending in the step does everything, the reset included, before it publishes.
//   (pw_kernels_actor16.hpp holds the same pass as a function for the kernels that are not built around it.)
// Bits: v_mfma_f32_16x16x4_f32 is a chain of fused multiply-adds over k = 0..3 in order (tools/mfma16_probe.hip: 0 of
// 512000 elements differ), as the 32x32x2 form is over its two k; the input projection feeds k in the order the other
// forms' 32x32x2 chains use (pairs {k, k + 4}), the bias is added to the finished sum, the recurrence continues the same
// accumulator with k ascending -- element for element the operation sequence of pw_bilstm_kernel.  (At the first timestep
// h = 0 and the other forms run the chain with zeros, which leaves every nonzero accumulator unchanged; here it is skipped.)
// LDS (N = 6, D = 16): 24 KB x1 fragments + 8 KB h exchange + 24 KB head input + 18 KB small = 74 KB.
// ------------------------------------------------------------------------------------------
struct Roll3Lds {
    float4 *s_xf;    // [N timesteps][4 j][64 lane]: element e of (j, lane (n, kq)) = x1[row (ts, n)][kpos(16 j + 4 e + kq)]
    float4 *s_hx;    // [2 buffers][2 dir][2 j][64 lane]: element e = h[seq n][16 j + 4 e + kq]
    float4 *s_hf;    // head input in B-fragment order: [rows / 16 tiles][4 j][64 lane], element e of (j, lane (n, kq)) =
                     // relu(h)[row 16 tile + n][16 j + 4 e + kq] (k < 32 forward, >= 32 reverse), rows env-major
    float *f_w1;     // [2 m][S1][64 lane]
    float *s_b1, *s_b2;
    float *s_noise;  // [rows * 5] Gumbel noise of the coming head
    float *s_obs;    // [rows][DS], DS = D + 2
    int32_t *s_act;  // [rows]
    float2 *s_posb;  // [8 env waves][64] (16 environments of N agents: up to 8 waves of whole environments)
    float2 *s_lmb;   // [E * L]
    double *s_fs;    // [16]
    int *s_fc;       // [16]
};
__host__ __device__ inline size_t roll3_lds_bytes(int E, int N, int L, int D, int S1)
{
    const size_t rows = (size_t)E * N;
    size_t fl = (size_t)N * 4 * 64 * 4 + 2 * 2 * 2 * 64 * 4 + ((rows + 15) / 16) * 1024 + (size_t)2 * S1 * 64 + 64 + 16 + rows * 5 + 1 +
                rows * (D + 2) + rows + 1;
    return fl * 4 + 8 * kWave * sizeof(float2) + (size_t)E * L * sizeof(float2) + 16 * (sizeof(double) + sizeof(int)) + 64;
}
__device__ __forceinline__ Roll3Lds roll3_carve(unsigned char *raw, int E, int N, int L, int D, int S1)
{
    // offsets in floats from the (16-byte aligned) base; no pointer <-> integer casts (LDS address space kept)
    const int rows = E * N;
    float *base = reinterpret_cast<float *>(raw);
    Roll3Lds S;
    int o = 0;
    S.s_xf = reinterpret_cast<float4 *>(base + o); o += N * 4 * 64 * 4;
    S.s_hx = reinterpret_cast<float4 *>(base + o); o += 2 * 2 * 2 * 64 * 4;
    S.s_hf = reinterpret_cast<float4 *>(base + o); o += ((rows + 15) / 16) * 1024;
    S.f_w1 = base + o; o += 2 * S1 * 64;
    S.s_b1 = base + o; o += 64;
    S.s_b2 = base + o; o += 16;
    S.s_noise = base + o; o += rows * 5;
    o = (o + 1) & ~1;
    S.s_obs = base + o; o += rows * (D + 2);
    S.s_act = reinterpret_cast<int32_t *>(base + o); o += rows;
    o = (o + 1) & ~1;
    S.s_posb = reinterpret_cast<float2 *>(base + o); o += 8 * kWave * 2;
    S.s_lmb = reinterpret_cast<float2 *>(base + o); o += E * L * 2;
    S.s_fs = reinterpret_cast<double *>(base + o); o += 32;
    S.s_fc = reinterpret_cast<int *>(base + o);
    return S;
}

template <int S1C, int NT, bool SINK, bool BF3 = false>  // BF3: the opt-in, not exact bf16x3 input projection (pw_kernels_actor16.hpp)
__global__ void __launch_bounds__(512) pw_policy_rollout3_kernel(const PolicyRolloutArgs P)
{
    constexpr int LT = NT;
    constexpr int S1 = 4 * S1C;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const ActorFusedArgs &A = P.A;
    const StreamParams &V = P.V;
    const int N = NT ? NT : A.N, L = LT ? LT : V.L, D = A.D, DS = D + 2, E = A.E;
    const Roll3Lds S = roll3_carve(smem_raw, E, N, L, D, S1);

    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, col = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long env0 = (long)blockIdx.x * E;
    const int envs_here = (int)((long)A.B - env0 < (long)E ? (long)A.B - env0 : (long)E);
    const int rows_here = envs_here * N;
    const int nblk = 2 * ((N + 1) >> 1);    // stage-1 blocks: (32-column tile = two timesteps of 16 sequences) x (hidden half)
    const long row_base = env0 * N;
    const size_t BN = (size_t)A.B * N;
    const uint64_t step0 = A.step_dev ? (uint64_t)*A.step_dev : A.step;
    constexpr int OUT = 5;  // one 5-logit head (checked on the host)

    // ---- constants -> LDS (once)
    {
        const float4 *src = reinterpret_cast<const float4 *>(A.frag + 8 * 2 * 4 * 64 * 4);
        for (int f = tid; f < (2 * S1 * 64) / 4; f += 512) reinterpret_cast<float4 *>(S.f_w1)[f] = src[f];
        if (tid < 64) S.s_b1[tid] = A.b1[tid];
        if (tid < OUT) S.s_b2[tid] = A.b2[tid];
    }

    // Gumbel noise of one head evaluation: value (row, logit o) = log(-log(u)), u = word (o & 3) of Philox block (o >> 2) keyed
    // (seed; step, global row) exactly as actor_forward_wg / pw_actor_head_kernel.  One thread per (row, block) -- the four
    // words of a block serve four logits -- called by `nthr` threads (t0 = their index)
    auto draw_noise = [&](const uint64_t step, const int t0, const int nthr) {
        constexpr int NB = (OUT + 3) / 4;
        for (int idx = t0; idx < rows_here * NB; idx += nthr) {
            const int rr = idx / NB;
            const uint32_t blk = (uint32_t)(idx - rr * NB), tag = ((blk & 1u) << 31) | ((blk >> 1) << 30);
            const long grow = row_base + rr;
            uint32_t u[4];
            pw_philox4x32_10((uint32_t)grow, (uint32_t)((uint64_t)grow >> 32) | tag, (uint32_t)step, (uint32_t)(step >> 32),
                             (uint32_t)A.seed, (uint32_t)(A.seed >> 32), u);
#pragma unroll
            for (int wq = 0; wq < 4; ++wq) {
                const int o = 4 * (int)blk + wq;
                if (o < OUT) {
                    const float uo = ((float)(u[wq] >> 8) + 0.5f) * 5.9604644775390625e-8f;  // (0, 1)
                    S.s_noise[rr * OUT + o] = __logf(-__logf(uo));
                }
            }
        }
    };

    // ---- environment lanes (as the other forms): wave w < n_env_waves owns local envs [w * epw, ...)
    const int epw_max = E < kWave / N ? E : kWave / N;
    const int waves_full = (E + epw_max - 1) / epw_max;
    const int epw = (E + waves_full - 1) / waves_full;
    const int n_env_waves = (envs_here + epw - 1) / epw;  // <= 8 (N <= 32)
    // the LAST waves: after a step they finish its rewards and stores while the first waves (dense1 blocks are dealt from
    // wave 0 up) already work on the next one
    const int ew = wave - (8 - n_env_waves);
    const bool env_wave = ew >= 0;
    int e_loc = lane / N, a = lane - e_loc * N;
    int el = ew * epw + e_loc;
    const bool live = env_wave && e_loc < epw && el < envs_here;
    if (!live) { e_loc = 0; a = 0; el = env_wave ? ew * epw : 0; }
    const int base = e_loc * N, r = el * N + a;
    const long env = env0 + el;
    const uint32_t g_lane = (uint32_t)env * (uint32_t)N + (uint32_t)a, g = g_lane;
    float2 *s_pos = S.s_posb + (env_wave ? ew : 0) * kWave;
    const float2 *pp = s_pos + base;
    float2 *lmv = S.s_lmb + el * L;
    const int la = a < L ? a : 0;

    float px = 0.f, py = 0.f, vx = 0.f, vy = 0.f, olx = 0.f, oly = 0.f, best = 0.f;
    int ep_step = 0;
    uint32_t ep_count = 0;
    uint64_t coll = 0, near = 0;
    float ep_ret = 0.f;
    double fin_sum = 0.0;
    int fin_cnt = 0;
    if (env_wave) {
        if (SINK && P.episode_return && live && a == 0) ep_ret = P.episode_return[env];
        px = V.pos_x[g]; py = V.pos_y[g]; vx = V.vel_x[g]; vy = V.vel_y[g];
        ep_step = V.ep_step[env];
        ep_count = V.ep_count[env];
        if (L > 0) {
            olx = V.lm_x[(size_t)env * L + la];
            oly = V.lm_y[(size_t)env * L + la];
            if (live) lmv[la] = make_float2(olx, oly);
        }
        if (live) s_pos[base + a] = make_float2(px, py);
        wave_lds_sync();
        stream_partner_pass<NT, uint64_t>(N, a, pp, px, py, olx, oly, V.coll_thr2, V.near_thr2, coll, near, best);
        if (live) lds_write_obs_row<LT>(S.s_obs + r * DS, L, lmv, px, py, vx, vy);
    }
    const float k = V.contact_margin, cf = V.contact_force, dt = V.dt, damp = V.damp, mass = V.mass;

    // ---- this wave's resident weights.  MFMA 16x16x4 lane roles: as A operand lane = (tile row ar = lane % 16, k quarter
    // kq = lane / 16); as B operand / accumulator lane = (column n = lane % 16, kq resp. row group rg = lane / 16).
    // Tile row ar = 4 * (unit within tile) + gate: W row gate * 32 + hq * 8 + 4 * tile + ar / 4 of direction dir.
    const int dir = wave >> 2, hq = wave & 3;
    const int n16 = lane & 15, kq = lane >> 4;
    float aih[2][16], ahh[2][8], bias[2][4];
    u32x2 ah[2][4], al[2][4];  // bf16x3 form of aih (opt-in, not exact: pw_kernels_actor16.hpp)
    actor16_load_ih<S1, BF3>(A.frag, wave, lane, aih, ah, al);
    {
        const float *whh = dir ? A.whh_r : A.whh_f;
#pragma unroll
        for (int T = 0; T < 2; ++T) {
            const int wrow = (n16 & 3) * 32 + hq * 8 + 4 * T + (n16 >> 2);  // within the direction: gate * 32 + unit
#pragma unroll
            for (int s = 0; s < 8; ++s) ahh[T][s] = whh[wrow * 32 + 4 * s + kq];
#pragma unroll
            for (int i = 0; i < 4; ++i) bias[T][i] = A.bih[dir * 128 + i * 32 + hq * 8 + 4 * T + kq];  // accumulator role: rg = kq
        }
    }
    float aw2[16];  // head weights as A fragments: tile row = logit (5 of 16 rows used)
#pragma unroll
    for (int sx = 0; sx < 16; ++sx) aw2[sx] = n16 < OUT ? A.w2[n16 * 64 + 4 * sx + kq] : 0.0f;
    const bool seq_ok = n16 < envs_here;
    // the Gumbel noise of the next step is drawn by the waves without environment duty while the environment waves advance
    // the agents (by everybody, first, when every wave has environment duty: N > 24)
    // With two environment waves (6 and 7: SIMDs 2 and 3) the noise goes to waves 0, 1, 4, 5 -- the other two SIMDs.
    const int noise_thr = n_env_waves == 2 ? 256 : n_env_waves < 8 ? (8 - n_env_waves) * 64 : 512;
    const int noise_t0 = n_env_waves == 2 ? ((wave & 2) ? 512 : (wave >> 2) * 128 + (wave & 1) * 64 + lane) : tid;
    wg_lds_barrier();  // constants in LDS
    if (noise_t0 < noise_thr) draw_noise(step0, noise_t0, noise_thr);
    wg_lds_barrier();  // first observation rows and first noise in LDS
    PW_R2_DECL;

    // the head on the matrix cores: logits^T [16 (5 used) x 16 rows] = W2 [16 x 64] * relu(h)^T per 16-row tile, C-in = b2, k
    // ascending -- the chain of actor_forward_wg's head (b2, then one fused multiply-add per hidden unit) -- then minus the
    // Gumbel noise and the arg-max (first maximum wins) in the lanes: row group 0 holds logits 0..3 of its row, row group 1
    // logit 4.  Wave w serves tiles w, w + 8, ..; one barrier afterwards.
    auto head = [&]() {
        const int ntile = (rows_here + 15) >> 4;
        for (int tile = wave; tile < ntile; tile += 8) {
            f32x4 lg;
#pragma unroll
            for (int i = 0; i < 4; ++i) lg[i] = 4 * kq + i < OUT ? S.s_b2[4 * kq + i] : 0.0f;
            const float4 *hf = S.s_hf + (tile * 4) * 64 + lane;
#pragma unroll
            for (int jx = 0; jx < 4; ++jx) {
                const float4 b = hf[jx * 64];
                lg = __builtin_amdgcn_mfma_f32_16x16x4f32(aw2[4 * jx + 0], b.x, lg, 0, 0, 0);
                lg = __builtin_amdgcn_mfma_f32_16x16x4f32(aw2[4 * jx + 1], b.y, lg, 0, 0, 0);
                lg = __builtin_amdgcn_mfma_f32_16x16x4f32(aw2[4 * jx + 2], b.z, lg, 0, 0, 0);
                lg = __builtin_amdgcn_mfma_f32_16x16x4f32(aw2[4 * jx + 3], b.w, lg, 0, 0, 0);
            }
            const int rr = tile * 16 + n16;
            const float *nz = S.s_noise + (rr < rows_here ? rr : 0) * OUT;
            const float p0 = lg[0] - nz[kq == 1 ? 4 : 0], p1 = lg[1] - nz[1], p2 = lg[2] - nz[2], p3 = lg[3] - nz[3];
            const float p4 = __shfl(p0, n16 + 16, kWave);  // logit 4 lives in register 0 of row group 1
            int best = 0;
            float bv = p0;
            if (p1 > bv) { bv = p1; best = 1; }
            if (p2 > bv) { bv = p2; best = 2; }
            if (p3 > bv) { bv = p3; best = 3; }
            if (p4 > bv) { bv = p4; best = 4; }
            if (kq == 0 && rr < rows_here) S.s_act[rr] = best;
        }
        wg_lds_barrier();
    };

    // input projection of timestep ts for this wave's two tiles (+ bias)
    auto inproj = [&](const int ts, f32x4 (&acc)[2]) { actor16_inproj<BF3>(S.s_xf, ts, lane, aih, ah, al, bias, acc); };

    // The rest of environment step t once positions and velocities are advanced, in two pieces so that each fits a window in
    // which the environment waves have nothing else to do:
    //   tail_compute  partner pass on the new positions (the next step's forces need its near mask), rewards, episode
    //                 bookkeeping -- beside the next step's dense1 blocks (the first waves)
    //   tail_stores   every global store of the step -- beside the next step's head tiles (the first waves again)
    // A wave with an episode ending in this step runs both, the reset and the post-reset partner pass before it publishes
    // the next observation rows (env_step_with_reset).
    int ai = 0;
    size_t slot = 0;
    float t_rw = 0.f, t_acc = 0.f;
    bool t_term = false;
    auto tail_compute = [&]() {
        wave_lds_sync();
        if (live) s_pos[base + a] = make_float2(px, py);
        wave_lds_sync();
        stream_partner_pass<NT, uint64_t>(N, a, pp, px, py, olx, oly, V.coll_thr2, V.near_thr2, coll, near, best);
        const float own = sqrtf(best);
        float rw = 0.0f, acc = 0.0f;
        if (NT > 0) {
#pragma unroll
            for (int l = 0; l < LT; ++l) rw -= __shfl(own, base + l, kWave);
#pragma unroll
            for (int j = 0; j < NT; ++j)
                if ((coll >> j) & 1) rw -= 1.0f;
#pragma unroll
            for (int i = 0; i < NT; ++i) acc += __shfl(rw, base + i, kWave);
        } else {  // runtime N: shuffles four at a time; "-1 per colliding agent" as many times as the mask has bits (equal subtrahends)
            rw = shfl_sub_ordered(rw, own, base, L);
            for (int c = __builtin_popcountll(coll); c > 0; --c) rw -= 1.0f;
            acc = shfl_add_ordered(acc, rw, base, N);
        }
        ep_step += 1;
        t_term = V.max_episode_len > 0 && ep_step >= V.max_episode_len;
        t_rw = rw;
        t_acc = acc;
        if (SINK && live && a == 0 && P.episode_return) {  // run.py:55-65, per env
            const float rsum = ep_ret + acc;
            if (t_term) { fin_sum += (double)rsum; fin_cnt += 1; ep_ret = 0.0f; }
            else ep_ret = rsum;
        }
    };
    // `opaque`: the lane's launch-invariant global index passes through an empty asm wherever it feeds an output address -- otherwise every
    // `pointer + g` of the step's stores is computed once per launch and kept in a register pair for the whole kernel (what pushed the widest
    // instantiations over the 256-register cap: pw_kernels_policy_tag.hpp has the same device).
    auto opaque = [](const uint32_t v) { uint32_t x = v; asm volatile("" : "+v"(x)); return x; };
    auto tail_stores = [&](const int t, const bool with_obs) {  // with_obs: V.obs too (no reset in between: the same row)
        const size_t tBN = (size_t)t * BN;
        const uint32_t g = opaque(g_lane);
        if (live) {
            if (P.act_out) P.act_out[tBN + g] = ai;
            if (V.rew) V.rew[tBN + g] = t_rw;
            if (V.done) V.done[tBN + g] = 0;
            if (a == 0) {
                if (V.rew_shared) V.rew_shared[(size_t)t * A.B + env] = t_acc;
                if (V.terminal) V.terminal[(size_t)t * A.B + env] = t_term ? 1 : 0;
            }
            if (SINK && P.has_ring) {  // next_obs is the PRE-reset observation (run.py:52 vs :60)
                if (P.ring.state_rows) sink_state_next(P.ring, slot, N, a, px, py, vx, vy);
                else stream_write_obs<LT>(P.ring.next_obs + (slot * N + a) * D, L, lmv, px, py, vx, vy);
                if (a == 0) { P.ring.rew[slot] = t_acc; P.ring.done[slot] = 0.0f; }
            }
            if (with_obs && V.obs) stream_write_obs<LT>(V.obs + (tBN + g) * D, L, lmv, px, py, vx, vy);
        }
    };
    auto env_step_with_reset = [&](const int t) {
        const size_t tBN = (size_t)t * BN;
        const uint32_t g = opaque(g_lane);
        tail_compute();
        tail_stores(t, false);
        const bool term = t_term;
        if (term && V.auto_reset) {  // same for every lane of an env
            if (live && V.final_obs) stream_write_obs<LT>(V.final_obs + (tBN + g) * D, L, lmv, px, py, vx, vy);
            wave_lds_sync();
            ep_count += 1;
            ep_step = 0;
            const uint64_t env_id = V.env_id_base + (uint64_t)env;
            pw_reset_xy(V.seed, env_id, ep_count, (uint32_t)a, -1.0f, 1.0f, &px, &py);
            vx = 0.f; vy = 0.f;
            if (L > 0) {
                pw_reset_xy(V.seed, env_id, ep_count, (uint32_t)(N + la), -1.0f, 1.0f, &olx, &oly);
                if (live) lmv[la] = make_float2(olx, oly);
            }
            if (live) s_pos[base + a] = make_float2(px, py);
        }
        wave_lds_sync();
        if (V.auto_reset && __any(term))
            stream_partner_pass<NT, uint64_t>(N, a, pp, px, py, olx, oly, V.coll_thr2, V.near_thr2, coll, near, best);
        if (live) {
            if (V.obs) stream_write_obs<LT>(V.obs + (tBN + g) * D, L, lmv, px, py, vx, vy);
            lds_write_obs_row<LT>(S.s_obs + r * DS, L, lmv, px, py, vx, vy);
        }
    };
    int tail_stage = 0;  // 0 nothing pending; 1: tail_compute of step t - 1 pending; 2: its tail_stores pending

    // dense1 blocks of this wave (bit = block), dealt once per launch: the environment waves run the previous step's tail_compute in the
    // same window -- ~2.4 blocks' worth of time, slowed by the matrix instructions their SIMD neighbours issue -- so a block goes to the
    // wave with the least load, an environment wave counting 2.4 ahead (stamps at N = 9 / 12, profiles/r4_policy_forms.txt: the
    // environment waves were the long pole of the window by 3-4 k cycles when every wave took block w, w + 8, ...)
    uint64_t my_blocks = 0;
    {
        int load10[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) load10[w] = w >= 8 - n_env_waves ? 24 : 0;   // tenths of a block
        for (int blk = 0; blk < nblk; ++blk) {
            int best_w = 0, best_v = load10[0];   // (the minimum's VALUE is carried along: load10[best_w] would index the array with a
#pragma unroll                            // run-time value and put it into scratch memory)
            for (int w = 1; w < 8; ++w)
                if (load10[w] < best_v) { best_v = load10[w]; best_w = w; }   // first minimum: waves from 0 up
#pragma unroll
            for (int w = 0; w < 8; ++w) load10[w] += w == best_w ? 10 : 0;
            if (best_w == wave) my_blocks |= 1ull << blk;
        }
    }

    for (int t = 0; t < P.T; ++t) {
        PW_R2_START;
        if (tail_stage == 1) { tail_compute(); tail_stage = 2; }  // step t - 1, beside the dense1 blocks of the other waves
        // ---- dense1 + ReLU: 32 x 32 blocks of relu(W1 X^T + b1), column rho = 16 * timestep + sequence
        for (uint64_t todo = PW_DBG(1) ? 0 : my_blocks; todo; todo &= todo - 1) {
            const int blk = __builtin_ctzll(todo);
            const int rt = blk >> 1, m = blk & 1;
            // the observation row behind column rt * 32 + col: agent ts of local env n (slots past N or past the envs of
            // this workgroup read a valid row; nobody uses their results)
            int ts = 2 * rt + (col >> 4), n = col & 15;
            if (ts >= N) ts = N - 1;
            if (n >= envs_here) n = 0;
            const float *xr = S.s_obs + (n * N + ts) * DS;
            float xb[S1];
#pragma unroll
            for (int sidx = 0; sidx < S1; ++sidx) {
                const int kk = 2 * sidx + half;
                xb[sidx] = kk < D ? xr[kk] : 0.0f;
            }
            f32x16 acc1;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc1[q] = 0.0f;
#pragma unroll
            for (int sidx = 0; sidx < S1; ++sidx)
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(S.f_w1[(m * S1 + sidx) * 64 + lane], xb[sidx], acc1, 0, 0, 0);
            float v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = fmaxf(acc1[q] + S.s_b1[m * 32 + mfma_row(q, half)], 0.0f);
            if (2 * rt + (col >> 4) < N) actor16_store_x1<BF3>(S.s_xf, 2 * rt + (col >> 4), m, half, col & 15, v);
        }
        PW_R2_STAMP(0);
        wg_lds_barrier();  // the x1 fragments are in LDS
        PW_R2_STAMP(1);
        // ---- the BiLSTM, one timestep per barrier
        if (!PW_DBG(2)) {
            f32x4 acc[2], accn[2];
            float c0 = 0.f, c1 = 0.f;
            inproj(dir ? N - 1 : 0, acc);
            for (int s2 = 0; s2 < N; ++s2) {
                const int ts = dir ? N - 1 - s2 : s2;
                if (s2 > 0) {
                    const float4 *hx = S.s_hx + ((((s2 - 1) & 1) * 2 + dir) * 2) * 64 + lane;
                    const float4 h0 = hx[0], h1 = hx[64];
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ahh[0][0], h0.x, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ahh[1][0], h0.x, acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ahh[0][1], h0.y, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ahh[1][1], h0.y, acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ahh[0][2], h0.z, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ahh[1][2], h0.z, acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ahh[0][3], h0.w, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ahh[1][3], h0.w, acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ahh[0][4], h1.x, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ahh[1][4], h1.x, acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ahh[0][5], h1.y, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ahh[1][5], h1.y, acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ahh[0][6], h1.z, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ahh[1][6], h1.z, acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ahh[0][7], h1.w, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ahh[1][7], h1.w, acc[1], 0, 0, 0);
                }
                // the two cells of this lane: accumulator registers = gates i, f, g, o
                float h0v, h1v;
                lstm_cell(acc[0][0], acc[0][1], acc[0][2], acc[0][3], c0, h0v);
                lstm_cell(acc[1][0], acc[1][1], acc[1][2], acc[1][3], c1, h1v);
                // h exchange: unit u = hq * 8 + 4 T + kq is k quarter kq of k step 2 hq + T: fragment j = hq / 2, elements
                // (2 hq) % 4 + T of this very lane slot
                reinterpret_cast<float2 *>(S.s_hx + (((s2 & 1) * 2 + dir) * 2 + (hq >> 1)) * 64 + lane)[hq & 1] = make_float2(h0v, h1v);
                if (seq_ok) {  // head input: row n16 * N + ts, k = dir * 32 + unit -> fragment 2 dir + hq / 2, same element pair
                    const int rr = n16 * N + ts;
                    reinterpret_cast<float2 *>(S.s_hf + ((rr >> 4) * 4 + 2 * dir + (hq >> 1)) * 64 + kq * 16 + (rr & 15))[hq & 1] =
                        make_float2(A.relu_out ? fmaxf(h0v, 0.0f) : h0v, A.relu_out ? fmaxf(h1v, 0.0f) : h1v);
                }
                if (s2 + 1 < N) inproj(dir ? N - 2 - s2 : s2 + 1, accn);  // before the barrier: work for the matrix pipe while the workgroup meets
                wg_lds_barrier();  // h(ts) of every unit is in LDS (the last one: Hs complete)
                if (s2 + 1 < N) { acc[0] = accn[0]; acc[1] = accn[1]; }
            }
        } else {
            wg_lds_barrier();
        }
        PW_R2_STAMP(2);
        if (tail_stage == 2) { tail_stores(t - 1, true); tail_stage = 0; }  // beside the head tiles of the first waves
        head();  // one barrier inside
        PW_R2_STAMP(3);

        // ---- environment step (pw_spread_stream_kernel's arithmetic), in two parts: advance the agents and publish the next
        // observation rows (everybody waits for those), then -- behind the barrier, while the other waves start the next
        // actor pass -- the partner pass, rewards, stores and bookkeeping (env_tail).  A wave with an episode ending this
        // step runs the tail first: the rows to publish are the post-reset ones.
        if (t + 1 < P.T && noise_t0 < noise_thr) draw_noise(step0 + (uint64_t)(t + 1), noise_t0, noise_thr);  // this step's is consumed
        if (env_wave) {
            ai = S.s_act[r];
            if (SINK && P.has_ring) {  // the observation the policy acted on (still in LDS) -> ring.obs
                slot = ring_slot(P.ring_start, t, A.B, (long)env, P.ring.capacity);
                if (live) {
                    if (P.ring.state_rows) {   // the state the policy acted on (still in the registers) + the episode's landmarks
                        sink_state_obs(P.ring, slot, N, a, L, lmv, px, py, vx, vy);
                    } else {
                        const float2 *src = reinterpret_cast<const float2 *>(S.s_obs + r * DS);
                        float2 *dst = reinterpret_cast<float2 *>(P.ring.obs + (slot * N + a) * D);
                        for (int c = 0; c < D / 2; ++c) dst[c] = src[c];
                    }
                    P.ring.act[slot * N + a] = (uint8_t)ai;
                }
            }
            float ux = 0.0f + ((ai == 1 ? 1.0f : 0.0f) - (ai == 2 ? 1.0f : 0.0f));
            float uy = 0.0f + ((ai == 3 ? 1.0f : 0.0f) - (ai == 4 ? 1.0f : 0.0f));
            ux *= V.sens; uy *= V.sens;
            if (V.fscale != 1.0f) { ux = V.fscale * ux; uy = V.fscale * uy; }
            float fx = ux + 0.0f, fy = uy + 0.0f;
            near_force_loop<uint64_t, float2>(live ? near : 0, pp, px, py, V.dist_min, k, cf, fx, fy);
            vx = vx * damp; vy = vy * damp;
            vx = vx + (fx / mass) * dt;
            vy = vy + (fy / mass) * dt;
            px = px + vx * dt;
            py = py + vy * dt;
            const bool ends = V.auto_reset && V.max_episode_len > 0 && ep_step + 1 >= V.max_episode_len;
            if (__any(ends)) env_step_with_reset(t);
            else {
                if (live) lds_write_obs_row<LT>(S.s_obs + r * DS, L, lmv, px, py, vx, vy);
                tail_stage = 1;
            }
        }
        PW_R2_STAMP(4);
        wg_lds_barrier();  // the next observation rows are in LDS
        PW_R2_STAMP(5);
        PW_R2_STAMP(6);
    }
    if (tail_stage == 1) tail_compute();
    if (tail_stage != 0) tail_stores(P.T - 1, true);
    if (wave == 0) PW_R2_FLUSH(0);
    if (wave == 7) PW_R2_FLUSH(8);

    if (live) {
        V.pos_x[g] = px; V.pos_y[g] = py;
        V.vel_x[g] = vx; V.vel_y[g] = vy;
        if (L > 0 && a < L) {
            V.lm_x[(size_t)env * L + la] = olx;
            V.lm_y[(size_t)env * L + la] = oly;
        }
        if (a == 0) {
            V.ep_step[env] = ep_step;
            V.ep_count[env] = ep_count;
            if (SINK && P.episode_return) P.episode_return[env] = ep_ret;
        }
    }
    if (SINK && P.episode_return) {
        wg_lds_barrier();
        if (live && a == 0) { S.s_fs[el] = fin_sum; S.s_fc[el] = fin_cnt; }
        wg_lds_barrier();
        rollout_finish_stats(envs_here, S.s_fs, S.s_fc, P.scratch, P.finished_sum, P.finished_count, smem_raw);
    }
}

}  // namespace
