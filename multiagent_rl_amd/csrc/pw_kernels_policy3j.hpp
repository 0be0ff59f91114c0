// pw_kernels_policy3j.hpp -- part of libpworld.so (translation unit csrc/pworld_policy.hip includes it).
// Policy-in-the-loop rollout, third form for LONG agent axes (N = 13 .. 50): the BiLSTM of pw_kernels_policy3.hpp with
// dense1 computed JUST IN TIME, one timestep per direction per loop iteration, and no observation rows in LDS.
#pragma once

namespace {

// ------------------------------------------------------------------------------------------
// Why.  pw_policy_rollout3_kernel keeps, per workgroup, the dense1 output of ALL N timesteps (4 KB each) and the observation
// rows of all E * N agents ((D + 2) * 4 bytes each) in LDS: at N = 24 (D = 52) that is 96 + 83 KB beside the 96 KB head input --
// it no longer fits 160 KB even with 8 environments per workgroup, the rollout fell back to the second form and cost 404 us per
// batched step at B = 4096 (N = 12: 32.5).  Two observations remove both blocks:
//   * a local observation row is {vx, vy, px, py, lm_0 - p, lm_1 - p, ...} (experiments/scenarios.py:6-20): the B operand of
//     dense1 can be COMPOSED while it is fed -- k < 4: the state itself; k >= 4: one float32 subtraction of a landmark
//     coordinate (constant over the agent axis: held in registers for the whole pass) and the agent's position.  LDS holds 16
//     bytes per agent instead of a row.
//   * dense1 of timestep ts is only needed by the input projection of timestep ts: wave (direction, hidden quarter hq)
//     computes rows 16 hq .. 16 hq + 15 of relu(W1 x(ts) + b1) for the timestep its direction reaches TWO iterations later,
//     as ONE 16x16x4 accumulator (K = D: up to 16 v_mfma_f32_16x16x4_f32), and scatters it into a two-deep ring of x1 fragments
//     per direction (16 KB in all instead of N * 4 KB).  W1's A fragments live in registers (16 VGPRs).
// Bits: the 32x32x2 dense1 chain of the other forms sums k ascending (instruction sidx holds k = 2 sidx, 2 sidx + 1); a
// 16x16x4 chain fed k = 4 s + kq sums the same k in the same order (tools/mfma16_probe.hip: a chain of fused multiply-adds over
// its four k, in order), zero padding leaves the accumulator unchanged, bias and ReLU follow.  The accumulator's rows land in
// the x1 fragment exactly where actor16_store_x1 puts them: hidden unit h = 16 hq + 4 rg + i is position
// p = 32 (h / 32) + 16 (hq & 1) + 8 (rg / 2) + 2 i + (rg & 1) of the projection's summation order (x1_kpos(p) = h): fragment
// j = hq, lane slot (2 (i & 1) + (rg & 1)) * 16 + n, element 2 (rg / 2) + i / 2 -- registers (0, 2) and (1, 3) are two 8-byte
// stores.  Everything downstream (input projection, recurrence, head, sampling, environment step) is pw_policy_rollout3_kernel's.
// The environment step runs undeferred here (every wave has environment duty at these N, and its work is a few per cent of a
// pass of 2 x N x 62 matrix instructions per wave).
// LDS (N = 24, L = 24, E = 16): 16 KB x1 ring (the sampled actions alias it) + 8 KB h exchange + 96 KB head input + 6 KB states + 3 KB
// landmarks + small = 134 KB; N = 30: 159.6 KB -- 16 environments per workgroup still fit (the Gumbel noise is drawn in the head's lanes).  Serves simple_spread with the local observation, D = 4 + 2 L <= 104 (W1's fragments: up to 26
// VGPRs), at most 8 environment waves of whole environments: 16 environments per workgroup up to N = 30, 8 at
// N = 33 .. 50 (BASELINE's C5 point N = L = 48: D = 100, the only form that holds such rows).
// Row stride NP (round 5).  A workgroup's rows sit in LDS env-major: state row / head-input row rr = n * NP + ts for sequence (environment)
// n and timestep (agent) ts.  With NP = N even, the sixteen sequences of an MFMA column set are a multiple of 8 rows apart: at N = 24 every
// lane of the head-input write of a timestep (one 8-byte store per lane into slot (rr & 15) of its tile: 16 x rr bytes modulo the 128-byte
// bank period) hit the SAME bank pair -- a 64-way conflict per wave and timestep -- and the sixteen state reads of dense1 (16 bytes x N apart)
// the same four banks: profiles/r4_policy_n24_summary.json counted 0.67 of all LDS cycles as conflict cycles.  An ODD stride makes
// n * NP + ts run through all residues modulo 16: the host passes NP = N | 1 wherever the 16 environments still fit (the padding
// rows are never read; the Gumbel noise stays keyed by the TRUE global row env * N + agent).
// ------------------------------------------------------------------------------------------
struct Roll3jLds {
    float4 *s_xf;    // [2 buffers][2 dir][4 j][64 lane]: x1 fragments of one timestep per (buffer, direction)
    float4 *s_hx;    // [2 buffers][2 dir][2 j][64 lane]
    float4 *s_hf;    // [rows / 16 tiles][4 j][64 lane]
    float *s_b2;     // [16]
    float4 *s_st;    // [rows] {vx, vy, px, py} of the agents as the policy sees them (row = env * N + agent)
    int32_t *s_act;  // [rows]: aliases the x1 ring (written by the head, read by the environment lanes: the ring is idle in between)
    float2 *s_posb;  // [8 env waves][64]
    float2 *s_lmb;   // [E * L]
    double *s_fs;    // [16]
    int *s_fc;       // [16]
};
__host__ __device__ inline size_t roll3j_lds_bytes(int E, int NP, int L)   // NP: the row stride (>= N)
{
    const size_t rows = (size_t)E * NP;
    const size_t fl = 2 * 2 * 4 * 64 * 4 + 2 * 2 * 2 * 64 * 4 + ((rows + 15) / 16) * 1024 + 16 + 3 + rows * 4 + 1;
    return fl * 4 + 8 * kWave * sizeof(float2) + (size_t)E * L * sizeof(float2) + 16 * (sizeof(double) + sizeof(int)) + 64;
}
__device__ __forceinline__ Roll3jLds roll3j_carve(unsigned char *raw, int E, int NP, int L)
{
    const int rows = E * NP;
    float *base = reinterpret_cast<float *>(raw);
    Roll3jLds S;
    int o = 0;
    S.s_xf = reinterpret_cast<float4 *>(base + o); o += 2 * 2 * 4 * 64 * 4;
    S.s_hx = reinterpret_cast<float4 *>(base + o); o += 2 * 2 * 2 * 64 * 4;
    S.s_hf = reinterpret_cast<float4 *>(base + o); o += ((rows + 15) / 16) * 1024;
    S.s_b2 = base + o; o += 16;
    o = (o + 3) & ~3;
    S.s_st = reinterpret_cast<float4 *>(base + o); o += rows * 4;
    S.s_act = reinterpret_cast<int32_t *>(S.s_xf);   // rows <= 512 ints of the ring's 4096 floats
    o = (o + 1) & ~1;
    S.s_posb = reinterpret_cast<float2 *>(base + o); o += 8 * kWave * 2;
    S.s_lmb = reinterpret_cast<float2 *>(base + o); o += E * L * 2;
    S.s_fs = reinterpret_cast<double *>(base + o); o += 32;
    S.s_fc = reinterpret_cast<int *>(base + o);
    return S;
}

template <int S1C, bool SINK>
__global__ void __launch_bounds__(512) pw_policy_rollout3j_kernel(const PolicyRolloutArgs P)
{
    constexpr int S1 = 4 * S1C;     // 32x32x2 k steps of the packed W1 image (2 k each)
    constexpr int KS = 2 * S1C;     // 16x16x4 k steps of dense1 (4 k each): K = 8 S1C >= D
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const ActorFusedArgs &A = P.A;
    const StreamParams &V = P.V;
    const int N = A.N, L = V.L, D = A.D, E = A.E, NP = P.NP;
    const Roll3jLds S = roll3j_carve(smem_raw, E, NP, L);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long env0 = (long)blockIdx.x * E;
    const int envs_here = (int)((long)A.B - env0 < (long)E ? (long)A.B - env0 : (long)E);
    const int rows_here = envs_here * NP;   // LDS rows incl. the padding rows (agent index >= N) of every environment
    const long row_base = env0 * N;
    const size_t BN = (size_t)A.B * N;
    const uint64_t step0 = A.step_dev ? (uint64_t)*A.step_dev : A.step;
    constexpr int OUT = 5;  // one 5-logit head (checked on the host)

    if (tid < OUT) S.s_b2[tid] = A.b2[tid];

    // ---- environment lanes: wave w >= 8 - n_env_waves owns local envs [ew * epw, ...), lane = (env, agent)
    const int epw_max = E < kWave / N ? E : kWave / N;
    const int waves_full = (E + epw_max - 1) / epw_max;
    const int epw = (E + waves_full - 1) / waves_full;
    const int n_env_waves = (envs_here + epw - 1) / epw;  // <= 8 (the host caps E at 8 * (64 / N))
    const int ew = wave - (8 - n_env_waves);
    const bool env_wave = ew >= 0;
    int e_loc = lane / N, a = lane - e_loc * N;
    int el = ew * epw + e_loc;
    const bool live = env_wave && e_loc < epw && el < envs_here;
    if (!live) { e_loc = 0; a = 0; el = env_wave ? ew * epw : 0; }
    const int base = e_loc * N, r = el * NP + a;   // r: this lane's row in LDS (stride NP)
    const long env = env0 + el;
    const uint32_t g = (uint32_t)env * (uint32_t)N + (uint32_t)a;
    float2 *s_pos = S.s_posb + (env_wave ? ew : 0) * kWave;
    const float2 *pp = s_pos + base;
    float2 *lmv = S.s_lmb + el * L;

    float px = 0.f, py = 0.f, vx = 0.f, vy = 0.f, olx = 0.f, oly = 0.f, best = 0.f;
    int ep_step = 0;
    uint32_t ep_count = 0;
    uint64_t coll = 0, near = 0;
    float ep_ret = 0.f;
    double fin_sum = 0.0;
    int fin_cnt = 0;
    const int la = a < L ? a : 0;
    // landmarks: lane a owns landmarks a, a + N, ... (L <= N on this path: one each; checked on the host)
    if (env_wave) {
        if (SINK && P.episode_return && live && a == 0) ep_ret = P.episode_return[env];
        px = V.pos_x[g]; py = V.pos_y[g]; vx = V.vel_x[g]; vy = V.vel_y[g];
        ep_step = V.ep_step[env];
        ep_count = V.ep_count[env];
        if (L > 0) {
            olx = V.lm_x[(size_t)env * L + la];
            oly = V.lm_y[(size_t)env * L + la];
            if (live && a < L) lmv[la] = make_float2(olx, oly);
        }
        if (live) s_pos[base + a] = make_float2(px, py);
        wave_lds_sync();
        stream_partner_pass<0, uint64_t>(N, a, pp, px, py, olx, oly, V.coll_thr2, V.near_thr2, coll, near, best);
        if (live) S.s_st[r] = make_float4(vx, vy, px, py);
    }
    const float k = V.contact_margin, cf = V.contact_force, dt = V.dt, damp = V.damp, mass = V.mass;

    // ---- this wave's resident weights (lane roles: pw_kernels_actor16.hpp)
    const int dir = wave >> 2, hq = wave & 3;
    const int n16 = lane & 15, kq = lane >> 4;
    float aih[2][16], ahh[2][8], bias[2][4];
    u32x2 ah[2][4], al[2][4];  // unused (exact form only)
    actor16_load_ih<S1, false>(A.frag, wave, lane, aih, ah, al);
    {
        const float *whh = dir ? A.whh_r : A.whh_f;
#pragma unroll
        for (int T = 0; T < 2; ++T) {
            const int wrow = (n16 & 3) * 32 + hq * 8 + 4 * T + (n16 >> 2);
#pragma unroll
            for (int s = 0; s < 8; ++s) ahh[T][s] = whh[wrow * 32 + 4 * s + kq];
#pragma unroll
            for (int i = 0; i < 4; ++i) bias[T][i] = A.bih[dir * 128 + i * 32 + hq * 8 + 4 * T + kq];
        }
    }
    float aw2[16];
#pragma unroll
    for (int sx = 0; sx < 16; ++sx) aw2[sx] = n16 < OUT ? A.w2[n16 * 64 + 4 * sx + kq] : 0.0f;
    // dense1: A fragments of rows 16 hq .. 16 hq + 15 of W1 (lane = (row n16, k quarter kq)) out of the packed 32x32x2 image
    // [2 m][S1][64 lane] (lane = row % 32 + 32 (k & 1), k step k / 2; zero beyond D), and the bias of this lane's accumulator rows
    float a1[KS], b1v[4];
    {
        const float *w1p = A.frag + 8 * 2 * 4 * 64 * 4;
        const int h = 16 * hq + n16;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int kk = 4 * s + kq;
            a1[s] = w1p[((h >> 5) * S1 + (kk >> 1)) * 64 + (h & 31) + 32 * (kk & 1)];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) b1v[i] = A.b1[16 * hq + 4 * kq + i];
    }
    const bool seq_ok = n16 < envs_here;
    const int nseq = seq_ok ? n16 : 0;   // columns past the environments of this workgroup read env 0; nobody uses their results
    wg_lds_barrier();  // constants, first states and landmarks in LDS

    // The head on the matrix cores (pw_policy_rollout3_kernel's), with the Gumbel noise drawn IN the lanes that subtract it: value
    // (row, logit o) = log(-log(u)), u = word (o & 3) of Philox block (o >> 2) keyed (seed; step, global row) -- the keying of every
    // other form -- so row group 0 (logits 0..3) needs block 0 and row group 1 (logit 4) block 1 of its row: no noise plane in LDS
    // (9.6 KB at N = 30: with it, 16 environments per workgroup would not fit).
    auto head = [&](const uint64_t step) {
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
            float nz[4] = {0.f, 0.f, 0.f, 0.f};
            if (kq < 2) {  // wave-divergent only by row group
                const int rq = rr < rows_here ? rr : 0, re = rq / NP, ra = rq - re * NP;   // LDS row -> (environment, agent)
                const long grow = row_base + (long)re * N + (ra < N ? ra : 0);            // the TRUE global row keys the noise
                const uint32_t blk = (uint32_t)kq, tag = ((blk & 1u) << 31) | ((blk >> 1) << 30);
                uint32_t u[4];
                pw_philox4x32_10((uint32_t)grow, (uint32_t)((uint64_t)grow >> 32) | tag, (uint32_t)step, (uint32_t)(step >> 32),
                                 (uint32_t)A.seed, (uint32_t)(A.seed >> 32), u);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float uo = ((float)(u[i] >> 8) + 0.5f) * 5.9604644775390625e-8f;  // (0, 1)
                    nz[i] = __logf(-__logf(uo));
                }
            }
            const float p0 = lg[0] - nz[0], p1 = lg[1] - nz[1], p2 = lg[2] - nz[2], p3 = lg[3] - nz[3];
            const float p4 = __shfl(p0, n16 + 16, kWave);  // logit 4 lives in register 0 of row group 1
            int bi = 0;
            float bv = p0;
            if (p1 > bv) { bv = p1; bi = 1; }
            if (p2 > bv) { bv = p2; bi = 2; }
            if (p3 > bv) { bv = p3; bi = 3; }
            if (p4 > bv) { bv = p4; bi = 4; }
            if (kq == 0 && rr < rows_here) S.s_act[rr] = bi;
        }
        wg_lds_barrier();
    };

    // dense1 of timestep ts (this wave's direction) into x1 ring buffer `buf`: rows 16 hq .. + 15, 16 sequences
    float lmk[KS > 1 ? KS - 1 : 1];  // landmark coordinate (kq & 1) of landmark 2 (s - 1) + kq / 2 of env n16: constant over the pass
    auto dense1 = [&](const int ts, const int buf) {
        const float *st = reinterpret_cast<const float *>(S.s_st + nseq * NP + ts);
        const float x0 = st[kq], pc = st[2 + (kq & 1)];
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[0], x0, acc, 0, 0, 0);
#pragma unroll
        for (int s = 1; s < KS; ++s) {
            const float x = 2 * (s - 1) + (kq >> 1) < L ? lmk[s - 1] - pc : 0.0f;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], x, acc, 0, 0, 0);
        }
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = fmaxf(acc[i] + b1v[i], 0.0f);
        float4 *dst = S.s_xf + ((buf * 2 + dir) * 4 + hq) * 64 + n16;
        reinterpret_cast<float2 *>(dst + (kq & 1) * 16)[kq >> 1] = make_float2(v[0], v[2]);
        reinterpret_cast<float2 *>(dst + (2 + (kq & 1)) * 16)[kq >> 1] = make_float2(v[1], v[3]);
    };
    auto inproj = [&](const int buf, f32x4 (&acc)[2]) { actor16_inproj<false>(S.s_xf, buf * 2 + dir, lane, aih, ah, al, bias, acc); };

    // environment step t after positions and velocities are advanced (pw_policy_rollout3_kernel's env_step_with_reset, run for
    // every step: partner pass, rewards, stores, bookkeeping, the reset where an episode ends, then the next states published)
    int ai = 0;
    size_t slot = 0;
    auto env_finish = [&](const int t) {
        const size_t tBN = (size_t)t * BN;
        wave_lds_sync();
        if (live) s_pos[base + a] = make_float2(px, py);
        wave_lds_sync();
        stream_partner_pass<0, uint64_t>(N, a, pp, px, py, olx, oly, V.coll_thr2, V.near_thr2, coll, near, best);
        const float own = sqrtf(best);
        float rw = shfl_sub_ordered(0.0f, own, base, L);   // shuffles four at a time, upstream's order
        for (int c = __builtin_popcountll(coll); c > 0; --c) rw -= 1.0f;   // "-1 per colliding agent": equal subtrahends, only their number matters
        const float acc = shfl_add_ordered(0.0f, rw, base, N);
        ep_step += 1;
        const bool term = V.max_episode_len > 0 && ep_step >= V.max_episode_len;
        if (SINK && live && a == 0 && P.episode_return) {  // run.py:55-65, per env
            const float rsum = ep_ret + acc;
            if (term) { fin_sum += (double)rsum; fin_cnt += 1; ep_ret = 0.0f; }
            else ep_ret = rsum;
        }
        if (live) {
            if (P.act_out) P.act_out[tBN + g] = ai;
            if (V.rew) V.rew[tBN + g] = rw;
            if (V.done) V.done[tBN + g] = 0;
            if (a == 0) {
                if (V.rew_shared) V.rew_shared[(size_t)t * A.B + env] = acc;
                if (V.terminal) V.terminal[(size_t)t * A.B + env] = term ? 1 : 0;
            }
            if (SINK && P.has_ring) {  // next_obs is the PRE-reset observation (run.py:52 vs :60)
                stream_write_obs<0>(P.ring.next_obs + (slot * N + a) * D, L, lmv, px, py, vx, vy);
                if (a == 0) { P.ring.rew[slot] = acc; P.ring.done[slot] = 0.0f; }
            }
        }
        if (term && V.auto_reset) {  // same for every lane of an env
            if (live && V.final_obs) stream_write_obs<0>(V.final_obs + (tBN + g) * D, L, lmv, px, py, vx, vy);
            wave_lds_sync();
            ep_count += 1;
            ep_step = 0;
            const uint64_t env_id = V.env_id_base + (uint64_t)env;
            pw_reset_xy(V.seed, env_id, ep_count, (uint32_t)a, -1.0f, 1.0f, &px, &py);
            vx = 0.f; vy = 0.f;
            if (L > 0) {
                pw_reset_xy(V.seed, env_id, ep_count, (uint32_t)(N + la), -1.0f, 1.0f, &olx, &oly);
                if (live && a < L) lmv[la] = make_float2(olx, oly);
            }
            if (live) s_pos[base + a] = make_float2(px, py);
        }
        wave_lds_sync();
        if (V.auto_reset && __any(term))
            stream_partner_pass<0, uint64_t>(N, a, pp, px, py, olx, oly, V.coll_thr2, V.near_thr2, coll, near, best);
        if (live) {
            if (V.obs) stream_write_obs<0>(V.obs + (tBN + g) * D, L, lmv, px, py, vx, vy);
            S.s_st[r] = make_float4(vx, vy, px, py);
        }
    };

    for (int t = 0; t < P.T; ++t) {
        // ---- the pass's landmark registers, the first two timesteps' dense1, the first input projection
#pragma unroll
        for (int s = 1; s < KS; ++s) {
            const int l = 2 * (s - 1) + (kq >> 1);
            lmk[s - 1] = l < L ? reinterpret_cast<const float *>(S.s_lmb + nseq * L + l)[kq & 1] : 0.0f;
        }
        dense1(dir ? N - 1 : 0, 0);
        wg_lds_barrier();
        // ---- the BiLSTM, one timestep per barrier; dense1 runs two timesteps ahead, the input projection one
        {
            f32x4 acc[2], accn[2];
            float c0 = 0.f, c1 = 0.f;
            inproj(0, acc);                          // reads ring buffer 0 ...
            if (N > 1) dense1(dir ? N - 2 : 1, 1);
            wg_lds_barrier();                        // ... before anybody's iteration 0 overwrites it
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
                float h0v, h1v;
                lstm_cell(acc[0][0], acc[0][1], acc[0][2], acc[0][3], c0, h0v);
                lstm_cell(acc[1][0], acc[1][1], acc[1][2], acc[1][3], c1, h1v);
                reinterpret_cast<float2 *>(S.s_hx + (((s2 & 1) * 2 + dir) * 2 + (hq >> 1)) * 64 + lane)[hq & 1] = make_float2(h0v, h1v);
                if (seq_ok) {
                    const int rr = n16 * NP + ts;
                    reinterpret_cast<float2 *>(S.s_hf + ((rr >> 4) * 4 + 2 * dir + (hq >> 1)) * 64 + kq * 16 + (rr & 15))[hq & 1] =
                        make_float2(A.relu_out ? fmaxf(h0v, 0.0f) : h0v, A.relu_out ? fmaxf(h1v, 0.0f) : h1v);
                }
                // x1 ring: buffer s2 & 1 held timestep s2's fragments, consumed by the input projection issued in iteration
                // s2 - 1 (or above), before that iteration's barrier: free for timestep s2 + 2
                if (s2 + 2 < N) dense1(dir ? N - 3 - s2 : s2 + 2, s2 & 1);
                if (s2 + 1 < N) inproj((s2 + 1) & 1, accn);  // its fragments were written one iteration ago, a barrier has passed
                wg_lds_barrier();
                if (s2 + 1 < N) { acc[0] = accn[0]; acc[1] = accn[1]; }
            }
        }
        head(step0 + (uint64_t)t);  // one barrier inside

        // ---- environment step (pw_spread_stream_kernel's arithmetic)
        if (env_wave) {
            ai = S.s_act[r];
            if (SINK && P.has_ring) {  // the observation the policy acted on: rebuilt from the (still pre-step) registers
                slot = ring_slot(P.ring_start, t, A.B, (long)env, P.ring.capacity);
                if (live) {
                    stream_write_obs<0>(P.ring.obs + (slot * N + a) * D, L, lmv, px, py, vx, vy);
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
            env_finish(t);
        }
        wg_lds_barrier();  // the next states (and, after a reset, landmarks) are in LDS
    }

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
