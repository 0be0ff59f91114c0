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
// rows are never read; the Gumbel noise stays keyed by the TRUE global row env * N + agent).  (Measured: the step time did not move --
// N = 24 74.3 us before and after -- the kernel does not wait for LDS; kept because it is free and removes 0.67 -> of the conflict cycles.)
// Sixteen environments per workgroup beyond N = 30 (round 5; template parameter HALF = rows longer than 64 numbers, N = L = 31 .. 50).  Two things
// capped the workgroup at 8 environments there -- half of the 16 MFMA columns: the head input (256 B per agent: 196 KB at N = 48) and the
// environment lanes (lane = (environment, agent): one environment per wave).
//   * HALF-STORAGE HEAD.  The head of timestep ts needs relu(h_fwd(ts)) and relu(h_bwd(ts)), i.e. it can run as soon as BOTH directions have
//     visited ts -- at iteration max(ts, N - 1 - ts) -- instead of after the whole pass.  Only the FIRST visitor's half has to be kept (the
//     second one's is still in the h exchange buffer when the head runs): s_hh[ts] = that half in the exchange buffer's own B-fragment layout
//     (2 KB per timestep), written beside the exchange write.  After the barrier of iteration s2 up to two timesteps are complete (ts = s2 if
//     the backward direction was there first, ts = N - 1 - s2 if the forward one was); two waves (rotating) run their heads -- logits chain
//     b2, k = 0 .. 31 (forward), k = 32 .. 63 (backward): the order of every other form -- draw the noise, take the arg-max and write the byte
//     action, while the others go on with iteration s2 + 1.  128 B per agent instead of 256, and no head phase after the loop.
//   * ENVIRONMENT SLOTS.  No environment state lives in registers across steps any more: {vel, pos} is the published s_st row, landmarks s_lmb,
//     the episode clock / counters small per-environment arrays in LDS.  An environment wave walks `slots` groups of environments one after the
//     other (N = 48, E = 16: two); the near mask the step's forces need is kept per row in LDS too (re-deriving it by a second partner
//     pass per step cost 4 us per step at N = 24: measured).  This also took ~20 registers off every instantiation.
//   HALF is chosen by the host wherever the full head input does not leave room for 16 environments (N >= 30) and compiled for rows of more
//   than 48 numbers (S1C >= 7); below that the full head (no head work inside the timestep loop) is ~5 % faster.
// ------------------------------------------------------------------------------------------
struct Roll3jLds {
    float4 *s_xf;    // [2 buffers][2 dir][4 j][64 lane]: x1 fragments of one timestep per (buffer, direction)
    float4 *s_hx;    // [2 buffers][2 dir][2 j][64 lane]: h exchange, element e of (j, lane (n, kq)) = h[seq n][16 j + 4 e + kq]
    float4 *s_hf;    // !HALF: head input [rows / 16 tiles][4 j][64 lane];  HALF: s_hh [NP timesteps][2 j][64 lane] = the first visitor's h(ts)
    float *s_b2;     // [16]
    float4 *s_st;    // [rows] {vx, vy, px, py} of the agents as the policy sees them (row = env * NP + agent)
    uint8_t *s_act;  // [rows] sampled actions (written by the head, read by the environment lanes)
    uint64_t *s_near; // [rows] near mask of every agent on the current positions (what the next step's contact forces visit)
    float2 *s_posb;  // [8 env waves][64]
    float2 *s_lmb;   // [E * L]
    int *s_eps;      // [16] episode step of every environment of the workgroup
    uint32_t *s_epc; // [16] episode number
    float *s_ret;    // [16] running episode return (SINK)
    double *s_fs;    // [16]
    int *s_fc;       // [16]
};
__host__ __device__ inline size_t roll3j_lds_bytes(int E, int NP, int L, bool half)   // NP: the row stride (>= N)
{
    const size_t rows = (size_t)E * NP;
    const size_t head = half ? (size_t)NP * 2 * 64 * 4 : ((rows + 15) / 16) * 1024;
    const size_t fl = 2 * 2 * 4 * 64 * 4 + 2 * 2 * 2 * 64 * 4 + head + 16 + 3 + rows * 4 + 1;
    // (!HALF: the byte actions alias the x1 ring -- idle between the head and the environment step -- and no near masks are kept: at
    // N = 30 the full-head form fits its 16 environments with 240 bytes to spare)
    return fl * 4 + (half ? ((rows + 15) & ~(size_t)15) + rows * sizeof(uint64_t) : 0) + 8 * kWave * sizeof(float2) + (size_t)E * L * sizeof(float2) + 16 * (3 * 4 + sizeof(double) + sizeof(int)) + 64;
}
__device__ __forceinline__ Roll3jLds roll3j_carve(unsigned char *raw, int E, int NP, int L, bool half)
{
    const int rows = E * NP;
    float *base = reinterpret_cast<float *>(raw);
    Roll3jLds S;
    int o = 0;
    S.s_xf = reinterpret_cast<float4 *>(base + o); o += 2 * 2 * 4 * 64 * 4;
    S.s_hx = reinterpret_cast<float4 *>(base + o); o += 2 * 2 * 2 * 64 * 4;
    S.s_hf = reinterpret_cast<float4 *>(base + o); o += half ? NP * 2 * 64 * 4 : ((rows + 15) / 16) * 1024;
    S.s_b2 = base + o; o += 16;
    o = (o + 3) & ~3;
    S.s_st = reinterpret_cast<float4 *>(base + o); o += rows * 4;
    if (half) {
        S.s_act = reinterpret_cast<uint8_t *>(base + o); o += ((rows + 15) & ~15) / 4;
        o = (o + 1) & ~1;
        S.s_near = reinterpret_cast<uint64_t *>(base + o); o += rows * 2;
    } else {
        S.s_act = reinterpret_cast<uint8_t *>(S.s_xf);   // rows <= 1024 bytes of the ring's 16 KB: written by the head, read by the environment lanes
        S.s_near = nullptr;
        o = (o + 1) & ~1;
    }
    S.s_posb = reinterpret_cast<float2 *>(base + o); o += 8 * kWave * 2;
    S.s_lmb = reinterpret_cast<float2 *>(base + o); o += E * L * 2;
    S.s_fs = reinterpret_cast<double *>(base + o); o += 32;
    S.s_fc = reinterpret_cast<int *>(base + o); o += 16;
    S.s_eps = reinterpret_cast<int *>(base + o); o += 16;
    S.s_epc = reinterpret_cast<uint32_t *>(base + o); o += 16;
    S.s_ret = base + o;
    return S;
}

template <int S1C, bool SINK, bool HALF>
__global__ void __launch_bounds__(512) pw_policy_rollout3j_kernel(const PolicyRolloutArgs P)
{
    constexpr int S1 = 4 * S1C;     // 32x32x2 k steps of the packed W1 image (2 k each)
    constexpr int KS = 2 * S1C;     // 16x16x4 k steps of dense1 (4 k each): K = 8 S1C >= D
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const ActorFusedArgs &A = P.A;
    const StreamParams &V = P.V;
    const int N = A.N, L = V.L, D = A.D, E = A.E, NP = P.NP;
    const Roll3jLds S = roll3j_carve(smem_raw, E, NP, L, HALF);

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

    // ---- environment lanes.  The environments of the workgroup are dealt to "virtual waves" of epw environments each (lane = (env, agent));
    // environment wave ew (the LAST n_env_waves waves) serves virtual waves ew, ew + 8, ... one after the other (slots).
    const int epw_max = E < kWave / N ? E : kWave / N;
    const int vwaves_full = (E + epw_max - 1) / epw_max;
    const int epw = (E + vwaves_full - 1) / vwaves_full;
    const int n_vwaves = (envs_here + epw - 1) / epw;
    const int n_env_waves = n_vwaves < 8 ? n_vwaves : 8;
    const int ew = wave - (8 - n_env_waves);
    const bool env_wave = ew >= 0;
    float2 *s_pos = S.s_posb + (env_wave ? ew : 0) * kWave;
    const float k = V.contact_margin, cf = V.contact_force, dt = V.dt, damp = V.damp, mass = V.mass;

    // one slot's lane coordinates
    struct EnvLane {
        int e_loc, a, el, base, r, la;
        bool live;
        long env;
        uint32_t g;
    };
    auto env_lane = [&](const int vw) {
        EnvLane q;
        q.e_loc = lane / N; q.a = lane - q.e_loc * N;
        q.el = vw * epw + q.e_loc;
        q.live = q.e_loc < epw && q.el < envs_here;
        if (!q.live) { q.e_loc = 0; q.a = 0; q.el = vw * epw; }   // idle lanes shadow lane 0 (vw * epw < envs_here: the caller checks)
        q.base = q.e_loc * N; q.r = q.el * NP + q.a; q.la = q.a < L ? q.a : 0;
        q.env = env0 + q.el;
        q.g = (uint32_t)q.env * (uint32_t)N + (uint32_t)q.a;
        return q;
    };
    // !HALF (one slot per environment wave: the host caps E at 8 x (64 / N) there): the lane's state stays in these registers for the whole
    // launch, as in every other rollout kernel (through LDS each step it cost 2.5 % at N = 16 .. 24: measured).  HALF: reloaded per slot.
    float r_px = 0.f, r_py = 0.f, r_vx = 0.f, r_vy = 0.f, r_olx = 0.f, r_oly = 0.f;
    int r_ep_step = 0;
    uint32_t r_ep_count = 0;
    uint64_t r_near = 0;
    if (env_wave) {   // the workgroup's state: global planes -> LDS
        for (int vw = ew; vw < n_vwaves; vw += 8) {
            const EnvLane q = env_lane(vw);
            if (q.live) {
                S.s_st[q.r] = make_float4(V.vel_x[q.g], V.vel_y[q.g], V.pos_x[q.g], V.pos_y[q.g]);
                if (L > 0 && q.a < L) S.s_lmb[q.el * L + q.la] = make_float2(V.lm_x[(size_t)q.env * L + q.la], V.lm_y[(size_t)q.env * L + q.la]);
                if (q.a == 0) {
                    S.s_eps[q.el] = V.ep_step[q.env];
                    S.s_epc[q.el] = V.ep_count[q.env];
                    S.s_ret[q.el] = (SINK && P.episode_return) ? P.episode_return[q.env] : 0.0f;
                    S.s_fs[q.el] = 0.0;
                    S.s_fc[q.el] = 0;
                }
            }
            // the near masks of the first step: a partner pass on the initial positions
            const float px = V.pos_x[q.g], py = V.pos_y[q.g];
            wave_lds_sync();
            if (q.live) s_pos[q.base + q.a] = make_float2(px, py);
            wave_lds_sync();
            uint64_t coll = 0, near = 0;
            float best = 0.f;
            stream_partner_pass<0, uint64_t>(N, q.a, s_pos + q.base, px, py, 0.0f, 0.0f, V.coll_thr2, V.near_thr2, coll, near, best);
            if (HALF && q.live) S.s_near[q.r] = near;
            if (!HALF) {
                r_px = px; r_py = py; r_vx = V.vel_x[q.g]; r_vy = V.vel_y[q.g];
                if (L > 0) { r_olx = V.lm_x[(size_t)q.env * L + q.la]; r_oly = V.lm_y[(size_t)q.env * L + q.la]; }
                r_ep_step = V.ep_step[q.env]; r_ep_count = V.ep_count[q.env];
                r_near = near;
            }
        }
    }

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

    // Gumbel noise of logits 4 kq .. 4 kq + 3 of global row `grow`, drawn IN the lanes that subtract it: value (row, logit o) = log(-log(u)),
    // u = word (o & 3) of Philox block (o >> 2) keyed (seed; step, global row) -- the keying of every other form -- so row group 0 (logits
    // 0..3) needs block 0 and row group 1 (logit 4) block 1 of its row: no noise plane in LDS.  Then the arg-max (first maximum wins).
    auto sample = [&](const f32x4 lg, const long grow, const uint64_t step) {
        float nz[4] = {0.f, 0.f, 0.f, 0.f};
        if (kq < 2) {  // wave-divergent only by row group
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
        const float p4 = lane_xor16(p0);              // logit 4 lives in register 0 of row group 1
        int bi = 0;
        float bv = p0;
        if (p1 > bv) { bv = p1; bi = 1; }
        if (p2 > bv) { bv = p2; bi = 2; }
        if (p3 > bv) { bv = p3; bi = 3; }
        if (p4 > bv) { bv = p4; bi = 4; }
        return bi;
    };
    auto head_b2 = [&]() {
        f32x4 lg;
#pragma unroll
        for (int i = 0; i < 4; ++i) lg[i] = 4 * kq + i < OUT ? S.s_b2[4 * kq + i] : 0.0f;
        return lg;
    };
    // !HALF: the head after the pass, on the tiles of the row-ordered head input (pw_policy_rollout3_kernel's)
    auto head_rows = [&](const uint64_t step) {
        const int ntile = (rows_here + 15) >> 4;
        for (int tile = wave; tile < ntile; tile += 8) {
            f32x4 lg = head_b2();
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
            const int rq = rr < rows_here ? rr : 0, re = rq / NP, ra = rq - re * NP;   // LDS row -> (environment, agent)
            const int bi = sample(lg, row_base + (long)re * N + (ra < N ? ra : 0), step);   // the TRUE global row keys the noise
            if (kq == 0 && rr < rows_here) S.s_act[rr] = (uint8_t)bi;
        }
        wg_lds_barrier();
    };
    // HALF: the head of ONE timestep (its 16 sequences are the 16 columns) from the two directions' h fragments ([2 j][64 lane] each)
    auto head_ts = [&](const int ts, const float4 *hF, const float4 *hB, const uint64_t step) {
        f32x4 lg = head_b2();
        auto half = [&](const float4 *hx, const int s0) {
#pragma unroll
            for (int jx = 0; jx < 2; ++jx) {
                float4 b = hx[jx * 64 + lane];
                if (A.relu_out) { b.x = fmaxf(b.x, 0.0f); b.y = fmaxf(b.y, 0.0f); b.z = fmaxf(b.z, 0.0f); b.w = fmaxf(b.w, 0.0f); }
                lg = __builtin_amdgcn_mfma_f32_16x16x4f32(aw2[s0 + 4 * jx + 0], b.x, lg, 0, 0, 0);
                lg = __builtin_amdgcn_mfma_f32_16x16x4f32(aw2[s0 + 4 * jx + 1], b.y, lg, 0, 0, 0);
                lg = __builtin_amdgcn_mfma_f32_16x16x4f32(aw2[s0 + 4 * jx + 2], b.z, lg, 0, 0, 0);
                lg = __builtin_amdgcn_mfma_f32_16x16x4f32(aw2[s0 + 4 * jx + 3], b.w, lg, 0, 0, 0);
            }
        };
        half(hF, 0);   // k = 0 .. 31: the forward direction's units
        half(hB, 8);   // k = 32 .. 63: the backward direction's
        // the logits replace the stored half (this wave just consumed it; nobody else reads it): [64 lane] float4 of the timestep's slot.
        // Noise and arg-max follow after the loop, on all eight waves at once (inside the loop they sat on the critical path of two).
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        reinterpret_cast<f32x4 *>(S.s_hf + ts * 128)[lane] = lg;
    };
    auto sample_all = [&](const uint64_t step) {   // HALF, after the pass: timestep ts = wave, wave + 8, ...
        for (int ts = wave; ts < N; ts += 8) {
            const f32x4 lg = reinterpret_cast<const f32x4 *>(S.s_hf + ts * 128)[lane];
            const int bi = sample(lg, row_base + (long)nseq * N + ts, step);
            if (kq == 0 && seq_ok) S.s_act[n16 * NP + ts] = (uint8_t)bi;
        }
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
            // no "landmark index < L ? .. : 0" here: beyond the row (k >= D) W1's packed image holds +0 weights and lmk holds 0, so the
            // operand is the finite -pc and the product a zero that leaves the accumulator as it is (the chain starts from +0 and a sum
            // of zeros or an exact cancellation is +0 in round-to-nearest: the accumulator is never -0, so the zero's sign cannot show).
            // The KS - 1 loop-invariant lane masks of that select were hoisted into SGPR pairs, spilled, and re-read in every timestep.
            const float x = lmk[s - 1] - pc;
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

    // Environment step t of one slot (pw_spread_stream_kernel's arithmetic): state from LDS, near mask from a partner pass on the
    // current positions, action force + contact forces + integration, then partner pass on the new positions, rewards, stores,
    // bookkeeping, the reset where an episode ends, and the next state published.
    auto env_step = [&](const int t, const int vw) {
        const EnvLane q = env_lane(vw);
        const int a = q.a, base = q.base, r = q.r, la = q.la, el = q.el;
        const bool live = q.live;
        const long env = q.env;
        const uint32_t g = q.g;
        const size_t tBN = (size_t)t * BN;
        const float2 *pp = s_pos + base;
        float2 *lmv = S.s_lmb + el * L;
        float vx = r_vx, vy = r_vy, px = r_px, py = r_py, olx = r_olx, oly = r_oly, best = 0.f;
        int ep_step = r_ep_step;
        uint32_t ep_count = r_ep_count;
        uint64_t coll = 0, near = r_near;   // on the current positions: left by the previous step's partner pass
        if (HALF) {
            const float4 st = S.s_st[r];
            vx = st.x; vy = st.y; px = st.z; py = st.w;
            if (L > 0) { const float2 o = lmv[la]; olx = o.x; oly = o.y; }
            ep_step = S.s_eps[el];
            ep_count = S.s_epc[el];
            near = S.s_near[r];
        }
        wave_lds_sync();
        if (live) s_pos[base + a] = make_float2(px, py);   // the contact forces read the partners' positions here
        wave_lds_sync();
        const int ai = S.s_act[r];
        size_t slot = 0;
        if (SINK && P.has_ring) {  // the observation the policy acted on: rebuilt from the (still pre-step) state
            slot = ring_slot(P.ring_start, t, A.B, (long)env, P.ring.capacity);
            if (live) {
                if (P.ring.state_rows) sink_state_obs(P.ring, slot, N, a, L, lmv, px, py, vx, vy);
                else stream_write_obs<0>(P.ring.obs + (slot * N + a) * D, L, lmv, px, py, vx, vy);
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
            const float rsum = S.s_ret[el] + acc;
            if (term) { S.s_fs[el] += (double)rsum; S.s_fc[el] += 1; S.s_ret[el] = 0.0f; }
            else S.s_ret[el] = rsum;
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
                if (P.ring.state_rows) sink_state_next(P.ring, slot, N, a, px, py, vx, vy);
                else stream_write_obs<0>(P.ring.next_obs + (slot * N + a) * D, L, lmv, px, py, vx, vy);
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
        }
        wave_lds_sync();
        if (V.auto_reset && __any(term)) {   // post-reset positions: the masks of the next step
            if (live) s_pos[base + a] = make_float2(px, py);
            wave_lds_sync();
            stream_partner_pass<0, uint64_t>(N, a, pp, px, py, olx, oly, V.coll_thr2, V.near_thr2, coll, near, best);
        }
        if (live) {
            if (V.obs) stream_write_obs<0>(V.obs + (tBN + g) * D, L, lmv, px, py, vx, vy);
            S.s_st[r] = make_float4(vx, vy, px, py);
            if (HALF) S.s_near[r] = near;
            if (a == 0) { S.s_eps[el] = ep_step; S.s_epc[el] = ep_count; }
        }
        if (!HALF) {
            r_vx = vx; r_vy = vy; r_px = px; r_py = py; r_olx = olx; r_oly = oly;
            r_ep_step = ep_step; r_ep_count = ep_count; r_near = near;
        }
    };

    for (int t = 0; t < P.T; ++t) {
        const uint64_t step = step0 + (uint64_t)t;
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
                if (HALF && s2 > 0) {  // the timesteps the barrier of iteration s2 - 1 completed: their heads, on two rotating waves
                    const int p = s2 - 1, tsF = p, tsB = N - 1 - p;
                    const float4 *hxp = S.s_hx + ((p & 1) * 2) * 2 * 64;   // [dir][2 j][64]: both directions' h of iteration p
                    if (2 * tsF > N - 1 && wave == ((2 * p) & 7)) head_ts(tsF, hxp, S.s_hf + tsF * 128, step);            // backward was there first
                    if (2 * tsB <= N - 1 && wave == ((2 * p + 5) & 7)) head_ts(tsB, S.s_hf + tsB * 128, hxp + 128, step);   // forward was there first
                }
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
                if (HALF) {
                    // this direction is the FIRST visitor of ts (forward: 2 ts <= N - 1, backward: 2 ts > N - 1): its h(ts) is kept for the head
                    if (dir ? 2 * ts > N - 1 : 2 * ts <= N - 1)
                        reinterpret_cast<float2 *>(S.s_hf + ts * 128 + (hq >> 1) * 64 + lane)[hq & 1] = make_float2(h0v, h1v);
                } else if (seq_ok) {
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
        if (HALF) {  // the last iteration completed timesteps N - 1 (forward) and 0 (backward)
            const int p = N - 1;
            const float4 *hxp = S.s_hx + ((p & 1) * 2) * 2 * 64;
            if (2 * p > N - 1 && wave == ((2 * p) & 7)) head_ts(p, hxp, S.s_hf + p * 128, step);
            if (0 <= N - 1 && wave == ((2 * p + 5) & 7)) head_ts(0, S.s_hf, hxp + 128, step);
            wg_lds_barrier();   // every timestep's logits are in LDS
            sample_all(step);
            wg_lds_barrier();
        } else {
            head_rows(step);  // one barrier inside
        }

        // ---- environment step: every slot of this wave
        if (env_wave)
            for (int vw = ew; vw < n_vwaves; vw += 8) env_step(t, vw);
        wg_lds_barrier();  // the next states (and, after a reset, landmarks) are in LDS
    }

    if (env_wave) {   // the workgroup's state: LDS -> global planes
        for (int vw = ew; vw < n_vwaves; vw += 8) {
            const EnvLane q = env_lane(vw);
            if (q.live) {
                const float4 st = S.s_st[q.r];
                V.vel_x[q.g] = st.x; V.vel_y[q.g] = st.y; V.pos_x[q.g] = st.z; V.pos_y[q.g] = st.w;
                if (L > 0 && q.a < L) {
                    const float2 o = S.s_lmb[q.el * L + q.la];
                    V.lm_x[(size_t)q.env * L + q.la] = o.x;
                    V.lm_y[(size_t)q.env * L + q.la] = o.y;
                }
                if (q.a == 0) {
                    V.ep_step[q.env] = S.s_eps[q.el];
                    V.ep_count[q.env] = S.s_epc[q.el];
                    if (SINK && P.episode_return) P.episode_return[q.env] = S.s_ret[q.el];
                }
            }
        }
    }
    if (SINK && P.episode_return) {
        wg_lds_barrier();
        rollout_finish_stats(envs_here, S.s_fs, S.s_fc, P.scratch, P.finished_sum, P.finished_count, smem_raw);
    }
}

}  // namespace
