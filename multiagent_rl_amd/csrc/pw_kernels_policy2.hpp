// pw_kernels_policy2.hpp -- part of libpworld.so (one translation unit: csrc/pworld.hip includes it).
// Policy-in-the-loop rollout, second form: role-specialised waves so that the matrix cores and the vector ALUs of
// a CU work at the same time instead of in turns.
#pragma once

namespace {

// ------------------------------------------------------------------------------------------
// pw_policy_rollout_kernel (pw_kernels_policy.hpp) runs the actor pass as a sequence of workgroup-wide phases:
// stage 1 -> [fill W_ih/W_hh(d) -> stage 2(d) on the matrix cores -> recurrence(d) on the vector ALUs] x 2 -> head.
// The MFMA pipes idle during the recurrences and the VALUs during stage 2: 47 k of the 52 k cycles of a step.
// Here the eight waves of a workgroup keep ONE role for the whole rollout:
//   waves 0-3  "matrix" waves, one per SIMD.  Wave w owns gate-unit tile w (32 of a direction's 128 units) of BOTH
//              directions: its W_ih fragments (64 VGPRs) are loaded once per launch and never leave registers -- no
//              per-direction weight refill through LDS.  The workgroup's rows are tiled AGENT-MAJOR (MFMA column
//              rho = agent * E + env), so a 32-row tile is "a few timesteps of every sequence": per step the wave
//              runs stage 1 (W1 X^T, relu; kept in registers per tile) and stage 2 for the forward direction on
//              tiles 0, 1, 2 and for the reverse direction on tiles 2, 1, 0 -- the order in which the two
//              recurrences need them -- two independent accumulator chains at a time (forward tile k with reverse
//              tile n-1-k), each written to its LDS tile (Gf, Gr) and announced through an LDS counter.
//   waves 4-5  forward LSTM waves, waves 6-7 reverse LSTM waves, one per SIMD.  The 128 W_hh weights of a lane's
//              hidden unit stay in VGPRs for the whole launch.  A wave advances EIGHT sequences at a time (four per
//              lane group, interleaved: the other sequences' FMAs hide a sequence's LDS round trips and
//              transcendentals, and every weight register is used four times per step); it takes timestep t as soon
//              as the matrix waves have announced the tile holding it, so recurrence steps overlap the stage-2 MFMAs
//              of the later timesteps on the same SIMD's other pipe, and both directions run concurrently.
//   all        head + Gumbel-argmax as before (the Gumbel noise of the NEXT step is drawn by the LSTM waves while
//              the environment step runs); the first matrix waves then advance the environments
//              (pw_spread_stream_kernel's arithmetic) and refresh the observation rows in LDS.
// The arithmetic of every output element is unchanged (same MFMA k-order, same FMA chains, same Philox keys), so the
// results are bit-identical to pw_policy_rollout_kernel, to pw_actor_fused + pw_step loops and to the three-launch
// chain (tests/test_gpu_engine.py).  Per step: ~15 k cycles of MFMA issue per SIMD (the f32 matrix rate is the
// floor) with the recurrences hidden beneath, + ~4 k LSTM tail + head + environment step.
// LDS: Gf + Gr (2 x rows x 129 floats) + Hs + observation rows + small; the host picks the number of environments
// per workgroup so that it fits 160 KB (16 at N = 6, fewer for long observation rows).
// ------------------------------------------------------------------------------------------
struct Roll2Lds {
    float *s_g[2];   // [GR][129] per direction, GR = rows rounded up to 32
    float *s_hid;    // [GR][68]
    float *f_w1;     // [2 m][S1][64 lane]
    float *s_b1, *s_bih, *s_w2, *s_b2;
    float *s_hx;     // [4 LSTM waves][8 sequences][32]
    float *s_noise;  // [GR * 5] Gumbel noise of the coming head
    float *s_obs;    // [GR][DS], DS = D + 2 (bank spread; rows 8-byte aligned)
    int32_t *s_act;  // [GR]
    float2 *s_posb;  // [2 env waves][64]
    float2 *s_lmb;   // [E * L]
    double *s_fs;    // [16]
    int *s_fc;       // [16]
    unsigned *s_flag;  // [2 dir][4 row tiles]
    float *s_lg;     // perturbed logits [rows * 16] -- aliases Gf, dead once both recurrences are done
};
__host__ __device__ inline size_t roll2_lds_bytes(int E, int N, int L, int D, int S1)
{
    const size_t GR = (size_t)((E * N + 31) / 32) * 32;
    size_t fl = 2 * GR * kGs + GR * kHs + (size_t)2 * S1 * 64 + 64 + 256 + 1024 + 16 + 4 * 8 * 32 + GR * 5 + GR * (D + 2) + GR;
    return fl * 4 + 2 * kWave * sizeof(float2) + (size_t)E * L * sizeof(float2) + 16 * (sizeof(double) + sizeof(int)) + 8 * 4 + 64;
}
__device__ __forceinline__ Roll2Lds roll2_carve(unsigned char *raw, int E, int N, int L, int D, int S1)
{
    // offsets in floats from the (16-byte aligned) base; no pointer <-> integer casts, so that every pointer keeps
    // its LDS address space (ds_* instructions, not flat_*)
    const int GR = ((E * N + 31) / 32) * 32;
    float *base = reinterpret_cast<float *>(raw);
    Roll2Lds S;
    int o = 0;
    S.s_g[0] = base + o; o += GR * kGs;
    S.s_g[1] = base + o; o += GR * kGs;
    o = (o + 3) & ~3;
    S.s_hid = base + o; o += GR * kHs;
    S.f_w1 = base + o; o += 2 * S1 * 64;
    S.s_b1 = base + o; o += 64;
    S.s_bih = base + o; o += 256;
    S.s_w2 = base + o; o += 1024;
    S.s_b2 = base + o; o += 16;
    S.s_hx = base + o; o += 4 * 8 * 32;
    S.s_noise = base + o; o += GR * 5;
    o = (o + 1) & ~1;
    S.s_obs = base + o; o += GR * (D + 2);
    S.s_act = reinterpret_cast<int32_t *>(base + o); o += GR;
    o = (o + 1) & ~1;
    S.s_posb = reinterpret_cast<float2 *>(base + o); o += 2 * kWave * 2;
    S.s_lmb = reinterpret_cast<float2 *>(base + o); o += E * L * 2;
    S.s_fs = reinterpret_cast<double *>(base + o); o += 32;
    S.s_fc = reinterpret_cast<int *>(base + o); o += 16;
    S.s_flag = reinterpret_cast<unsigned *>(base + o);
    S.s_lg = S.s_g[0];
    return S;
}

// observation row into LDS with 8-byte stores (the LDS rows are only 8-byte aligned: stride D + 2)
template <int LT>
__device__ __forceinline__ void lds_write_obs_row(float *o, const int L, const float2 *lm, float px, float py, float vx,
                                                  float vy)
{
    float2 *o2 = reinterpret_cast<float2 *>(o);
    o2[0] = make_float2(vx, vy);
    o2[1] = make_float2(px, py);
#pragma unroll(LT > 0 ? LT : 1)
    for (int l = 0; l < (LT ? LT : L); ++l) {
        const float2 q = lm[l];
        o2[2 + l] = make_float2(q.x - px, q.y - py);
    }
}

// one hidden unit's W_hh rows straight from global memory (once per launch): whh [4 gates x 32 units][32] row-major
__device__ __forceinline__ void lstm_load_unit_global(const float *whh, const int j, LstmUnitW &w)
{
    const float4 *wh = reinterpret_cast<const float4 *>(whh);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float4 a = wh[(0 * 32 + j) * 8 + q], b = wh[(1 * 32 + j) * 8 + q];
        const float4 c = wh[(2 * 32 + j) * 8 + q], d = wh[(3 * 32 + j) * 8 + q];
        w.wif[4 * q] = f32x2{a.x, b.x}; w.wif[4 * q + 1] = f32x2{a.y, b.y};
        w.wif[4 * q + 2] = f32x2{a.z, b.z}; w.wif[4 * q + 3] = f32x2{a.w, b.w};
        w.wgo[4 * q] = f32x2{c.x, d.x}; w.wgo[4 * q + 1] = f32x2{c.y, d.y};
        w.wgo[4 * q + 2] = f32x2{c.z, d.z}; w.wgo[4 * q + 3] = f32x2{c.w, d.w};
    }
}

// lstm_step for FOUR independent sequences sharing the lane's weights: the same FMA chains per sequence (k
// ascending), interleaved so that no sequence waits for its own LDS reads or transcendentals.
// hs: the wave's h exchange [4 slots x 2 lane groups][32]; slot i of this lane group at hs + i * 64
// lstm_step for FOUR independent sequences sharing the lane's weights: the same FMA chains per sequence (k
// ascending), interleaved so that no sequence waits for its own LDS reads or transcendentals.
// hs: the wave's h exchange [4 slots x 2 lane groups][32]; slot i of this lane group at hs + i * 64
__device__ __forceinline__ void lstm_step4(const LstmUnitW &w, const float *hs, const float (&g)[4][4], float (&h)[4],
                                           float (&c)[4])
{
    f32x2 aif[4], ago[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { aif[i] = f32x2{g[i][0], g[i][1]}; ago[i] = f32x2{g[i][2], g[i][3]}; }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        float4 u[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) u[i] = reinterpret_cast<const float4 *>(hs + i * 64)[q];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            aif[i] = __builtin_elementwise_fma(w.wif[4 * q], f32x2{u[i].x, u[i].x}, aif[i]);
            ago[i] = __builtin_elementwise_fma(w.wgo[4 * q], f32x2{u[i].x, u[i].x}, ago[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            aif[i] = __builtin_elementwise_fma(w.wif[4 * q + 1], f32x2{u[i].y, u[i].y}, aif[i]);
            ago[i] = __builtin_elementwise_fma(w.wgo[4 * q + 1], f32x2{u[i].y, u[i].y}, ago[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            aif[i] = __builtin_elementwise_fma(w.wif[4 * q + 2], f32x2{u[i].z, u[i].z}, aif[i]);
            ago[i] = __builtin_elementwise_fma(w.wgo[4 * q + 2], f32x2{u[i].z, u[i].z}, ago[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            aif[i] = __builtin_elementwise_fma(w.wif[4 * q + 3], f32x2{u[i].w, u[i].w}, aif[i]);
            ago[i] = __builtin_elementwise_fma(w.wgo[4 * q + 3], f32x2{u[i].w, u[i].w}, ago[i]);
        }
    }
    wave_lds_sync();  // all reads of h done before the caller overwrites it
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        lstm_cell(aif[i].x, aif[i].y, ago[i].x, ago[i].y, c[i], h[i]);
    }
}

#ifdef PW_STAMPS
__device__ int g_pw_debug;  // probe builds only: bit 0 skips the MFMAs (tiles still announced), bit 1 the recurrences
#define PW_DBG(bit) (g_pw_debug & (bit))
#else
#define PW_DBG(bit) 0
#endif

__device__ __forceinline__ unsigned lds_flag_load(const unsigned *f)
{
    return __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

#ifdef PW_STAMPS
#define PW_R2_DECL unsigned long long rs[8] = {0, 0, 0, 0, 0, 0, 0, 0}, r0_ = 0, r1_ = 0
#define PW_R2_START asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0_)::"memory")
#define PW_R2_STAMP(i) do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1_)::"memory"); rs[i] += r1_ - r0_; r0_ = r1_; } while (0)
#define PW_R2_FLUSH(base) do { if (blockIdx.x == 0 && lane == 0) for (int i_ = 0; i_ < 8; ++i_) g_pw_stamps[(base) + i_] = rs[i_]; } while (0)
#else
#define PW_R2_DECL
#define PW_R2_START
#define PW_R2_STAMP(i)
#define PW_R2_FLUSH(base)
#endif

template <int S1C, int NT, bool SINK>
__global__ void __launch_bounds__(512) pw_policy_rollout2_kernel(const PolicyRolloutArgs P)
{
    constexpr int LT = NT;
    constexpr int S1 = 4 * S1C;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const ActorFusedArgs &A = P.A;
    const StreamParams &V = P.V;
    const int N = NT ? NT : A.N, L = LT ? LT : V.L, D = A.D, DS = D + 2, E = A.E;
    const Roll2Lds S = roll2_carve(smem_raw, E, N, L, D, S1);

    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, col = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long env0 = (long)blockIdx.x * E;
    const int envs_here = (int)((long)A.B - env0 < (long)E ? (long)A.B - env0 : (long)E);
    const int rows_here = envs_here * N;
    const int nrho = N * E;                 // agent-major MFMA columns rho = a * E + e (slots of absent envs included)
    const int nrt = (nrho + 31) >> 5;       // 32-column tiles in use (<= 3)
    const long row_base = env0 * N;
    const size_t BN = (size_t)A.B * N;
    const uint64_t step0 = A.step_dev ? (uint64_t)*A.step_dev : A.step;
    constexpr int OUT = 5;  // one 5-logit head (checked on the host)

    // ---- constants -> LDS (once)
    {
        const float4 *src = reinterpret_cast<const float4 *>(A.frag + 8 * 2 * 4 * 64 * 4);
        for (int f = tid; f < (2 * S1 * 64) / 4; f += 512) reinterpret_cast<float4 *>(S.f_w1)[f] = src[f];
        if (tid < 64) S.s_b1[tid] = A.b1[tid];
        if (tid < 256) S.s_bih[tid] = A.bih[tid];
        for (int f = tid; f < OUT * 64; f += 512) S.s_w2[f] = A.w2[f];
        if (tid < OUT) S.s_b2[tid] = A.b2[tid];
        if (tid < 8) S.s_flag[tid] = 0u;
    }

    // Gumbel noise of one head evaluation: value (row r, logit o) = log(-log(u)), u = word (o & 3) of Philox block
    // (o >> 2) keyed (seed; step, global row) exactly as actor_forward_wg / pw_actor_head_kernel.  Called by the 256
    // threads of the LSTM waves (t0 = their index) while the environment step runs.
    auto draw_noise = [&](const uint64_t step, const int t0) {
        for (int idx = t0; idx < rows_here * OUT; idx += 256) {
            const int r = idx / OUT, o = idx - r * OUT;
            const long grow = row_base + r;
            const uint32_t blk = (uint32_t)o >> 2, tag = ((blk & 1u) << 31) | ((blk >> 1) << 30);
            uint32_t u[4];
            pw_philox4x32_10((uint32_t)grow, (uint32_t)((uint64_t)grow >> 32) | tag, (uint32_t)step, (uint32_t)(step >> 32),
                             (uint32_t)A.seed, (uint32_t)(A.seed >> 32), u);
            const int wq = o & 3;
            const uint32_t uw = wq == 0 ? u[0] : wq == 1 ? u[1] : wq == 2 ? u[2] : u[3];
            const float uo = ((float)(uw >> 8) + 0.5f) * 5.9604644775390625e-8f;  // (0, 1)
            S.s_noise[idx] = __logf(-__logf(uo));
        }
    };

    // the head: thread = (row, logit), then one thread per row picks the arg-max (actor_forward_wg's arithmetic)
    auto head = [&]() {
        for (int idx = tid; idx < rows_here * OUT; idx += 512) {
            const int r = idx / OUT, o = idx - r * OUT;
            float acc = S.s_b2[o];
            const float4 *hv = reinterpret_cast<const float4 *>(S.s_hid + r * kHs), *wv = reinterpret_cast<const float4 *>(S.s_w2 + o * 64);
#pragma unroll 4
            for (int q = 0; q < 16; ++q) {  // partially unrolled: the LSTM waves run this with 128 weight registers live
                const float4 hq = hv[q], wq = wv[q];
                acc = __builtin_fmaf(wq.x, hq.x, acc);
                acc = __builtin_fmaf(wq.y, hq.y, acc);
                acc = __builtin_fmaf(wq.z, hq.z, acc);
                acc = __builtin_fmaf(wq.w, hq.w, acc);
            }
            S.s_lg[idx] = acc - S.s_noise[idx];
        }
        wg_lds_barrier();
        for (int r = tid; r < rows_here; r += 512) {
            const float *v = S.s_lg + r * OUT;
            int best = 0;
            float bv = v[0];
#pragma unroll
            for (int o = 1; o < OUT; ++o)
                if (v[o] > bv) { bv = v[o]; best = o; }
            S.s_act[r] = best;
        }
        wg_lds_barrier();
    };

    if (wave < 4) {
        // =========================== matrix waves (+ the environment step) ===========================
        // environment lanes (as pw_policy_rollout_kernel): wave w < n_env_waves owns local envs [w * epw, ...)
        const int epw_max = E < kWave / N ? E : kWave / N;
        const int waves_full = (E + epw_max - 1) / epw_max;
        const int epw = (E + waves_full - 1) / waves_full;
        const int n_env_waves = (envs_here + epw - 1) / epw;  // <= 2
        const bool env_wave = wave < n_env_waves;
        int e_loc = lane / N, a = lane - e_loc * N;
        int el = wave * epw + e_loc;
        const bool live = env_wave && e_loc < epw && el < envs_here;
        if (!live) { e_loc = 0; a = 0; el = env_wave ? wave * epw : 0; }
        const int base = e_loc * N, r = el * N + a;
        const long env = env0 + el;
        const uint32_t g = (uint32_t)env * (uint32_t)N + (uint32_t)a;
        float2 *s_pos = S.s_posb + (env_wave ? wave : 0) * kWave;
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

        // this wave's W_ih fragments: gate-unit tile `wave` of both directions, resident for the whole launch
        float4 wf[2][2][4];
        {
            const float4 *frag4 = reinterpret_cast<const float4 *>(A.frag);
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) wf[d][m][rq] = frag4[(((d * 4 + wave) * 2 + m) * 4 + rq) * 64 + lane];
        }
        // observation row behind MFMA column rho = rt * 32 + col: agent a = rho / E of local env e = rho % E
        // (columns past N * E, and envs this workgroup does not have, read a valid row; nobody uses their results)
        int obs_off[3];
#pragma unroll
        for (int rt = 0; rt < 3; ++rt) {
            int rho = rt * 32 + col;
            if (rho >= nrho) rho = nrho - 1;
            const int aa = rho / E;
            int ee = rho - aa * E;
            if (ee >= envs_here) ee = 0;
            obs_off[rt] = (ee * N + aa) * DS;
        }
        wg_lds_barrier();  // constants, flags, the first observation rows and the first noise are in LDS
        PW_R2_DECL;

        // stage 1 of column tile RT into registers
        auto stage1 = [&](f32x16 (&acc1)[2], const int off) {
            const float *xr = S.s_obs + off;
            float xb[S1];
#pragma unroll
            for (int sidx = 0; sidx < S1; ++sidx) {
                const int kk = 2 * sidx + half;
                xb[sidx] = kk < D ? xr[kk] : 0.0f;
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc1[m][q] = 0.0f;
#pragma unroll
            for (int sidx = 0; sidx < S1; ++sidx) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    acc1[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(S.f_w1[(m * S1 + sidx) * 64 + lane], xb[sidx], acc1[m], 0, 0, 0);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc1[m][q] = fmaxf(acc1[m][q] + S.s_b1[m * 32 + mfma_row(q, half)], 0.0f);
        };
        // stage 2: forward direction on tile rtf and reverse direction on tile rtr, two independent accumulator chains
        // interleaved (a lone chain of dependent MFMAs leaves the pipe idle between its links), then both tiles are
        // written to LDS and announced
        auto stage2_pair = [&](const f32x16 (&xf)[2], const int rtf, const f32x16 (&xr_)[2], const int rtr) {
            f32x16 af, ar;
#pragma unroll
            for (int q = 0; q < 16; ++q) { af[q] = 0.0f; ar[q] = 0.0f; }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const float4 wa = wf[0][m][rq], wb = wf[1][m][rq];
                    af = __builtin_amdgcn_mfma_f32_32x32x2f32(wa.x, xf[m][4 * rq + 0], af, 0, 0, 0);
                    ar = __builtin_amdgcn_mfma_f32_32x32x2f32(wb.x, xr_[m][4 * rq + 0], ar, 0, 0, 0);
                    af = __builtin_amdgcn_mfma_f32_32x32x2f32(wa.y, xf[m][4 * rq + 1], af, 0, 0, 0);
                    ar = __builtin_amdgcn_mfma_f32_32x32x2f32(wb.y, xr_[m][4 * rq + 1], ar, 0, 0, 0);
                    af = __builtin_amdgcn_mfma_f32_32x32x2f32(wa.z, xf[m][4 * rq + 2], af, 0, 0, 0);
                    ar = __builtin_amdgcn_mfma_f32_32x32x2f32(wb.z, xr_[m][4 * rq + 2], ar, 0, 0, 0);
                    af = __builtin_amdgcn_mfma_f32_32x32x2f32(wa.w, xf[m][4 * rq + 3], af, 0, 0, 0);
                    ar = __builtin_amdgcn_mfma_f32_32x32x2f32(wb.w, xr_[m][4 * rq + 3], ar, 0, 0, 0);
                }
            }
            // the tiles go out WITHOUT the bias: the LSTM lanes add it when they pick a value up (their four biases are
            // lane constants in registers; here it would be 32 just-in-time LDS reads per pair -- 2.7 k cycles of
            // round trips at this kernel's register budget).  acc + bias either way: the same bits.
            float *df = S.s_g[0] + (rtf * 32 + col) * kGs + wave * 32, *dr = S.s_g[1] + (rtr * 32 + col) * kGs + wave * 32;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                df[mfma_row(q, half)] = af[q];
                dr[mfma_row(q, half)] = ar[q];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the tiles have landed before they are announced
            if (lane == 0) {
                __hip_atomic_fetch_add(S.s_flag + rtf, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(S.s_flag + 4 + rtr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        };

        for (int t = 0; t < P.T; ++t) {
            PW_R2_START;
            // ---- stage 1 + stage 2: forward tiles ascending paired with reverse tiles descending
            f32x16 x0[2], x1[2], x2[2];
            if (PW_DBG(1)) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0)
                    for (int q = 0; q < nrt; ++q) {
                        __hip_atomic_fetch_add(S.s_flag + q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(S.s_flag + 4 + q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
            } else if (nrt == 3) {
                stage1(x0, obs_off[0]);
                stage1(x2, obs_off[2]);
                PW_R2_STAMP(0);
                stage2_pair(x0, 0, x2, 2);
                PW_R2_STAMP(1);
                stage1(x1, obs_off[1]);
                PW_R2_STAMP(2);
                stage2_pair(x1, 1, x1, 1);
                PW_R2_STAMP(3);
                stage2_pair(x2, 2, x0, 0);
            } else if (nrt == 2) {
                stage1(x0, obs_off[0]);
                stage1(x1, obs_off[1]);
                stage2_pair(x0, 0, x1, 1);
                stage2_pair(x1, 1, x0, 0);
            } else {
                stage1(x0, obs_off[0]);
                stage2_pair(x0, 0, x0, 0);
            }
            PW_R2_STAMP(4);
            wg_lds_barrier();  // B1: both recurrences done, Hs complete
            PW_R2_STAMP(5);
            head();  // B2, B3 inside
            PW_R2_STAMP(6);

            // ---- environment step (pw_spread_stream_kernel's arithmetic)
            if (env_wave) {
                const size_t tBN = (size_t)t * BN;
                const int ai = S.s_act[r];
                size_t slot = 0;
                if (SINK && P.has_ring) {  // the observation the policy acted on (still in LDS) -> ring.obs
                    slot = (size_t)((P.ring_start + (int64_t)t * A.B + env) % P.ring.capacity);
                    if (live) {
                        const float2 *src = reinterpret_cast<const float2 *>(S.s_obs + r * DS);
                        float2 *dst = reinterpret_cast<float2 *>(P.ring.obs + (slot * N + a) * D);
                        for (int c = 0; c < D / 2; ++c) dst[c] = src[c];
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
                stream_partner_pass<NT, uint64_t>(N, a, pp, px, py, olx, oly, V.coll_thr2, V.near_thr2, coll, near, best);
                const float own = sqrtf(best);
                float rw = 0.0f;
#pragma unroll(LT > 0 ? LT : 1)
                for (int l = 0; l < L; ++l) rw -= __shfl(own, base + l, kWave);
#pragma unroll(NT > 0 ? NT : 1)
                for (int j = 0; j < N; ++j)
                    if ((coll >> j) & 1) rw -= 1.0f;
                float acc = 0.0f;
#pragma unroll(NT > 0 ? NT : 1)
                for (int i = 0; i < N; ++i) acc += __shfl(rw, base + i, kWave);
                ep_step += 1;
                const bool term = V.max_episode_len > 0 && ep_step >= V.max_episode_len;
                if (live) {
                    if (P.act_out) P.act_out[tBN + g] = ai;
                    if (V.rew) V.rew[tBN + g] = rw;
                    if (V.done) V.done[tBN + g] = 0;
                    if (a == 0) {
                        if (V.rew_shared) V.rew_shared[(size_t)t * A.B + env] = acc;
                        if (V.terminal) V.terminal[(size_t)t * A.B + env] = term ? 1 : 0;
                    }
                    if (SINK && P.has_ring) {  // next_obs is the PRE-reset observation (run.py:52 vs :60)
                        stream_write_obs<LT>(P.ring.next_obs + (slot * N + a) * D, L, lmv, px, py, vx, vy);
                        if (a == 0) { P.ring.rew[slot] = acc; P.ring.done[slot] = 0.0f; }
                    }
                    if (SINK && a == 0 && P.episode_return) {  // run.py:55-65, per env
                        const float rsum = ep_ret + acc;
                        if (term) { fin_sum += (double)rsum; fin_cnt += 1; ep_ret = 0.0f; }
                        else ep_ret = rsum;
                    }
                }
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
            }
            wg_lds_barrier();  // B4: the next observation rows (and the next noise) are in LDS
            PW_R2_STAMP(7);
        }
        if (wave == 0) PW_R2_FLUSH(0);

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
            if (tid == 0) {
                double ws = 0.0;
                long long wc = 0;
                for (int i = 0; i < envs_here; ++i) { ws += S.s_fs[i]; wc += S.s_fc[i]; }
                double *part_sum = reinterpret_cast<double *>(P.scratch);
                long long *part_cnt = reinterpret_cast<long long *>(P.scratch + gridDim.x);
                unsigned long long *ticket = P.scratch + 2 * gridDim.x;
                part_sum[blockIdx.x] = ws;
                part_cnt[blockIdx.x] = wc;
                __threadfence();
                if (atomicAdd(ticket, 1ull) == (unsigned long long)gridDim.x - 1) {
                    __threadfence();
                    double ssum = 0.0;
                    long long scnt = 0;
                    for (unsigned i = 0; i < gridDim.x; ++i) {
                        ssum += __builtin_nontemporal_load(part_sum + i);
                        scnt += __builtin_nontemporal_load(part_cnt + i);
                    }
                    *P.finished_sum += ssum;
                    *P.finished_count += scnt;
                    *ticket = 0;
                }
            }
        }
    } else {
        // =========================== LSTM waves ===========================
        const int lw = wave - 4, dir = lw >> 1, kk = lw & 1;  // waves 4,5 forward; 6,7 reverse
        const int j = lane & 31;
        LstmUnitW w;
        lstm_load_unit_global(dir ? A.whh_r : A.whh_f, j, w);
        // sequences (= local envs) of this wave: kk * 8 + 2 * i + half, i = 0..3 interleaved in every lane
        float *hs = S.s_hx + (lw * 8 + half) * 32;  // slot i of this lane group at hs + i * 64
        const float *G = S.s_g[dir];
        const unsigned *flag = S.s_flag + dir * 4;
        const int s_first = kk * 8;
        int se[4];
        bool ok[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int sq = s_first + 2 * i + half;
            ok[i] = sq < envs_here;
            se[i] = ok[i] ? sq : 0;  // idle slots shadow sequence 0 (valid G rows; they write nothing)
        }
        const bool busy = s_first < envs_here;  // wave-uniform
        draw_noise(step0, tid - 256);
        wg_lds_barrier();
        float bq[4];  // this unit's input-projection biases (gates i, f, g, o), added to the matrix waves' tiles
#pragma unroll
        for (int q = 0; q < 4; ++q) bq[q] = S.s_bih[dir * 128 + q * 32 + j];
        PW_R2_DECL;

        for (int t = 0; t < P.T; ++t) {
            PW_R2_START;
            const unsigned target = 4u * (unsigned)(t + 1);  // every matrix wave announces each tile once per step
            if (busy && !PW_DBG(2)) {
                float h[4] = {0.f, 0.f, 0.f, 0.f}, c[4] = {0.f, 0.f, 0.f, 0.f};
                for (int s2 = 0; s2 < N; ++s2) {
                    const int ts = dir ? N - 1 - s2 : s2;
                    {   // timestep ts of every sequence = MFMA columns [ts * E, ts * E + E): wait for their tiles
                        const int rt_lo = (ts * E) >> 5, rt_hi = (ts * E + E - 1) >> 5;
                        for (int rt = rt_lo; rt <= rt_hi; ++rt)
                            while (lds_flag_load(flag + rt) < target) __builtin_amdgcn_s_sleep(1);
                        asm volatile("" ::: "memory");
                    }
                    PW_R2_STAMP(0);
                    float gt[4][4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float *gp = G + (ts * E + se[i]) * kGs;
                        gt[i][0] = gp[j] + bq[0]; gt[i][1] = gp[32 + j] + bq[1];
                        gt[i][2] = gp[64 + j] + bq[2]; gt[i][3] = gp[96 + j] + bq[3];
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) hs[i * 64 + j] = h[i];
                    wave_lds_sync();
                    lstm_step4(w, hs, gt, h, c);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (ok[i]) S.s_hid[((s_first + 2 * i + half) * N + ts) * kHs + dir * 32 + j] = A.relu_out ? fmaxf(h[i], 0.0f) : h[i];
                    PW_R2_STAMP(1);
                }
            }
            wg_lds_barrier();  // B1
            PW_R2_STAMP(2);
            head();            // B2, B3
            PW_R2_STAMP(3);
            if (t + 1 < P.T) draw_noise(step0 + (uint64_t)(t + 1), tid - 256);  // under the environment step
            wg_lds_barrier();  // B4
            PW_R2_STAMP(4);
        }
        if (wave == 4) PW_R2_FLUSH(8);
        if (SINK && P.episode_return) {
            wg_lds_barrier();
            wg_lds_barrier();
        }
    }
}

}  // namespace
