// pw_kernels_policy.hpp -- part of libpworld.so (one translation unit: csrc/pworld.hip includes it).
// Policy forward (BiLSTM recurrence, head + Gumbel sampling) and episode bookkeeping.
#pragma once

namespace {

// ------------------------------------------------------------------------------------------
// Policy forward (rls/model/ac_network_multi_gumbel.py:24-67) pieces that MIOpen serves badly:
// its RNN path issues ~45 tiny kernels for a length-6 sequence (260-400 us per batched step at
// B = 4096, 500x the environment step).  The two dense input GEMMs stay in rocBLAS (they are real
// GEMMs: [B*N, 64] x [64, 256]); the recurrence and the output head + Gumbel sampling are fused here.
//
// pw_bilstm_kernel: lane = (env, direction, hidden unit j), 32 lanes per sequence.  A lane keeps the
// four W_hh rows of its unit (i, f, g, o gates; 128 weights) in VGPRs for the whole kernel, so a
// recurrence step is 128 FMAs + 5 activations per lane; h is exchanged through wave-private LDS
// (one write, eight broadcast ds_read_b128).  G holds x*W_ih^T + b_ih + b_hh for every timestep.
// ------------------------------------------------------------------------------------------
// v_exp_f32 / v_rcp_f32 (1 ulp): the policy net is ordinary float32 inference, not part of the
// bit-exact environment contract; tests compare against PyTorch's float32 LSTM with a 2e-5 bound.
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x)
{
    const float e = __expf(-2.0f * fabsf(x));  // in (0, 1]: no overflow
    const float t = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
    return copysignf(t, x);
}

__global__ void __launch_bounds__(256) pw_bilstm_kernel(const float *__restrict__ G, const float *__restrict__ w_fw,
                                                        const float *__restrict__ w_bw, const int B, const int N,
                                                        const int relu_out, float *__restrict__ H)
{
    // W_hh of both directions staged once per workgroup (32 KB), laid out [dir][gate][k/4][unit] as
    // float4 so that the 32 lanes of a sequence read consecutive 16-B slots (conflict-free); the lanes
    // then keep their 128 weights in VGPRs.  Global weight traffic: 32 KB per workgroup instead of
    // 512 B per lane (4x less), read with fully coalesced float4 loads.
    __shared__ float4 s_w[2 * 4 * 8 * 32];
    __shared__ __attribute__((aligned(16))) float s_h[256];  // [8 sequences per workgroup][32]
    for (int f = threadIdx.x; f < 2048; f += 256) {
        const int d = f >> 10, r = f & 1023, row = r >> 3, q = r & 7, gate = row >> 5, unit = row & 31;
        s_w[((d * 4 + gate) * 8 + q) * 32 + unit] = reinterpret_cast<const float4 *>(d ? w_bw : w_fw)[r];
    }
    __syncthreads();
    const int j = threadIdx.x & 31, grp = threadIdx.x >> 5;   // hidden unit, sequence slot in the workgroup
    const long seq = (long)blockIdx.x * 8 + grp;              // sequence id = env * 2 + dir
    const bool valid = seq < 2L * B;
    const long env = valid ? seq >> 1 : 0;
    const int dir = valid ? (int)(seq & 1) : 0;
    float wi[32], wf[32], wg[32], wo[32];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float4 a = s_w[((dir * 4 + 0) * 8 + q) * 32 + j];
        const float4 b = s_w[((dir * 4 + 1) * 8 + q) * 32 + j];
        const float4 c = s_w[((dir * 4 + 2) * 8 + q) * 32 + j];
        const float4 d = s_w[((dir * 4 + 3) * 8 + q) * 32 + j];
        wi[4 * q] = a.x; wi[4 * q + 1] = a.y; wi[4 * q + 2] = a.z; wi[4 * q + 3] = a.w;
        wf[4 * q] = b.x; wf[4 * q + 1] = b.y; wf[4 * q + 2] = b.z; wf[4 * q + 3] = b.w;
        wg[4 * q] = c.x; wg[4 * q + 1] = c.y; wg[4 * q + 2] = c.z; wg[4 * q + 3] = c.w;
        wo[4 * q] = d.x; wo[4 * q + 1] = d.y; wo[4 * q + 2] = d.z; wo[4 * q + 3] = d.w;
    }
    float h = 0.0f, c = 0.0f;
    float *hs = s_h + grp * 32;
    const float *g0 = G + (((size_t)env * N + (dir ? N - 1 : 0)) * 2 + dir) * 128;
    float ni = g0[j], nf = g0[32 + j], ng = g0[64 + j], no = g0[96 + j];
    for (int s = 0; s < N; ++s) {
        const int t = dir ? N - 1 - s : s;
        float ai = ni, af = nf, ag = ng, ao = no;
        if (s + 1 < N) {  // prefetch the next timestep's pre-activations under this step's FMAs
            const float *g = G + (((size_t)env * N + (dir ? t - 1 : t + 1)) * 2 + dir) * 128;
            ni = g[j]; nf = g[32 + j]; ng = g[64 + j]; no = g[96 + j];
        }
        hs[j] = h;
        wave_lds_sync();  // a sequence's 32 lanes sit in one wave
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 hv = reinterpret_cast<const float4 *>(hs)[q];
            ai = __builtin_fmaf(wi[4 * q], hv.x, ai); af = __builtin_fmaf(wf[4 * q], hv.x, af);
            ag = __builtin_fmaf(wg[4 * q], hv.x, ag); ao = __builtin_fmaf(wo[4 * q], hv.x, ao);
            ai = __builtin_fmaf(wi[4 * q + 1], hv.y, ai); af = __builtin_fmaf(wf[4 * q + 1], hv.y, af);
            ag = __builtin_fmaf(wg[4 * q + 1], hv.y, ag); ao = __builtin_fmaf(wo[4 * q + 1], hv.y, ao);
            ai = __builtin_fmaf(wi[4 * q + 2], hv.z, ai); af = __builtin_fmaf(wf[4 * q + 2], hv.z, af);
            ag = __builtin_fmaf(wg[4 * q + 2], hv.z, ag); ao = __builtin_fmaf(wo[4 * q + 2], hv.z, ao);
            ai = __builtin_fmaf(wi[4 * q + 3], hv.w, ai); af = __builtin_fmaf(wf[4 * q + 3], hv.w, af);
            ag = __builtin_fmaf(wg[4 * q + 3], hv.w, ag); ao = __builtin_fmaf(wo[4 * q + 3], hv.w, ao);
        }
        wave_lds_sync();  // all reads of h done before the next step overwrites it
        c = fast_sigmoid(af) * c + fast_sigmoid(ai) * fast_tanh(ag);
        h = fast_sigmoid(ao) * fast_tanh(c);
        if (valid) H[((size_t)env * N + t) * 64 + dir * 32 + j] = relu_out ? fmaxf(h, 0.0f) : h;
    }
}

// Dense layers with a tiny reduction dimension (K <= 64): Y = act(X * W^T + b), W [out, K] row-major.
// Used for dense1 (D -> 64, ReLU) and for the LSTM input projection of both directions (64 -> 256).
// At these shapes a library GEMM is launch- and tile-quantisation-bound (12 + 4.5 us and 18 us at
// B*N = 24576 rows).  Weight-stationary mapping instead: lane = one output unit (its K weights and bias
// stay in VGPRs for the whole kernel), a wave covers 64 outputs and walks over rows whose inputs are
// wave-uniform and arrive through the scalar path (s_load), and each row's 64 outputs leave as one
// coalesced 256-B store.
template <int KT, bool RELU>
__global__ void __launch_bounds__(256) pw_dense_kernel(const float *__restrict__ X, const float *__restrict__ W,
                                                       const float *__restrict__ bvec, const long rows, const int K,
                                                       const int out_dim, const int rows_per_wave,
                                                       float *__restrict__ Y)
{
    constexpr int KM = KT > 0 ? KT : 64;
    const int Kd = KT > 0 ? KT : K;
    const int chunks = out_dim >> 6;
    const long wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    const int chunk = (int)(wave % chunks);
    const long rg = wave / chunks;
    const int o = chunk * 64 + (threadIdx.x & 63);
    float w[KM];
#pragma unroll
    for (int k = 0; k < KM; ++k) w[k] = k < Kd ? W[(size_t)o * Kd + k] : 0.0f;
    const float bias = bvec[o];
    const long r0 = rg * rows_per_wave;
    const long r1 = r0 + rows_per_wave < rows ? r0 + rows_per_wave : rows;
#pragma unroll 2
    for (long r = r0; r < r1; ++r) {
        const float *xr = X + (size_t)r * Kd;  // wave-uniform address: scalar loads
        float acc = bias;
#pragma unroll
        for (int k = 0; k < KM; ++k)
            if (KT > 0 || k < Kd) acc = __builtin_fmaf(w[k], xr[k], acc);
        Y[(size_t)r * out_dim + o] = RELU ? fmaxf(acc, 0.0f) : acc;
    }
}

// ------------------------------------------------------------------------------------------
// Fused actor front end on the matrix cores: G = relu(X * W1^T + b1) * Wih^T + bih in ONE launch
// (ActorNetwork.dense1 + F.relu + the input projections of both LSTM directions), exact float32
// (v_mfma_f32_32x32x2_f32 is bit-for-bit a k-ordered fmaf chain).
//
// Everything is computed TRANSPOSED so that the hidden activations never leave registers:
//   stage 1   X1^T [64 hidden x 32 rows] = W1 [64 x D] * X^T [D x 32]      (2 tiles, D/2 k-steps)
//   stage 2   G^T  [256 units x 32 rows] = Wih [256 x 64] * X1^T [64 x 32] (8 tiles, 32 k-steps each)
// A 32x32 accumulator tile has its column (here: the data row) on the lane and its rows (hidden unit)
// in the 16 registers, which is exactly the B-operand layout of the next MFMA when that MFMA sums over
// the tile's ROW index: register `reg` of lane half h holds hidden unit hloc(reg) + 4h, so one MFMA
// k-step consumes the pair {hloc(reg), hloc(reg) + 4} and the A operand (weights) is simply fetched in
// that k order.  Weights are pre-swizzled into LDS in fragment order once per workgroup; the output tile
// is transposed through a 33-float-stride LDS patch so that G leaves as 128-byte row segments.
// One wave owns 32 data rows; a workgroup of 4 waves shares the 64 KB + 4 KB of weight fragments.
// ------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int mfma_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

// fragment order of the stage-2 weights: [8 n][2 m][4 rq][64 lane] float4, element e of lane ln =
// Wih[n*32 + (ln & 31)][m*32 + 8*rq + 4*(ln >> 5) + e]; stage 1: [2 m][S1][64 lane], lane ln of step s =
// W1[m*32 + (ln & 31)][2s + (ln >> 5)] (0 past in_dim), S1 = the k-step count rounded up to a multiple of 4 (the
// kernel is instantiated per S1 / 4 so that stage 1 is straight-line code; a zero k-step adds +0 to the
// accumulators, which never hold -0).  pw_actor_front_pack writes both once per weight update.
__global__ void pw_actor_front_pack_kernel(const float *__restrict__ w1, const float *__restrict__ wih, const int D,
                                           float *__restrict__ frag)
{
    const int S1 = ((D + 7) >> 3) * 4;
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < 8 * 2 * 4 * 64) {
        const int ln = f & 63, rq = (f >> 6) & 3, m = (f >> 8) & 1, n = f >> 9;
        const int u = n * 32 + (ln & 31), h = m * 32 + 8 * rq + 4 * (ln >> 5);
        reinterpret_cast<float4 *>(frag)[f] = *reinterpret_cast<const float4 *>(wih + (size_t)u * 64 + h);
    }
    if (f < 2 * S1 * 64) {
        const int ln = f & 63, sidx = (f >> 6) % S1, m = (f >> 6) / S1;
        const int k = 2 * sidx + (ln >> 5);
        frag[8 * 2 * 4 * 64 * 4 + f] = k < D ? w1[(size_t)(m * 32 + (ln & 31)) * D + k] : 0.0f;
    }
}

template <int S1C>
__global__ void __launch_bounds__(256) pw_actor_front_kernel(const float *__restrict__ X, const float *__restrict__ frag,
                                                             const float *__restrict__ b1, const float *__restrict__ bih,
                                                             const long rows, const int D, float *__restrict__ G)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int S1 = 4 * S1C;                                      // stage-1 k-steps (K = 2 each), zero padded
    float4 *f_wih = reinterpret_cast<float4 *>(smem_raw);            // [8 n][2 m][4 rq][64 lane] float4 (e = reg & 3)
    float *f_w1 = reinterpret_cast<float *>(f_wih + 8 * 2 * 4 * 64);  // [2 m][S1][64 lane]
    float *s_b1 = f_w1 + 2 * S1 * 64;                                // [64]
    float *s_bih = s_b1 + 64;                                        // [256]
    float *s_t = s_bih + 256;                                        // [4 waves][32][33] transpose patches

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
    // ---- weight fragments -> LDS (once per workgroup): a linear, fully coalesced copy of the packed image
    {
        const float4 *src = reinterpret_cast<const float4 *>(frag);
        const int n4 = 8 * 2 * 4 * 64 + (2 * S1 * 64) / 4;  // f_w1 follows f_wih contiguously, (2*S1*64) % 4 == 0
        for (int f = tid; f < n4; f += 256) f_wih[f] = src[f];
    }
    if (tid < 64) s_b1[tid] = b1[tid];
    s_bih[tid] = bih[tid];
    __syncthreads();

    const long row0 = ((long)blockIdx.x * 4 + wave) * 32;
    if (row0 >= rows) return;
    long myrow = row0 + col;
    const bool row_ok = myrow < rows;
    if (!row_ok) myrow = rows - 1;
    // ---- stage 1: two 32x32 tiles of X1^T, bias + ReLU applied in registers
    f32x16 acc1[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[m][r] = 0.0f;
    const float *xr = X + (size_t)myrow * D;
    float xb[S1];  // this lane's B operands of all k-steps, fetched before the first MFMA (loads in flight together)
#pragma unroll
    for (int sidx = 0; sidx < S1; ++sidx) {
        const int k = 2 * sidx + half;
        xb[sidx] = k < D ? xr[k] : 0.0f;                           // B[kk = half][j = col] = X[row][k]
    }
#pragma unroll
    for (int sidx = 0; sidx < S1; ++sidx) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
            acc1[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(f_w1[(m * S1 + sidx) * 64 + lane], xb[sidx], acc1[m], 0, 0, 0);
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[m][r] = fmaxf(acc1[m][r] + s_b1[m * 32 + mfma_row(r, half)], 0.0f);

    // ---- stage 2: eight 32x32 tiles of G^T, one at a time (acc1 stays resident as the B operands)
    float *patch = s_t + wave * 32 * 33;
#pragma unroll 1
    for (int n = 0; n < 8; ++n) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const float4 a = f_wih[((n * 2 + m) * 4 + rq) * 64 + lane];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, acc1[m][4 * rq + 0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, acc1[m][4 * rq + 1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, acc1[m][4 * rq + 2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, acc1[m][4 * rq + 3], acc, 0, 0, 0);
            }
        }
        // transpose: lane (col = data row, half) holds units mfma_row(r, half); patch[row][unit], stride 33
#pragma unroll
        for (int r = 0; r < 16; ++r) patch[col * 33 + mfma_row(r, half)] = acc[r];
        wave_lds_sync();
        // each lane emits 4 float4 = 16 consecutive units of one row: rows (lane >> 1) and halves (lane & 1)
        {
            const int rr = lane >> 1, u0 = (lane & 1) * 16;
            const long orow = row0 + rr;
            if (orow < rows) {
                float4 *dst = reinterpret_cast<float4 *>(G + (size_t)orow * 256 + n * 32 + u0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float *src = patch + rr * 33 + u0 + 4 * q;
                    const float *bb = s_bih + n * 32 + u0 + 4 * q;
                    dst[q] = make_float4(src[0] + bb[0], src[1] + bb[1], src[2] + bb[2], src[3] + bb[3]);
                }
            }
        }
        wave_lds_sync();
    }
}

// Output head: logits = H * W2^T + b2 (64 -> 5) for one (env, agent) row per lane, then the hard
// Gumbel-softmax sample of ddpg_gumbel_fix.py:109-116 as argmax(logits + g), g = -log(-log(u)),
// u from Philox4x32-10 keyed (seed; step, row) -- the action stays an int32 index in HBM.
__global__ void __launch_bounds__(256) pw_actor_head_kernel(const float *__restrict__ H, const float *__restrict__ w2,
                                                            const float *__restrict__ b2, const long rows,
                                                            const uint64_t seed, uint64_t step,
                                                            const int64_t *__restrict__ step_dev,
                                                            float *__restrict__ logits, int32_t *__restrict__ act)
{
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    if (step_dev) step = (uint64_t)*step_dev;
    float acc[5];
#pragma unroll
    for (int o = 0; o < 5; ++o) acc[o] = b2[o];
    const float4 *h4 = reinterpret_cast<const float4 *>(H + (size_t)r * 64);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const float4 hv = h4[q];
#pragma unroll
        for (int o = 0; o < 5; ++o) {  // w2 indices are uniform: scalar loads
            acc[o] = __builtin_fmaf(w2[o * 64 + 4 * q], hv.x, acc[o]);
            acc[o] = __builtin_fmaf(w2[o * 64 + 4 * q + 1], hv.y, acc[o]);
            acc[o] = __builtin_fmaf(w2[o * 64 + 4 * q + 2], hv.z, acc[o]);
            acc[o] = __builtin_fmaf(w2[o * 64 + 4 * q + 3], hv.w, acc[o]);
        }
    }
    if (logits) {
#pragma unroll
        for (int o = 0; o < 5; ++o) logits[(size_t)r * 5 + o] = acc[o];
    }
    if (act) {
        uint32_t u[8];
        pw_philox4x32_10((uint32_t)r, (uint32_t)((uint64_t)r >> 32), (uint32_t)step, (uint32_t)(step >> 32),
                         (uint32_t)seed, (uint32_t)(seed >> 32), u);
        pw_philox4x32_10((uint32_t)r, (uint32_t)((uint64_t)r >> 32) | 0x80000000u, (uint32_t)step, (uint32_t)(step >> 32),
                         (uint32_t)seed, (uint32_t)(seed >> 32), u + 4);
        int best = 0;
        float bv = 0.0f;
#pragma unroll
        for (int o = 0; o < 5; ++o) {
            const float uo = ((float)(u[o] >> 8) + 0.5f) * 5.9604644775390625e-8f;  // (0, 1)
            const float v = acc[o] - __logf(-__logf(uo));
            if (o == 0 || v > bv) { bv = v; best = o; }
        }
        act[r] = best;
    }
}

// Test hook: evaluate one device math primitive element-wise so that tests can compare the exact bits
// against the CPU contract (include/pworld_math.h, restated in oracle/pworld_oracle.c) over millions of
// inputs.  fn: 0 sqrt_rn_fast, 1 softplus_branchless, 2 pw_softplus, 3 pw_exp, 4 sqrtf, 5 x / aux (IEEE)
__global__ void pw_debug_math_kernel(const int fn, const float *x, const float aux, float *y, const long n)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    float r;
    switch (fn) {
    case 0: r = sqrt_rn_fast(v); break;
    case 1: r = softplus_branchless(v); break;
    case 2: r = pw_softplus(v); break;
    case 3: r = pw_exp(v); break;
    case 4: r = sqrtf(v); break;
    default: r = v / aux; break;
    }
    y[i] = r;
}

// Episode bookkeeping of the rollout loop (experiments/run.py:55-65, vectorised): return += shared
// reward; on terminal the return is added to (sum, count) and cleared.  ONE workgroup with a
// fixed-order tree reduction, so the statistics are bit-reproducible (no float atomics).
__global__ void __launch_bounds__(1024) pw_episode_stats_kernel(const float *rew_shared, const uint8_t *terminal,
                                                                const int B, float *episode_return,
                                                                double *finished_sum, int64_t *finished_count)
{
    __shared__ double s_sum[1024];
    __shared__ int s_cnt[1024];
    double acc = 0.0;
    int cnt = 0;
    for (int e = threadIdx.x; e < B; e += 1024) {
        const float r = episode_return[e] + rew_shared[e];
        if (terminal[e]) { acc += (double)r; cnt += 1; episode_return[e] = 0.0f; }
        else episode_return[e] = r;
    }
    s_sum[threadIdx.x] = acc;
    s_cnt[threadIdx.x] = cnt;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            s_sum[threadIdx.x] += s_sum[threadIdx.x + w];
            s_cnt[threadIdx.x] += s_cnt[threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *finished_sum += s_sum[0];
        *finished_count += s_cnt[0];
    }
}

}  // namespace
