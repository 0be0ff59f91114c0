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
// tanh(x) = 2 / (1 + exp(-2x)) - 1: five instructions (mul, exp2, add, rcp, fma) instead of the nine of (1 - e) / (1 + e) on |x| with the
// sign copied back -- the cell update is vector work that cannot overlap the exact-f32 matrix instructions, so every instruction of it
// is on the timestep's path (round 5: 2 % of a step).  x -> -inf: exp = inf, rcp = 0, result -1; x -> +inf: exp = 0, result 1; no NaN from
// finite input.  Same absolute accuracy as the other form (both are limited by the rounding of a number near 1: ~1e-7).
__device__ __forceinline__ float fast_tanh(float x)
{
    return fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -2.8853900817779268f)), -1.0f);   // exp(-2x) = 2^(-2 log2(e) x): one multiply
}

// One LSTM cell (PyTorch's gate order i, f, g, o): pre-activations -> new cell state c and output h.  Every kernel form calls this
// one function, so the forms agree bit for bit whatever the activations' rounding is.
// (The four gates' exponent arguments and denominators are formed two at a time -- (i, f) and (g, o) sit in adjacent accumulator registers --
// so that they compile to packed multiplies / adds; the operations and their bits are those of fast_sigmoid / fast_tanh.)
__device__ __forceinline__ void lstm_cell(const float gi, const float gf, const float gg, const float go, float &c, float &h)
{
    typedef float v2 __attribute__((ext_vector_type(2)));
    const v2 a = v2{gi, gf} * v2{-1.4426950408889634f, -1.4426950408889634f};   // exp(-x) = 2^(-log2(e) x)
    const v2 b = v2{gg, go} * v2{-2.8853900817779268f, -1.4426950408889634f};   // tanh's exp(-2x) for g
    const v2 d1 = v2{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)} + v2{1.0f, 1.0f};
    const v2 d2 = v2{__builtin_amdgcn_exp2f(b.x), __builtin_amdgcn_exp2f(b.y)} + v2{1.0f, 1.0f};
    const float si = __builtin_amdgcn_rcpf(d1.x), sf = __builtin_amdgcn_rcpf(d1.y), so = __builtin_amdgcn_rcpf(d2.y);
    const float tg = fmaf(2.0f, __builtin_amdgcn_rcpf(d2.x), -1.0f);
    c = sf * c + si * tg;
    h = so * fast_tanh(c);
}

// One hidden unit's four W_hh rows, gate pairs (i, f) and (g, o) packed so that a recurrence step is 64
// v_pk_fma_f32 (two gates per instruction, h broadcast through op_sel) instead of 128 scalar FMAs.  Each
// gate still accumulates over k in ascending order with fused multiply-adds: the same bits either way.
struct LstmUnitW {
    f32x2 wif[32], wgo[32];
};

// sw: one direction's W_hh staged as [gate][k/4][unit] float4
__device__ __forceinline__ void lstm_load_unit(const float4 *sw, const int j, LstmUnitW &w)
{
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float4 a = sw[(0 * 8 + q) * 32 + j], b = sw[(1 * 8 + q) * 32 + j];
        const float4 c = sw[(2 * 8 + q) * 32 + j], d = sw[(3 * 8 + q) * 32 + j];
        w.wif[4 * q] = f32x2{a.x, b.x}; w.wif[4 * q + 1] = f32x2{a.y, b.y};
        w.wif[4 * q + 2] = f32x2{a.z, b.z}; w.wif[4 * q + 3] = f32x2{a.w, b.w};
        w.wgo[4 * q] = f32x2{c.x, d.x}; w.wgo[4 * q + 1] = f32x2{c.y, d.y};
        w.wgo[4 * q + 2] = f32x2{c.z, d.z}; w.wgo[4 * q + 3] = f32x2{c.w, d.w};
    }
}

// gates = pre-activations (x W_ih^T + biases) of this unit; hs = the sequence's previous h [32] in LDS
__device__ __forceinline__ void lstm_step(const LstmUnitW &w, const float *hs, float gi, float gf, float gg, float go,
                                          float &h, float &c)
{
    f32x2 aif = {gi, gf}, ago = {gg, go};
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float4 hv = reinterpret_cast<const float4 *>(hs)[q];
        aif = __builtin_elementwise_fma(w.wif[4 * q], f32x2{hv.x, hv.x}, aif);
        ago = __builtin_elementwise_fma(w.wgo[4 * q], f32x2{hv.x, hv.x}, ago);
        aif = __builtin_elementwise_fma(w.wif[4 * q + 1], f32x2{hv.y, hv.y}, aif);
        ago = __builtin_elementwise_fma(w.wgo[4 * q + 1], f32x2{hv.y, hv.y}, ago);
        aif = __builtin_elementwise_fma(w.wif[4 * q + 2], f32x2{hv.z, hv.z}, aif);
        ago = __builtin_elementwise_fma(w.wgo[4 * q + 2], f32x2{hv.z, hv.z}, ago);
        aif = __builtin_elementwise_fma(w.wif[4 * q + 3], f32x2{hv.w, hv.w}, aif);
        ago = __builtin_elementwise_fma(w.wgo[4 * q + 3], f32x2{hv.w, hv.w}, ago);
    }
    wave_lds_sync();  // all reads of h done before the caller overwrites it
    lstm_cell(aif.x, aif.y, ago.x, ago.y, c, h);
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
    LstmUnitW w;
    lstm_load_unit(s_w + dir * 1024, j, w);
    float h = 0.0f, c = 0.0f;
    float *hs = s_h + grp * 32;
    const float *g0 = G + (((size_t)env * N + (dir ? N - 1 : 0)) * 2 + dir) * 128;
    float ni = g0[j], nf = g0[32 + j], ng = g0[64 + j], no = g0[96 + j];
    for (int s = 0; s < N; ++s) {
        const int t = dir ? N - 1 - s : s;
        const float ai = ni, af = nf, ag = ng, ao = no;
        if (s + 1 < N) {  // prefetch the next timestep's pre-activations under this step's FMAs
            const float *g = G + (((size_t)env * N + (dir ? t - 1 : t + 1)) * 2 + dir) * 128;
            ni = g[j]; nf = g[32 + j]; ng = g[64 + j]; no = g[96 + j];
        }
        hs[j] = h;
        wave_lds_sync();  // a sequence's 32 lanes sit in one wave
        lstm_step(w, hs, ai, af, ag, ao, h, c);
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
// accumulators, which never hold -0).  Third section (pw_kernels_actor16.hpp): the same W_ih as A fragments of
// v_mfma_f32_16x16x4_f32, [8 wave][2 tile][4 j][64 lane] float4: element e of lane ln of wave (dir, hq) = Wih[dir * 128 +
// (ln & 3) * 32 + hq * 8 + 4 tile + ((ln & 15) >> 2)][x1_kpos(4 * (4 j + e) + ln / 16)] -- tile rows ordered (unit, gate), k in the
// order the 32x32x2 chains sum.  pw_actor_front_pack writes all three once per weight update.
// hidden index of position p (0..63) in the summation order of the input projection: the 32x32x2 chains walk (m, rq, e) and sum
// the pair {m * 32 + 8 rq + e, + 4} per instruction
__host__ __device__ __forceinline__ int x1_kpos(int p) { return ((p >> 5) << 5) + (((p >> 3) & 3) << 3) + ((p >> 1) & 3) + ((p & 1) << 2); }
__host__ __device__ inline size_t actor_frag16_offset(int S1) { return (size_t)8 * 2 * 4 * 64 * 4 + (size_t)2 * S1 * 64; }
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
    if (f < 8 * 2 * 4 * 64) {
        const int ln = f & 63, jx = (f >> 6) & 3, T = (f >> 8) & 1, wv = f >> 9;
        const int n16 = ln & 15, kq = ln >> 4;
        const size_t R = (size_t)(wv >> 2) * 128 + (n16 & 3) * 32 + (wv & 3) * 8 + 4 * T + (n16 >> 2);
        float4 v;
        v.x = wih[R * 64 + x1_kpos(4 * (4 * jx + 0) + kq)];
        v.y = wih[R * 64 + x1_kpos(4 * (4 * jx + 1) + kq)];
        v.z = wih[R * 64 + x1_kpos(4 * (4 * jx + 2) + kq)];
        v.w = wih[R * 64 + x1_kpos(4 * (4 * jx + 3) + kq)];
        reinterpret_cast<float4 *>(frag + actor_frag16_offset(S1))[f] = v;
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

// ------------------------------------------------------------------------------------------
// The whole actor in ONE launch: obs [B,N,D] -> Gumbel-sampled action index [B,N] (and/or logits, H).
// A workgroup of 8 waves owns E = min(16, 96 / N) environments (R = E*N <= 96 rows = up to three 32-row MFMA
// tiles) and nothing but the observation rows and the outputs touches HBM:
//   stage 1   X1^T = relu(W1 X^T + b1) on the matrix cores, kept in registers (each wave: its own row tile)
//   per direction d (forward, reverse):
//     fill    W_ih(d) fragments (32 KB) + W_hh(d) (16 KB) -> LDS
//     stage 2 G(d)^T = W_ih(d) X1^T + b on the matrix cores -> LDS tile Gs [R][128] (row stride 129)
//     LSTM    lane = (sequence, hidden unit): 8 waves x 2 sequences, the unit's 128 W_hh weights in VGPRs,
//             N recurrence steps of 64 packed FMAs; relu(h) -> LDS tile Hs [R][64] (row stride 68)
//   head      thread = (row, logit): 64 FMAs, Gumbel noise from Philox4x32-10 keyed (seed; step, row) exactly
//             as pw_actor_head_kernel; argmax per row and head (one head of <= 16 logits, or the two heads of a
//             MultiDiscrete actor, main.py:52-54: act [rows, 2]).
// Same arithmetic, in the same order, as pw_actor_front_kernel + pw_bilstm_kernel + pw_actor_head_kernel
// (the tests demand identical bits); what is removed is the HBM round trip of G (1 KB per row, twice), of H,
// and two launches.  LDS: 6-16 KB stage-1 fragments + 32 + 16 + 48.4 (Gs) + 25.5 (Hs) + 7 KB small = <= 145 KB.
// ------------------------------------------------------------------------------------------
struct ActorFusedArgs {
    const float *X, *frag, *b1, *bih, *whh_f, *whh_r, *w2, *b2;
    int B, N, D, E, relu_out;
    int n_out0, n_out1;  // logits of head 0 / head 1 (0 = single head); n_out0 + n_out1 <= 16
    int bf16x3;          // opt-in, not exact: input projection on bf16 MFMAs, three products per k step (16x16x4-core kernels only)
    uint64_t seed, step;
    const int64_t *step_dev;
    float *H, *logits;
    int32_t *act;
};
constexpr int kFusedRows = 96, kGs = 129, kHs = 68;  // kHs: 16-byte aligned rows for the head's float4 reads

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also carries a workgroup-scope fence for GLOBAL
// memory, i.e. an s_waitcnt vmcnt(0): every wave would sit out the full HBM latency of its outstanding stores at
// each of the ~8 barriers of a pass.  Inside these kernels waves hand data to each other through LDS alone, and
// what they store to global memory is only read after the kernel (or behind an explicit __threadfence).
__device__ __forceinline__ void wg_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// LDS of one actor workgroup (carved from dynamic shared memory; kActorLdsFloats(S1) floats in total)
struct ActorLds {
    float4 *f_wih;  // [4 n][2 m][4 rq][64 lane] float4, one direction
    float4 *s_whh;  // [4 gate][8 q][32 unit] float4, one direction
    float *f_w1;    // [2 m][S1][64 lane]
    float *s_g;     // [96][129]
    float *s_hid;   // [96][68]
    float *s_b1, *s_bih;  // [64], [256]
    float *s_w2, *s_b2;   // [16][64] (n_out0 + n_out1 rows used), [16]
    float *s_hx;    // [16 sequences][32]
    float *s_lg;    // [96 * 16] perturbed logits -- aliases s_g, which is dead once the last recurrence has finished
    unsigned char *end;
};
__host__ __device__ constexpr size_t actor_lds_bytes(int S1)
{
    return (size_t)(4 * 2 * 4 * 64 + 4 * 8 * 32) * 16 +
           (size_t)(2 * S1 * 64 + kFusedRows * kGs + kFusedRows * kHs + 64 + 256 + 1024 + 16 + 512) * 4;
}
__device__ __forceinline__ ActorLds actor_carve(unsigned char *raw, const int S1)
{
    ActorLds S;
    S.f_wih = reinterpret_cast<float4 *>(raw);
    S.s_whh = S.f_wih + 4 * 2 * 4 * 64;
    S.f_w1 = reinterpret_cast<float *>(S.s_whh + 4 * 8 * 32);
    S.s_g = S.f_w1 + 2 * S1 * 64;
    S.s_hid = S.s_g + kFusedRows * kGs;
    S.s_b1 = S.s_hid + kFusedRows * kHs;
    S.s_bih = S.s_b1 + 64;
    S.s_w2 = S.s_bih + 256;
    S.s_b2 = S.s_w2 + 1024;
    S.s_hx = S.s_b2 + 16;
    S.s_lg = S.s_g;
    S.end = reinterpret_cast<unsigned char *>(S.s_hx + 512);
    return S;
}

// one direction's weights -> LDS: W_ih fragments (32 KB) + W_hh (16 KB, re-laid as [gate][k/4][unit]); called by
// `nthr` threads numbered t0 = 0 .. nthr - 1
__device__ __forceinline__ void actor_fill_dir(const ActorFusedArgs &A, const ActorLds &S, const int dir, const int t0,
                                               const int nthr)
{
    const float4 *src = reinterpret_cast<const float4 *>(A.frag) + dir * 2048;
    for (int f = t0; f < 2048; f += nthr) S.f_wih[f] = src[f];
    const float4 *wh = reinterpret_cast<const float4 *>(dir ? A.whh_r : A.whh_f);
    for (int f = t0; f < 1024; f += nthr) {
        const int row = f >> 3, q = f & 7, gate = row >> 5, unit = row & 31;
        S.s_whh[(gate * 8 + q) * 32 + unit] = wh[f];
    }
}

// One forward pass of the actor for the rows of this workgroup (all 512 threads call it).
//   xrows      observation rows of this workgroup, row-major [rows_here][A.D] -- global memory or LDS
//   load_const stage-1 fragments, biases and head weights -> LDS (needed once per kernel)
//   fill0      fetch the forward direction's weights here (false: the caller already did, behind a barrier)
//   step       Philox step of the Gumbel noise
//   act_g      global sink of the sampled indices [rows_here * nheads] (or NULL), act_l the same in LDS (or NULL)
// On return every thread has passed a barrier after the last LDS write of the pass.
template <int S1C>
__device__ __forceinline__ void actor_forward_wg(const ActorFusedArgs &A, const ActorLds &S, const float *xrows,
                                                 const int rows_here, const int envs_here, const long row_base,
                                                 const bool load_const, const bool fill0, const uint64_t step,
                                                 int32_t *act_g, int32_t *act_l)
{
    constexpr int S1 = 4 * S1C;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, col = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = A.N;
    PW_STAMP_DECL;
    PW_STAMP_START;

    // Matrix-core job map: 3 row tiles x 4 unit tiles per direction = 12 jobs over 4 SIMDs (waves w and w + 4
    // share one): waves 0-3 take two unit tiles of row tiles 0 / 1, waves 4-7 one unit tile of row tile 2, so
    // every SIMD carries three jobs.  Each wave needs stage 1 of its own row tile only.
    const int rt = wave < 4 ? wave >> 1 : 2;
    const int n_lo = wave < 4 ? (wave & 1) * 2 : wave - 4, n_cnt = wave < 4 ? 2 : 1;
    const bool front = rt * 32 < rows_here;  // wave-uniform
    float xb[S1];  // this lane's stage-1 B operands, requested before the weights so that the loads overlap the fill
    {
        int lr = rt * 32 + col;
        if (lr >= rows_here) lr = rows_here - 1;
        const float *xr = xrows + (size_t)lr * A.D;
#pragma unroll
        for (int sidx = 0; sidx < S1; ++sidx) {
            const int k = 2 * sidx + half;
            xb[sidx] = k < A.D ? xr[k] : 0.0f;
        }
    }
    if (load_const) {  // stage-1 fragments, the small vectors
        const float4 *src = reinterpret_cast<const float4 *>(A.frag + 8 * 2 * 4 * 64 * 4);
        for (int f = tid; f < (2 * S1 * 64) / 4; f += 512) reinterpret_cast<float4 *>(S.f_w1)[f] = src[f];
        if (tid < 64) S.s_b1[tid] = A.b1[tid];
        if (tid < 256) S.s_bih[tid] = A.bih[tid];
        const int OUTc = A.n_out0 + A.n_out1;
        for (int f = tid; f < OUTc * 64; f += 512) S.s_w2[f] = A.w2[f];
        if (tid < OUTc) S.s_b2[tid] = A.b2[tid];
    }
    if (fill0) actor_fill_dir(A, S, 0, tid, 512);
    if (fill0 || load_const) wg_lds_barrier();
    PW_STAMP(0);

    // ---- stage 1
    f32x16 acc1[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[m][r] = 0.0f;
    if (front) {
#pragma unroll
        for (int sidx = 0; sidx < S1; ++sidx) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
                acc1[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(S.f_w1[(m * S1 + sidx) * 64 + lane], xb[sidx], acc1[m], 0, 0, 0);
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[m][r] = fmaxf(acc1[m][r] + S.s_b1[m * 32 + mfma_row(r, half)], 0.0f);
    }
    PW_STAMP(1);
    // ---- LSTM lane identity: 2 sequences per wave, sequence slot = local environment
    const int j = lane & 31, sl = wave * 2 + (lane >> 5);
    const bool seq_ok = sl < envs_here;
    const int se = seq_ok ? sl : 0;  // idle half-waves shadow environment 0 (they write nothing)
    float *hs = S.s_hx + sl * 32;

#pragma unroll 1
    for (int dir = 0; dir < 2; ++dir) {
        if (dir == 1) {
            actor_fill_dir(A, S, 1, tid, 512);
            wg_lds_barrier();
        }
        PW_STAMP(2);
        if (front) {
#pragma unroll 1
            for (int nn = 0; nn < n_cnt; ++nn) {
                const int n = n_lo + nn;
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        const float4 a = S.f_wih[((n * 2 + m) * 4 + rq) * 64 + lane];
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, acc1[m][4 * rq + 0], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, acc1[m][4 * rq + 1], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, acc1[m][4 * rq + 2], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, acc1[m][4 * rq + 3], acc, 0, 0, 0);
                    }
                }
                float *dst = S.s_g + (rt * 32 + col) * kGs + n * 32;
                const float *bb = S.s_bih + dir * 128 + n * 32;
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[mfma_row(r, half)] = acc[r] + bb[mfma_row(r, half)];
            }
        }
        PW_STAMP(3);
        wg_lds_barrier();
        PW_STAMP(4);
        {   // recurrence over the agent axis
            LstmUnitW w;
            lstm_load_unit(S.s_whh, j, w);
            float h = 0.0f, c = 0.0f;
            for (int s = 0; s < N; ++s) {
                const int t = dir ? N - 1 - s : s;
                const float *g = S.s_g + (se * N + t) * kGs;
                const float gi = g[j], gf = g[32 + j], gg = g[64 + j], go = g[96 + j];
                hs[j] = h;
                wave_lds_sync();
                lstm_step(w, hs, gi, gf, gg, go, h, c);
                if (seq_ok) S.s_hid[(se * N + t) * kHs + dir * 32 + j] = A.relu_out ? fmaxf(h, 0.0f) : h;
            }
        }
        PW_STAMP(5);
        wg_lds_barrier();  // Gs / weights are overwritten by the next direction; Hs complete after the last
        PW_STAMP(6);
    }

    // ---- head
    if (A.H) {
        for (int idx = tid; idx < rows_here * 64; idx += 512)
            A.H[(size_t)row_base * 64 + idx] = S.s_hid[(idx >> 6) * kHs + (idx & 63)];
    }
    // thread = (row, logit); the heads' logits are concatenated ([n_out0 | n_out1], run.py:39-41 order)
    const int OUT = A.n_out0 + A.n_out1, nheads = A.n_out1 > 0 ? 2 : 1;
    const bool sample = act_g != nullptr || act_l != nullptr;
    for (int idx = tid; idx < rows_here * OUT; idx += 512) {
        const int r = idx / OUT, o = idx - r * OUT;
        float acc = S.s_b2[o];
        const float4 *hv = reinterpret_cast<const float4 *>(S.s_hid + r * kHs), *wv = reinterpret_cast<const float4 *>(S.s_w2 + o * 64);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float4 hq = hv[q], wq = wv[q];
            acc = __builtin_fmaf(wq.x, hq.x, acc);
            acc = __builtin_fmaf(wq.y, hq.y, acc);
            acc = __builtin_fmaf(wq.z, hq.z, acc);
            acc = __builtin_fmaf(wq.w, hq.w, acc);
        }
        const long grow = row_base + r;
        if (A.logits) A.logits[(size_t)grow * OUT + o] = acc;
        if (sample) {
            // uniform o of a row = word (o & 3) of Philox block (o >> 2); the block index sits in the two top bits
            // of counter word 1 (block 1 = 0x80000000, as pw_actor_head_kernel's second call)
            const uint32_t blk = (uint32_t)o >> 2, tag = ((blk & 1u) << 31) | ((blk >> 1) << 30);
            uint32_t u[4];
            pw_philox4x32_10((uint32_t)grow, (uint32_t)((uint64_t)grow >> 32) | tag, (uint32_t)step, (uint32_t)(step >> 32),
                             (uint32_t)A.seed, (uint32_t)(A.seed >> 32), u);
            const int w = o & 3;
            const uint32_t uw = w == 0 ? u[0] : w == 1 ? u[1] : w == 2 ? u[2] : u[3];
            const float uo = ((float)(uw >> 8) + 0.5f) * 5.9604644775390625e-8f;  // (0, 1)
            S.s_lg[idx] = acc - __logf(-__logf(uo));
        }
    }
    if (sample) {
        wg_lds_barrier();
        for (int idx = tid; idx < rows_here * nheads; idx += 512) {
            const int r = idx / nheads, hd = idx - r * nheads;
            const int lo = hd ? A.n_out0 : 0, cnt = hd ? A.n_out1 : A.n_out0;
            const float *v = S.s_lg + r * OUT + lo;
            int best = 0;
            float bv = v[0];
            for (int o = 1; o < cnt; ++o)
                if (v[o] > bv) { bv = v[o]; best = o; }
            if (act_g) act_g[idx] = best;
            if (act_l) act_l[idx] = best;
        }
    }
    PW_STAMP(7);
    PW_STAMP_FLUSH;
}

template <int S1C>
__global__ void __launch_bounds__(512) pw_actor_fused_kernel(const ActorFusedArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const ActorLds S = actor_carve(smem_raw, 4 * S1C);
    const int N = A.N, E = A.E;
    const long env0 = (long)blockIdx.x * E;
    const int envs_here = (int)((long)A.B - env0 < (long)E ? (long)A.B - env0 : (long)E);
    const long row_base = env0 * N;
    const int nheads = A.n_out1 > 0 ? 2 : 1;
    const uint64_t step = (A.act && A.step_dev) ? (uint64_t)*A.step_dev : A.step;
    actor_forward_wg<S1C>(A, S, A.X + (size_t)row_base * A.D, envs_here * N, envs_here, row_base, true, true, step,
                          A.act ? A.act + row_base * nheads : nullptr, nullptr);
}

// ------------------------------------------------------------------------------------------
// Policy-in-the-loop rollout as ONE launch: T x (actor forward + Gumbel sampling + environment step) with the
// observations, the sampled actions and the world state of a workgroup's 16 environments never leaving the
// CU between steps (pw_kernels_policy2.hpp, pw_kernels_policy3.hpp, pw_kernels_policy3j.hpp hold the kernels).  The
// environment lanes advance their envs exactly as pw_spread_stream_kernel does (lane = (env, agent); same expressions,
// same order, same bits) and write the step's outputs -- and the next observation rows (or states) back into LDS.  HBM
// sees the outputs of a step once; there is no launch and no kernel boundary between steps.
// simple_spread fast-path configurations (local observation, homogeneous agents, L <= N), one 5-logit head.
// ------------------------------------------------------------------------------------------
struct PolicyRolloutArgs {
    ActorFusedArgs A;   // weights, B, N, D, E, heads, seed, step / step_dev (Philox step of the FIRST pass)
    StreamParams V;     // world constants, state planes, outputs (V.act unused)
    int T;
    int32_t *act_out;   // [T,B,N] sampled action indices (or NULL)
    // optional direct sink: the transitions go straight into the replay ring (slot (ring_start + t*B + env) %
    // capacity, the order of T pw_replay_add calls) and the episode returns are kept here, so no second launch
    // has to re-read the step outputs (which then may all be NULL)
    pw_replay_store ring;
    int has_ring;
    int64_t ring_start;
    float *episode_return;          // [B] running return per env (or NULL)
    double *finished_sum;
    int64_t *finished_count;
    unsigned long long *scratch;    // [2 * gridDim.x + 1] words, zero before first use
    int NP;                         // just-in-time form: row stride of an environment in LDS (>= N; odd: see pw_kernels_policy3j.hpp)
};

// (The first, phase-by-phase form of the rollout kernel -- pw_policy_rollout_kernel: actor_forward_wg per step, the environment
// on the first waves -- was retired in round 4: since the third form and its just-in-time variant nothing selected it
// automatically, and the forms that run are each compared with the CPU oracle directly.)

// Test hook: evaluate one device math primitive element-wise so that tests can compare the exact bits
// against the CPU contract (include/pworld_math.h, restated in oracle/pworld_oracle.c) over millions of
// inputs.  fn: 0 sqrt_rn_fast, 1 softplus_branchless, 2 pw_softplus, 3 pw_exp, 4 sqrtf, 5 x / aux (IEEE),
// 6 div_chain(x, aux), 7 div_chain(aux, x) (the scaling-free division of the hot loops), 8 softplus_branchless (kept: the round-2 chain form had its own number),
// 9 aux / x (IEEE), 10 div_chain1(x, aux) (one correction; exact only for divisors pw_margin_one_correction accepts),
// 11 sqrt_rn_core(x) (the hot loops' one-correction sqrt; x in [2^-90, 2^90)), 12 sqrt_rn_core_tests(x) (the two-test form)
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
    case 6: r = div_chain(v, aux, div_refined_rcp(aux)); break;
    case 7: r = div_chain(aux, v, div_refined_rcp(v)); break;
    case 8: r = softplus_branchless(v); break;
    case 9: r = aux / v; break;
    case 10: r = div_chain1(v, aux, div_refined_rcp(aux)); break;
    case 11: r = sqrt_rn_core(v); break;
    case 12: r = sqrt_rn_core_tests(v); break;
    default: r = v / aux; break;
    }
    y[i] = r;
}

// Episode bookkeeping of the rollout loop (experiments/run.py:55-65, vectorised): return += shared
// reward; on terminal the return is added to (sum, count) and cleared.  ONE workgroup with a
// fixed-order tree reduction, so the statistics are bit-reproducible (no float atomics).
// The same launch can advance up to two device-side counters (replay ring cursor, Philox step) once every
// earlier kernel of the step has read them: stream order puts this launch last in a rollout step.
struct TailCounters {
    int64_t *c0, *c1;
    int64_t d0, m0, d1, m1;
};
__global__ void __launch_bounds__(1024) pw_episode_stats_kernel(const float *rew_shared, const uint8_t *terminal,
                                                                const int B, float *episode_return,
                                                                double *finished_sum, int64_t *finished_count,
                                                                const TailCounters tc)
{
    if (threadIdx.x == 0) {
        if (tc.c0) { int64_t v = *tc.c0 + tc.d0; if (tc.m0 > 0) v %= tc.m0; *tc.c0 = v; }
        if (tc.c1) { int64_t v = *tc.c1 + tc.d1; if (tc.m1 > 0) v %= tc.m1; *tc.c1 = v; }
    }
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
