// pw_kernels_actor16.hpp -- part of libpworld.so (translation unit csrc/pworld_policy.hip includes it).
// One forward pass of the actor (dense1 + ReLU, BiLSTM over the agent axis, head(s) + Gumbel arg-max) for the <= 16
// environments of a workgroup with the whole BiLSTM on v_mfma_f32_16x16x4_f32 -- the core of the third rollout form
// (pw_kernels_policy3.hpp explains the design) as a function, for the kernels that run an actor pass between other work:
// pw_actor_fused16_kernel (one pass per launch), the simple_reference and simple_tag one-launch rollouts.
#pragma once

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------
// Opt-in, NOT exact (ActorFusedArgs.bf16x3; pw_set_actor_precision; never the default, never a headline):
// the input projection W_ih * x1 -- two thirds of the pass's matrix time -- on v_mfma_f32_16x16x16_bf16 with both operands split
// into bfloat16 high and low parts and three products per k step (lo*hi + hi*lo + hi*hi, f32 accumulate; the lo*lo term and the
// parts' rounding are ~2^-16 relative).  dense1, the recurrence and the head stay exact f32.  tests/test_gpu_engine.py holds the
// mode to the 2e-5 bound of the PyTorch comparison on the reference's own weights; it does NOT reproduce the exact form's bits.
// Fragment layout of the 16x16x16 form: a lane supplies four consecutive k (16 s + 4 kq .. + 3) of its row / column per k step.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t bf16_rn(float x)
{
    const uint32_t u = __float_as_uint(x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;  // round to nearest even (finite inputs)
}
__device__ __forceinline__ void bf16_split4(const float a, const float b, const float c, const float d, u32x2 &hi, u32x2 &lo)
{
    const uint32_t ha = bf16_rn(a), hb = bf16_rn(b), hc = bf16_rn(c), hd = bf16_rn(d);
    const uint32_t la = bf16_rn(a - __uint_as_float(ha << 16)), lb = bf16_rn(b - __uint_as_float(hb << 16));
    const uint32_t lc = bf16_rn(c - __uint_as_float(hc << 16)), ld = bf16_rn(d - __uint_as_float(hd << 16));
    hi = u32x2{ha | (hb << 16), hc | (hd << 16)};
    lo = u32x2{la | (lb << 16), lc | (ld << 16)};
}

// A fragments of this wave's two W_ih tiles: exact form out of the third section of pw_actor_front_pack's image (k in the
// projection's summation order); bf16x3 form: natural k, four consecutive k per k step = one float4 of the first section
template <int S1, bool BF3>
__device__ __forceinline__ void actor16_load_ih(const float *frag, const int wave, const int lane,
                                                float (&aih)[2][16], u32x2 (&ah)[2][4], u32x2 (&al)[2][4])
{
    const int dir = wave >> 2, hq = wave & 3, n16 = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int T = 0; T < 2; ++T) {
        if (!BF3) {
            const float4 *f16 = reinterpret_cast<const float4 *>(frag + actor_frag16_offset(S1)) + ((wave * 2 + T) * 4) * 64 + lane;
#pragma unroll
            for (int jx = 0; jx < 4; ++jx) {
                const float4 q = f16[jx * 64];
                aih[T][4 * jx + 0] = q.x; aih[T][4 * jx + 1] = q.y; aih[T][4 * jx + 2] = q.z; aih[T][4 * jx + 3] = q.w;
            }
        } else {
            const int R = dir * 128 + (n16 & 3) * 32 + hq * 8 + 4 * T + (n16 >> 2);  // row of W_ih [256][64]
#pragma unroll
            for (int sx = 0; sx < 4; ++sx) {
                const int k0 = 16 * sx + 4 * kq, kl = k0 & 31;
                // W_ih[R][k0 .. k0 + 3] in the 32x32x2 fragment array: [8 n][2 m][4 rq][64 lane] float4
                const float4 q = reinterpret_cast<const float4 *>(frag)[(((R >> 5) * 2 + (k0 >> 5)) * 4 + (kl >> 3)) * 64 + (R & 31) + 32 * ((kl >> 2) & 1)];
                bf16_split4(q.x, q.y, q.z, q.w, ah[T][sx], al[T][sx]);
            }
        }
    }
}

// dense1 block (32-column tile rt, hidden half m) -> LDS in the B-fragment order of the timestep loop.  Register q of lane half
// `half` is hidden unit h = m * 32 + (q & 3) + 8 (q >> 2) + 4 half.  Exact form: position 32 m + 2 q + half of the projection's
// summation order -> fragment j = 2 m + q / 8, element (q / 2) % 4, lane (2 (q & 1) + half) * 16 + sequence.  bf16x3 form: natural k,
// registers 4 a .. 4 a + 3 are k step 2 m + a / 2, k quarter 2 (a & 1) + half: one 16-byte {hi, lo} group per a.
template <bool BF3>
__device__ __forceinline__ void actor16_store_x1(float4 *s_xf, const int ts_w, const int m, const int half, const int n,
                                                 const float (&v)[16])
{
    if (!BF3) {
        float4 *dst = s_xf + (ts_w * 4 + 2 * m) * 64 + half * 16 + n;
#pragma unroll
        for (int qh = 0; qh < 2; ++qh)
#pragma unroll
            for (int ql = 0; ql < 2; ++ql)
                dst[qh * 64 + ql * 32] = make_float4(v[8 * qh + ql], v[8 * qh + 2 + ql], v[8 * qh + 4 + ql], v[8 * qh + 6 + ql]);
    } else {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            u32x2 hi, lo;
            bf16_split4(v[4 * a], v[4 * a + 1], v[4 * a + 2], v[4 * a + 3], hi, lo);
            reinterpret_cast<uint4 *>(s_xf)[(ts_w * 4 + 2 * m + (a >> 1)) * 64 + (2 * (a & 1) + half) * 16 + n] = make_uint4(hi.x, hi.y, lo.x, lo.y);
        }
    }
}

// input projection of timestep ts for this wave's two tiles (+ bias)
template <bool BF3>
__device__ __forceinline__ void actor16_inproj(const float4 *s_xf, const int ts, const int lane,
                                               const float (&aih)[2][16], const u32x2 (&ah)[2][4], const u32x2 (&al)[2][4],
                                               const float (&bias)[2][4], f32x4 (&acc)[2])
{
    acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!BF3) {
        // all four x1 fragments are requested before the first product (left alone, the compiler reads each one in front of its
        // eight matrix instructions through ONE register quad: four LDS round trips in a row on the timestep's chain)
        const float4 *xf = s_xf + (ts * 4) * 64 + lane;
        const float4 xq[4] = {xf[0], xf[64], xf[128], xf[192]};
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jx = 0; jx < 4; ++jx) {
            const float4 b = xq[jx];
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(aih[0][4 * jx + 0], b.x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(aih[1][4 * jx + 0], b.x, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(aih[0][4 * jx + 1], b.y, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(aih[1][4 * jx + 1], b.y, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(aih[0][4 * jx + 2], b.z, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(aih[1][4 * jx + 2], b.z, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(aih[0][4 * jx + 3], b.w, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(aih[1][4 * jx + 3], b.w, acc[1], 0, 0, 0);
        }
    } else {
        const uint4 *xb = reinterpret_cast<const uint4 *>(s_xf) + (ts * 4) * 64 + lane;
#pragma unroll
        for (int sx = 0; sx < 4; ++sx) {
            const uint4 q = xb[sx * 64];
            const bf16x4 bh = __builtin_bit_cast(bf16x4, u32x2{q.x, q.y}), bl = __builtin_bit_cast(bf16x4, u32x2{q.z, q.w});
#pragma unroll
            for (int T = 0; T < 2; ++T) {
                const bf16x4 wh = __builtin_bit_cast(bf16x4, ah[T][sx]), wl = __builtin_bit_cast(bf16x4, al[T][sx]);
                acc[T] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wl, bh, acc[T], 0, 0, 0);
                acc[T] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wh, bl, acc[T], 0, 0, 0);
                acc[T] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wh, bh, acc[T], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[T][i] += bias[T][i];
}

// The value of lane ^ 16 / lane ^ 32 by v_permlane16_swap / v_permlane32_swap (gfx950): the instruction swaps the odd rows (halves) of
// its first operand with the even rows (halves) of the second, so with both operands the same value the first holds, in every
// even row (half), its own value and the second its neighbour's -- one VALU instruction and a select instead of a ds_bpermute
// round trip through the LDS crossbar (~120 cycles, exposed on the head's dependent chain).
__device__ __forceinline__ uint32_t lane_xor16(const uint32_t v)
{
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return (threadIdx.x & 16) ? r[0] : r[1];
}
__device__ __forceinline__ uint32_t lane_xor32(const uint32_t v)
{
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return (threadIdx.x & 32) ? r[0] : r[1];
}
__device__ __forceinline__ float lane_xor16(const float v) { return __uint_as_float(lane_xor16(__float_as_uint(v))); }
__device__ __forceinline__ float lane_xor32(const float v) { return __uint_as_float(lane_xor32(__float_as_uint(v))); }

// LDS of the pass (floats from a 16-byte aligned base): dense1 output and head input in B-fragment order, the h exchange,
// the dense1 constants
struct Actor16Lds {
    float4 *s_xf;  // [N timesteps][4 j][64 lane]: element e of (j, lane (n, kq)) = x1[row (ts, n)][x1_kpos(16 j + 4 e + kq)]
    float4 *s_hx;  // [2 buffers][2 dir][2 j][64 lane]: element e = h[seq n][16 j + 4 e + kq]
    float4 *s_hf;  // [rows / 16 tiles][4 j][64 lane]: element e = relu(h)[row 16 tile + n][16 j + 4 e + kq], rows env-major
    float *f_w1;   // [2 m][S1][64 lane]
    float *s_b1;   // [64]
    float *end;
};
__host__ __device__ inline size_t actor16_lds_floats(int N, int rows, int S1)
{
    return (size_t)N * 1024 + 2048 + (size_t)((rows + 15) / 16) * 1024 + (size_t)2 * S1 * 64 + 64;
}
__device__ __forceinline__ Actor16Lds actor16_carve(float *base, int N, int rows, int S1)
{
    Actor16Lds S;
    int o = 0;
    S.s_xf = reinterpret_cast<float4 *>(base + o); o += N * 1024;
    S.s_hx = reinterpret_cast<float4 *>(base + o); o += 2048;
    S.s_hf = reinterpret_cast<float4 *>(base + o); o += ((rows + 15) / 16) * 1024;
    S.f_w1 = base + o; o += 2 * S1 * 64;
    S.s_b1 = base + o; o += 64;
    S.end = base + o;
    return S;
}

// A wave's resident weights (wave = (direction wave / 4, hidden quarter wave % 4) of a 512-thread workgroup).  MFMA 16x16x4
// lane roles: as A operand lane = (tile row ar = lane % 16, k quarter kq = lane / 16); as B operand / accumulator lane =
// (column n = lane % 16, kq resp. row group rg = lane / 16).  LSTM tile row ar = 4 * (unit within tile) + gate: W row
// gate * 32 + hq * 8 + 4 * tile + ar / 4 of the direction; head tile row = logit.
struct Actor16W {
    float aih[2][16], ahh[2][8], bias[2][4];
    float aw2[16], b2c[4];
    u32x2 ah[2][4], al[2][4];  // bf16x3 form of aih (one of the two is live)
};

// weights -> registers, dense1 constants -> LDS; a workgroup barrier must follow before the first pass
template <int S1C, bool BF3 = false>
__device__ __forceinline__ void actor16_load(const ActorFusedArgs &A, const Actor16Lds &S, Actor16W &W)
{
    constexpr int S1 = 4 * S1C;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = wave >> 2, hq = wave & 3, n16 = lane & 15, kq = lane >> 4;
    const int OUT = A.n_out0 + A.n_out1;
    {
        const float4 *src = reinterpret_cast<const float4 *>(A.frag + 8 * 2 * 4 * 64 * 4);
        for (int f = tid; f < (2 * S1 * 64) / 4; f += 512) reinterpret_cast<float4 *>(S.f_w1)[f] = src[f];
        if (tid < 64) S.s_b1[tid] = A.b1[tid];
    }
    const float *whh = dir ? A.whh_r : A.whh_f;
    actor16_load_ih<S1, BF3>(A.frag, wave, lane, W.aih, W.ah, W.al);
#pragma unroll
    for (int T = 0; T < 2; ++T) {
        const int wrow = (n16 & 3) * 32 + hq * 8 + 4 * T + (n16 >> 2);  // within the direction: gate * 32 + unit
#pragma unroll
        for (int sx = 0; sx < 8; ++sx) W.ahh[T][sx] = whh[wrow * 32 + 4 * sx + kq];
#pragma unroll
        for (int i = 0; i < 4; ++i) W.bias[T][i] = A.bih[dir * 128 + i * 32 + hq * 8 + 4 * T + kq];  // accumulator role: rg = kq
    }
#pragma unroll
    for (int sx = 0; sx < 16; ++sx) W.aw2[sx] = n16 < OUT ? A.w2[n16 * 64 + 4 * sx + kq] : 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) W.b2c[i] = 4 * kq + i < OUT ? A.b2[4 * kq + i] : 0.0f;
}

// dense1 by 16 x 16 TILES (optional; short agent axes): wave (w / 4, hq = w % 4) computes rows 16 hq .. 16 hq + 15 of relu(W1 x(ts) + b1) for
// the 16 sequences of timestep ts = w / 4, w / 4 + 2, .. as ONE v_mfma_f32_16x16x4_f32 accumulator (K = D) -- pw_kernels_policy3j.hpp's
// dense1, whose header has the argument for the bits: fed k = 4 s + kq the chain sums the same k in the same order as the 32x32x2 blocks.
// With 32 x 32 blocks a pass has 2 ceil(N / 2) of them: at N = 2 (simple_reference) two waves work and six wait (stamps: 2.0 k of a
// step's 14.7 k cycles); as tiles the same matrix instructions are spread over all eight waves.  W1's A fragments: 2 S1C registers.
template <int S1C>
struct Actor16D1 {
    float a1[2 * S1C], b1v[4];
};
template <int S1C>
__device__ __forceinline__ void actor16_load_d1(const ActorFusedArgs &A, Actor16D1<S1C> &T)
{
    constexpr int S1 = 4 * S1C;
    const int lane = threadIdx.x & 63, n16 = lane & 15, kq = lane >> 4;
    const int hq = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) & 3;
    const float *w1p = A.frag + 8 * 2 * 4 * 64 * 4;   // the packed 32x32x2 image [2 m][S1][64 lane]: lane = row % 32 + 32 (k & 1), k step k / 2
    const int h = 16 * hq + n16;
#pragma unroll
    for (int sx = 0; sx < 2 * S1C; ++sx) {
        const int kk = 4 * sx + kq;
        T.a1[sx] = w1p[((h >> 5) * S1 + (kk >> 1)) * 64 + (h & 31) + 32 * (kk & 1)];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) T.b1v[i] = A.b1[16 * hq + 4 * kq + i];
}

// One pass for the rows of this workgroup (all 512 threads call it; E <= 16 environments, rows env-major r = e * N + agent).
//   xrows, xstride  observation rows [rows_here][xstride >= A.D] -- global memory or LDS (readable on entry)
//   step            Philox step of the Gumbel noise: value (row, logit o) = log(-log(u)), u = word (o & 3) of Philox block
//                   (o >> 2) keyed (seed; step, global row), as pw_actor_head_kernel -- block rg is exactly what the lanes of
//                   row group rg need for their four logits
//   act_g / act_l   sinks of the sampled indices [rows_here * nheads], global / LDS (either may be NULL); A.H, A.logits too
//   noise_l         this pass's Gumbel noise in LDS (actor16_draw_noise, same step), or NULL: drawn here
//   d1              W1 as 16 x 16 tile fragments (actor16_load_d1): dense1 runs as tiles on all eight waves (exact form only), or NULL: 32 x 32 blocks
// Arithmetic: element for element the operation sequence of actor_forward_wg (see pw_kernels_policy3.hpp, "Bits").
// On return every thread has passed a barrier after the last LDS access of the pass.  BF3: the opt-in bf16x3 input projection.
// pre() / mid(): called by every wave before its dense1 blocks / before its head tiles -- the two windows in which waves without
// a block or a tile idle; a rollout kernel parks the tail of its previous environment step there (no LDS of the pass touched).
struct Actor16NoHook {
    __device__ __forceinline__ void operator()() const {}
};

// The Gumbel noise of ONE pass's head(s), drawn ahead of the pass into LDS [rows_here][NB = ceil(OUT / 4) blocks][4] (16-byte aligned; the
// lanes of row group kq read block kq of their row as one float4): value (row, logit o) = log(-log(u)), u = word (o & 3) of Philox
// block (o >> 2) keyed (seed; step, global row) -- the numbers actor16_forward draws inline when it is given no noise.  One thread
// per (row, block); called by `nthr` threads with indices t0 = 0 .. nthr - 1.  A rollout kernel runs it for step t + 1 on the waves
// that wait while the environment waves advance step t: ten Philox rounds and two logarithms per logit leave the head's dependent
// chain (stamps, simple_reference: 3.8 k of a step's 16 k cycles sat there).
__host__ __device__ inline int actor16_noise_floats(int rows, int out) { return rows * 4 * ((out + 3) >> 2) + 4; }   // + alignment slack
__device__ __forceinline__ void actor16_draw_noise(const ActorFusedArgs &A, float *s_noise, const int rows_here, const long row_base,
                                                   const uint64_t step, const int t0, const int nthr)
{
    const int OUT = A.n_out0 + A.n_out1, NB = (OUT + 3) >> 2;
    for (int idx = t0; idx < rows_here * NB; idx += nthr) {
        const int rr = idx / NB;
        const uint32_t blk = (uint32_t)(idx - rr * NB), tag = ((blk & 1u) << 31) | ((blk >> 1) << 30);
        const long grow = row_base + rr;
        uint32_t u[4];
        pw_philox4x32_10((uint32_t)grow, (uint32_t)((uint64_t)grow >> 32) | tag, (uint32_t)step, (uint32_t)(step >> 32),
                         (uint32_t)A.seed, (uint32_t)(A.seed >> 32), u);
        float nz[4];
#pragma unroll
        for (int wq = 0; wq < 4; ++wq) {   // (the words past OUT serve no logit: their values are never looked at)
            const float uo = ((float)(u[wq] >> 8) + 0.5f) * 5.9604644775390625e-8f;  // (0, 1)
            nz[wq] = __logf(-__logf(uo));
        }
        reinterpret_cast<float4 *>(s_noise)[idx] = make_float4(nz[0], nz[1], nz[2], nz[3]);
    }
}
#ifdef PW_STAMPS   // probe builds only (tools/*_probe.hip): shader cycles per phase into the caller's accumulators
#define PW_A16_STAMP_ARGS , unsigned long long *a16_st = nullptr, unsigned long long *a16_t0 = nullptr
#define PW_A16_STAMP(i) do { if (a16_st) { unsigned long long n_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(n_)::"memory"); \
                                          a16_st[i] += n_ - *a16_t0; *a16_t0 = n_; } } while (0)
#else
#define PW_A16_STAMP_ARGS
#define PW_A16_STAMP(i)
#endif
template <int S1C, bool BF3 = false, class Pre = Actor16NoHook, class Mid = Actor16NoHook>
__device__ __forceinline__ void actor16_forward(const ActorFusedArgs &A, const Actor16Lds &S, const Actor16W &W,
                                                const float *xrows, const int xstride, const int rows_here,
                                                const int envs_here, const long row_base, const uint64_t step,
                                                int32_t *act_g, int32_t *act_l, Pre pre = Pre(), Mid mid = Mid(),
                                                const float *noise_l = nullptr, const Actor16D1<S1C> *d1 = nullptr PW_A16_STAMP_ARGS)
{
    constexpr int S1 = 4 * S1C;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, col = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = wave >> 2, hq = wave & 3, n16 = lane & 15, kq = lane >> 4;
    const int N = A.N, D = A.D;
    const int nblk = 2 * ((N + 1) >> 1);  // dense1 blocks: (32-column tile = two timesteps of 16 sequences) x (hidden half)
    const bool seq_ok = n16 < envs_here;

    pre();
    PW_A16_STAMP(0);
    // ---- dense1 + ReLU as 16 x 16 tiles (timestep, hidden quarter of this wave), all eight waves
    if (!BF3 && d1) {
        constexpr int KS = 2 * S1C;
        const int nq = seq_ok ? n16 : 0;   // columns past the environments of this workgroup read a valid row; nobody uses their results
        for (int ts = wave >> 2; ts < N; ts += 2) {
            const float *xr = xrows + (size_t)(nq * N + ts) * xstride;
            float xb[KS];
#pragma unroll
            for (int sx = 0; sx < KS; ++sx) {
                const int kk = 4 * sx + kq;
                xb[sx] = kk < D ? xr[kk] : 0.0f;
            }
            f32x4 acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sx = 0; sx < KS; ++sx) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(d1->a1[sx], xb[sx], acc1, 0, 0, 0);
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = fmaxf(acc1[i] + d1->b1v[i], 0.0f);
            // hidden unit 16 hq + 4 kq + i in the x1 fragment (pw_kernels_policy3j.hpp): fragment j = hq, lane slot (2 (i & 1) + (kq & 1)) * 16 + n,
            // element 2 (kq / 2) + i / 2 -- registers (0, 2) and (1, 3) are two 8-byte stores
            float4 *dst = S.s_xf + (ts * 4 + hq) * 64 + n16;
            reinterpret_cast<float2 *>(dst + (kq & 1) * 16)[kq >> 1] = make_float2(v[0], v[2]);
            reinterpret_cast<float2 *>(dst + (2 + (kq & 1)) * 16)[kq >> 1] = make_float2(v[1], v[3]);
        }
    } else
    // ---- dense1 + ReLU: 32 x 32 blocks of relu(W1 X^T + b1), column rho = 16 * timestep + sequence
    for (int blk = wave; blk < nblk; blk += 8) {
        const int rt = blk >> 1, m = blk & 1;
        int ts = 2 * rt + (col >> 4), n = col & 15;
        if (ts >= N) ts = N - 1;
        if (n >= envs_here) n = 0;  // slots past the environments of this workgroup read a valid row; nobody uses their results
        const float *xr = xrows + (size_t)(n * N + ts) * xstride;
        // rows longer than 32 numbers: the B operands are read in two batches with a scheduling fence in between, so that at most
        // S1 / 2 of them are live at a time (all S1 at once is what took the simple_tag rollout past the 256-register cap at D > 32)
        constexpr int KC = S1 > 16 ? S1 / 2 : S1;
        f32x16 acc1;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc1[q] = 0.0f;
#pragma unroll
        for (int k0 = 0; k0 < S1; k0 += KC) {
            float xb[KC];
#pragma unroll
            for (int sidx = 0; sidx < KC; ++sidx) {
                const int kk = 2 * (k0 + sidx) + half;
                xb[sidx] = kk < D ? xr[kk] : 0.0f;
            }
#pragma unroll
            for (int sidx = 0; sidx < KC; ++sidx)
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(S.f_w1[(m * S1 + k0 + sidx) * 64 + lane], xb[sidx], acc1, 0, 0, 0);
            if (k0 + KC < S1) __builtin_amdgcn_sched_barrier(0);
        }
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = fmaxf(acc1[q] + S.s_b1[m * 32 + mfma_row(q, half)], 0.0f);
        if (2 * rt + (col >> 4) < N) actor16_store_x1<BF3>(S.s_xf, 2 * rt + (col >> 4), m, half, col & 15, v);
    }
    PW_A16_STAMP(1);
    wg_lds_barrier();  // the x1 fragments are in LDS
    PW_A16_STAMP(2);

    // ---- the BiLSTM, one timestep per barrier
    auto inproj = [&](const int ts, f32x4 (&acc)[2]) { actor16_inproj<BF3>(S.s_xf, ts, lane, W.aih, W.ah, W.al, W.bias, acc); };
    {
        f32x4 acc[2], accn[2];
        float c0 = 0.f, c1 = 0.f;
        inproj(dir ? N - 1 : 0, acc);
        for (int s2 = 0; s2 < N; ++s2) {
            const int ts = dir ? N - 1 - s2 : s2;
            if (s2 > 0) {
                const float4 *hx = S.s_hx + ((((s2 - 1) & 1) * 2 + dir) * 2) * 64 + lane;
                const float4 h0 = hx[0], h1 = hx[64];
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(W.ahh[0][0], h0.x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(W.ahh[1][0], h0.x, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(W.ahh[0][1], h0.y, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(W.ahh[1][1], h0.y, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(W.ahh[0][2], h0.z, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(W.ahh[1][2], h0.z, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(W.ahh[0][3], h0.w, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(W.ahh[1][3], h0.w, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(W.ahh[0][4], h1.x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(W.ahh[1][4], h1.x, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(W.ahh[0][5], h1.y, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(W.ahh[1][5], h1.y, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(W.ahh[0][6], h1.z, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(W.ahh[1][6], h1.z, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(W.ahh[0][7], h1.w, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(W.ahh[1][7], h1.w, acc[1], 0, 0, 0);
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
            wg_lds_barrier();  // h(ts) of every unit is in LDS (the last one: the head input is complete)
            if (s2 + 1 < N) { acc[0] = accn[0]; acc[1] = accn[1]; }
        }
    }

    PW_A16_STAMP(3);
    mid();
    PW_A16_STAMP(4);
    // ---- optional: the hidden state H [rows][64] back in row order
    if (A.H) {
        for (int idx = tid; idx < rows_here * 64; idx += 512) {
            const int rr = idx >> 6, kk = idx & 63;
            const float4 q = S.s_hf[((rr >> 4) * 4 + (kk >> 4)) * 64 + (kk & 3) * 16 + (rr & 15)];
            const int e = (kk >> 2) & 3;
            A.H[(size_t)row_base * 64 + idx] = e == 0 ? q.x : e == 1 ? q.y : e == 2 ? q.z : q.w;
        }
    }
    // ---- the head(s) on the matrix cores: logits^T [16 x 16 rows] = W2 [16 x 64] * relu(h)^T per 16-row tile, C-in = b2, k
    // ascending (the chain of actor_forward_wg's head); the heads' logits are concatenated ([n_out0 | n_out1], run.py:39-41
    // order): row group rg of a column holds logits 4 rg .. 4 rg + 3.  Then minus the Gumbel noise and one arg-max per head
    // (first maximum wins): a local scan and two exchanges across the four row groups.
    const int OUT = A.n_out0 + A.n_out1, nheads = A.n_out1 > 0 ? 2 : 1;
    const bool sample = act_g != nullptr || act_l != nullptr;
    const int ntile = (rows_here + 15) >> 4;
    const int NB = (OUT + 3) >> 2;
    for (int tile = wave; tile < ntile; tile += 8) {
        f32x4 lg;
#pragma unroll
        for (int i = 0; i < 4; ++i) lg[i] = W.b2c[i];
        const int rr = tile * 16 + n16;
        const bool row_ok = rr < rows_here;
        const long grow = row_base + (row_ok ? rr : 0);
        // every LDS operand of the tile is requested before the first matrix instruction (read one by one in front of its four
        // products, each read is a round trip of its own on the chain)
        const float4 *hf = S.s_hf + (tile * 4) * 64 + lane;
        const float4 hb[4] = {hf[0], hf[64], hf[128], hf[192]};
        float4 nzv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (noise_l) nzv = reinterpret_cast<const float4 *>(noise_l)[(row_ok ? rr : 0) * NB + (kq < NB ? kq : 0)];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jx = 0; jx < 4; ++jx) {
            lg = __builtin_amdgcn_mfma_f32_16x16x4f32(W.aw2[4 * jx + 0], hb[jx].x, lg, 0, 0, 0);
            lg = __builtin_amdgcn_mfma_f32_16x16x4f32(W.aw2[4 * jx + 1], hb[jx].y, lg, 0, 0, 0);
            lg = __builtin_amdgcn_mfma_f32_16x16x4f32(W.aw2[4 * jx + 2], hb[jx].z, lg, 0, 0, 0);
            lg = __builtin_amdgcn_mfma_f32_16x16x4f32(W.aw2[4 * jx + 3], hb[jx].w, lg, 0, 0, 0);
        }
        if (A.logits && row_ok) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (4 * kq + i < OUT) A.logits[(size_t)grow * OUT + 4 * kq + i] = lg[i];
        }
        if (sample) {
            float p[4] = {lg[0], lg[1], lg[2], lg[3]};
            if (noise_l) {   // (logits past OUT: rows of W2 / b2 that are zero, never candidates below)
                p[0] = lg[0] - nzv.x; p[1] = lg[1] - nzv.y; p[2] = lg[2] - nzv.z; p[3] = lg[3] - nzv.w;
            } else if (4 * kq < OUT) {  // wave-divergent only by row group
                const uint32_t blk = (uint32_t)kq, tag = ((blk & 1u) << 31) | ((blk >> 1) << 30);
                uint32_t u[4];
                pw_philox4x32_10((uint32_t)grow, (uint32_t)((uint64_t)grow >> 32) | tag, (uint32_t)step, (uint32_t)(step >> 32),
                                 (uint32_t)A.seed, (uint32_t)(A.seed >> 32), u);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float uo = ((float)(u[i] >> 8) + 0.5f) * 5.9604644775390625e-8f;  // (0, 1)
                    p[i] = lg[i] - __logf(-__logf(uo));
                }
            }
            // one arg-max per head (first maximum wins), written as selects: both heads' scans and exchanges are straight-line code
            int bests[2] = {0, 0};
#pragma unroll
            for (int hd = 0; hd < 2; ++hd) {
                const int lo = hd ? A.n_out0 : 0, cnt = hd ? A.n_out1 : A.n_out0;   // (one head: cnt = 0 for the second, nothing stored)
                float bv = -INFINITY;
                int best = 0x7fffffff;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int o = 4 * kq + i - lo;
                    const bool take = o >= 0 && o < cnt && (best == 0x7fffffff || p[i] > bv);
                    bv = take ? p[i] : bv;
                    best = take ? o : best;
                }
#pragma unroll
                for (int sh = 16; sh <= 32; sh <<= 1) {  // the lower logit index wins a tie: the first maximum, as a scan would find
                    const float ov = sh == 16 ? lane_xor16(bv) : lane_xor32(bv);
                    const int ob = (int)(sh == 16 ? lane_xor16((uint32_t)best) : lane_xor32((uint32_t)best));
                    const bool take = ob != 0x7fffffff && (best == 0x7fffffff || ov > bv || (ov == bv && ob < best));
                    bv = take ? ov : bv;
                    best = take ? ob : best;
                }
                bests[hd] = best;
                if (nheads == 1) break;
            }
            if (kq == 0 && row_ok) {
                if (act_g) { act_g[rr * nheads] = bests[0]; if (nheads == 2) act_g[rr * nheads + 1] = bests[1]; }
                if (act_l) { act_l[rr * nheads] = bests[0]; if (nheads == 2) act_l[rr * nheads + 1] = bests[1]; }
            }
        }
    }
    PW_A16_STAMP(5);
    wg_lds_barrier();
    PW_A16_STAMP(6);
}

// The whole actor in ONE launch (pw_actor_fused) on the 16x16x4 core: 16 environments per workgroup at any N <= 16 that fits LDS
template <int S1C, bool BF3 = false>
__global__ void __launch_bounds__(512) pw_actor_fused16_kernel(const ActorFusedArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int N = A.N, E = A.E;
    const Actor16Lds S = actor16_carve(reinterpret_cast<float *>(smem_raw), N, E * N, 4 * S1C);
    const long env0 = (long)blockIdx.x * E;
    const int envs_here = (int)((long)A.B - env0 < (long)E ? (long)A.B - env0 : (long)E);
    const long row_base = env0 * N;
    const int nheads = A.n_out1 > 0 ? 2 : 1;
    const uint64_t step = (A.act && A.step_dev) ? (uint64_t)*A.step_dev : A.step;
    Actor16W W;
    actor16_load<S1C, BF3>(A, S, W);
    wg_lds_barrier();
    actor16_forward<S1C, BF3>(A, S, W, A.X + (size_t)row_base * A.D, A.D, envs_here * N, envs_here, row_base, step,
                         A.act ? A.act + row_base * nheads : nullptr, nullptr);
}

}  // namespace
