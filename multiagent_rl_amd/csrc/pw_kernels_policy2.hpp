// pw_kernels_policy2.hpp -- part of libpworld.so (translation unit csrc/pworld_policy.hip includes it).
// Helpers shared by the one-launch policy rollout kernels (pw_kernels_policy3.hpp, pw_kernels_policy3j.hpp): the LDS observation
// row writer and the stamp / debug macros of the PW_STAMPS probe builds.
// (The second form of the rollout -- role-specialised waves: four "matrix" waves keeping W_ih in registers, four LSTM waves running
// the recurrence on packed vector FMAs beneath them -- lived here in rounds 2 and 3.  It was retired in round 4 together with the
// first form: with form 3 (N <= 12) and its just-in-time variant (N >= 13) nothing selected it automatically, it was 1.5-5x slower at
// every N (profiles/r4_policy_forms.txt), and the forms that run are each compared with the CPU oracle directly.)
#pragma once

namespace {

// observation row into LDS with 8-byte stores (the LDS rows are only 8-byte aligned: stride D + 2)
template <int LT>
__device__ __forceinline__ void lds_write_obs_row(float *o, const int L, const float2 *lm, float px, float py, float vx,
                                                  float vy)
{
    float2 *o2 = reinterpret_cast<float2 *>(o);
    o2[0] = make_float2(vx, vy);
    o2[1] = make_float2(px, py);
    if (LT > 0) {
#pragma unroll
        for (int l = 0; l < LT; ++l) {
            const float2 q = lm[l];
            o2[2 + l] = make_float2(q.x - px, q.y - py);
        }
    } else {  // runtime L: four landmark reads in flight per round (one LDS round trip per landmark otherwise: ~1.2 k cycles at L = 12)
        int l = 0;
        for (; l + 4 <= L; l += 4) {
            const float2 q0 = lm[l], q1 = lm[l + 1], q2 = lm[l + 2], q3 = lm[l + 3];
            o2[2 + l] = make_float2(q0.x - px, q0.y - py);
            o2[3 + l] = make_float2(q1.x - px, q1.y - py);
            o2[4 + l] = make_float2(q2.x - px, q2.y - py);
            o2[5 + l] = make_float2(q3.x - px, q3.y - py);
        }
        for (; l < L; ++l) {
            const float2 q = lm[l];
            o2[2 + l] = make_float2(q.x - px, q.y - py);
        }
    }
}

#ifdef PW_STAMPS
__device__ int g_pw_debug;  // probe builds only: bit 0 skips the MFMAs (tiles still announced), bit 1 the recurrences
#define PW_DBG(bit) (g_pw_debug & (bit))
#else
#define PW_DBG(bit) 0
#endif

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

// Ordered per-env reductions by wave shuffle for a runtime agent count: acc (+/-)= shfl(v, base + i) for i = 0 .. n - 1 IN THAT ORDER
// (the upstream summation order: the bits depend on it), with four shuffles in flight per round -- issued one at a time, each
// ds_bpermute's ~120-cycle round trip is exposed: 36 of them per step at N = 12, 72 at N = 24, on the environment waves' critical path.
__device__ __forceinline__ float shfl_sub_ordered(float acc, const float v, const int base, const int n)
{
    int i = 0;
    for (; i + 4 <= n; i += 4) {
        const float a0 = __shfl(v, base + i, kWave), a1 = __shfl(v, base + i + 1, kWave);
        const float a2 = __shfl(v, base + i + 2, kWave), a3 = __shfl(v, base + i + 3, kWave);
        acc -= a0; acc -= a1; acc -= a2; acc -= a3;
    }
    for (; i < n; ++i) acc -= __shfl(v, base + i, kWave);
    return acc;
}
__device__ __forceinline__ float shfl_add_ordered(float acc, const float v, const int base, const int n)
{
    int i = 0;
    for (; i + 4 <= n; i += 4) {
        const float a0 = __shfl(v, base + i, kWave), a1 = __shfl(v, base + i + 1, kWave);
        const float a2 = __shfl(v, base + i + 2, kWave), a3 = __shfl(v, base + i + 3, kWave);
        acc += a0; acc += a1; acc += a2; acc += a3;
    }
    for (; i < n; ++i) acc += __shfl(v, base + i, kWave);
    return acc;
}

// Ring sink of the one-launch rollouts into a STATE ring (pw_replay_store.state_rows: the planes hold {vx, vy, px, py} per agent and the episode's
// landmarks per transition; pw_replay_gather rebuilds the rows): the pre-step state + this lane's landmarks, and the post-step (pre-reset) state.
__device__ __forceinline__ void sink_state_obs(const pw_replay_store &ring, const size_t slot, const int N, const int a, const int L,
                                               const float2 *lmv, const float px, const float py, const float vx, const float vy)
{
    reinterpret_cast<float4 *>(ring.obs)[slot * N + a] = make_float4(vx, vy, px, py);
    for (int l = a; l < L; l += N) reinterpret_cast<float2 *>(ring.lm)[slot * L + l] = lmv[l];
}
__device__ __forceinline__ void sink_state_next(const pw_replay_store &ring, const size_t slot, const int N, const int a, const float px,
                                                const float py, const float vx, const float vy)
{
    reinterpret_cast<float4 *>(ring.next_obs)[slot * N + a] = make_float4(vx, vy, px, py);
}

// Ring slot of transition (t, env) of a chunk that starts at ring_start: (ring_start + t * B + env) mod capacity WITHOUT the 64-bit
// division (~150 instructions on the environment waves' critical path, every step): the host guarantees 0 <= ring_start < capacity and
// T * B <= capacity, so the sum is below 2 * capacity and one conditional subtraction is the modulo.
__device__ __forceinline__ size_t ring_slot(const int64_t ring_start, const int t, const int B, const long env, const int64_t capacity)
{
    const int64_t x = ring_start + (int64_t)t * B + env;
    return (size_t)(x >= capacity ? x - capacity : x);
}

// Finished-episode statistics of a rollout launch (run.py:55-65 bookkeeping summed over all envs): every workgroup leaves its
// partial (sum, count) in `scratch` ([gridDim.x] doubles, [gridDim.x] counts, one ticket word); the workgroup that takes the last
// ticket adds the partials up -- with ALL its threads: 512 partials per round through LDS and a fixed binary tree (deterministic:
// the pairing depends on gridDim.x alone).  (Until round 4 one thread walked the partials one global load after the other: ~40 us at
// the end of every launch with 256 workgroups -- 3 % of a 100-step C2 chunk.)  Call with the per-env partials in s_fs / s_fc
// (LDS, complete: a barrier has passed); `red` = 8 KB of LDS that nothing else uses any more (the start of the block).
__device__ __forceinline__ void rollout_finish_stats(const int envs_here, double *s_fs, int *s_fc, unsigned long long *scratch,
                                                     double *finished_sum, int64_t *finished_count, unsigned char *red)
{
    const int tid = threadIdx.x;
    double *part_sum = reinterpret_cast<double *>(scratch);
    long long *part_cnt = reinterpret_cast<long long *>(scratch + gridDim.x);
    unsigned long long *ticket = scratch + 2 * gridDim.x;
    if (tid == 0) {
        double ws = 0.0;
        long long wc = 0;
        for (int i = 0; i < envs_here; ++i) { ws += s_fs[i]; wc += s_fc[i]; }
        part_sum[blockIdx.x] = ws;
        part_cnt[blockIdx.x] = wc;
        __threadfence();
        s_fc[0] = atomicAdd(ticket, 1ull) == (unsigned long long)gridDim.x - 1 ? 1 : 0;  // the per-env partials are consumed: reuse a slot
    }
    wg_lds_barrier();
    if (!s_fc[0]) return;   // workgroup-uniform
    __threadfence();
    double *red_s = reinterpret_cast<double *>(red);             // [512]
    long long *red_c = reinterpret_cast<long long *>(red) + 512;  // [512]
    double ssum = 0.0;
    long long scnt = 0;
    for (unsigned base = 0; base < gridDim.x; base += 512) {
        const unsigned i = base + (unsigned)tid;
        red_s[tid] = i < gridDim.x ? __builtin_nontemporal_load(part_sum + i) : 0.0;
        red_c[tid] = i < gridDim.x ? __builtin_nontemporal_load(part_cnt + i) : 0ll;
        wg_lds_barrier();
        for (int stride = 256; stride >= 1; stride >>= 1) {
            if (tid < stride) { red_s[tid] += red_s[tid + stride]; red_c[tid] += red_c[tid + stride]; }
            wg_lds_barrier();
        }
        if (tid == 0) { ssum += red_s[0]; scnt += red_c[0]; }
        wg_lds_barrier();
    }
    if (tid == 0) {
        *finished_sum += ssum;
        *finished_count += scnt;
        *ticket = 0;
    }
}

}  // namespace
