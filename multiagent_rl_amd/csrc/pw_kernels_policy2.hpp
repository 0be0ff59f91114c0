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
#pragma unroll(LT > 0 ? LT : 1)
    for (int l = 0; l < (LT ? LT : L); ++l) {
        const float2 q = lm[l];
        o2[2 + l] = make_float2(q.x - px, q.y - py);
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

}  // namespace
