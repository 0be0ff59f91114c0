// pw_experiments.hpp -- TIMING-ONLY overlays of the product kernels.  NOT part of libpworld.so: this file lives under tools/, the
// product headers include it only under -DPW_EXPERIMENTS (a flag multiagent_rl_amd/build_native.py never passes), and every
// switch below makes the kernels compute WRONG results on purpose -- they bound what a change of schedule could gain before anyone
// builds it.  Build e.g.:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
//         -DPW_EXPERIMENTS -DPW_EXP_NO_FORCE -I include -I tools/experiments tools/step_time.hip -o tools/step_time_noforce.bin
// Switches (results: profiles/r4_n3_pair_bound.txt, profiles/r4_c2_bounds.txt):
//   PW_EXP_NO_NT_STORES   outputs that leave through nt_store are computed (kept alive by an empty asm) but not stored: what the
//                         write stream costs a kernel
//   PW_EXP_NO_FORCE       near_force_loop sees an empty near mask: a step without any contact force
//   PW_EXP_ONE_PARTNER    ... with at most ONE evaluation per lane (the bound of a pair-parallel form's gain)
//   PW_EXP_BARRIER2 / 0   the quad kernel's workgroup meets every SECOND step only / never
//   (PW_QUAD_ACT_AHEAD=8|16 is a plain parameter of pw_kernels_spread_quad.hpp, not an experiment: results stay exact)
#pragma once
#ifndef PW_EXPERIMENTS
#error "pw_experiments.hpp is a timing-only overlay: compile with -DPW_EXPERIMENTS (never for libpworld.so)"
#endif
#warning "PW_EXPERIMENTS build: kernels may compute WRONG results by design (timing only)"

#if defined(PW_EXP_NO_NT_STORES)
#define PW_HAVE_EXP_NT_STORE 1
template <typename T>
__device__ __forceinline__ void nt_store(T *p, const T v)
{
    if constexpr (sizeof(T) == 8) {
        const unsigned long long u = (unsigned long long)v;
        asm volatile("" :: "v"(p), "v"((unsigned)u), "v"((unsigned)(u >> 32)));
    } else if constexpr (sizeof(T) < 4) {
        asm volatile("" :: "v"(p), "v"((unsigned)v));
    } else {
        asm volatile("" :: "v"(p), "v"(v));
    }
}
__device__ __forceinline__ void nt_store(float4 *p, const float4 v) { asm volatile("" :: "v"(p), "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); }
__device__ __forceinline__ void nt_store(float2 *p, const float2 v) { asm volatile("" :: "v"(p), "v"(v.x), "v"(v.y)); }
#endif

#if defined(PW_EXP_NO_FORCE)
#define PW_NEAR_MASK_HOOK(m) do { (m) = 0; } while (0)
#elif defined(PW_EXP_ONE_PARTNER)
#define PW_NEAR_MASK_HOOK(m) do { (m) &= (decltype(m))0 - (m); } while (0)
#endif

#if defined(PW_EXP_BARRIER2)
#define PW_QUAD_BARRIER(t) do { if (!((t) & 1)) duo_barrier(); else wave_lds_sync(); } while (0)
#elif defined(PW_EXP_BARRIER0)
#define PW_QUAD_BARRIER(t) wave_lds_sync()
#endif
