// pw_kernels_tag.hpp -- part of libpworld.so (one translation unit: csrc/pworld.hip includes it).
// simple_tag (predator-prey) streaming kernel: the "asymmetric collision path" of BASELINE.json configs[2].
#pragma once

namespace {

// ------------------------------------------------------------------------------------------
// simple_tag in the streaming style of pw_spread_stream_kernel: one wave per EPW envs, state in
// registers + LDS across T steps, every store unconditional (idle lanes shadow lane 0), exact
// far-pair skip, sqrt-free collision masks.  What is different from simple_spread:
//   * two agent classes (adversary: a < A, good: a >= A) with their own size / accel / max_speed, so
//     dist_min and the exact d2 thresholds are 2x2 tables indexed by (class_i, class_j);
//   * landmarks collide (immovable, size 0.2): agent-landmark pairs join the near set, after the agents,
//     which is upstream's entity order;
//   * integrate_state clamps |v| to max_speed;
//   * rewards: good agents -10 per colliding adversary and the boundary penalty; adversaries +10 per
//     colliding (good, adversary) pair (read from the good lanes' masks through LDS);
//   * observation rows [vel, pos, landmark - pos, other - pos, velocities of the OTHER good agents],
//     zero-padded to the adversaries' width D (8-byte stores: every component is an (x, y) pair).
// Requires each role to be homogeneous (the canonical scenario); otherwise the generic kernel runs.
// ------------------------------------------------------------------------------------------
struct TagParams {
    int B, N, L, A, D, epw, max_episode_len, auto_reset;
    int p_prio;     // duo kernel: raise the physics wave's issue priority (set by the launch: small and mid-size grids)
    int obs_block;  // duo kernel: stage the wave's observation rows in LDS and store them as one contiguous block (0 / 2 / 4 = chunk floats)
    uint64_t seed, env_id_base;
    float dt, damp, contact_force, contact_margin, mass;
    float sens[2], fscale[2], max_speed[2];
    float dist_min[2][2], coll_thr2[2][2], near_thr2[2][2];  // [class_i][class_j]
    float dist_min_lm[2], near_thr2_lm[2];                   // agent class vs landmark
    float *pos_x, *pos_y, *vel_x, *vel_y, *lm_x, *lm_y;
    int32_t *ep_step;
    uint32_t *ep_count;
    const int32_t *act;
    float *obs, *final_obs, *rew, *rew_shared;
    uint8_t *done, *terminal;
    uint64_t *coll;  // [T,B,N] collision masks; written only by the COLL instantiations
};

__device__ __forceinline__ bool bits_near(float d2, float near_thr2)
{
    // far <=> near_thr2 <= d2 < +inf on the raw bits (d2 is a sum of squares, never -0); NaN/inf stay near
    const uint32_t lo = __float_as_uint(near_thr2);
    return __float_as_uint(d2) - lo >= 0x7F800000u - lo;
}

template <int NT, int AT, int LT>
__device__ __forceinline__ void tag_write_obs(float *__restrict__ o, const int N, const int A, const int L,
                                              const int D, const int a, const float2 *lm, const float2 *pp,
                                              const float2 *vv, float px, float py, float vx, float vy)
{
    float2 *o2 = reinterpret_cast<float2 *>(o);
    int k = 0;
    o2[k++] = make_float2(vx, vy);
    o2[k++] = make_float2(px, py);
#pragma unroll(LT > 0 ? LT : 1)
    for (int l = 0; l < (LT ? LT : L); ++l) {
        const float2 q = lm[l];
        o2[k++] = make_float2(q.x - px, q.y - py);
    }
    for (int j = 0; j < (NT ? NT : N); ++j) {
        if (j == a) continue;
        const float2 q = pp[j];
        o2[k++] = make_float2(q.x - px, q.y - py);
    }
    for (int j = (AT >= 0 ? AT : A); j < (NT ? NT : N); ++j) {
        if (j == a) continue;
        o2[k++] = vv[j];
    }
    while (2 * k < D) o2[k++] = make_float2(0.0f, 0.0f);
}

// NT / AT / LT: compile-time N / A / L (0 / -1 / 0 = runtime)
template <int NT, int AT, int LT, bool UNIT_MASS, bool COLL = false>
__global__ void __launch_bounds__(kWave) pw_tag_stream_kernel(const TagParams P, const int T)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int N = NT ? NT : P.N, A = AT >= 0 ? AT : P.A, L = LT ? LT : P.L, D = P.D;
    float2 *s_pos = reinterpret_cast<float2 *>(smem_raw);   // [64]
    float2 *s_vel = s_pos + kWave;                           // [64]
    uint32_t *s_mlo = reinterpret_cast<uint32_t *>(s_vel + kWave);  // [64] collision mask, low / high words
    uint32_t *s_mhi = s_mlo + kWave;
    float *s_rew = reinterpret_cast<float *>(s_mhi + kWave);  // [64]
    float2 *s_lm = reinterpret_cast<float2 *>(s_rew + kWave); // [epw * L]

    int e_local = (int)threadIdx.x / N;
    int a = (int)threadIdx.x - e_local * N;
    int env = blockIdx.x * P.epw + e_local;
    if (e_local >= P.epw || env >= P.B) {  // idle lane: shadow lane 0
        e_local = 0; a = 0; env = blockIdx.x * P.epw;
    }
    const int base = e_local * N, me = base + a;
    const int cls = a >= A ? 1 : 0;
    const uint32_t g = (uint32_t)env * (uint32_t)N + (uint32_t)a;
    const size_t BN = (size_t)P.B * N;
    const float2 *pp = s_pos + base, *vv = s_vel + base;
    float2 *lmv = s_lm + e_local * L;
    const uint64_t env_id = P.env_id_base + (uint64_t)env;
    const uint64_t adv_bits = A >= 64 ? ~0ull : ((1ull << A) - 1ull);

    const float my_sens = P.sens[cls], my_fscale = P.fscale[cls], my_maxspeed = P.max_speed[cls];
    const float dmin_adv = P.dist_min[cls][0], dmin_good = P.dist_min[cls][1], dmin_lm = P.dist_min_lm[cls];
    const float cthr_adv = P.coll_thr2[cls][0], cthr_good = P.coll_thr2[cls][1];
    const float nthr_adv = P.near_thr2[cls][0], nthr_good = P.near_thr2[cls][1], nthr_lm = P.near_thr2_lm[cls];

    float px = P.pos_x[g], py = P.pos_y[g], vx = P.vel_x[g], vy = P.vel_y[g];
    int ep_step = P.ep_step[env];
    uint32_t ep_count = P.ep_count[env];
    for (int l = a; l < L; l += N) lmv[l] = make_float2(P.lm_x[(size_t)env * L + l], P.lm_y[(size_t)env * L + l]);
    s_pos[me] = make_float2(px, py);
    s_vel[me] = make_float2(vx, vy);
    wave_lds_sync();

    // collision mask of the current state + near sets of the next force evaluation
    uint64_t coll = 0, near_a = 0, near_l = 0;
    auto partner_pass = [&]() {
        coll = 0; near_a = 0; near_l = 0;
#pragma unroll(NT > 0 ? NT : 1)
        for (int j = 0; j < (NT ? NT : N); ++j) {
            const float2 q = pp[j];
            const float dx = q.x - px, dy = q.y - py;
            const float d2 = dx * dx + dy * dy;
            const bool jg = j >= A;
            if (d2 < (jg ? cthr_good : cthr_adv)) coll |= 1ull << j;
            if (bits_near(d2, jg ? nthr_good : nthr_adv)) near_a |= 1ull << j;
        }
        near_a &= ~(1ull << a);
#pragma unroll(LT > 0 ? LT : 1)
        for (int l = 0; l < (LT ? LT : L); ++l) {
            const float2 q = lmv[l];
            const float dx = q.x - px, dy = q.y - py;
            if (bits_near(dx * dx + dy * dy, nthr_lm)) near_l |= 1ull << l;
        }
    };
    partner_pass();

    const float k = P.contact_margin, cf = P.contact_force, dt = P.dt, damp = P.damp, mass = P.mass;
    int act_next = P.act[g];
    constexpr int kStoresPerStep = (NT > 0 && AT >= 0 && LT > 0) ? 4 + (COLL ? 1 : 0) + (4 + 2 * LT + 2 * (NT - 1) + 2 * (NT - AT)) / 2 : 0;
    constexpr int kVm = kStoresPerStep < 63 ? kStoresPerStep : 63;
    __builtin_amdgcn_s_waitcnt(0x0F70);

    for (int t = 0; t < T; ++t) {
        const size_t tBN = (size_t)t * BN;
        const int ai = act_next;
        {
            const int tn = t + 1 < T ? t + 1 : t;
            act_next = P.act[(size_t)tn * BN + g];
        }
        float ux = 0.0f + ((ai == 1 ? 1.0f : 0.0f) - (ai == 2 ? 1.0f : 0.0f));
        float uy = 0.0f + ((ai == 3 ? 1.0f : 0.0f) - (ai == 4 ? 1.0f : 0.0f));
        ux *= my_sens; uy *= my_sens;
        if (my_fscale != 1.0f) { ux = my_fscale * ux; uy = my_fscale * uy; }
        float fx = ux + 0.0f, fy = uy + 0.0f;
        // ---- U5: near agents (ascending j), then near landmarks (ascending l): upstream's entity order
        for (uint64_t m = near_a; m; m &= m - 1) {
            const int j = __builtin_ctzll(m);
            const float2 q = pp[j];
            collision_force<true>(px, py, q.x, q.y, j >= A ? dmin_good : dmin_adv, k, cf, fx, fy);
        }
        for (uint64_t m = near_l; m; m &= m - 1) {
            const float2 q = lmv[__builtin_ctzll(m)];
            collision_force<true>(px, py, q.x, q.y, dmin_lm, k, cf, fx, fy);
        }
        // ---- U6 with the max_speed clamp
        vx = vx * damp; vy = vy * damp;
        vx = vx + div_mass<UNIT_MASS>(fx, mass) * dt;
        vy = vy + div_mass<UNIT_MASS>(fy, mass) * dt;
        if (my_maxspeed >= 0.0f) {
            const float speed = sqrtf(vx * vx + vy * vy);
            if (speed > my_maxspeed) {
                vx = vx / speed * my_maxspeed;
                vy = vy / speed * my_maxspeed;
            }
        }
        px = px + vx * dt;
        py = py + vy * dt;
        wave_lds_sync();
        s_pos[me] = make_float2(px, py);
        s_vel[me] = make_float2(vx, vy);
        wave_lds_sync();
        partner_pass();

        // ---- simple_tag.reward
        s_mlo[me] = (uint32_t)coll;
        s_mhi[me] = (uint32_t)(coll >> 32);
        wave_lds_sync();
        float r = 0.0f;
        if (cls) {
            for (int q = 0; q < A; ++q)
                if ((coll >> q) & 1) r -= 10.0f;
            r -= tag_bound(fabsf(px));
            r -= tag_bound(fabsf(py));
        } else {
            for (int gj = A; gj < N; ++gj) {  // +10 per colliding (good, adversary) pair: exact small integers
                const uint64_t mg = ((uint64_t)s_mhi[base + gj] << 32) | s_mlo[base + gj];
                r += 10.0f * (float)__builtin_popcountll(mg & adv_bits);
            }
        }
        s_rew[me] = r;
        wave_lds_sync();
        float acc = 0.0f;
        for (int i = 0; i < N; ++i) acc += s_rew[base + i];
        nt_store(&P.rew[tBN + g], r);
        nt_store(&P.done[tBN + g], (uint8_t)0);
        if (COLL) nt_store(&P.coll[tBN + g], coll);
        nt_store(&P.rew_shared[(size_t)t * P.B + env], acc);
        ep_step += 1;
        const bool term = P.max_episode_len > 0 && ep_step >= P.max_episode_len;
        nt_store(&P.terminal[(size_t)t * P.B + env], (uint8_t)(term ? 1 : 0));
        if (term && P.auto_reset) {
            if (P.final_obs)
                tag_write_obs<NT, AT, LT>(P.final_obs + (tBN + g) * D, N, A, L, D, a, lmv, pp, vv, px, py, vx, vy);
            wave_lds_sync();
            ep_count += 1;
            ep_step = 0;
            pw_reset_xy(P.seed, env_id, ep_count, (uint32_t)a, -1.0f, 1.0f, &px, &py);
            vx = 0.f; vy = 0.f;
            for (int l = a; l < L; l += N) {
                float x, y;
                pw_reset_xy(P.seed, env_id, ep_count, (uint32_t)(N + l), -0.9f, 0.9f, &x, &y);
                lmv[l] = make_float2(x, y);
            }
            s_pos[me] = make_float2(px, py);
            s_vel[me] = make_float2(0.f, 0.f);
        }
        wave_lds_sync();
        if (P.auto_reset && __any(term)) partner_pass();
        tag_write_obs<NT, AT, LT>(P.obs + (tBN + g) * D, N, A, L, D, a, lmv, pp, vv, px, py, vx, vy);
        if (kStoresPerStep > 0) __builtin_amdgcn_s_waitcnt((kVm & 0xF) | 0x0F70 | ((kVm >> 4) << 14));
    }

    P.pos_x[g] = px; P.pos_y[g] = py;
    P.vel_x[g] = vx; P.vel_y[g] = vy;
    for (int l = a; l < L; l += N) {
        const float2 q = lmv[l];
        P.lm_x[(size_t)env * L + l] = q.x;
        P.lm_y[(size_t)env * L + l] = q.y;
    }
    P.ep_step[env] = ep_step;
    P.ep_count[env] = ep_count;
}

// ------------------------------------------------------------------------------------------
// Two-wave (duo) form of the simple_tag kernel, same split as pw_spread_duo_kernel: wave P does the
// physics (action, near-pair forces against agents then landmarks, integrate + max_speed clamp, publish
// {pos, vel} into the 3-slot LDS ring, near sets of the next step), wave O one step behind does the
// collision masks, rewards, shared reward, terminal, observation rows and every store.  Each wave keeps
// its own copy of the landmarks (both draw them from Philox at a reset), so the only cross-wave traffic is
// the ring.  Used while the grid is small enough to be latency bound.
// ------------------------------------------------------------------------------------------
template <int NT, int AT, int LT, bool UNIT_MASS, bool COLL = false, bool TRIO = false>
__global__ void __launch_bounds__((TRIO ? 3 : 2) * kWave) pw_tag_duo_kernel(const TagParams P, const int T)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int N = NT ? NT : P.N, A = AT >= 0 ? AT : P.A, L = LT ? LT : P.L, D = P.D;
    float4 *s_ring = reinterpret_cast<float4 *>(smem_raw);                 // [3][64] {px, py, vx, vy}
    uint32_t *s_mlo = reinterpret_cast<uint32_t *>(s_ring + 3 * kWave);    // [64] (wave O)
    uint32_t *s_mhi = s_mlo + kWave;
    float *s_rew = reinterpret_cast<float *>(s_mhi + kWave);               // [64] (wave O)
    float2 *s_lm_p = reinterpret_cast<float2 *>(s_rew + kWave);            // [epw * L] wave P's landmarks
    float2 *s_lm_o = s_lm_p + P.epw * L;                                   // [epw * L] wave O's landmarks
    // [4][64] wave P's action indices, fetched four steps ahead by LDS-direct loads (pw_common.hpp act_fetch_issue)
    int32_t *s_act = reinterpret_cast<int32_t *>(smem_raw + ((3 * kWave * sizeof(float4) + 3 * kWave * sizeof(float) +
                                                              2 * (size_t)P.epw * L * sizeof(float2) + 15) & ~(size_t)15));
    // [64][D] the wave's observation rows, staged for the block-wise store (only when P.obs_block; 16-byte aligned)
    float *s_rows = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(s_act) + kActRingBytes);

    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int lane = (int)threadIdx.x & 63;
    int e_local = lane / N;
    int a = lane - e_local * N;
    int env = blockIdx.x * P.epw + e_local;
    if (e_local >= P.epw || env >= P.B) {  // idle lane: shadow lane 0
        e_local = 0; a = 0; env = blockIdx.x * P.epw;
    }
    const int base = e_local * N, me = base + a;
    const int cls = a >= A ? 1 : 0;
    const uint32_t g = (uint32_t)env * (uint32_t)N + (uint32_t)a;
    const size_t BN = (size_t)P.B * N;
    const uint64_t env_id = P.env_id_base + (uint64_t)env;
    int ep_step = P.ep_step[env];
    uint32_t ep_count = P.ep_count[env];
    int cur = 0;

    if (wave == 0) {
        // ================================ wave P: physics ================================
        if (P.p_prio) __builtin_amdgcn_s_setprio(3);  // serve the physics wave first where it shares a SIMD with output waves
        float2 *lmv = s_lm_p + e_local * L;
        const float my_sens = P.sens[cls], my_fscale = P.fscale[cls], my_maxspeed = P.max_speed[cls];
        const float dmin_adv = P.dist_min[cls][0], dmin_good = P.dist_min[cls][1], dmin_lm = P.dist_min_lm[cls];
        const float nthr_adv = P.near_thr2[cls][0], nthr_good = P.near_thr2[cls][1], nthr_lm = P.near_thr2_lm[cls];
        float px = P.pos_x[g], py = P.pos_y[g], vx = P.vel_x[g], vy = P.vel_y[g];
        for (int l = a; l < L; l += N) lmv[l] = make_float2(P.lm_x[(size_t)env * L + l], P.lm_y[(size_t)env * L + l]);
        s_ring[me] = make_float4(px, py, vx, vy);
        wave_lds_sync();
        uint64_t near_a = 0, near_l = 0;
        auto near_pass = [&](const float4 *slot) {
            near_a = 0; near_l = 0;
#pragma unroll(NT > 0 ? NT : 1)
            for (int j = 0; j < (NT ? NT : N); ++j) {
                const float2 q = *reinterpret_cast<const float2 *>(slot + j);
                const float dx = q.x - px, dy = q.y - py;
                if (bits_near(dx * dx + dy * dy, j >= A ? nthr_good : nthr_adv)) near_a |= 1ull << j;
            }
            near_a &= ~(1ull << a);
#pragma unroll(LT > 0 ? LT : 1)
            for (int l = 0; l < (LT ? LT : L); ++l) {
                const float2 q = lmv[l];
                const float dx = q.x - px, dy = q.y - py;
                if (bits_near(dx * dx + dy * dy, nthr_lm)) near_l |= 1ull << l;
            }
        };
        near_pass(s_ring + base);
        const float k = P.contact_margin, cf = P.contact_force, dt = P.dt, damp = P.damp, mass = P.mass;
        // Action indices: four steps ahead by LDS-direct loads, as in pw_spread_quad_kernel.  Stamps at C3 (B = 8192,
        // profiles/r3_tag_prefetch.txt): the index loaded ONE step ahead into a register still cost this wave 1020 of its
        // 3140 cycles per step -- HBM latency under the output waves' write stream exceeds a step.  This wave has no other
        // vector memory operation in its loop; every load the compiler counts is consumed before the first fetch.
        const int32_t *act_g = P.act + g;
        const uint32_t act_lds = __builtin_amdgcn_readfirstlane((uint32_t)reinterpret_cast<uintptr_t>(s_act));
        auto fetch_act = [&](int t) {  // indices of step t (clamped: the tail re-fetches the last step) -> slot t & 3
            act_fetch_issue(act_g + (size_t)(t < T ? t : T - 1) * BN, act_lds + (uint32_t)(t & 3) * (kWave * 4));
        };
        // Measured over B (profiles/r3_tag_prefetch.txt): B = 8192 -7 %, 4096 -13 %; 16384 ... 65536 unchanged (several
        // workgroups per SIMD already cover the load); only a chip that is not full pays (B = 2048: +3.7 %).  One form for
        // every grid: a run-time choice between the two fetches inside this loop compiled badly (the counted load's wait then
        // also drains the other form's fetches: 1.82 instead of 1.37 us per step at C3).
        asm volatile("" :: "v"(ep_step), "v"(ep_count), "v"(px), "v"(py), "v"(vx), "v"(vy), "v"(near_a), "v"(near_l) : "memory");
        fetch_act(0); fetch_act(1); fetch_act(2); fetch_act(3);
        PW_STAMP_DECL;
        for (int t = 0; t < T; ++t) {
            PW_STAMP_START;
            act_fetch_wait3();  // step t's indices are in LDS
            const int ai = s_act[(t & 3) * kWave + lane];
            fetch_act(t + 4);   // into the slot just read
            float ux = 0.0f + ((ai == 1 ? 1.0f : 0.0f) - (ai == 2 ? 1.0f : 0.0f));
            float uy = 0.0f + ((ai == 3 ? 1.0f : 0.0f) - (ai == 4 ? 1.0f : 0.0f));
            ux *= my_sens; uy *= my_sens;
            if (my_fscale != 1.0f) { ux = my_fscale * ux; uy = my_fscale * uy; }
            float fx = ux + 0.0f, fy = uy + 0.0f;
            const float4 *pp = s_ring + cur * kWave + base;
            PW_STAMP(0);
            // ONE loop over the lane's near entities -- agents (ascending j), then landmarks (ascending l): upstream's entity
            // order per lane.  A wave runs as many iterations as its busiest lane needs: in two loops that was
            // max(agents) + max(landmarks) over the lanes, in one it is max(agents + landmarks) -- a lane with an agent
            // contact and another with a landmark contact now share an iteration (~600 cycles each).
            if (N + L <= 64) {
                for (uint64_t m = near_a | (near_l << N); m; m &= m - 1) {
                    const int j = __builtin_ctzll(m);
                    const bool is_lm = j >= N;
                    const float2 q = is_lm ? lmv[j - N] : *reinterpret_cast<const float2 *>(pp + j);
                    collision_force<true>(px, py, q.x, q.y, is_lm ? dmin_lm : (j >= A ? dmin_good : dmin_adv), k, cf, fx, fy);
                }
            } else {
                for (uint64_t m = near_a; m; m &= m - 1) {
                    const int j = __builtin_ctzll(m);
                    const float2 q = *reinterpret_cast<const float2 *>(pp + j);
                    collision_force<true>(px, py, q.x, q.y, j >= A ? dmin_good : dmin_adv, k, cf, fx, fy);
                }
                for (uint64_t m = near_l; m; m &= m - 1) {
                    const float2 q = lmv[__builtin_ctzll(m)];
                    collision_force<true>(px, py, q.x, q.y, dmin_lm, k, cf, fx, fy);
                }
            }
            PW_STAMP(1);
            vx = vx * damp; vy = vy * damp;
            vx = vx + div_mass<UNIT_MASS>(fx, mass) * dt;
            vy = vy + div_mass<UNIT_MASS>(fy, mass) * dt;
            if (my_maxspeed >= 0.0f) {
                const float speed = sqrtf(vx * vx + vy * vy);
                if (speed > my_maxspeed) {
                    vx = vx / speed * my_maxspeed;
                    vy = vy / speed * my_maxspeed;
                }
            }
            px = px + vx * dt;
            py = py + vy * dt;
            int nxt = cur + 1; nxt = nxt == 3 ? 0 : nxt;
            s_ring[nxt * kWave + me] = make_float4(px, py, vx, vy);
            ep_step += 1;
            if (P.auto_reset && P.max_episode_len > 0 && ep_step >= P.max_episode_len) {
                ep_count += 1;
                ep_step = 0;
                pw_reset_xy(P.seed, env_id, ep_count, (uint32_t)a, -1.0f, 1.0f, &px, &py);
                vx = 0.f; vy = 0.f;
                for (int l = a; l < L; l += N) {
                    float x, y;
                    pw_reset_xy(P.seed, env_id, ep_count, (uint32_t)(N + l), -0.9f, 0.9f, &x, &y);
                    lmv[l] = make_float2(x, y);
                }
                nxt = nxt + 1; nxt = nxt == 3 ? 0 : nxt;
                s_ring[nxt * kWave + me] = make_float4(px, py, 0.f, 0.f);
            }
            cur = nxt;
            PW_STAMP(2);
            duo_barrier();
            PW_STAMP(3);
            near_pass(s_ring + cur * kWave + base);
            PW_STAMP(4);
        }
        act_fetch_drain();  // the tail's fetches have landed before the wave ends
        PW_STAMP_FLUSH;
        P.pos_x[g] = px; P.pos_y[g] = py;
        P.vel_x[g] = vx; P.vel_y[g] = vy;
        P.ep_step[env] = ep_step;
        P.ep_count[env] = ep_count;
    } else {
        // ================================ wave O: outputs ================================
        // One wave does both halves below -- or two do one each (TRIO: wave 1 the collision masks, rewards and the
        // per-agent / per-env planes, wave 2 the observation rows and the landmarks they need).  At C3 the single output
        // wave was the step's critical path (stamps: 3400 busy cycles against the physics wave's 2680); the halves share
        // nothing but the ring slot they read, so each follows the slot sequence and the episode clock itself.
        const bool do_rew = !TRIO || wave == 1, do_obs = !TRIO || wave == 2;  // compile-time true in the two-wave form
        if (TRIO && wave == 2 && P.p_prio) __builtin_amdgcn_s_setprio(2);  // the observation wave is the longer of the two output waves
        float2 *lmv = s_lm_o + e_local * L;
        const float cthr_adv = P.coll_thr2[cls][0], cthr_good = P.coll_thr2[cls][1];
        const uint64_t adv_bits = A >= 64 ? ~0ull : ((1ull << A) - 1ull);
        for (int l = a; l < L; l += N) lmv[l] = make_float2(P.lm_x[(size_t)env * L + l], P.lm_y[(size_t)env * L + l]);
        constexpr int kStoresPerStep = (NT > 0 && AT >= 0 && LT > 0) ? 4 + (COLL ? 1 : 0) + (4 + 2 * LT + 2 * (NT - 1) + 2 * (NT - AT)) / 2 : 0;
        constexpr int kVm = kStoresPerStep < 63 ? kStoresPerStep : 63;
        // observation row from a ring slot ({pos, vel} of every agent of the env) + this wave's landmarks
        auto write_row = [&](float *o, const float4 *slot, float px, float py, float vx, float vy) {
            float2 *o2 = reinterpret_cast<float2 *>(o);
            int kk = 0;
            o2[kk++] = make_float2(vx, vy);
            o2[kk++] = make_float2(px, py);
#pragma unroll(LT > 0 ? LT : 1)
            for (int l = 0; l < (LT ? LT : L); ++l) {
                const float2 q = lmv[l];
                o2[kk++] = make_float2(q.x - px, q.y - py);
            }
            for (int j = 0; j < (NT ? NT : N); ++j) {
                if (j == a) continue;
                const float4 q = slot[j];
                o2[kk++] = make_float2(q.x - px, q.y - py);
            }
            for (int j = (AT >= 0 ? AT : A); j < (NT ? NT : N); ++j) {
                if (j == a) continue;
                const float4 q = slot[j];
                o2[kk++] = make_float2(q.z, q.w);
            }
            while (2 * kk < D) o2[kk++] = make_float2(0.0f, 0.0f);
        };
        // Direct composition of the block (compile-time rosters): a row is DE = D / 2 float2 entries for BOTH roles (good
        // rows end in a zero entry), so the wave's block is a regular rows x DE grid and entry n = lane + 64 i has a fixed
        // (row, column) for the whole launch.  Every entry is "A - B" of two 8-byte LDS reads: A = the row's own velocity /
        // position, a landmark, another agent's position or velocity, B = the row's position or a zero that lives in LDS
        // (x - 0 is x) -- the subtractions write_row does, without staging the rows: per step one {pos, vel} publish per
        // lane, then NIT x (2 reads, 1 packed subtraction, 1 store of 512 contiguous bytes per wave).
        constexpr bool kDirect = NT > 0 && AT >= 0 && LT > 0;
        constexpr int DE = kDirect ? (4 + 2 * LT + 2 * (NT - 1) + 2 * (NT - AT)) / 2 : 1;
        constexpr int kMaxRows = kDirect ? (kWave / NT) * NT : 1;
        constexpr int NIT = kDirect ? (kMaxRows * DE + kWave - 1) / kWave : 1;
        float4 *s_state = reinterpret_cast<float4 *>(s_rows);                 // [64] {pos, vel} of every row, current slot
        float2 *s_zero = reinterpret_cast<float2 *>(s_state + kWave);         // {0, 0}
        int offA[NIT], offB[NIT];   // byte offsets into the workgroup's LDS
        const int envs_here_o = P.B - (int)blockIdx.x * P.epw < P.epw ? P.B - (int)blockIdx.x * P.epw : P.epw;
        const int n_entries = envs_here_o * N * DE;
        if (kDirect && do_obs && P.obs_block) {
            if (lane == 0) *s_zero = make_float2(0.0f, 0.0f);
            const int zoff = (int)(reinterpret_cast<unsigned char *>(s_zero) - smem_raw);
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int n = lane + kWave * it;
                const int nn = n < n_entries ? n : 0;
                const int r = nn / DE, kk = nn - r * DE;
                const int e = r / NT, ar = r - e * NT;
                const int self = (int)(reinterpret_cast<unsigned char *>(s_state + r) - smem_raw);  // {px, py, vx, vy}
                int oa, ob = zoff;
                if (kk == 0) oa = self + 8;                        // vel
                else if (kk == 1) oa = self;                       // pos
                else if (kk < 2 + LT) { oa = (int)(reinterpret_cast<unsigned char *>(s_lm_o + e * LT + (kk - 2)) - smem_raw); ob = self; }
                else if (kk < 2 + LT + NT - 1) {
                    const int m = kk - 2 - LT, j = m < ar ? m : m + 1;
                    oa = (int)(reinterpret_cast<unsigned char *>(s_state + e * NT + j) - smem_raw);
                    ob = self;
                } else {
                    int j = AT + (kk - (2 + LT + NT - 1));          // good agents in order, skipping the row's own
                    if (ar >= AT && j >= ar) j += 1;
                    oa = j < NT ? (int)(reinterpret_cast<unsigned char *>(s_state + e * NT + j) - smem_raw) + 8 : zoff;
                }
                offA[it] = oa; offB[it] = ob;
            }
        }
        PW_STAMP_DECL;
        for (int t = 0; t < T; ++t) {
            const size_t tBN = (size_t)t * BN;
            PW_STAMP_START;
            duo_barrier();
            PW_STAMP(0);
            int nxt = cur + 1; nxt = nxt == 3 ? 0 : nxt;
            const float4 *slot = s_ring + nxt * kWave + base;
            const float4 mine = slot[a];
            float px = mine.x, py = mine.y, vx = mine.z, vy = mine.w;
            ep_step += 1;
            const bool term = P.max_episode_len > 0 && ep_step >= P.max_episode_len;
            if (do_rew) {
                uint64_t coll = 0;
#pragma unroll(NT > 0 ? NT : 1)
                for (int j = 0; j < (NT ? NT : N); ++j) {
                    const float4 q = slot[j];
                    const float dx = q.x - px, dy = q.y - py;
                    if (dx * dx + dy * dy < (j >= A ? cthr_good : cthr_adv)) coll |= 1ull << j;
                }
                s_mlo[me] = (uint32_t)coll;
                s_mhi[me] = (uint32_t)(coll >> 32);
                wave_lds_sync();
                float r = 0.0f;
                if (cls) {
                    for (int q = 0; q < A; ++q)
                        if ((coll >> q) & 1) r -= 10.0f;
                    r -= tag_bound(fabsf(px));
                    r -= tag_bound(fabsf(py));
                } else {
                    for (int gj = A; gj < N; ++gj) {
                        const uint64_t mg = ((uint64_t)s_mhi[base + gj] << 32) | s_mlo[base + gj];
                        r += 10.0f * (float)__builtin_popcountll(mg & adv_bits);
                    }
                }
                s_rew[me] = r;
                wave_lds_sync();
                float acc = 0.0f;
                for (int i = 0; i < N; ++i) acc += s_rew[base + i];
                PW_STAMP(1);
                nt_store(&P.rew[tBN + g], r);
                nt_store(&P.done[tBN + g], (uint8_t)0);
                if (COLL) nt_store(&P.coll[tBN + g], coll);
                nt_store(&P.rew_shared[(size_t)t * P.B + env], acc);
                nt_store(&P.terminal[(size_t)t * P.B + env], (uint8_t)(term ? 1 : 0));
            }
            if (term && P.auto_reset) {
                if (do_obs && P.final_obs) write_row(P.final_obs + (tBN + g) * D, slot, px, py, vx, vy);
                wave_lds_sync();
                ep_count += 1;
                ep_step = 0;
                if (do_obs) {
                    for (int l = a; l < L; l += N) {
                        float x, y;
                        pw_reset_xy(P.seed, env_id, ep_count, (uint32_t)(N + l), -0.9f, 0.9f, &x, &y);
                        lmv[l] = make_float2(x, y);
                    }
                }
                nxt = nxt + 1; nxt = nxt == 3 ? 0 : nxt;
                slot = s_ring + nxt * kWave + base;  // post-reset state published by P
                const float4 fresh = slot[a];
                px = fresh.x; py = fresh.y; vx = fresh.z; vy = fresh.w;
            }
            cur = nxt;
            wave_lds_sync();
            if (!do_obs) {
                // (the rewards wave of a trio: nothing more in this step)
            } else if (kDirect && P.obs_block) {
                s_state[me] = make_float4(px, py, vx, vy);
                wave_lds_sync();
                float2 *blk2 = reinterpret_cast<float2 *>(P.obs + (tBN + (size_t)blockIdx.x * P.epw * N) * D);
                float2 va[NIT], vb[NIT];
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    va[it] = *reinterpret_cast<const float2 *>(smem_raw + offA[it]);
                    vb[it] = *reinterpret_cast<const float2 *>(smem_raw + offB[it]);
                }
#pragma unroll
                for (int it = 0; it < NIT; ++it)
                    if (lane + kWave * it < n_entries) nt_store(blk2 + lane + kWave * it, make_float2(va[it].x - vb[it].x, va[it].y - vb[it].y));
                wave_lds_sync();  // the reads are done before the states change again
            } else if (P.obs_block) {
                // A wave's rows are contiguous in the obs plane.  Row-per-lane they leave as 8-byte pieces at a stride of D
                // floats (every store instruction touches 48 cache lines); staged in LDS they leave as ONE block, 1 KiB
                // (or 512 B) contiguous per store instruction -- what pw_kernels_spread.hpp's stream_write_obs_block does
                // for simple_spread, here through LDS because a tag row is ragged (pw_common.hpp, nt_store: the hint).
                write_row(s_rows + me * D, slot, px, py, vx, vy);
                wave_lds_sync();
                const int envs_here = P.B - (int)blockIdx.x * P.epw < P.epw ? P.B - (int)blockIdx.x * P.epw : P.epw;
                const int total = envs_here * N * D;  // floats
                float *blk = P.obs + (tBN + (size_t)blockIdx.x * P.epw * N) * D;
                if (P.obs_block == 4) {
                    for (int q = lane; 4 * q < total; q += kWave)
                        nt_store(reinterpret_cast<float4 *>(blk) + q, reinterpret_cast<const float4 *>(s_rows)[q]);
                } else {
                    for (int q = lane; 2 * q < total; q += kWave)
                        nt_store(reinterpret_cast<float2 *>(blk) + q, reinterpret_cast<const float2 *>(s_rows)[q]);
                }
                wave_lds_sync();  // the block's LDS reads are done before the rows change again
            } else {
                write_row(P.obs + (tBN + g) * D, slot, px, py, vx, vy);
                if (kStoresPerStep > 0) __builtin_amdgcn_s_waitcnt((kVm & 0xF) | 0x0F70 | ((kVm >> 4) << 14));
            }
            PW_STAMP(2);
        }
#ifdef PW_STAMPS
        if (blockIdx.x == 0 && lane == 0) {
            if (do_rew) { g_pw_stamps[8] = st_acc[0]; g_pw_stamps[9] = st_acc[1]; }
            if (do_obs) { g_pw_stamps[10] = st_acc[2]; g_pw_stamps[11] = st_acc[0]; }
        }
#endif
        if (do_obs) {
            for (int l = a; l < L; l += N) {
                const float2 q = lmv[l];
                P.lm_x[(size_t)env * L + l] = q.x;
                P.lm_y[(size_t)env * L + l] = q.y;
            }
        }
    }
}

}  // namespace
