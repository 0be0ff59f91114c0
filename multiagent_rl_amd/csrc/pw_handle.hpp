// pw_handle.hpp -- part of libpworld.so: the handle and the host-side helpers both translation units use.
#pragma once

struct pw_handle {
    pw_config cfg;
    KParams kp;
    pw_state_layout layout;
    bool bound;
    bool fast;      // pw_spread_fast_kernel applies
    FastConsts fc;
    bool tag_fast;  // pw_tag_stream_kernel applies
    TagParams tp;   // its constant part (pointers are filled per launch)
    float *comm;    // simple_reference planes inside the bound state block
    int32_t *goal;
    const char *last_kernel;  // name of the kernel the last pw_step / pw_rollout launched (pw_rollout_kernel)
    pw_dispatch disp;         // kernel selection: fixed by pw_create / pw_set_dispatch, never read from the environment at launch
    int actor_bf16x3;         // pw_set_actor_precision: 1 = the opt-in, NOT exact bf16x3 input projection of the one-launch rollouts
};

namespace {

// kernel parameter block of the communication scenarios (pw_kernels_reference.hpp) from a handle
RefParams ref_params(const pw_handle *h)
{
    const KParams &kp = h->kp;
    RefParams R;
    std::memset(&R, 0, sizeof(R));
    R.B = kp.B; R.L = kp.L; R.D = kp.D; R.max_episode_len = kp.max_episode_len; R.auto_reset = kp.auto_reset;
    R.force_discrete = kp.force_discrete; R.seed = kp.seed; R.env_id_base = kp.env_id_base;
    R.dt = kp.dt; R.damp = kp.damp; R.mass = kp.mass; R.sens = kp.agent_sens[0];
    R.pos_x = kp.pos_x; R.pos_y = kp.pos_y; R.vel_x = kp.vel_x; R.vel_y = kp.vel_y; R.lm_x = kp.lm_x; R.lm_y = kp.lm_y;
    R.comm = h->comm; R.goal = h->goal; R.ep_step = kp.ep_step; R.ep_count = kp.ep_count;
    return R;
}

int check_ready(const pw_handle *h)
{
    if (!h) return fail(PW_EINVAL, "null handle");
    if (!h->bound) return fail(PW_ESTATE, "state block not bound: call pw_bind_state first");
    return PW_OK;
}

// Kernels that ask for more than 64 KB of dynamic LDS need the opt-in once per (kernel, DEVICE): a process driving
// several GPUs must not skip it on the second one.  The device's bit is set only AFTER hipFuncSetAttribute succeeded
// (lds_optin_done), atomically: a failed call is retried by the next launch, and two host threads on different devices
// cannot lose each other's bit.
bool lds_optin_needed(const unsigned long long *done_mask, int *dev_out)
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { *dev_out = -1; return true; }  // unknown: set it every time (cheap)
    *dev_out = dev;
    return !(__atomic_load_n(done_mask, __ATOMIC_ACQUIRE) >> dev & 1ull);
}

void lds_optin_done(unsigned long long *done_mask, int dev)
{
    if (dev >= 0) __atomic_fetch_or(done_mask, 1ull << dev, __ATOMIC_RELEASE);
}

// hipFuncSetAttribute(kernel, MaxDynamicSharedMemorySize, 160 KB) once per (kernel, device); returns from the caller on failure
#define PW_LDS_OPTIN(mask_ptr, kernel_expr)                                                                              \
    do {                                                                                                                 \
        int optin_dev_;                                                                                                  \
        if (lds_optin_needed((mask_ptr), &optin_dev_)) {                                                                 \
            PW_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel_expr),                                \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                   \
            lds_optin_done((mask_ptr), optin_dev_);                                                                      \
        }                                                                                                                \
    } while (0)

// The chunk / tail / packed / wire entry points and the rollout sink write the plain ring layout only.
int plain_ring_only(const pw_replay_store *st, const char *who)
{
    if (st && (st->act_heads > 1 || st->per_agent))
        return fail(PW_EINVAL, std::string(who) + ": two-head / per-agent rings are served by pw_replay_add and pw_replay_gather only");
    if (st && st->state_rows)
        return fail(PW_EINVAL, std::string(who) + ": a STATE ring (pw_replay_store.state_rows) is filled by pw_replay_add_state_wire and read by "
                                                  "pw_replay_gather only -- observation rows cannot be turned back into the landmarks they were built from");
    return PW_OK;
}

// A state ring's own consistency (pw_replay_add_state_wire, pw_replay_gather).
int state_ring_ok(const pw_replay_store *st, const char *who)
{
    if (st->act_heads > 1 || st->per_agent) return fail(PW_EINVAL, std::string(who) + ": a STATE ring is a plain single-head, shared-reward ring");
    const int N = st->num_agents, L = st->num_landmarks, A = st->num_adversaries;
    if (!st->lm && L > 0) return fail(PW_EINVAL, std::string(who) + ": STATE ring without a landmark plane");
    if (N < 1 || L < 0 || A < 0 || A > N) return fail(PW_EINVAL, std::string(who) + ": bad STATE ring shape");
    const int D = st->scenario == PW_SIMPLE_SPREAD ? 4 + 2 * L : st->scenario == PW_SIMPLE_TAG ? 4 + 2 * L + 2 * (N - 1) + 2 * (N - A) : -1;
    if (D != st->obs_dim)
        return fail(PW_EINVAL, std::string(who) + ": STATE rings serve simple_spread (local observation, D = 4 + 2L) and simple_tag (D = 4 + 2L + 2(N - 1) + 2(N - A))");
    if ((reinterpret_cast<uintptr_t>(st->obs) | reinterpret_cast<uintptr_t>(st->next_obs)) & 15) return fail(PW_EINVAL, std::string(who) + ": STATE ring planes must be 16-byte aligned");
    if (reinterpret_cast<uintptr_t>(st->lm) & 7) return fail(PW_EINVAL, std::string(who) + ": STATE ring landmark plane must be 8-byte aligned");
    return PW_OK;
}
}  // namespace
