/* pworld.h -- C ABI of libpworld.so: the MI355X (gfx950) batched particle-world.
 *
 * Drop-in boundary for the environment surface the reference drives:
 *   MultiAgentEnv.reset()/step()  call sites  experiments/run.py:28,44,60
 *   make_env() configuration                  experiments/scenarios.py:124-192
 *   local observation layout                  experiments/scenarios.py:6-20
 *   episode length / terminal rule            experiments/run.py:49-50, rls/arglist.py:5
 *   transition sink (replay)                  rls/replay_buffer.py:30-52
 * The arithmetic itself (World.step & friends) lives in the third-party
 * `multiagent` package the reference imports (experiments/scenarios.py:2-3);
 * each entry point names the upstream function it replaces.
 *
 * Conventions
 *   - plain C types only; every pointer marked "device" is caller-owned HBM
 *     (PyTorch allocates it); the library never allocates or frees device memory
 *   - every launch is asynchronous on the caller's hipStream_t (passed as void*,
 *     e.g. torch.cuda.current_stream().cuda_stream); no hidden synchronisation,
 *     so calls are hipGraph-capturable
 *   - return 0 on success, a negative PW_E* code otherwise; pw_last_error() gives
 *     the thread-local message of the last failure
 *   - one handle per (device, env shard); a handle is not thread-safe
 *
 * Layouts (B envs, N agents, L landmarks, D = pw_obs_dim):
 *   state block (device, pw_state_bytes, 256-B aligned planes, SoA over [B x N]):
 *     pos_x[B*N] pos_y[B*N] vel_x[B*N] vel_y[B*N] lm_x[B*L] lm_y[B*L]
 *     ep_step[B] (int32) ep_count[B] (uint32)      -- offsets: pw_state_layout
 *   obs [B,N,D] f32 row-major (rows of ragged simple_tag agents zero-padded)
 *   rew [B,N] f32; done [B,N] u8 (always 0: upstream done_callback is None);
 *   terminal [B] u8 (episode_step >= max_episode_len, run.py:50);
 *   coll [B,N] u64, bit j = is_collision(agent j, agent i) (bit i is always set,
 *   exactly as upstream's reward loop counts it)
 *
 * Reset RNG (device path): Philox4x32-10, counter = (entity, episode, env_id lo,
 *   env_id hi), key = (seed lo, seed hi); entity = agent index, or N + landmark
 *   index; x = out[0], y = out[1]; u = (r >> 8) * 2^-24; value = (hi-lo)*u + lo in
 *   float32 without FMA.  env_id = env_id_base + local env index, so a batch
 *   split over GPUs draws the same states as the unsplit batch.
 *   (The B = 1 compatibility env instead draws from NumPy's global legacy stream on
 *   the host, as upstream reset_world does -- main.py:47 -- and uploads via pw_set_state.)
 *
 * Numerics: IEEE float32, no IMPLICIT FMA contraction, correctly rounded / and sqrt, and
 *   a libm-free deterministic softplus/exp whose polynomial steps are explicit fmaf (definitions:
 *   pworld_math.h, revision 3).  A CPU implementation following pworld_math.h reproduces every
 *   output bit for bit.
 */
#ifndef PWORLD_H
#define PWORLD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PW_VERSION 106 /* 0.1.6: + STATE rings (pw_replay_store.state_rows: the ring keeps {vel, pos} + the episode's landmarks, pw_replay_gather rebuilds the rows);
                          state-only wire blocks for simple_tag (pw_state_wire_layout_scn), compact-row wire blocks for simple_reference (pw_ref_wire_*); pw_replay_store and pw_state_wire grew (appended fields, zero = before);
                          PWORLD_POLICY_V2 no longer read.  0.1.5: + pw_state_wire_* / pw_replay_add_state_wire (state-only wire blocks); PW_ACTOR_BF16X3 environment switch removed; 0.1.4: + pw_set_actor_precision / pw_actor_set_bf16x3 (opt-in bf16x3 input projection); pw_actor_front_pack's
                          image grew a third section.  0.1.3: + pw_dispatch (kernel selection frozen in the handle; no environment reads at launch)
                          (0.1.2: + pw_rollout_kernel; 0.1.1: pw_replay_store grew act_heads / per_agent / head_width; wire-block entry points) */
#define PW_MAX_AGENTS 64
#define PW_MAX_LANDMARKS 64

enum pw_scenario { PW_SIMPLE_SPREAD = 0, PW_SIMPLE_TAG = 1, PW_SIMPLE_REFERENCE = 2, PW_SIMPLE_SPEAKER_LISTENER = 3 };
#define PW_DIM_C 10   /* simple_reference: world.dim_c communication symbols */
#define PW_SL_DIM_C 3 /* simple_speaker_listener: world.dim_c */
enum pw_obs_mode { PW_OBS_LOCAL = 0, PW_OBS_FULL = 1 };
enum pw_error {
    PW_OK = 0,
    PW_EINVAL = -1,   /* bad argument / unsupported configuration */
    PW_ESTATE = -2,   /* state block not bound */
    PW_EHIP = -3,     /* a HIP runtime call failed (message has hipGetErrorString) */
    PW_ENOMEM = -4
};

typedef struct pw_handle pw_handle;

/* Mirrors what make_env()/make_world() fix for one env (experiments/scenarios.py:124-192)
 * plus the World constants (upstream core.py World.__init__). */
typedef struct pw_config {
    uint32_t struct_size;            /* = sizeof(pw_config); checked */
    int32_t scenario;                /* pw_scenario */
    int32_t num_envs;                /* B of THIS handle (the local shard) */
    int32_t num_agents;              /* N <= PW_MAX_AGENTS */
    int32_t num_landmarks;           /* L <= PW_MAX_LANDMARKS */
    int32_t num_adversaries;         /* simple_tag: agents [0, A) are adversaries */
    int32_t obs_mode;                /* pw_obs_mode (simple_spread only) */
    int32_t max_episode_len;         /* rls/arglist.py:5 (25); 0 = never terminal */
    int32_t auto_reset;              /* reset an env inside the step that made it terminal (run.py:59-60) */
    int32_t force_discrete_action;   /* experiments/scenarios.py:191; applies to act_vec input */
    int32_t landmark_collide;        /* simple_spread 0, simple_tag 1 */
    int32_t action_force_uses_accel; /* fork knob: p_force = mass*accel*u; canonical 0 */
    uint64_t seed;                   /* Philox key */
    uint64_t env_id_base;            /* global index of local env 0 (multi-GPU sharding) */
    float dt, damping, contact_force, contact_margin, default_sensitivity, mass, landmark_size;
    float agent_size[PW_MAX_AGENTS];
    float agent_accel[PW_MAX_AGENTS];     /* < 0: None (sensitivity = default_sensitivity) */
    float agent_max_speed[PW_MAX_AGENTS]; /* < 0: None */
} pw_config;

typedef struct pw_state_layout {
    size_t pos_x, pos_y, vel_x, vel_y, lm_x, lm_y, ep_step, ep_count; /* byte offsets */
    size_t total_bytes;
    size_t comm, goal; /* communication scenarios only: state.c [B*N*dim_c] f32 (dim_c = PW_DIM_C for simple_reference,
                          PW_SL_DIM_C for simple_speaker_listener), goal_b index [B*N] i32 */
} pw_state_layout;

/* Buffers of one step (T = 1) or of a T-step rollout (leading dimension T).
 * Any output pointer may be NULL (that output is skipped). Exactly one of
 * act_idx / act_vec is non-NULL. All pointers are device pointers. */
typedef struct pw_step_io {
    const int32_t *act_idx; /* [T,B,N] action index 0..4 (0 noop,1 +x,2 -x,3 +y,4 -y) */
    const float *act_vec;   /* [T,B,N,5] one-hot / soft action, as run.py:38 passes it */
    float *obs;             /* [T,B,N,D] what the policy sees next (post-reset where auto-reset fired) */
    float *final_obs;       /* [T,B,N,D] pre-reset observation; written only for envs that reset */
    float *rew;             /* [T,B,N] per-agent reward (world.collaborative = False) */
    float *rew_shared;      /* [T,B]   sum over agents in agent order (run.py:46) */
    uint8_t *done;          /* [T,B,N] always 0 */
    uint8_t *terminal;      /* [T,B] */
    uint64_t *coll;         /* [T,B,N] */
    const int32_t *act_comm; /* [T,B,N] communication symbol 0..PW_DIM_C-1; simple_reference with act_idx.
                                There act_vec is [T,B,N,5+PW_DIM_C]: the concatenated MultiDiscrete action of
                                experiments/run.py:39-41 (movement one-hot | communication vector).
                                simple_speaker_listener has one action head per agent (upstream environment.py:
                                Discrete(3) for the speaker, Discrete(5) for the listener): act_idx [T,B,N] =
                                (symbol 0..2 of agent 0, movement 0..4 of agent 1), or act_vec [T,B,N,5] with the
                                speaker's vector in the first three entries; act_comm stays NULL */
} pw_step_io;

int pw_version(void);
const char *pw_last_error(void);

/* Canonical upstream constants for a scenario: simple_spread (N agents, L = N
 * landmarks), simple_tag (num_adversaries + good, L = 2), simple_reference (2 speaking agents, L = 3;
 * obs = [p_vel, landmark - pos, goal_b colour, other agent's c], D = 21) or simple_speaker_listener (a fixed
 * speaker + a silent moving listener, L = 3; obs = [p_vel, landmark - pos, goal_b colour or zeros], D = 11, the
 * observation experiments/scenarios.py:45-64 patches in). Replaces Scenario.make_world() + World.__init__(). */
int pw_config_default(pw_config *cfg, int scenario, int num_envs, int num_agents,
                      int num_landmarks /* <0: scenario default */, int num_adversaries);

int pw_create(const pw_config *cfg, pw_handle **out); /* MultiAgentEnv.__init__ */
void pw_destroy(pw_handle *h);
int pw_obs_dim(const pw_handle *h);                   /* observation_space[i].shape[0] */
int pw_get_config(const pw_handle *h, pw_config *out);
/* env.force_discrete_action = ... after construction (experiments/scenarios.py:191) */
int pw_set_force_discrete_action(pw_handle *h, int on);
int pw_get_state_layout(const pw_handle *h, pw_state_layout *out);
size_t pw_state_bytes(const pw_handle *h);
int pw_bind_state(pw_handle *h, void *device_state_block);

/* AoS <-> SoA: pos/vel [B,N,2], lm [B,L,2] f32 device (upstream p_pos / p_vel arrays);
 * ep_step/ep_count [B] may be NULL (set: zeroed / get: skipped). */
int pw_set_state(pw_handle *h, const float *pos, const float *vel, const float *lm,
                 const int32_t *ep_step, const uint32_t *ep_count, void *stream);
int pw_get_state(pw_handle *h, float *pos, float *vel, float *lm,
                 int32_t *ep_step, uint32_t *ep_count, void *stream);

/* Communication scenarios only: the communication state (agent.state.c, [B,N,dim_c] f32) and each agent's goal
 * landmark index (goal_b, [B,N] i32; the listener's entry is unused).  Either pointer may be NULL. */
int pw_set_comm_state(pw_handle *h, const float *comm, const int32_t *goal, void *stream);
int pw_get_comm_state(pw_handle *h, float *comm, int32_t *goal, void *stream);

/* MultiAgentEnv.reset(): env_mask [B] u8 device or NULL (= all). Masked-in envs get
 * ep_count += 1, ep_step = 0, Philox initial state; obs (may be NULL) is written for ALL envs. */
int pw_reset(pw_handle *h, const uint8_t *env_mask, float *obs, void *stream);
/* scenario.observation for every agent from the current state. */
int pw_observe(pw_handle *h, float *obs, void *stream);
/* scenario.reward / is_collision from the current state (no step). */
int pw_reward(pw_handle *h, float *rew, uint64_t *coll, void *stream);

/* MultiAgentEnv.step(): _set_action, World.step (apply_action_force,
 * apply_environment_force/get_collision_force, integrate_state), then per agent
 * observation, reward, done -- one fused launch for all B envs. */
int pw_step(pw_handle *h, const pw_step_io *io, void *stream);
/* T consecutive steps in ONE launch (state stays in registers/LDS between steps). */
int pw_rollout(pw_handle *h, const pw_step_io *io, int num_steps, void *stream);

/* ---- kernel selection ------------------------------------------------------------------------------------------
 * The dispatcher picks a kernel form per (scenario, N, L, B, outputs requested) from measured crossovers.  Every field
 * below overrides one of those choices for ONE handle; the selection lives in the handle, is fixed by pw_create /
 * pw_set_dispatch, and nothing on the pw_step / pw_rollout / pw_policy_rollout path reads the process environment.
 * pw_create initialises it from pw_dispatch_default() overlaid, once, with the PWORLD_* environment variables of the
 * creating process (for A/B runs of unmodified host programs: PWORLD_FORCE_GENERIC, PWORLD_NO_STREAM, PWORLD_NO_DUO,
 * PWORLD_FORCE_DUO, PWORLD_NO_QUAD, PWORLD_FORCE_QUAD, PWORLD_OBS_BLOCK, PWORLD_SPREAD_TRIO / PWORLD_TAG_TRIO,
 * PWORLD_P_PRIO, PWORLD_EPW, PWORLD_POLICY_V3 / PWORLD_POLICY_V3J); pw_set_dispatch replaces it.  Results never depend
 * on it: every form produces the same bits (tests/test_gpu_parity.py runs them all against the oracle). */
typedef struct pw_dispatch {
    uint32_t struct_size;  /* = sizeof(pw_dispatch); checked */
    int32_t force_generic; /* 1: the generic kernel (pw_rollout_kernel) for every launch */
    int32_t no_stream;     /* 1: no streaming family (quad / duo / stream): pw_spread_fast_kernel, generic simple_tag */
    int32_t duo;           /* -1 auto (by grid size); 0: the one-wave stream form (and no quad form); 1: the two-wave duo form on any grid */
    int32_t quad;          /* -1 auto; 0: off; 1: pw_spread_quad_kernel wherever it applies (N = L = 6, unit mass) */
    int32_t obs_block;     /* -1 auto; 0: row-wise observation stores; 1: block-wise */
    int32_t trio;          /* -1 auto; 0 / 1: the three-wave variants of the duo kernels (spread: block-store forms, N >= 6) */
    int32_t p_prio;        /* -1 auto; >= 0: issue-priority bits of the duo kernels' waves (2 bits per wave; simple_tag: 0 / 1) */
    int32_t envs_per_wave; /* 0 auto; n >= 1: envs per wave, clamped to 64 / N */
    int32_t policy_form;   /* 0 auto; 3: pw_policy_rollout3_kernel; 4: pw_policy_rollout3j_kernel; 1, 2: retired (PW_EINVAL)
                            * (the third form with dense1 just in time: long agent axes, N <= 32) */
} pw_dispatch;
int pw_dispatch_default(pw_dispatch *d);                    /* every choice automatic */
int pw_set_dispatch(pw_handle *h, const pw_dispatch *d);    /* between launches; the bound state is untouched */
int pw_get_dispatch(const pw_handle *h, pw_dispatch *out);

/* Arithmetic of the actor inside the one-launch policy rollouts of this handle (pw_policy_rollout).  PW_ACTOR_F32 (default): exact
 * float32, bit-identical to every other form and to the pw_actor_fused + pw_step loop.  PW_ACTOR_BF16X3: OPT-IN and NOT exact -- the
 * LSTM input projection runs on bfloat16 matrix instructions with both operands split in high and low halves, three products per k
 * step, float32 accumulation; dense1, the recurrence and the head stay float32.  Within the 2e-5 bound of the PyTorch float32
 * comparison on the reference's weights (tests/test_gpu_engine.py), but sampled actions can differ from the exact form's.
 * Selected by this call only (no environment variable changes results: a process-wide switch would silently turn every "exact"
 * rollout of the process into this mode); never a default, never a headline figure.
 * Served by the simple_spread rollout in its third kernel form and by pw_actor_fused at N <= 16; anything else returns PW_EINVAL
 * rather than run in float32 unannounced.
 * pw_actor_set_bf16x3: the same switch for the handle-less pw_actor_fused (process-wide, off until set; N <= 16 only). */
#define PW_ACTOR_F32 0
#define PW_ACTOR_BF16X3 1
int pw_set_actor_precision(pw_handle *h, int32_t mode);
int pw_get_actor_precision(const pw_handle *h);
int pw_actor_set_bf16x3(int32_t on);   /* returns the previous value */

/* Name of the device kernel the last pw_step / pw_rollout on this handle launched (a static string; "" before the
 * first launch).  The dispatcher picks a kernel per (scenario, N, L, B, outputs requested): measurement tools name the
 * dominant kernel from this, not from a table of their own. */
const char *pw_rollout_kernel(const pw_handle *h);

/* Algorithmic HBM bytes of one env-step: 57N + 8L + 8NL for local obs. */
size_t pw_algorithmic_bytes_per_env_step(const pw_handle *h);

/* ---- device replay ring (rls/replay_buffer.py:9-91 ReplayBuffer) -----------------
 * Storage is caller-owned SoA: obs/next_obs [cap,N,D] f32, act [cap,N] u8 index,
 * rew [cap] f32 (shared reward), done [cap] f32.
 * Variants (zero = the plain ring above, so a memset struct keeps its old meaning):
 *   act_heads = 2: act is [cap,N,2] u8 = (movement, communication symbol) -- the MultiDiscrete action of
 *     experiments/run.py:39-41; head_width = {5, dim_c} sizes the one-hot rows pw_replay_gather returns
 *     (head_width[0] = 0 means 5);
 *   per_agent = 1: rew and done are [cap,N] f32 -- the BiCNet tuple of experiments/run_BIC.py:46,50.
 * pw_replay_add and pw_replay_gather serve every variant, pw_replay_add_rollout the plain and the two-head ring; the tail /
 * packed / wire entry points and the pw_policy_rollout sink take the plain ring only (PW_EINVAL otherwise).
 *   state_rows = 1 (0.1.6): a STATE ring.  obs / next_obs are [cap,N,4] f32 planes holding {vx, vy, px, py} of every agent (the
 *     first four columns of its observation row) and lm [cap,L,2] the landmarks of the transition's episode: 32N + 8L bytes per
 *     transition instead of 8ND (C2: 240 instead of 768) -- what the learner rank writes per gathered transition.  obs_dim stays D:
 *     pw_replay_gather REBUILDS the rows it returns ([b,N,D]; every entry is the state itself or one float32 subtraction, so the
 *     batch is bit-identical to the row ring's).  scenario / num_landmarks / num_adversaries name the row layout: PW_SIMPLE_SPREAD
 *     with the local observation (D = 4 + 2L) or PW_SIMPLE_TAG (D = 4 + 2L + 2(N - 1) + 2(N - A), good agents' rows zero-padded).
 *     Filled by pw_replay_add_state_wire and by the pw_policy_rollout ring sink (which hold the state itself); the writers that are
 *     handed ROWS return PW_EINVAL (a row cannot be turned back into the landmarks it was built from).  Plain single-head, shared-reward rings only. */
typedef struct pw_replay_store {
    float *obs, *next_obs, *rew, *done;
    uint8_t *act;
    int64_t capacity;
    int32_t num_agents, obs_dim;
    int32_t act_heads, per_agent;
    int32_t head_width[2];
    int32_t state_rows, num_landmarks, scenario, num_adversaries; /* 0.1.6; all zero = the row ring of before */
    float *lm;                                                    /* state ring: [cap,L,2] */
} pw_replay_store;

/* add(): append B transitions at ring positions (start + i) % capacity.
 * next_obs row i comes from final_obs where terminal[i] != 0 (and final_obs != NULL).
 * act_idx is [B,N] int32 ([B,N,2] when st->act_heads = 2); rew_shared / done are [B] ([B,N] when st->per_agent). */
int pw_replay_add(const pw_replay_store *st, int64_t start, const int64_t *start_dev /* device, or NULL */,
                  int32_t B, const float *obs, const int32_t *act_idx, const float *rew_shared,
                  const float *next_obs, const float *final_obs, const uint8_t *terminal,
                  const float *done /* [B] or NULL = 0 */, void *stream);
/* hipGraph support: a captured launch freezes by-value arguments; the two values that change every
 * step (ring position, Philox step) can instead live in device memory (start_dev / step_dev override the
 * by-value argument when non-NULL) and be advanced by this one-thread launch inside the same graph:
 * *counter = (*counter + delta) % modulo (modulo <= 0: no wrap). */
int pw_counter_add(int64_t *counter, int64_t delta, int64_t modulo, void *stream);
/* sample_index() / _encode_sample(): gather rows idx[0..b) into dense batch tensors;
 * out_act is one-hot f32 [b,N,5] exactly as the reference's trainer consumes it ([b,N,head_width[0]+head_width[1]]
 * for a two-head ring); out_rew / out_done are [b] ([b,N] for a per-agent ring).  (Declared below.) */
/* pw_replay_add + pw_episode_stats in ONE launch (the last launch of a captured rollout step).  The ring
 * position comes from `start` or, if non-NULL, *start_dev; (position + B) % capacity is written to
 * *next_start_dev (optional; must not alias start_dev: double-buffer the cursor) and *step_counter (optional,
 * e.g. the policy's Philox step) is incremented.  Episode-return arithmetic is identical to pw_episode_stats. */
int pw_replay_add_tail(const pw_replay_store *st, int64_t start, const int64_t *start_dev, int64_t *next_start_dev,
                       int32_t B, const float *obs, const int32_t *act_idx, const float *rew_shared,
                       const float *next_obs, const float *final_obs, const uint8_t *terminal, const float *done,
                       float *episode_return, double *finished_sum, int64_t *finished_count, int64_t *step_counter,
                       void *stream);
/* A whole rollout chunk (pw_policy_rollout / pw_rollout outputs, [T, ...]) into the ring in one launch: transition
 * (t, e) goes to slot (start + t*B + e) % capacity -- the order of T pw_replay_add calls -- with obs = obs0 [B,N,D]
 * for t = 0 and the chunk's obs[t-1] after that, next_obs = final_obs where terminal.  io needs obs, rew_shared,
 * terminal (final_obs optional); act [T,B,N] int32 ([T,B,N,2] for a two-head ring, st->act_heads = 2: the MultiDiscrete chunks
 * of pw_policy_rollout on simple_reference).  episode_return / finished_sum / finished_count / scratch
 * (all or none): the chunk's episode-return bookkeeping, as T pw_episode_stats calls up to float64 summation
 * order (fixed, so reproducible); scratch = pw_replay_add_rollout_scratch_bytes(B) device bytes, zeroed once. */
int pw_replay_add_rollout(const pw_replay_store *st, int64_t start, int32_t B, int32_t T, const float *obs0,
                          const pw_step_io *io, const int32_t *act, float *episode_return, double *finished_sum,
                          int64_t *finished_count, void *scratch, void *stream);
size_t pw_replay_add_rollout_scratch_bytes(int32_t B);
int pw_replay_gather(const pw_replay_store *st, const int64_t *idx, int32_t b,
                     float *out_obs, float *out_act, float *out_rew, float *out_next_obs,
                     float *out_done, void *stream);

/* ---- multi-GPU exchange (one rank per GPU; the collective itself is RCCL, driven by the host) ----
 * A transition row is f32 [obs N*D | next_obs N*D | act N | rew | done], width 2*N*D + N + 2.
 * pw_pack_transitions: row r = transition (sel_t[r] >= 1, sel_e[r]) of a T-step chunk described by
 * io (obs[t-1] is the observation acted on; next_obs is final_obs where terminal, else obs[t]).
 * pw_replay_add_packed: ReplayBuffer.add() of R received rows at ring positions (start + r) % capacity. */
int pw_pack_transitions(const pw_step_io *io, int32_t B, int32_t N, int32_t D, const int32_t *sel_t,
                        const int32_t *sel_e, int32_t R, float *rows, void *stream);
int pw_replay_add_packed(const pw_replay_store *st, int64_t start, int32_t R, const float *rows, void *stream);
/* Both of the above in ONE launch (either half may be disabled: st/rows_in NULL or io/rows_out NULL):
 * append the R_in rows of the previous collective to the ring, pack R_out rows of this chunk for the next. */
int pw_exchange(const pw_replay_store *st, int64_t start, int32_t R_in, const float *rows_in, const pw_step_io *io,
                int32_t B, int32_t N, int32_t D, const int32_t *sel_t, const int32_t *sel_e, int32_t R_out,
                float *rows_out, void *stream);

/* ---- full gather of transitions to the learner rank (north_star: "RCCL-over-xGMI gather of transitions into
 * rls/replay_buffer"; consumer experiments/run.py:20-21,52) ------------------------------------------------------
 * One WIRE BLOCK per rank per T-step rollout chunk carries EVERY transition of the chunk once: the observation
 * is sent once per step (next_obs of step t is obs of step t+1), plus the pre-reset observation only for the steps
 * that ended an episode (run.py:52 stores new_obs_n BEFORE env.reset(), :60).  Planes, 256-B aligned:
 *   obs0 [B,N,D] f32        observation the policy acted on at step 0
 *   obs [T,B,N,D] f32       post-step (post-reset) observations -- the rollout kernels write them HERE, no copy
 *   final_rows [F,B,N,D] f32  pre-reset observation of env e's k-th episode end inside the chunk,
 *                           F = ceil(T / max_episode_len) (0 when episodes never end)
 *   rew_shared [T,B] f32    written in place by the rollout as well
 *   act [T,B,N] u8          action index
 *   fin_slot [T,B] u8       k where step (t, e) ended an episode, 0xFF elsewhere
 * = N*D*4 + N + 5 bytes per env-step (C2: 395 B) + (1 + F) observation batches per chunk (C2,
 * T = 100: 414 B/env-step in all).
 * The collective itself is RCCL, driven by the host: direct peer -> root sends, one block per peer per chunk. */
typedef struct pw_chunk_wire {
    int32_t T, B, N, D, F, reserved;
    size_t obs0, obs, final_rows, rew_shared, act, fin_slot; /* byte offsets into the block */
    size_t total_bytes;
} pw_chunk_wire;
int pw_chunk_wire_layout(int32_t T, int32_t B, int32_t N, int32_t D, int32_t max_episode_len, pw_chunk_wire *out);
/* Sender, after the chunk's rollout wrote obs / rew_shared into the block: fills obs0, final_rows, act, fin_slot
 * from obs0 [B,N,D], the rollout's dense final_obs [T,B,N,D] (may be NULL when F = 0), terminal [T,B] and
 * act [T,B,N] int32.  One launch. */
int pw_chunk_wire_finalize(const pw_chunk_wire *w, void *wire, const float *obs0, const float *final_obs,
                           const uint8_t *terminal, const int32_t *act, void *stream);
/* Root: ReplayBuffer.add() of the block's T*B transitions; transition (t, e) goes to ring slot
 * (start + t*B + e) % capacity -- the order T pw_replay_add calls would have used -- bit-identical to
 * pw_replay_add_rollout on the sender's buffers.  One launch. */
int pw_replay_add_wire(const pw_replay_store *st, int64_t start, const pw_chunk_wire *w, const void *wire, void *stream);

/* ---- the same gather on a diet: STATE-ONLY wire blocks (simple_spread, local observation) ---------------------------
 * A local observation row is a pure function of the agent's {vel, pos} and the episode's landmarks
 * (experiments/scenarios.py:6-20: [p_vel, p_pos, landmark.p_pos - p_pos ...]); each entry is ONE float32 operation, so the
 * root can rebuild obs / next_obs rows bit for bit from 16 bytes per agent instead of receiving 4 + 2L floats per agent
 * (C2: 384 B of the 395 B per env-step of the row block above).  Planes, 256-B aligned:
 *   state0 [B,N] float4 {vx, vy, px, py}      the state the policy acted on at step 0 (written by pw_state_wire_begin)
 *   state [T,B,N] float4                      post-step (post-reset) state = columns 0..3 of the rollout's obs rows
 *   final_state [F,B,N] float4                pre-reset state of env e's k-th episode end inside the chunk
 *   lm [(F+1),B,L] float2                     landmarks of the episode in progress at step 0 (k = 0, copied from the state
 *                                             block by pw_state_wire_begin) and after the k-th reset (k >= 1: re-derived by
 *                                             pw_state_wire_finalize from the reset's Philox key (seed, global env id,
 *                                             episode number) -- the in-kernel auto-reset draws exactly these)
 *   ep0 [B] u32                               episode number of every env at step 0 (pw_state_wire_begin; sender-side only,
 *                                             it travels with the block and lets the root audit the landmark planes)
 *   rew_shared [T,B] f32                      written in place by the rollout
 *   act [T,B,N] u8
 *   epi [T,B] u8                              bits 0..6: k = episode ends of env e before step t (which lm / final_state
 *                                             plane applies), bit 7: step (t, e) ended an episode
 * = 17N + 5 bytes per env-step + (1 + F) state and landmark batches per chunk (C2, T = 100: 107 + 7 = 114 B per
 * env-step, 28 % of the row block's 414).
 * Sender:  pw_state_wire_begin (BEFORE the chunk's rollout launch, same stream)  ->  rollout with rew_shared pointing into
 * the block  ->  pw_state_wire_finalize.  Root: pw_replay_add_state_wire = ReplayBuffer.add() of the block's T*B transitions,
 * bit-identical to pw_replay_add_rollout on the sender's buffers (tests/test_gpu_engine.py). */
typedef struct pw_state_wire {
    int32_t T, B, N, L, D, F;
    size_t state0, state, final_state, lm, ep0, rew_shared, act, epi; /* byte offsets into the block */
    size_t total_bytes;
    int32_t scenario, num_adversaries; /* 0.1.6: PW_SIMPLE_SPREAD (0: as before) or PW_SIMPLE_TAG with A adversaries */
} pw_state_wire;
/* Host arithmetic only (as pw_chunk_wire_layout).  begin / finalize return PW_EINVAL unless h is a handle of the block's scenario,
 * B, N, L (and A) whose rows are a function of the state: simple_spread with the local observation (D = 4 + 2L), or simple_tag
 * (0.1.6; D = 4 + 2L + 2(N - 1) + 2(N - A): [vel, pos, landmark - pos .., other pos - pos .., other good agents' vel ..], rows of good
 * agents zero-padded; landmarks drawn from U(-0.9, 0.9): C3's 4 + 2 roster ships 17N + 5 = 107 B per env-step + (1 + F) state /
 * landmark batches per chunk instead of the row block's 4ND + N + 5 = 539). */
int pw_state_wire_layout(int32_t T, int32_t B, int32_t N, int32_t L, int32_t max_episode_len, pw_state_wire *out); /* simple_spread */
int pw_state_wire_layout_scn(int32_t scenario, int32_t T, int32_t B, int32_t N, int32_t L, int32_t num_adversaries,
                             int32_t max_episode_len, pw_state_wire *out);
int pw_state_wire_begin(const pw_handle *h, const pw_state_wire *w, void *wire, void *stream);
/* obs [T,B,N,D] and final_obs [T,B,N,D] (may be NULL when F = 0) are the rollout's outputs, terminal [T,B], act [T,B,N]
 * int32.  One launch. */
int pw_state_wire_finalize(const pw_handle *h, const pw_state_wire *w, void *wire, const float *obs, const float *final_obs,
                           const uint8_t *terminal, const int32_t *act, void *stream);
/* Root; needs no handle: the rebuild is the observation's own arithmetic on the block's data.  One launch.  st may be a row ring
 * (rows rebuilt here) or a STATE ring of the same scenario / N / L / A (states and landmarks copied; rows rebuilt by
 * pw_replay_gather when a batch is sampled). */
int pw_replay_add_state_wire(const pw_replay_store *st, int64_t start, const pw_state_wire *w, const void *wire, void *stream);

/* ---- the gather for simple_reference (the MultiDiscrete scenario of main.py:24,52-54; 0.1.6): COMPACT-ROW wire blocks ------------
 * Its 21-number row is [p_vel (2), landmark - p_pos (3 x 2), goal_b colour (3), the other agent's communication state (10)]
 * (experiments/scenarios.py:23-42).  The last 13 numbers are not arithmetic at all: the colour is a constant of the agent's goal
 * landmark (fixed for an episode) and the communication state is the one-hot of the symbol the OTHER agent sampled in this step
 * (zeros after a reset; upstream update_agent_state copies action.c for a speaking agent).  So a block carries the first EIGHT
 * numbers of every row as they are (the position itself is not in the row, and landmark - pos cannot be turned back into it), one
 * goal byte per agent and episode, and both action heads as bytes; the root rebuilds obs / next_obs bit for bit.  Planes, 256-B aligned:
 *   head0 [B,2,8] f32         columns 0..7 of the rows the policy acted on at step 0
 *   head [T,B,2,8] f32        columns 0..7 of the post-step (post-reset) rows
 *   final_head [F,B,2,8] f32  columns 0..7 of the pre-reset rows of env e's k-th episode end inside the chunk
 *   goal [(F+1),B,2] u8       goal landmark of each agent in the episode in progress at step 0 (k = 0) and after the k-th reset
 *   comm0 [B,2] u8            the symbol visible in agent a's row at step 0 (0xFF: none, i.e. zeros)
 *   rew_shared [T,B] f32      written in place by the rollout
 *   act [T,B,2,2] u8          (movement, symbol) of every agent
 *   epi [T,B] u8              as in pw_state_wire
 * = 73 bytes per env-step + (1 + F) head / goal batches per chunk (T = 100: 77 B; the row block: 181).
 * Sender: pw_ref_wire_finalize after the chunk's rollout (rew_shared pointing into the block); it reads the rollout's outputs only
 * (no handle).  Root: pw_replay_add_ref_wire into the TWO-HEAD ring (act_heads = 2, head widths 5 | PW_DIM_C), bit-identical to
 * pw_replay_add_rollout on the sender's buffers.  The rows must come from HARD symbol indices (pw_policy_rollout, or pw_rollout with
 * act_idx / act_comm): a soft communication vector (act_vec) is not representable -- use pw_chunk_wire_* then. */
typedef struct pw_ref_wire {
    int32_t T, B, F, reserved;
    size_t head0, head, final_head, goal, comm0, rew_shared, act, epi; /* byte offsets into the block */
    size_t total_bytes;
} pw_ref_wire;
int pw_ref_wire_layout(int32_t T, int32_t B, int32_t max_episode_len, pw_ref_wire *out);
/* obs0 [B,2,21], obs / final_obs [T,B,2,21] (final_obs may be NULL when F = 0), terminal [T,B], act [T,B,2,2] int32.  One launch. */
int pw_ref_wire_finalize(const pw_ref_wire *w, void *wire, const float *obs0, const float *obs, const float *final_obs,
                         const uint8_t *terminal, const int32_t *act, void *stream);
int pw_replay_add_ref_wire(const pw_replay_store *st, int64_t start, const pw_ref_wire *w, const void *wire, void *stream);

/* Episode bookkeeping of the rollout loop (experiments/run.py:55-65) over B envs in one launch:
 * episode_return[b] += rew_shared[b]; where terminal[b]: *finished_sum += return (double),
 * *finished_count += 1, return cleared.  Deterministic (single workgroup, fixed-order reduction). */
int pw_episode_stats(const float *rew_shared, const uint8_t *terminal, int32_t B, float *episode_return,
                     double *finished_sum, int64_t *finished_count, void *stream);
/* pw_episode_stats that also advances up to two device-side counters (NULL to skip; value = (value + delta) %
 * modulo, modulo 0 = no wrap) -- e.g. the replay ring cursor and the policy's Philox step of a captured
 * rollout step, whose earlier launches have all read them by the time this one runs (stream order). */
int pw_rollout_tail(const float *rew_shared, const uint8_t *terminal, int32_t B, float *episode_return,
                    double *finished_sum, int64_t *finished_count, int64_t *counter0, int64_t delta0, int64_t modulo0,
                    int64_t *counter1, int64_t delta1, int64_t modulo1, void *stream);

/* ---- action producer (rls/model/ac_network_multi_gumbel.py:24-67, ddpg_gumbel_fix.py:86-116) --------
 * The actor is Linear(D,64)-ReLU-BiLSTM(64->2x32 over the AGENT axis)-ReLU-Linear(64,5).  The two input
 * GEMMs stay in rocBLAS; these two launches replace MIOpen's many-kernel RNN path and the sampling:
 * pw_bilstm_forward: G [B,N,2,128] = x*W_ih^T + b_ih + b_hh per direction (PyTorch gate order i,f,g,o;
 *   direction 1 = reverse), w_hh_* [128,32] row-major -> H [B,N,64] = [h_forward | h_reverse], ReLU'd
 *   if relu_out (the next layer applies F.relu).  Hidden size is the reference's fixed 32.
 * pw_actor_head: logits [rows,5] = H*W2^T + b2 (optional output) and act[rows] = argmax(logits + g),
 *   g = -log(-log(u)) Gumbel noise from Philox4x32-10 keyed (seed; step, row): the hard one-hot of
 *   F.gumbel_softmax(hard=True) kept as an int32 index on the device. */
/* Y [rows,out_dim] = act(X [rows,in_dim] * W^T + b), W [out_dim,in_dim] row-major, in_dim <= 64, out_dim a
 * multiple of 64, act = ReLU if relu.  ActorNetwork.dense1 + F.relu, and the LSTM input projection
 * x * [W_ih; W_ih_reverse]^T + (b_ih + b_hh) that pw_bilstm_forward consumes. */
int pw_dense(const float *X, const float *W, const float *b, int64_t rows, int32_t in_dim, int32_t out_dim,
             int32_t relu, float *Y, void *stream);
/* G [rows,256] = relu(X [rows,in_dim] * W1^T + b1) * Wih^T + bih in one launch on the matrix cores
 * (v_mfma_f32_32x32x2_f32, exact float32): dense1 + F.relu + the input projections of both LSTM directions.
 * The weights are consumed in MFMA fragment order: pw_actor_front_pack writes that image (w1 [64,in_dim],
 * w_ih [256,64] = [W_ih; W_ih_reverse] row-major -> frag[pw_actor_front_pack_floats(in_dim)]) once per weight
 * update; b_ih [256] = b_ih + b_hh per direction. */
size_t pw_actor_front_pack_floats(int32_t in_dim);
int pw_actor_front_pack(const float *w1, const float *w_ih, int32_t in_dim, float *frag, void *stream);
int pw_actor_front(const float *X, const float *frag, const float *b1, const float *b_ih, int64_t rows, int32_t in_dim,
                   float *G, void *stream);
int pw_bilstm_forward(const float *G, const float *w_hh_fw, const float *w_hh_bw, int32_t B, int32_t N,
                      int32_t relu_out, float *H, void *stream);
int pw_actor_head(const float *H, const float *w2, const float *b2, int64_t rows, uint64_t seed, uint64_t step,
                  const int64_t *step_dev /* device, or NULL */, float *logits, int32_t *act, void *stream);

/* The whole actor in ONE launch: X [B,N,in_dim] observations -> act (Gumbel-argmax index per head), and/or
 * logits, H [B,N,64] (each optional, NULL to skip).  One head: w2 [n_out0,64], b2 [n_out0], n_out1 = 0,
 * logits [B,N,n_out0], act [B,N].  Two heads (MultiDiscrete actors, main.py:52-54: dense2_1 / dense2_2): w2 / b2 =
 * the two layers concatenated, logits [B,N,n_out0+n_out1] in that order (run.py:39-41), act [B,N,2].
 * n_out0 + n_out1 <= 16.  With n_out0 = 5, n_out1 = 0: same arithmetic and the same Philox keying as
 * pw_actor_front + pw_bilstm_forward + pw_actor_head chained (identical results); G and H stay in LDS.
 * frag = pw_actor_front_pack's image; N <= 96. */
int pw_actor_fused(const float *X, const float *frag, const float *b1, const float *b_ih, const float *w_hh_fw,
                   const float *w_hh_bw, const float *w2, const float *b2, int32_t n_out0, int32_t n_out1, int64_t B,
                   int32_t N, int32_t in_dim, int32_t relu_out, uint64_t seed, uint64_t step,
                   const int64_t *step_dev /* device, or NULL */, float *H, float *logits, int32_t *act, void *stream);

/* Optional direct sink of pw_policy_rollout: the chunk's transitions go straight into the replay ring (slot
 * (ring_start + t*B + env) % capacity -- the order of T pw_replay_add calls; next_obs = the PRE-reset observation)
 * and the episode returns are kept in the same launch (episode_return [B] running returns, finished_sum /
 * finished_count accumulated reproducibly; scratch = pw_policy_rollout_scratch_bytes(h) device bytes, zeroed once).
 * ring may be NULL (bookkeeping only) and episode_return may be NULL (ring only).  0.1.6: the ring may be a STATE ring
 * (pw_replay_store.state_rows) of the handle's scenario (simple_spread with the local observation, simple_tag): the launch then leaves
 * {vel, pos} of every agent before / after the step and the episode's landmarks per transition -- 254 B at C2 instead of 782 -- and
 * pw_replay_gather rebuilds the rows when a batch is sampled (bit-identical to the row ring's batch). */
typedef struct pw_rollout_sink {
    const pw_replay_store *ring;
    int64_t ring_start;
    float *episode_return;
    double *finished_sum;
    int64_t *finished_count;
    void *scratch;
} pw_rollout_sink;

/* Policy-in-the-loop rollout as ONE launch: num_steps x (actor forward + Gumbel sampling + environment step +
 * auto-reset) on the handle's bound state, starting from the observation of the current state; observations,
 * sampled actions and world state stay on the CU between steps.  io: the pw_rollout outputs ([num_steps, ...];
 * no action inputs, no coll); act_out [num_steps,B,N] int32 receives the sampled indices.  Without a ring sink
 * act_out and obs, rew, rew_shared, done, terminal are required (final_obs optional); with one every output is
 * optional.  The Gumbel noise of step t is keyed (seed; step + t, row) exactly as pw_actor_fused / pw_actor_head, so
 * the results equal a loop of pw_actor_fused + pw_step (+ pw_replay_add_tail).
 * simple_spread fast-path configurations (local observation, homogeneous agents, L <= N; observation rows up to
 * D = 104, i.e. N = L <= 50: rows longer than 64 numbers on the just-in-time kernel form only) and simple_tag with homogeneous roles (9 <= D <= 48; good agents' rows zero-padded to D), one
 * 5-logit head; weights as for pw_actor_fused.
 * simple_reference (the MultiDiscrete scenario of main.py:24,52-54; 3 landmarks, D = 21): the two-head actor -- w2 [5 + PW_DIM_C,
 * 64] / b2 = dense2_1 and dense2_2 concatenated, one Gumbel-argmax per head exactly as pw_actor_fused(n_out0 = 5, n_out1 =
 * PW_DIM_C) -- and act_out [num_steps,B,N,2] = (movement, symbol); the ring sink must be the two-head ring (act_heads = 2, head widths
 * 5 | PW_DIM_C; 0.1.5 -- before, the chunk went into it with a second launch, pw_replay_add_rollout); results equal a loop of
 * pw_actor_fused + pw_step(act_idx, act_comm). */
int pw_policy_rollout(pw_handle *h, const float *frag, const float *b1, const float *b_ih, const float *w_hh_fw,
                      const float *w_hh_bw, const float *w2, const float *b2, int32_t relu_out, uint64_t seed,
                      uint64_t step, const int64_t *step_dev /* device, or NULL */, const pw_step_io *io,
                      int32_t *act_out, int32_t num_steps, const pw_rollout_sink *sink /* or NULL */, void *stream);
size_t pw_policy_rollout_scratch_bytes(const pw_handle *h);

/* Test hook: y[i] = f(x[i]) with the DEVICE implementation of one math primitive, so its bits can be compared
 * with a CPU implementation of pworld_math.h.  fn: 0 the kernels' fast correctly-rounded sqrt, 1 their
 * branch-free softplus, 2 pw_softplus, 3 pw_exp, 4 sqrtf, 5 x / aux (IEEE division), 6 / 7 the hot loops'
 * scaling-free division chain for x / aux and aux / x, 8 their softplus, 9 aux / x (IEEE division), 10 x / aux with ONE
 * correction step (what the C2 kernel runs for the division by the contact margin when pw_margin_one_correction(aux)),
 * 11 / 12 the hot loops' correctly rounded sqrt for x in [2^-90, 2^90) (one fused correction) / the two-test form it replaced. */
int pw_debug_math(int32_t fn, const float *x, float aux, float *y, int64_t n, void *stream);
/* 1 if dividing by this contact margin with one Newton correction is IEEE division (its refined reciprocal is the correctly
 * rounded one, from every 1-ulp-accurate starting value; significand not all ones; inside the division chain's range): the
 * host-side decision behind the K1 kernel instantiations.  The canonical margin 1e-3 qualifies. */
int pw_margin_one_correction(float contact_margin);

#ifdef __cplusplus
}
#endif
#endif /* PWORLD_H */
