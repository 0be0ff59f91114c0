"""ctypes binding of libpworld.so (the C ABI declared in include/pworld.h).

There is no CPU fallback: if the HIP library is missing or fails to load this
raises, loudly.  Build it with ``python -m multiagent_rl_amd.build_native`` (or
``__graft_entry__.build()``).
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libpworld.so')

PW_MAX_AGENTS = 64
PW_MAX_LANDMARKS = 64
PW_SIMPLE_SPREAD, PW_SIMPLE_TAG, PW_SIMPLE_REFERENCE, PW_SIMPLE_SPEAKER_LISTENER = 0, 1, 2, 3
PW_DIM_C = 10
PW_SL_DIM_C = 3
PW_OBS_LOCAL, PW_OBS_FULL = 0, 1
SCENARIOS = {'simple_spread': PW_SIMPLE_SPREAD, 'simple_tag': PW_SIMPLE_TAG, 'simple_reference': PW_SIMPLE_REFERENCE,
             'simple_speaker_listener': PW_SIMPLE_SPEAKER_LISTENER}


class PwConfig(C.Structure):
    _fields_ = [
        ('struct_size', C.c_uint32), ('scenario', C.c_int32), ('num_envs', C.c_int32),
        ('num_agents', C.c_int32), ('num_landmarks', C.c_int32), ('num_adversaries', C.c_int32),
        ('obs_mode', C.c_int32), ('max_episode_len', C.c_int32), ('auto_reset', C.c_int32),
        ('force_discrete_action', C.c_int32), ('landmark_collide', C.c_int32),
        ('action_force_uses_accel', C.c_int32),
        ('seed', C.c_uint64), ('env_id_base', C.c_uint64),
        ('dt', C.c_float), ('damping', C.c_float), ('contact_force', C.c_float),
        ('contact_margin', C.c_float), ('default_sensitivity', C.c_float), ('mass', C.c_float),
        ('landmark_size', C.c_float),
        ('agent_size', C.c_float * PW_MAX_AGENTS),
        ('agent_accel', C.c_float * PW_MAX_AGENTS),
        ('agent_max_speed', C.c_float * PW_MAX_AGENTS),
    ]


class PwStateLayout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in
                ('pos_x', 'pos_y', 'vel_x', 'vel_y', 'lm_x', 'lm_y', 'ep_step', 'ep_count', 'total_bytes', 'comm', 'goal')]


class PwStepIO(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ('act_idx', 'act_vec', 'obs', 'final_obs', 'rew', 'rew_shared', 'done', 'terminal', 'coll', 'act_comm')]


class PwDispatch(C.Structure):
    """pw_dispatch: the kernel selection of one handle (include/pworld.h); -1 / 0 = automatic."""
    _fields_ = [('struct_size', C.c_uint32)] + [(n, C.c_int32) for n in
                ('force_generic', 'no_stream', 'duo', 'quad', 'obs_block', 'trio', 'p_prio', 'envs_per_wave', 'policy_form')]


class PwRolloutSink(C.Structure):
    _fields_ = [('ring', C.c_void_p), ('ring_start', C.c_int64), ('episode_return', C.c_void_p),
                ('finished_sum', C.c_void_p), ('finished_count', C.c_void_p), ('scratch', C.c_void_p)]


class PwReplayStore(C.Structure):
    _fields_ = [('obs', C.c_void_p), ('next_obs', C.c_void_p), ('rew', C.c_void_p), ('done', C.c_void_p),
                ('act', C.c_void_p), ('capacity', C.c_int64), ('num_agents', C.c_int32), ('obs_dim', C.c_int32),
                ('act_heads', C.c_int32), ('per_agent', C.c_int32), ('head_width', C.c_int32 * 2),
                # 0.1.6: STATE ring (obs / next_obs = [cap,N,4] states, lm = [cap,L,2]; pw_replay_gather rebuilds the rows)
                ('state_rows', C.c_int32), ('num_landmarks', C.c_int32), ('scenario', C.c_int32), ('num_adversaries', C.c_int32),
                ('lm', C.c_void_p)]


class PwChunkWire(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('T', 'B', 'N', 'D', 'F', 'reserved')] + \
               [(n, C.c_size_t) for n in ('obs0', 'obs', 'final_rows', 'rew_shared', 'act', 'fin_slot', 'total_bytes')]


class PwStateWire(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('T', 'B', 'N', 'L', 'D', 'F')] + \
               [(n, C.c_size_t) for n in ('state0', 'state', 'final_state', 'lm', 'ep0', 'rew_shared', 'act', 'epi', 'total_bytes')] + \
               [('scenario', C.c_int32), ('num_adversaries', C.c_int32)]


class PwRefWire(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('T', 'B', 'F', 'reserved')] + \
               [(n, C.c_size_t) for n in ('head0', 'head', 'final_head', 'goal', 'comm0', 'rew_shared', 'act', 'epi', 'total_bytes')]


# name -> (restype, argtypes): every symbol include/pworld.h declares
SIGNATURES = {
    'pw_version': (C.c_int, []),
    'pw_last_error': (C.c_char_p, []),
    'pw_config_default': (C.c_int, [C.POINTER(PwConfig), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    'pw_create': (C.c_int, [C.POINTER(PwConfig), C.POINTER(C.c_void_p)]),
    'pw_destroy': (None, [C.c_void_p]),
    'pw_obs_dim': (C.c_int, [C.c_void_p]),
    'pw_get_config': (C.c_int, [C.c_void_p, C.POINTER(PwConfig)]),
    'pw_set_force_discrete_action': (C.c_int, [C.c_void_p, C.c_int]),
    'pw_dispatch_default': (C.c_int, [C.POINTER(PwDispatch)]),
    'pw_set_dispatch': (C.c_int, [C.c_void_p, C.POINTER(PwDispatch)]),
    'pw_get_dispatch': (C.c_int, [C.c_void_p, C.POINTER(PwDispatch)]),
    'pw_set_actor_precision': (C.c_int, [C.c_void_p, C.c_int32]),
    'pw_get_actor_precision': (C.c_int, [C.c_void_p]),
    'pw_actor_set_bf16x3': (C.c_int, [C.c_int32]),
    'pw_get_state_layout': (C.c_int, [C.c_void_p, C.POINTER(PwStateLayout)]),
    'pw_state_bytes': (C.c_size_t, [C.c_void_p]),
    'pw_bind_state': (C.c_int, [C.c_void_p, C.c_void_p]),
    'pw_set_state': (C.c_int, [C.c_void_p] + [C.c_void_p] * 5 + [C.c_void_p]),
    'pw_get_state': (C.c_int, [C.c_void_p] + [C.c_void_p] * 5 + [C.c_void_p]),
    'pw_set_comm_state': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'pw_get_comm_state': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'pw_reset': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'pw_observe': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    'pw_reward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'pw_step': (C.c_int, [C.c_void_p, C.POINTER(PwStepIO), C.c_void_p]),
    'pw_rollout': (C.c_int, [C.c_void_p, C.POINTER(PwStepIO), C.c_int, C.c_void_p]),
    'pw_rollout_kernel': (C.c_char_p, [C.c_void_p]),
    'pw_algorithmic_bytes_per_env_step': (C.c_size_t, [C.c_void_p]),
    'pw_replay_add': (C.c_int, [C.POINTER(PwReplayStore), C.c_int64, C.c_void_p, C.c_int32] + [C.c_void_p] * 7 + [C.c_void_p]),
    'pw_counter_add': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    'pw_pack_transitions': (C.c_int, [C.POINTER(PwStepIO), C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                     C.c_int32, C.c_void_p, C.c_void_p]),
    'pw_replay_add_packed': (C.c_int, [C.POINTER(PwReplayStore), C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    'pw_exchange': (C.c_int, [C.POINTER(PwReplayStore), C.c_int64, C.c_int32, C.c_void_p, C.POINTER(PwStepIO), C.c_int32,
                             C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    'pw_chunk_wire_layout': (C.c_int, [C.c_int32] * 5 + [C.POINTER(PwChunkWire)]),
    'pw_chunk_wire_finalize': (C.c_int, [C.POINTER(PwChunkWire)] + [C.c_void_p] * 6),
    'pw_replay_add_wire': (C.c_int, [C.POINTER(PwReplayStore), C.c_int64, C.POINTER(PwChunkWire), C.c_void_p, C.c_void_p]),
    'pw_state_wire_layout': (C.c_int, [C.c_int32] * 5 + [C.POINTER(PwStateWire)]),
    'pw_state_wire_layout_scn': (C.c_int, [C.c_int32] * 7 + [C.POINTER(PwStateWire)]),
    'pw_state_wire_begin': (C.c_int, [C.c_void_p, C.POINTER(PwStateWire), C.c_void_p, C.c_void_p]),
    'pw_state_wire_finalize': (C.c_int, [C.c_void_p, C.POINTER(PwStateWire)] + [C.c_void_p] * 6),
    'pw_replay_add_state_wire': (C.c_int, [C.POINTER(PwReplayStore), C.c_int64, C.POINTER(PwStateWire), C.c_void_p, C.c_void_p]),
    'pw_ref_wire_layout': (C.c_int, [C.c_int32] * 3 + [C.POINTER(PwRefWire)]),
    'pw_ref_wire_finalize': (C.c_int, [C.POINTER(PwRefWire)] + [C.c_void_p] * 7),
    'pw_replay_add_ref_wire': (C.c_int, [C.POINTER(PwReplayStore), C.c_int64, C.POINTER(PwRefWire), C.c_void_p, C.c_void_p]),
    'pw_dense': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                          C.c_void_p]),
    'pw_actor_front_pack_floats': (C.c_size_t, [C.c_int32]),
    'pw_actor_front_pack': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    'pw_actor_front': (C.c_int, [C.c_void_p] * 4 + [C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    'pw_actor_fused': (C.c_int, [C.c_void_p] * 8 + [C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                                    C.c_uint64, C.c_uint64] + [C.c_void_p] * 5),
    'pw_policy_rollout': (C.c_int, [C.c_void_p] * 8 + [C.c_int32, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                                       C.c_int32, C.c_void_p, C.c_void_p]),
    'pw_policy_rollout_scratch_bytes': (C.c_size_t, [C.c_void_p]),
    'pw_bilstm_forward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                   C.c_void_p]),
    'pw_actor_head': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_uint64, C.c_uint64, C.c_void_p,
                               C.c_void_p, C.c_void_p, C.c_void_p]),
    'pw_episode_stats': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'pw_replay_add_rollout': (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32] + [C.c_void_p] * 8),
    'pw_replay_add_rollout_scratch_bytes': (C.c_size_t, [C.c_int32]),
    'pw_replay_add_tail': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32] + [C.c_void_p] * 12),
    'pw_rollout_tail': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    'pw_debug_math': (C.c_int, [C.c_int32, C.c_void_p, C.c_float, C.c_void_p, C.c_int64, C.c_void_p]),
    'pw_margin_one_correction': (C.c_int, [C.c_float]),
    'pw_replay_gather': (C.c_int, [C.POINTER(PwReplayStore), C.c_void_p, C.c_int32] + [C.c_void_p] * 5 + [C.c_void_p]),
}


class PworldError(RuntimeError):
    pass


_lib = None


def load():
    """Load libpworld.so and bind every exported symbol.  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PworldError(
            'libpworld.so not found at %s: the HIP extension is required (no CPU fallback). '
            'Build it with `python -m multiagent_rl_amd.build_native`.' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise PworldError('libpworld error %d: %s' % (rc, load().pw_last_error().decode()))
    return rc
