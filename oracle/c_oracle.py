"""ctypes binding of oracle/libpworld_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this.  See pworld_oracle.c for what the library restates and why parity is
"unpinned" with respect to the reference.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'libpworld_oracle.so')

PO_MAX_AGENTS = 64
SIMPLE_SPREAD, SIMPLE_TAG, SIMPLE_REFERENCE, SIMPLE_SPEAKER_LISTENER = 0, 1, 2, 3
DIM_C = 10      # simple_reference
SL_DIM_C = 3    # simple_speaker_listener
OBS_LOCAL, OBS_FULL = 0, 1


class PoConfig(C.Structure):
    _fields_ = [
        ('scenario', C.c_int32), ('num_agents', C.c_int32), ('num_landmarks', C.c_int32),
        ('num_adversaries', C.c_int32), ('obs_mode', C.c_int32), ('max_episode_len', C.c_int32),
        ('auto_reset', C.c_int32), ('force_discrete_action', C.c_int32),
        ('landmark_collide', C.c_int32), ('action_force_uses_accel', C.c_int32),
        ('seed', C.c_uint64), ('env_id_base', C.c_uint64),
        ('dt', C.c_double), ('damping', C.c_double), ('contact_force', C.c_double),
        ('contact_margin', C.c_double), ('default_sensitivity', C.c_double), ('mass', C.c_double),
        ('landmark_size', C.c_double),
        ('agent_size', C.c_double * PO_MAX_AGENTS),
        ('agent_accel', C.c_double * PO_MAX_AGENTS),
        ('agent_max_speed', C.c_double * PO_MAX_AGENTS),
    ]


def build(force=False):
    if force or not os.path.exists(_SO) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_SO)
            for f in ('pworld_oracle.c', 'pworld_oracle_impl.h')):
        subprocess.check_call(['make', '-C', _HERE, '-B', 'libpworld_oracle.so'],
                              stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        assert _lib.po_config_size() == C.sizeof(PoConfig), 'po_config layout mismatch'
        _lib.po_exp_det_f32.restype = C.c_float
        _lib.po_exp_det_f32.argtypes = [C.c_float]
        _lib.po_log1p_det_f32.restype = C.c_float
        _lib.po_log1p_det_f32.argtypes = [C.c_float]
        _lib.po_softplus_det_f32.restype = C.c_float
        _lib.po_softplus_det_f32.argtypes = [C.c_float]
    return _lib


def make_config(scenario='simple_spread', num_agents=3, num_landmarks=None, num_adversaries=0,
                obs_mode='local', max_episode_len=25, auto_reset=False, force_discrete_action=True,
                seed=12345678, env_id_base=0, action_force_uses_accel=False,
                dt=0.1, damping=0.25, contact_force=100.0, contact_margin=1e-3,
                default_sensitivity=5.0, mass=1.0):
    """Canonical-upstream defaults."""
    c = PoConfig()
    N = num_agents
    if scenario == 'simple_spread':
        c.scenario = SIMPLE_SPREAD
        L = N if num_landmarks is None else num_landmarks
        c.landmark_collide = 0
        c.landmark_size = 0.05
        for i in range(N):
            c.agent_size[i], c.agent_accel[i], c.agent_max_speed[i] = 0.15, -1.0, -1.0
        c.num_adversaries = 0
    elif scenario == 'simple_tag':
        c.scenario = SIMPLE_TAG
        L = 2 if num_landmarks is None else num_landmarks
        c.landmark_collide = 1
        c.landmark_size = 0.2
        c.num_adversaries = num_adversaries
        for i in range(N):
            adv = i < num_adversaries
            c.agent_size[i] = 0.075 if adv else 0.05
            c.agent_accel[i] = 3.0 if adv else 4.0
            c.agent_max_speed[i] = 1.0 if adv else 1.3
    elif scenario == 'simple_reference':
        c.scenario = SIMPLE_REFERENCE
        N, L = 2, 3
        c.landmark_collide = 0
        c.landmark_size = 0.05
        c.num_adversaries = 0
        for i in range(N):
            c.agent_size[i], c.agent_accel[i], c.agent_max_speed[i] = 0.05, -1.0, -1.0
    elif scenario == 'simple_speaker_listener':
        c.scenario = SIMPLE_SPEAKER_LISTENER
        N, L = 2, 3
        c.landmark_collide = 0
        c.landmark_size = 0.04
        c.num_adversaries = 0
        for i in range(N):
            c.agent_size[i], c.agent_accel[i], c.agent_max_speed[i] = 0.075, -1.0, -1.0
    else:
        raise ValueError(scenario)
    c.num_agents, c.num_landmarks = N, L
    c.obs_mode = OBS_FULL if obs_mode == 'full' else OBS_LOCAL
    c.max_episode_len = max_episode_len
    c.auto_reset = int(auto_reset)
    c.force_discrete_action = int(force_discrete_action)
    c.action_force_uses_accel = int(action_force_uses_accel)
    c.seed, c.env_id_base = seed, env_id_base
    c.dt, c.damping, c.contact_force, c.contact_margin = dt, damping, contact_force, contact_margin
    c.default_sensitivity, c.mass = default_sensitivity, mass
    return c


def obs_dim(cfg):
    return lib().po_obs_dim(C.byref(cfg))


def _p(a, ct):
    return None if a is None else a.ctypes.data_as(C.POINTER(ct))


class COracle(object):
    """B independent worlds advanced by the C restatement; dtype float32 or float64."""

    def __init__(self, cfg, B, dtype=np.float32):
        self.cfg, self.B = cfg, B
        self.dtype = np.dtype(dtype)
        self.sfx = '_f32' if self.dtype == np.float32 else '_f64'
        self.ct = C.c_float if self.dtype == np.float32 else C.c_double
        self.N, self.L = cfg.num_agents, cfg.num_landmarks
        self.D = lib().po_obs_dim(C.byref(cfg))
        self.pos = np.zeros((B, self.N, 2), self.dtype)
        self.vel = np.zeros((B, self.N, 2), self.dtype)
        self.lm = np.zeros((B, self.L, 2), self.dtype)
        self.ep_step = np.zeros(B, np.int32)
        self.ep_count = np.zeros(B, np.uint32)

    def set_state(self, pos, vel, lm, ep_step=None, ep_count=None):
        self.pos[...] = pos
        self.vel[...] = vel
        self.lm[...] = lm
        if ep_step is not None:
            self.ep_step[...] = ep_step
        if ep_count is not None:
            self.ep_count[...] = ep_count

    def reset(self, mask=None):
        obs = np.zeros((self.B, self.N, self.D), self.dtype)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        rc = getattr(lib(), 'po_reset' + self.sfx)(
            C.byref(self.cfg), self.B, _p(self.pos, self.ct), _p(self.vel, self.ct), _p(self.lm, self.ct),
            _p(self.ep_step, C.c_int32), _p(self.ep_count, C.c_uint32), _p(m, C.c_uint8), _p(obs, self.ct))
        assert rc == 0
        return obs

    def observe(self):
        obs = np.zeros((self.B, self.N, self.D), self.dtype)
        getattr(lib(), 'po_observe' + self.sfx)(
            C.byref(self.cfg), self.B, _p(self.pos, self.ct), _p(self.vel, self.ct), _p(self.lm, self.ct),
            _p(obs, self.ct))
        return obs

    def reward(self):
        rew = np.zeros((self.B, self.N), self.dtype)
        coll = np.zeros((self.B, self.N), np.uint64)
        getattr(lib(), 'po_reward' + self.sfx)(
            C.byref(self.cfg), self.B, _p(self.pos, self.ct), _p(self.lm, self.ct),
            _p(rew, self.ct), _p(coll, C.c_uint64))
        return rew, coll

    def step(self, act_idx=None, act_vec=None):
        """-> dict(obs, final_obs, rew, rew_shared, done, terminal, coll)"""
        B, N, D = self.B, self.N, self.D
        out = dict(obs=np.zeros((B, N, D), self.dtype), final_obs=np.zeros((B, N, D), self.dtype),
                   rew=np.zeros((B, N), self.dtype), done=np.zeros((B, N), np.uint8),
                   terminal=np.zeros(B, np.uint8), coll=np.zeros((B, N), np.uint64))
        ai = None if act_idx is None else np.ascontiguousarray(act_idx, np.int32).reshape(B, N)
        av = None if act_vec is None else np.ascontiguousarray(act_vec, self.dtype).reshape(B, N, 5)
        rc = getattr(lib(), 'po_step' + self.sfx)(
            C.byref(self.cfg), B, _p(self.pos, self.ct), _p(self.vel, self.ct), _p(self.lm, self.ct),
            _p(self.ep_step, C.c_int32), _p(self.ep_count, C.c_uint32),
            _p(ai, C.c_int32), _p(av, self.ct),
            _p(out['obs'], self.ct), _p(out['final_obs'], self.ct), _p(out['rew'], self.ct),
            _p(out['done'], C.c_uint8), _p(out['terminal'], C.c_uint8), _p(out['coll'], C.c_uint64))
        assert rc == 0
        # run.py:46 rew_shared = np.sum(rew_n): the agent-order sum, one rounding per addition in this dtype
        shared = np.zeros(B, self.dtype)
        for i in range(N):
            shared = shared + out['rew'][:, i]
        out['rew_shared'] = shared
        return out


class CRefOracle(object):
    """simple_reference / simple_speaker_listener worlds (comm channel + goals) advanced by the C restatement."""

    def __init__(self, cfg, B, dtype=np.float32):
        assert cfg.scenario in (SIMPLE_REFERENCE, SIMPLE_SPEAKER_LISTENER)
        self.sl = cfg.scenario == SIMPLE_SPEAKER_LISTENER
        self.dim_c = SL_DIM_C if self.sl else DIM_C
        self.act_width = 5 if self.sl else 5 + DIM_C
        self.cfg, self.B = cfg, B
        self.dtype = np.dtype(dtype)
        self.sfx = '_f32' if self.dtype == np.float32 else '_f64'
        self.ct = C.c_float if self.dtype == np.float32 else C.c_double
        self.N, self.L = cfg.num_agents, cfg.num_landmarks
        self.D = obs_dim(cfg)
        self.pos = np.zeros((B, self.N, 2), self.dtype)
        self.vel = np.zeros((B, self.N, 2), self.dtype)
        self.lm = np.zeros((B, self.L, 2), self.dtype)
        self.comm = np.zeros((B, self.N, self.dim_c), self.dtype)
        self.goal = np.zeros((B, self.N), np.int32)
        self.ep_step = np.zeros(B, np.int32)
        self.ep_count = np.zeros(B, np.uint32)

    def _state(self):
        return (_p(self.pos, self.ct), _p(self.vel, self.ct), _p(self.lm, self.ct), _p(self.comm, self.ct),
                _p(self.goal, C.c_int32))

    def set_state(self, pos, vel, lm, comm, goal):
        self.pos[...], self.vel[...], self.lm[...], self.comm[...], self.goal[...] = pos, vel, lm, comm, goal

    def reset(self):
        obs = np.zeros((self.B, self.N, self.D), self.dtype)
        rc = getattr(lib(), 'po_ref_reset' + self.sfx)(C.byref(self.cfg), self.B, *self._state(),
                                                       _p(self.ep_step, C.c_int32), _p(self.ep_count, C.c_uint32),
                                                       _p(obs, self.ct))
        assert rc == 0
        return obs

    def observe(self):
        obs = np.zeros((self.B, self.N, self.D), self.dtype)
        getattr(lib(), 'po_ref_observe' + self.sfx)(C.byref(self.cfg), self.B, *self._state(), _p(obs, self.ct))
        return obs

    def step(self, act_idx=None, act_comm=None, act_vec=None):
        B, N, D = self.B, self.N, self.D
        out = dict(obs=np.zeros((B, N, D), self.dtype), final_obs=np.zeros((B, N, D), self.dtype),
                   rew=np.zeros((B, N), self.dtype), done=np.zeros((B, N), np.uint8), terminal=np.zeros(B, np.uint8))
        ai = None if act_idx is None else np.ascontiguousarray(act_idx, np.int32).reshape(B, N)
        ac = None if act_comm is None else np.ascontiguousarray(act_comm, np.int32).reshape(B, N)
        av = None if act_vec is None else np.ascontiguousarray(act_vec, self.dtype).reshape(B, N, self.act_width)
        rc = getattr(lib(), 'po_ref_step' + self.sfx)(
            C.byref(self.cfg), B, *self._state(), _p(self.ep_step, C.c_int32), _p(self.ep_count, C.c_uint32),
            _p(ai, C.c_int32), _p(ac, C.c_int32), _p(av, self.ct), _p(out['obs'], self.ct),
            _p(out['final_obs'], self.ct), _p(out['rew'], self.ct), _p(out['done'], C.c_uint8),
            _p(out['terminal'], C.c_uint8))
        assert rc == 0
        return out


def math_v(fn, x, aux=1.0):
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty_like(x)
    lib().po_math_v(C.c_int(fn), _p(x, C.c_float), C.c_float(aux), _p(y, C.c_float), C.c_long(x.size))
    return y


def philox4x32_10(counter, key):
    ctr = (C.c_uint32 * 4)(*counter)
    k = (C.c_uint32 * 2)(*key)
    out = (C.c_uint32 * 4)()
    lib().po_philox4x32_10(ctr, k, out)
    return tuple(out)
