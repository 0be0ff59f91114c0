"""Host-side mirror of the environment surface the reference drives.

``BatchedParticleEnv``  -- tensor API over B envs resident in HBM; every call is one
    launch of libpworld.so on the current torch stream (no host sync).
``MultiAgentEnv``       -- B = 1, list-of-ndarray API: a drop-in for the object
    ``experiments/run.py:11-103`` receives as ``env`` (``.n``, ``.observation_space``,
    ``.action_space``, ``reset()``, ``step(action_n)``, ``seed()``, ``render()``), built by
    ``multiagent_rl_amd.scenarios.make_env`` the way ``experiments/scenarios.py:124-192`` does.

PyTorch is plumbing here (device memory + streams); the arithmetic is in
csrc/pworld.hip.  There is no CPU path.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import PwConfig, PwStateLayout, PwStepIO, check


class Discrete(object):
    """gym.spaces.Discrete stand-in: ``.n`` and deliberately NO ``.high``
    (main.py:51-58 decides Discrete vs MultiDiscrete by hasattr(space, 'high'))."""

    def __init__(self, n):
        self.n = n
        self.shape = ()

    def sample(self):
        return int(np.random.randint(self.n))

    def __repr__(self):
        return 'Discrete(%d)' % self.n


class MultiDiscrete(object):
    """multiagent.multi_discrete.MultiDiscrete stand-in: ``.low`` / ``.high`` arrays.  main.py:52-54 reads
    ``space.high + 1`` as the per-head action sizes; run.py:39-41 concatenates the heads' one-hots."""

    def __init__(self, array_of_param_array):
        self.low = np.array([x[0] for x in array_of_param_array])
        self.high = np.array([x[1] for x in array_of_param_array])
        self.num_discrete_space = self.low.shape[0]
        self.shape = (self.num_discrete_space,)

    def sample(self):
        return [int(np.random.randint(lo, hi + 1)) for lo, hi in zip(self.low, self.high)]

    def __repr__(self):
        return 'MultiDiscrete' + str(self.num_discrete_space)


class Box(object):
    def __init__(self, shape, dtype=np.float32):
        self.shape = tuple(shape)
        self.dtype = dtype
        self.low, self.high_ = -np.inf, np.inf

    def __repr__(self):
        return 'Box%s' % (self.shape,)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def make_config(scenario_name, num_envs, num_agents=None, num_landmarks=None, num_adversaries=None,
                num_good=None, local_observation=True, max_episode_len=25, auto_reset=False,
                force_discrete_action=True, seed=12345678, env_id_base=0, action_force_uses_accel=False,
                **constants):
    """pw_config with the canonical constants of Scenario.make_world()/World.__init__().

    simple_spread: ``num_agents`` (default 3, as upstream) and L = N unless given --
    the ``make_world(num_agents=n)`` call of experiments/scenarios.py:170.
    simple_tag: ``num_adversaries`` (3) + ``num_good`` (1), L = 2.
    ``constants`` may override dt, damping, contact_force, contact_margin,
    default_sensitivity, mass, landmark_size.
    """
    lib = _lib.load()
    if scenario_name not in _lib.SCENARIOS:
        raise ValueError('unsupported scenario: %r (supported: %s)' % (scenario_name, sorted(_lib.SCENARIOS)))
    scen = _lib.SCENARIOS[scenario_name]
    if scen in (_lib.PW_SIMPLE_REFERENCE, _lib.PW_SIMPLE_SPEAKER_LISTENER):
        adv, n = 0, 2
    elif scen == _lib.PW_SIMPLE_TAG:
        adv = 3 if num_adversaries is None else int(num_adversaries)
        if num_good is None:
            num_good = 1 if num_agents is None else int(num_agents) - adv
        n = adv + int(num_good)
    else:
        adv = 0
        n = 3 if num_agents is None else int(num_agents)
    cfg = PwConfig()
    check(lib.pw_config_default(C.byref(cfg), scen, int(num_envs), n,
                                -1 if num_landmarks is None else int(num_landmarks), adv))
    cfg.obs_mode = _lib.PW_OBS_LOCAL if local_observation else _lib.PW_OBS_FULL
    cfg.max_episode_len = int(max_episode_len)
    cfg.auto_reset = int(bool(auto_reset))
    cfg.force_discrete_action = int(bool(force_discrete_action))
    cfg.action_force_uses_accel = int(bool(action_force_uses_accel))
    cfg.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    cfg.env_id_base = int(env_id_base)
    for k, v in constants.items():
        if k not in ('dt', 'damping', 'contact_force', 'contact_margin', 'default_sensitivity', 'mass',
                     'landmark_size'):
            raise TypeError('unknown world constant %r' % k)
        setattr(cfg, k, float(v))
    return cfg


class BatchedParticleEnv(object):
    """B particle worlds advanced together on one MI355X.

    obs [B,N,D] f32, rew [B,N] f32, done [B,N] bool (always False, as upstream's
    done_callback=None), info: terminal [B] bool (episode_step >= max_episode_len,
    run.py:50), rew_shared [B] (run.py:46), final_obs [B,N,D] (pre-reset observation of
    envs that auto-reset this step; other rows are stale), coll [B,N] uint64-as-int64.
    """

    def __init__(self, scenario_name='simple_spread', num_envs=1, device=None, config=None, want_coll=False,
                 dispatch=None, **kw):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.PworldError('BatchedParticleEnv needs a GPU: libpworld has no CPU fallback')
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.cfg = config if config is not None else make_config(scenario_name, num_envs, **kw)
        self.scenario_name = scenario_name
        h = C.c_void_p()
        check(self.lib.pw_create(C.byref(self.cfg), C.byref(h)))
        self._h = h
        if dispatch:
            self.set_dispatch(**dispatch)
        self.num_envs, self.n = self.cfg.num_envs, self.cfg.num_agents
        self.num_landmarks = self.cfg.num_landmarks
        self.obs_dim = self.lib.pw_obs_dim(h)
        self.want_coll = want_coll
        self.observation_space = [Box((self.obs_dim,)) for _ in range(self.n)]
        # simple_reference agents move AND speak: MultiDiscrete [5 movement | dim_c symbols];
        # simple_speaker_listener: a fixed speaker (Discrete(dim_c = 3)) and a silent listener (Discrete(5)),
        # the per-agent spaces upstream's environment.py builds from agent.movable / agent.silent
        self.speaker_listener = self.cfg.scenario == _lib.PW_SIMPLE_SPEAKER_LISTENER
        self.dim_c = {_lib.PW_SIMPLE_REFERENCE: _lib.PW_DIM_C,
                      _lib.PW_SIMPLE_SPEAKER_LISTENER: _lib.PW_SL_DIM_C}.get(self.cfg.scenario, 0)
        if self.speaker_listener:
            if self.cfg.obs_mode != _lib.PW_OBS_LOCAL:
                raise NotImplementedError('simple_speaker_listener: only the observation the reference patches in '
                                          '(experiments/scenarios.py:45-64, local_observation=True) is built; '
                                          "upstream's has a different length per agent")
            self.action_space = [Discrete(self.dim_c), Discrete(5)]
            self.act_width = 5
        else:
            self.action_space = [MultiDiscrete([[0, 4], [0, self.dim_c - 1]]) if self.dim_c else Discrete(5)
                                 for _ in range(self.n)]
            self.act_width = 5 + self.dim_c
        with torch.cuda.device(self.device):
            self._state = torch.zeros(self.lib.pw_state_bytes(h), dtype=torch.uint8, device=self.device)
        check(self.lib.pw_bind_state(h, _ptr(self._state)))
        lay = PwStateLayout()
        check(self.lib.pw_get_state_layout(h, C.byref(lay)))
        self.layout = lay
        self.bytes_per_env_step = self.lib.pw_algorithmic_bytes_per_env_step(h)

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            self.lib.pw_destroy(h)

    # -- kernel selection (pw_dispatch): results never depend on it, only which kernel form runs
    def get_dispatch(self):
        d = _lib.PwDispatch()
        check(self.lib.pw_get_dispatch(self._h, C.byref(d)))
        return {n: int(getattr(d, n)) for n, _ in _lib.PwDispatch._fields_ if n != 'struct_size'}

    def set_dispatch(self, **fields):
        """Override the dispatcher's choices for this handle, e.g. ``set_dispatch(duo=0)`` (one-wave stream form),
        ``set_dispatch(quad=1)``, ``set_dispatch(envs_per_wave=64)``; unnamed fields keep their current value."""
        d = _lib.PwDispatch()
        check(self.lib.pw_get_dispatch(self._h, C.byref(d)))
        for k, v in fields.items():
            if k == 'struct_size' or not hasattr(d, k):
                raise TypeError('unknown pw_dispatch field %r' % k)
            setattr(d, k, int(v))
        check(self.lib.pw_set_dispatch(self._h, C.byref(d)))

    def set_actor_precision(self, mode):
        """Arithmetic of the actor inside ``FusedActor.rollout`` on this env: ``'f32'`` (default, exact, bit-identical to the
        step loop) or ``'bf16x3'`` -- opt-in and NOT exact: the LSTM input projection on bfloat16 matrix instructions, three
        products per k step (pw_set_actor_precision in include/pworld.h)."""
        check(self.lib.pw_set_actor_precision(self._h, {'f32': 0, 'bf16x3': 1}[mode]))

    def get_actor_precision(self):
        return ('f32', 'bf16x3')[self.lib.pw_get_actor_precision(self._h)]

    # -- helpers
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _f32(self, *shape):
        return torch.empty(shape, dtype=torch.float32, device=self.device)

    def _dev(self, t, dtype):
        if t is None:
            return None
        t = torch.as_tensor(t)
        return t.to(device=self.device, dtype=dtype).contiguous()

    # -- state access (upstream entity.state.p_pos / p_vel)
    def set_state(self, pos, vel=None, landmarks=None, ep_step=None, ep_count=None, comm=None, goal=None):
        """pos/vel [B,N,2], landmarks [B,L,2]; None leaves that part unchanged (vel: zeros).
        simple_reference: comm [B,N,10] (agent.state.c) and goal [B,N] (landmark index of goal_b)."""
        B, N, L = self.num_envs, self.n, self.num_landmarks
        if comm is not None or goal is not None:
            cm, gl = self._dev(comm, torch.float32), self._dev(goal, torch.int32)
            assert (cm is None or cm.shape == (B, N, self.dim_c)) and (gl is None or gl.shape == (B, N))
            check(self.lib.pw_set_comm_state(self._h, _ptr(cm), _ptr(gl), self._stream()))
        pos = self._dev(pos, torch.float32)
        vel = torch.zeros(B, N, 2, device=self.device) if vel is None else self._dev(vel, torch.float32)
        lm = self._dev(landmarks, torch.float32)
        assert pos.shape == (B, N, 2) and vel.shape == (B, N, 2) and (lm is None or lm.shape == (B, L, 2))
        es, ec = self._dev(ep_step, torch.int32), self._dev(ep_count, torch.int32)
        if (es is None) != (ec is None) or es is None:
            # keep counters unless both are given: read them back first
            cur = self.get_state()
            es = cur['ep_step'] if es is None else es
            ec = cur['ep_count'] if ec is None else ec
        check(self.lib.pw_set_state(self._h, _ptr(pos), _ptr(vel), _ptr(lm), _ptr(es), _ptr(ec), self._stream()))

    def get_state(self):
        B, N, L = self.num_envs, self.n, self.num_landmarks
        out = dict(pos=self._f32(B, N, 2), vel=self._f32(B, N, 2), landmarks=self._f32(B, L, 2),
                   ep_step=torch.empty(B, dtype=torch.int32, device=self.device),
                   ep_count=torch.empty(B, dtype=torch.int32, device=self.device))
        check(self.lib.pw_get_state(self._h, _ptr(out['pos']), _ptr(out['vel']), _ptr(out['landmarks']),
                                    _ptr(out['ep_step']), _ptr(out['ep_count']), self._stream()))
        if self.dim_c:
            out['comm'] = self._f32(B, N, self.dim_c)
            out['goal'] = torch.empty(B, N, dtype=torch.int32, device=self.device)
            check(self.lib.pw_get_comm_state(self._h, _ptr(out['comm']), _ptr(out['goal']), self._stream()))
        return out

    # -- MultiAgentEnv.reset
    def reset(self, mask=None):
        """Philox-seeded reset of all (or masked-in) envs -> obs [B,N,D]."""
        obs = self._f32(self.num_envs, self.n, self.obs_dim)
        m = self._dev(mask, torch.uint8)
        check(self.lib.pw_reset(self._h, _ptr(m), _ptr(obs), self._stream()))
        return obs

    def observe(self):
        obs = self._f32(self.num_envs, self.n, self.obs_dim)
        check(self.lib.pw_observe(self._h, _ptr(obs), self._stream()))
        return obs

    def reward(self):
        rew = self._f32(self.num_envs, self.n)
        if self.dim_c:  # the communication scenarios have no collisions
            coll = torch.zeros(self.num_envs, self.n, dtype=torch.int64, device=self.device)
            check(self.lib.pw_reward(self._h, _ptr(rew), None, self._stream()))
            return rew, coll
        coll = torch.empty(self.num_envs, self.n, dtype=torch.int64, device=self.device)
        check(self.lib.pw_reward(self._h, _ptr(rew), _ptr(coll), self._stream()))
        return rew, coll

    # -- MultiAgentEnv.step
    def alloc_outputs(self, T=None, coll=None, final_obs=True):
        B, N, D = self.num_envs, self.n, self.obs_dim
        lead = () if T is None else (T,)
        out = dict(obs=self._f32(*lead, B, N, D), rew=self._f32(*lead, B, N), rew_shared=self._f32(*lead, B),
                   done=torch.empty(*lead, B, N, dtype=torch.bool, device=self.device),
                   terminal=torch.empty(*lead, B, dtype=torch.bool, device=self.device))
        if final_obs and self.cfg.auto_reset:
            out['final_obs'] = self._f32(*lead, B, N, D)
        if self.want_coll if coll is None else coll:
            out['coll'] = torch.empty(*lead, B, N, dtype=torch.int64, device=self.device)
        return out

    def _io(self, actions, out, T):
        B, N = self.num_envs, self.n
        io = PwStepIO()
        lead = (B, N) if T is None else (T, B, N)
        actions = torch.as_tensor(actions)
        if actions.is_floating_point():
            a = actions.to(device=self.device, dtype=torch.float32).contiguous()
            assert a.shape == lead + (self.act_width,), 'float actions must be [..., B, N, %d]' % self.act_width
            io.act_vec = a.data_ptr()
        elif self.dim_c and not self.speaker_listener:  # simple_reference: [..., B, N, 2] = (movement, symbol)
            assert actions.shape == lead + (2,), 'index actions must be [..., B, N, 2] (move, symbol)'
            a = actions.to(device=self.device, dtype=torch.int32)
            a = (a[..., 0].contiguous(), a[..., 1].contiguous())
            io.act_idx, io.act_comm = a[0].data_ptr(), a[1].data_ptr()
        else:
            a = actions.to(device=self.device, dtype=torch.int32).contiguous()
            assert a.shape == lead, 'index actions must be [..., B, N]'
            io.act_idx = a.data_ptr()
        for name in ('obs', 'final_obs', 'rew', 'rew_shared', 'done', 'terminal', 'coll'):
            t = out.get(name)
            if t is not None:
                assert t.is_contiguous() and t.device == self.device
                setattr(io, name, t.data_ptr())
        return io, a

    def plan_rollout(self, actions, out):
        """Pre-binds the buffers of one pw_rollout launch -> ``launch()`` (a bare ctypes call on the
        current stream).  For tight host loops that replay fixed buffers: a C host would keep its
        pw_step_io structs around in exactly this way."""
        T = int(actions.shape[0])
        io, keep = self._io(actions, out, T)
        fn, h, ref = self.lib.pw_rollout, self._h, C.byref(io)
        stream_of = self._stream

        def launch():
            rc = fn(h, ref, T, stream_of())
            if rc:
                check(rc)
        launch.keepalive = (io, keep, out)
        return launch

    def step(self, actions, out=None):
        """actions: int [B,N] indices (0 noop, 1 +x, 2 -x, 3 +y, 4 -y) or float [B,N,5]
        one-hot/soft vectors as run.py:38 builds them -> (obs, rew, done, info)."""
        out = self.alloc_outputs() if out is None else out
        io, keep = self._io(actions, out, None)
        check(self.lib.pw_step(self._h, C.byref(io), self._stream()))
        info = {k: out[k] for k in ('terminal', 'rew_shared', 'final_obs', 'coll') if k in out}
        return out['obs'], out['rew'], out['done'], info

    def last_kernel(self):
        """Name of the device kernel the last step / rollout launched, e.g. 'pw_spread_quad_kernel<true>' (pw_rollout_kernel)."""
        return self.lib.pw_rollout_kernel(self._h).decode().strip('()').replace(', ', ',')

    def rollout(self, actions, out=None):
        """T steps in one launch; actions [T,B,N] int (or [T,B,N,5] float) -> dict of [T,...] tensors."""
        T = int(actions.shape[0])
        out = self.alloc_outputs(T) if out is None else out
        io, keep = self._io(actions, out, T)
        check(self.lib.pw_rollout(self._h, C.byref(io), T, self._stream()))
        return out


class MultiAgentEnv(object):
    """Drop-in for the ``env`` argument of experiments/run.py:11 (B = 1).

    Like upstream, ``reset()`` draws the initial state from NumPy's *global* legacy
    stream in upstream order (all agents, then all landmarks, two uniforms each --
    main.py:47 seeds that stream; ``env.seed()`` itself seeds nothing, as
    gym.Env.seed's default), then the state is uploaded and every step runs on the GPU.
    """

    metadata = {'render.modes': ['human', 'rgb_array']}

    def __init__(self, scenario_name='simple_spread', n=None, local_observation=True, benchmark=False,
                 discrete_action=True, device=None, **kw):
        if scenario_name == 'simple_tag':
            kw.setdefault('num_agents', n)
        elif scenario_name == 'simple_spread':
            kw['num_agents'] = n
        self.batched = BatchedParticleEnv(scenario_name, 1, device=device, local_observation=local_observation,
                                          auto_reset=False, want_coll=benchmark, force_discrete_action=False,
                                          **kw)
        self.scenario_name = scenario_name
        self.n = self.batched.n
        cfg = self.batched.cfg
        self._A = cfg.num_adversaries if scenario_name == 'simple_tag' else 0
        self._L = cfg.num_landmarks
        D = self.batched.obs_dim
        if scenario_name == 'simple_tag':
            # ragged rows: good agents do not see their own velocity among "other good" ones
            dims = [D if i < self._A or self._A == 0 else D - 2 for i in range(self.n)]
        else:
            dims = [D] * self.n
        self._dims = dims
        self.observation_space = [Box((d,)) for d in dims]
        self.action_space = list(self.batched.action_space)
        self._act_width = self.batched.act_width
        self.benchmark = benchmark
        self.discrete_action_space = discrete_action
        self.discrete_action_input = False
        self.force_discrete_action = False  # make_env sets True (experiments/scenarios.py:191)
        self.shared_reward = False          # world.collaborative = False (experiments/scenarios.py:171)
        self.time = 0
        self._reset_world()  # make_world() calls reset_world once, consuming global draws (main.py:39)

    def seed(self, seed=None):
        return []

    def _reset_world(self):
        N, L = self.n, self._L
        goal = None
        if self.scenario_name == 'simple_reference':
            # upstream reset_world: goal_b = np.random.choice(world.landmarks) for agent 0, then agent 1,
            # BEFORE any position is drawn (same global stream)
            idx = list(range(L))
            goal = np.array([[np.random.choice(idx), np.random.choice(idx)]], dtype=np.int32)
        elif self.scenario_name == 'simple_speaker_listener':
            # upstream reset_world: ONE choice (the speaker's goal_b) before the positions
            goal = np.array([[np.random.choice(list(range(L))), 0]], dtype=np.int32)
        lo = 0.9 if self.scenario_name == 'simple_tag' else 1.0
        pos = np.stack([np.random.uniform(-1, +1, 2) for _ in range(N)]) if N else np.zeros((0, 2))
        lm = np.stack([np.random.uniform(-lo, +lo, 2) for _ in range(L)]) if L else np.zeros((0, 2))
        z = torch.zeros(1, dtype=torch.int32)
        kw = {}
        if goal is not None:
            kw = dict(goal=torch.from_numpy(goal), comm=torch.zeros(1, N, self.batched.dim_c))
        self.batched.set_state(torch.from_numpy(pos[None].astype(np.float32)), None,
                               torch.from_numpy(lm[None].astype(np.float32)), ep_step=z, ep_count=z, **kw)

    def _rows(self, obs):
        obs = obs[0].cpu().numpy().astype(np.float64)
        return [obs[i, :d].copy() for i, d in enumerate(self._dims)]

    def reset(self):
        self._reset_world()
        return self._rows(self.batched.observe())

    def step(self, action_n):
        if self.scenario_name == 'simple_speaker_listener':
            # per-agent action lengths (speaker 3, listener 5): rows are padded to the common width
            a = np.zeros((1, self.n, self._act_width), np.float32)
            for i, x in enumerate(action_n):
                x = np.asarray(x, dtype=np.float32).reshape(self.action_space[i].n)
                a[0, i, :x.size] = x
        else:
            a = np.stack([np.asarray(x, dtype=np.float32).reshape(self._act_width) for x in action_n])[None]
        self._sync_force_discrete()
        obs, rew, done, info = self.batched.step(torch.from_numpy(a))
        rew_n = list(rew[0].cpu().numpy().astype(np.float64))  # np.float64 scalars, as upstream's reward()
        done_n = [bool(x) for x in done[0].cpu().numpy()]
        if self.shared_reward:
            rew_n = [np.sum(rew_n)] * self.n
        info_n = {'n': [{} for _ in range(self.n)]}
        if self.benchmark:
            info_n = {'n': self._benchmark_data(rew_n, info['coll'][0].cpu().numpy())}
        return self._rows(obs), rew_n, done_n, info_n

    def _sync_force_discrete(self):
        want = int(bool(self.force_discrete_action))
        if getattr(self, '_fd_applied', None) != want:
            check(self.batched.lib.pw_set_force_discrete_action(self.batched._h, want))
            self._fd_applied = want

    def _benchmark_data(self, rew_n, coll):
        """scenario.benchmark_data (info_callback when make_env(benchmark=True))."""
        out = []
        if self.scenario_name == 'simple_tag':
            for i in range(self.n):  # collisions of this adversary with good agents
                c = int(coll[i]) & ((1 << self.n) - 1)
                out.append(bin(c >> self._A).count('1') if i < self._A else 0)
            return out
        st = self.batched.get_state()
        pos = st['pos'][0].cpu().numpy().astype(np.float64)
        lm = st['landmarks'][0].cpu().numpy().astype(np.float64)
        mins = [min(np.sqrt(np.sum(np.square(p - l))) for p in pos) for l in lm]
        min_dists = float(sum(mins))
        occupied = int(sum(1 for m in mins if m < 0.1))
        for i in range(self.n):
            c = int(coll[i]) & ((1 << self.n) - 1)
            out.append((rew_n[i], bin(c).count('1'), min_dists, occupied))
        return out

    def render(self, mode='human'):
        return []

    def close(self):
        pass
