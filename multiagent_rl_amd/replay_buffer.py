"""Device-resident replay ring with the API of ``rls/replay_buffer.py:9-91`` (ReplayBuffer).

Same method names, argument meaning and ring semantics as the reference
(``add`` overwrites at ``_next_idx`` modulo ``_maxsize``; ``make_index`` draws
``random.randint(0, len - 1)`` from Python's global ``random`` exactly like
``rls/replay_buffer.py:51-52``; ``sample_index`` returns
``(obs[b,N,D], act[b,N,5], rew[b], obs'[b,N,D], done[b])`` as
``_encode_sample`` does), but the storage is SoA tensors in HBM and the
encode step is one HIP gather launch (``pw_replay_gather``) instead of a Python
loop over 1024 tuples.  ``add_batch`` appends B transitions of a batched step in
one launch (``pw_replay_add``).

Deliberate deviation: the reference's ``_encode_sample`` calls
``np.array(list, copy=False)``, which raises under NumPy >= 2;
the intended NumPy-1 behaviour (``np.asarray``) is what is reproduced.
Returned batches are float32 torch tensors on the device (the reference's
trainer converts to float32 tensors right away, ddpg_gumbel_fix.py:121-127).

Ring variants (what ``add`` receives decides, or pass them to the constructor):
  * ``act_heads=(5, 10)`` -- MultiDiscrete scenarios: each agent's action is the concatenation of one one-hot
    per head (experiments/run.py:39-41); the ring keeps one index per head and ``sample_index`` returns the
    concatenated one-hot rows ``act[b, N, 15]`` the reference stored;
  * ``per_agent=True`` -- the BiCNet tuple (experiments/run_BIC.py:46,50): per-agent ``rew[b, N]`` / ``done[b, N]``.
  * ``state_ring=dict(scenario='simple_spread' | 'simple_tag', num_landmarks=L, num_adversaries=A)`` -- a STATE ring
    (``pw_replay_store.state_rows``): the planes ``obs`` / ``next_obs`` hold ``{vel, pos}`` of every agent ([cap, N, 4]) and ``lm``
    the landmarks of the transition's episode ([cap, L, 2]) -- 32 N + 8 L bytes per transition instead of 8 N D (C2: 240 instead of
    768) -- and ``sample_index`` REBUILDS the observation rows it returns (every entry is the state itself or one float32
    subtraction: the batch is bit-identical to the row ring's).  It is the learner rank's ring of the multi-GPU gather
    (``dist.FullTransitionGather(..., ring='state')`` fills it from state-only wire blocks); ``add`` / ``add_batch`` /
    ``add_rollout`` raise on it -- a row cannot be turned back into the landmarks it was built from.
The ring holds HARD one-hot actions as indices (what force_discrete_action / hard Gumbel-softmax produce); rows that
are not one-hot per head, or whose lengths differ between agents, are rejected loudly instead of being squeezed
through an argmax.  The object pickles (experiments/run.py:186-191 stores the memory in the test history): the
filled slots travel as CPU arrays and the device storage is rebuilt on first use after loading.
"""
import ctypes as C
import random

import numpy as np
import torch

from . import _lib
from ._lib import PwReplayStore, check


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class ReplayBuffer(object):
    def __init__(self, size, num_agents=None, obs_dim=None, device=None, act_heads=None, per_agent=None, state_ring=None,
                 device_index=False):
        """size: max number of transitions (``ReplayBuffer(size=1e+6)``, experiments/run.py:20).
        Storage is allocated on the first add (when N and D are known) unless given here.
        ``act_heads``: sizes of the action heads, ``(5,)`` (default) or ``(5, dim_c)``; ``per_agent``: per-agent
        reward / done planes.  Left None they are taken from the first ``add`` (a scalar reward = shared).
        ``device_index=True``: ``make_index`` draws the batch's indices ON the device (one ``torch.randint``; uniform with replacement
        like the reference's ``batch_size`` calls of Python's never-seeded ``random.randint``, rls/replay_buffer.py:51-52) and returns
        the int64 tensor ``sample_index`` takes as it is -- 1024 host-side draws, a list and an upload cost a learner ~0.7 ms of every
        update."""
        self._maxsize = int(size)
        self.device_index = bool(device_index)
        self._next_idx = 0
        self._len = 0
        self._device = None if device is None else torch.device(device)
        self._store = None
        self._lib = None
        self._host_state = None
        self.act_heads = None if act_heads is None else tuple(int(h) for h in act_heads)
        if self.act_heads is not None and (len(self.act_heads) not in (1, 2) or min(self.act_heads) < 1):
            raise ValueError('act_heads must be one or two positive head sizes, got %r' % (act_heads,))
        self.per_agent = None if per_agent is None else bool(per_agent)
        self.state_ring = None
        if state_ring is not None:
            sr = dict(state_ring)
            if sr.get('scenario') not in ('simple_spread', 'simple_tag'):
                raise ValueError("state_ring: scenario must be 'simple_spread' (local observation) or 'simple_tag'")
            if (self.act_heads or (5,)) != (5,) or self.per_agent:
                raise ValueError('a STATE ring is a plain single-head, shared-reward ring')
            if num_agents is None or obs_dim is None:
                raise ValueError('a STATE ring needs num_agents and obs_dim at construction')
            self.state_ring = dict(scenario=sr['scenario'], num_landmarks=int(sr['num_landmarks']),
                                   num_adversaries=int(sr.get('num_adversaries', 0)))
        self.num_agents, self.obs_dim = num_agents, obs_dim
        if num_agents is not None and obs_dim is not None:
            self._allocate(num_agents, obs_dim)

    # -- storage
    def _allocate(self, N, D):
        if not torch.cuda.is_available():
            raise _lib.PworldError('ReplayBuffer needs a GPU: libpworld has no CPU fallback')
        self._lib = _lib.load()
        if self._device is None:
            self._device = torch.device('cuda', torch.cuda.current_device())
        cap, dev = self._maxsize, self._device
        self.num_agents, self.obs_dim = int(N), int(D)
        if self.act_heads is None:
            self.act_heads = (5,)
        if self.per_agent is None:
            self.per_agent = False
        H = len(self.act_heads)
        sr = self.state_ring
        Dp = 4 if sr else D                      # a STATE ring keeps {vx, vy, px, py} per agent; the rows are rebuilt when sampled
        self.obs = torch.empty(cap, N, Dp, dtype=torch.float32, device=dev)
        self.next_obs = torch.empty(cap, N, Dp, dtype=torch.float32, device=dev)
        self.lm = torch.empty(cap, max(sr['num_landmarks'], 1), 2, dtype=torch.float32, device=dev) if sr else None
        self.act = torch.empty((cap, N) if H == 1 else (cap, N, H), dtype=torch.uint8, device=dev)
        self.rew = torch.empty((cap, N) if self.per_agent else (cap,), dtype=torch.float32, device=dev)
        self.done = torch.empty((cap, N) if self.per_agent else (cap,), dtype=torch.float32, device=dev)
        st = PwReplayStore()
        st.obs, st.next_obs, st.rew, st.done = (self.obs.data_ptr(), self.next_obs.data_ptr(),
                                                self.rew.data_ptr(), self.done.data_ptr())
        st.act = self.act.data_ptr()
        st.capacity, st.num_agents, st.obs_dim = cap, N, D
        st.act_heads, st.per_agent = H, int(self.per_agent)
        st.head_width[0], st.head_width[1] = self.act_heads[0], (self.act_heads[1] if H == 2 else 0)
        if sr:
            st.state_rows, st.num_landmarks, st.num_adversaries = 1, sr['num_landmarks'], sr['num_adversaries']
            st.scenario = _lib.SCENARIOS[sr['scenario']]
            st.lm = self.lm.data_ptr()
        self._store = st
        # device copies of _next_idx (hipGraph mode): cell [0] for add_batch(device_cursor=True); add_batch_tail
        # ping-pongs between [0] and [1] (it reads one cell in every workgroup and writes the other)
        self._cursor = torch.zeros(2, dtype=torch.int64, device=dev)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self._device).cuda_stream)

    def _rows_only(self, who):
        if self.state_ring is not None:
            raise _lib.PworldError('ReplayBuffer.%s: a STATE ring is filled from state-only wire blocks (dist.FullTransitionGather, '
                                   'pw_replay_add_state_wire) -- observation rows do not determine the landmarks they were built from' % who)

    def __len__(self):
        return self._len

    def clear(self):
        self._next_idx = 0
        self._len = 0

    # -- pickling (experiments/run.py:186-191 puts the memory into the test history it pickles)
    def __getstate__(self):
        host = self._host_state
        if self._store is not None:
            sl = slice(0, self._len)  # add() fills slots [0, len) before it ever wraps
            host = {k: getattr(self, k)[sl].cpu().numpy() for k in ('obs', 'next_obs', 'act', 'rew', 'done') + (('lm',) if self.state_ring else ())}
        return dict(maxsize=self._maxsize, next_idx=self._next_idx, len=self._len, num_agents=self.num_agents,
                    obs_dim=self.obs_dim, act_heads=self.act_heads, per_agent=self.per_agent, planes=host, state_ring=self.state_ring,
                    device_index=self.device_index)

    def __setstate__(self, st):
        self.__init__(st['maxsize'], act_heads=st['act_heads'], per_agent=st['per_agent'])
        self._next_idx, self._len = st['next_idx'], st['len']
        self.num_agents, self.obs_dim = st['num_agents'], st['obs_dim']
        self.state_ring = st.get('state_ring')
        self.device_index = bool(st.get('device_index', False))
        self._host_state = st['planes']   # uploaded by _ensure_device() on first use (unpickling needs no GPU)

    def _ensure_device(self):
        """Rebuild the device storage of an unpickled buffer."""
        if self._store is None and self._host_state is not None:
            host, self._host_state = self._host_state, None
            self._allocate(self.num_agents, self.obs_dim)
            for k, v in host.items():
                getattr(self, k)[:v.shape[0]].copy_(torch.from_numpy(v))

    def host_planes(self):
        """The filled part of the ring as CPU arrays: {obs, next_obs, act, rew, done} (works without a GPU for an
        unpickled buffer)."""
        return self.__getstate__()['planes']

    # -- rls/replay_buffer.py:30-37
    def _action_indices(self, action):
        """list of N one-hot rows -> int32 [N] (one head) or [N, 2]; rejects what the index ring cannot hold."""
        rows = [np.asarray(a, dtype=np.float64).reshape(-1) for a in action]
        widths = sorted(set(r.size for r in rows))
        if len(widths) != 1:
            raise ValueError('ReplayBuffer: agents have action vectors of different lengths %s (e.g. '
                             'simple_speaker_listener); the device ring needs one shape per agent -- use a host '
                             'memory for this scenario' % (widths,))
        w = widths[0]
        if self.act_heads is None:
            self.act_heads = (w,)
        if sum(self.act_heads) != w:
            raise ValueError('ReplayBuffer: action vectors have length %d but act_heads=%r' % (w, self.act_heads))
        a = np.stack(rows)
        out, lo = [], 0
        for h in self.act_heads:
            part = a[:, lo:lo + h]
            onehot = ((part == 0) | (part == 1)).all() and (part.sum(-1) == 1).all()
            if not onehot:
                hint = ' (a %d-wide row with two ones is a MultiDiscrete action: pass act_heads=(5, %d))' % (w, w - 5) \
                    if len(self.act_heads) == 1 and w > 5 else ''
                raise ValueError('ReplayBuffer: the device ring stores HARD one-hot actions as indices; got a row '
                                 'that is not one-hot per head%s' % hint)
            out.append(part.argmax(-1).astype(np.int32))
            lo += h
        return out[0] if len(out) == 1 else np.stack(out, -1)

    def add(self, obs_t, action, reward, obs_tp1, done):
        """One transition in the reference's host format: lists of N arrays (D,), list of N one-hot action rows
        ((5,), or the concatenated heads of a MultiDiscrete action), reward float (or list of N: the BiCNet tuple),
        lists of N arrays, done float (or list of N)."""
        obs = torch.as_tensor(np.stack([np.asarray(o, dtype=np.float32) for o in obs_t]))[None]
        nxt = torch.as_tensor(np.stack([np.asarray(o, dtype=np.float32) for o in obs_tp1]))[None]
        act = torch.as_tensor(self._action_indices(action))[None]
        per_agent = np.ndim(reward) > 0
        if self.per_agent is None:
            self.per_agent = per_agent
        if per_agent != self.per_agent or (np.ndim(done) > 0) != self.per_agent:
            raise ValueError('ReplayBuffer: this ring holds %s rewards / dones' %
                             ('per-agent (BiCNet)' if self.per_agent else 'one shared scalar'))
        rew = torch.as_tensor(np.asarray(reward, dtype=np.float32).reshape(1, -1) if per_agent
                              else np.asarray([float(reward)], dtype=np.float32))
        dn = torch.as_tensor(np.asarray(done, dtype=np.float32).reshape(1, -1) if per_agent
                             else np.asarray([float(done)], dtype=np.float32))
        self.add_batch(obs, act, rew, nxt, done=dn)

    def sync_cursor(self):
        """Copy the host ring position to the device cursor (call before capturing a graph)."""
        self._cursor.fill_(self._next_idx)

    def note_graph_adds(self, num_transitions):
        """Host bookkeeping for adds that ran inside a replayed graph (device cursor already advanced)."""
        self._next_idx = (self._next_idx + num_transitions) % self._maxsize
        self._len = min(self._len + num_transitions, self._maxsize)

    def add_batch(self, obs, act_idx, rew_shared, next_obs, final_obs=None, terminal=None, done=None,
                  device_cursor=False, advance_cursor=True):
        """B transitions of one batched step: obs/next_obs [B,N,D], act_idx [B,N] int, rew_shared [B].
        Where ``terminal[b]`` is set, next_obs is taken from ``final_obs`` (the pre-reset
        observation: the reference stores new_obs_n BEFORE env.reset(), run.py:52 vs :60).
        ``device_cursor=True`` (hipGraph capture): the ring position is read from, and advanced in,
        device memory by the captured launches; the caller accounts for it with ``note_graph_adds``
        (``advance_cursor=False``: the caller advances ``_cursor`` itself, e.g. in pw_rollout_tail)."""
        B, N, D = obs.shape
        self._ensure_device()
        self._rows_only('add / add_batch')
        if self._store is None:
            self._device = obs.device if obs.is_cuda and self._device is None else self._device
            if self.act_heads is None and act_idx.dim() == 3:
                raise ValueError('ReplayBuffer: [B,N,2] action indices need act_heads=(5, dim_c) at construction')
            self._allocate(N, D)
        assert (N, D) == (self.num_agents, self.obs_dim) and B <= self._maxsize
        H = len(self.act_heads)
        assert tuple(act_idx.shape) == ((B, N) if H == 1 else (B, N, H)), 'action indices must be [B,N] or [B,N,2] (two heads)'
        assert tuple(rew_shared.shape) == ((B, N) if self.per_agent else (B,)), 'reward must be [B] ([B,N] per-agent ring)'
        assert done is None or tuple(done.shape) == tuple(rew_shared.shape)
        dev = self._device
        f32 = lambda t: None if t is None else t.to(device=dev, dtype=torch.float32).contiguous()  # noqa: E731
        obs, next_obs, final_obs = f32(obs), f32(next_obs), f32(final_obs)
        rew_shared, done = f32(rew_shared), f32(done)
        act_idx = act_idx.to(device=dev, dtype=torch.int32).contiguous()
        term = None if terminal is None else terminal.to(device=dev).contiguous().view(torch.uint8)
        if device_cursor:
            check(self._lib.pw_replay_add(C.byref(self._store), 0, _ptr(self._cursor), B, _ptr(obs), _ptr(act_idx),
                                          _ptr(rew_shared), _ptr(next_obs), _ptr(final_obs), _ptr(term), _ptr(done),
                                          self._stream()))
            if advance_cursor:
                check(self._lib.pw_counter_add(_ptr(self._cursor), B, self._maxsize, self._stream()))
            return
        check(self._lib.pw_replay_add(C.byref(self._store), self._next_idx, None, B, _ptr(obs), _ptr(act_idx),
                                      _ptr(rew_shared), _ptr(next_obs), _ptr(final_obs), _ptr(term), _ptr(done),
                                      self._stream()))
        self._next_idx = (self._next_idx + B) % self._maxsize
        self._len = min(self._len + B, self._maxsize)

    def add_batch_tail(self, obs, act_idx, rew_shared, next_obs, final_obs, terminal, episode_return, finished_sum,
                       finished_count, step_counter=None, parity=None):
        """``add_batch`` + the rollout's episode-return bookkeeping (``pw_episode_stats``) in ONE launch.
        ``parity`` None: host ring position.  ``parity`` 0/1 (hipGraph capture): the position is read from
        ``_cursor[parity]`` and the next one written to ``_cursor[1 - parity]``; ``step_counter`` (a device
        int64, e.g. the policy's Philox step) is incremented by the same launch; the caller accounts for the
        adds with ``note_graph_adds``.  All tensors must already be float32 / int32 / uint8 device tensors."""
        B, N, D = obs.shape
        self._ensure_device()
        self._rows_only('add_batch_tail')
        if self._store is None:
            self._device = obs.device if self._device is None else self._device
            self._allocate(N, D)
        assert (N, D) == (self.num_agents, self.obs_dim) and B <= self._maxsize
        term = terminal.view(torch.uint8)
        for t, dt in ((obs, torch.float32), (next_obs, torch.float32), (rew_shared, torch.float32),
                      (act_idx, torch.int32), (term, torch.uint8), (episode_return, torch.float32)):
            assert t.is_cuda and t.dtype == dt and t.is_contiguous()
        if parity is None:
            start, cur, nxt = self._next_idx, None, None
        else:
            start, cur, nxt = 0, _ptr(self._cursor[parity:]), _ptr(self._cursor[1 - parity:])
        check(self._lib.pw_replay_add_tail(C.byref(self._store), start, cur, nxt, B, _ptr(obs), _ptr(act_idx),
                                           _ptr(rew_shared), _ptr(next_obs), _ptr(final_obs), _ptr(term), None,
                                           _ptr(episode_return), _ptr(finished_sum), _ptr(finished_count),
                                           _ptr(step_counter), self._stream()))
        if parity is None:
            self._next_idx = (self._next_idx + B) % self._maxsize
            self._len = min(self._len + B, self._maxsize)

    def add_rollout(self, obs0, out, episode_return=None, finished_sum=None, finished_count=None):
        """A whole rollout chunk in ONE launch: ``out`` = the [T, ...] outputs of ``FusedActor.rollout`` /
        ``BatchedParticleEnv.rollout`` (obs, rew_shared, terminal, final_obs, act), ``obs0`` [B,N,D] the
        observation before its first step.  Stored in the order T ``add_batch`` calls would have used; optionally
        does the chunk's episode-return bookkeeping in the same launch."""
        from ._lib import PwStepIO
        T, B, N, D = out['obs'].shape
        self._ensure_device()
        self._rows_only('add_rollout')
        if self._store is None:
            self._device = obs0.device if self._device is None else self._device
            self._allocate(N, D)
        assert (N, D) == (self.num_agents, self.obs_dim) and T * B <= self._maxsize
        io = PwStepIO()
        for name in ('obs', 'final_obs', 'rew_shared', 'terminal'):
            t = out.get(name)
            if t is not None:
                assert t.is_cuda and t.is_contiguous()
                setattr(io, name, t.data_ptr())
        act = out['act']
        assert act.dtype == torch.int32 and act.is_contiguous() and obs0.is_contiguous() and obs0.dtype == torch.float32
        assert tuple(act.shape) == ((T, B, N, 2) if len(self.act_heads) == 2 else (T, B, N)), \
            'chunk actions must be [T,B,N] ([T,B,N,2] for a two-head ring)' 
        scratch = None
        if episode_return is not None:
            if getattr(self, '_roll_scratch_B', None) != B:
                n = self._lib.pw_replay_add_rollout_scratch_bytes(B)
                self._roll_scratch, self._roll_scratch_B = torch.zeros(n, dtype=torch.uint8, device=self._device), B
            scratch = self._roll_scratch
        check(self._lib.pw_replay_add_rollout(C.byref(self._store), self._next_idx, B, T, _ptr(obs0), C.byref(io),
                                              _ptr(act), _ptr(episode_return), _ptr(finished_sum),
                                              _ptr(finished_count), _ptr(scratch), self._stream()))
        self._next_idx = (self._next_idx + T * B) % self._maxsize
        self._len = min(self._len + T * B, self._maxsize)

    # -- rls/replay_buffer.py:51-57
    def make_index(self, batch_size):
        if self.device_index and self._store is not None:
            return torch.randint(0, self._len, (int(batch_size),), dtype=torch.int64, device=self._device)
        return [random.randint(0, self._len - 1) for _ in range(batch_size)]

    def make_latest_index(self, batch_size):
        idx = [(self._next_idx - 1 - i) % self._maxsize for i in range(batch_size)]
        np.random.shuffle(idx)
        return idx

    # -- rls/replay_buffer.py:39-49, 59-60
    def _encode_sample(self, idxes):
        self._ensure_device()
        idx = torch.as_tensor(list(idxes) if not torch.is_tensor(idxes) else idxes, dtype=torch.int64,
                              device=self._device).contiguous()
        b, N, D, dev = idx.numel(), self.num_agents, self.obs_dim, self._device
        rd = (b, N) if self.per_agent else (b,)
        out = (torch.empty(b, N, D, device=dev), torch.empty(b, N, sum(self.act_heads), device=dev),
               torch.empty(rd, device=dev), torch.empty(b, N, D, device=dev), torch.empty(rd, device=dev))
        check(self._lib.pw_replay_gather(C.byref(self._store), _ptr(idx), b, _ptr(out[0]), _ptr(out[1]),
                                         _ptr(out[2]), _ptr(out[3]), _ptr(out[4]), self._stream()))
        return out

    def sample_index(self, idxes):
        return self._encode_sample(idxes)

    def sample(self, batch_size):
        if batch_size > 0:
            idxes = self.make_index(batch_size)
        else:
            idxes = range(0, self._len)
        return self._encode_sample(idxes)

    def collect(self):
        return self.sample(-1)
