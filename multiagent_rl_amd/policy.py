"""The action producer of the rollout: the reference's actor architecture and its
Gumbel-softmax exploration, batched over [B envs x N agents] and kept on the device.

``ActorNetwork`` has the layer structure AND parameter names of
``rls/model/ac_network_multi_gumbel.py:24-67`` (``dense1.module``, ``bilstm``,
``dense2.module``), so state_dicts saved by the reference's Trainer
(``ddpg_gumbel_fix.py:221-229``) load unchanged.  It stays stock PyTorch-ROCm
(rocBLAS / MIOpen on MFMA) as BASELINE.json's north_star prescribes; the
hand-written HIP work is the environment, not the policy GEMMs.

``GumbelPolicy`` is the batched form of ``Trainer.get_exploration_action`` /
``gumbel_softmax(hard=True)`` (``ddpg_gumbel_fix.py:86-116``): the hard one-hot of
``F.gumbel_softmax`` is ``argmax(logits + g)`` with ``g = -log(Exponential(1))``, drawn here with
the same RNG call, but the result stays an int32 index tensor [B,N] in HBM -- no
``.cpu().numpy()`` round trip per env-step.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class TimeDistributed(nn.Module):
    """Applies ``module`` over the last axis of [batch, agents, features]."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, x):
        if x.dim() <= 2:
            return self.module(x)
        lead = x.shape[:2]
        return self.module(x.reshape(lead[0] * lead[1], x.shape[2])).reshape(lead[0], lead[1], -1)


class ActorNetwork(nn.Module):
    """Linear(D, 64) -> ReLU -> BiLSTM(64 -> 2x32) over the AGENT axis -> ReLU -> Linear(64, 5).
    ``out_dim`` may be a list of two sizes (MultiDiscrete scenarios, main.py:52-54): then there are two
    heads, ``dense2_1`` / ``dense2_2``, and ``forward`` returns the list of their logits."""

    def __init__(self, input_dim, out_dim):
        super().__init__()
        self.out_dim = out_dim
        self.dense1 = TimeDistributed(nn.Linear(input_dim, 64))
        self.bilstm = nn.LSTM(64, 32, num_layers=1, batch_first=True, bidirectional=True)
        if type(out_dim) is list:
            self.dense2_1 = TimeDistributed(nn.Linear(64, out_dim[0]))
            self.dense2_2 = TimeDistributed(nn.Linear(64, out_dim[1]))
        else:
            self.dense2 = TimeDistributed(nn.Linear(64, out_dim))

    def forward(self, obs):
        hid = F.relu(self.dense1(obs))
        hid, _ = self.bilstm(hid, None)
        hid = F.relu(hid)
        if type(self.out_dim) is list:
            return [self.dense2_1(hid), self.dense2_2(hid)]
        return self.dense2(hid)


class GumbelPolicy(object):
    """obs [B,N,D] (device) -> action index [B,N] int32 (device)."""

    def __init__(self, actor, generator=None):
        self.actor = actor
        self.generator = generator

    @torch.no_grad()
    def logits(self, obs):
        return self.actor(obs)

    def _sample(self, logits):
        gumbels = -torch.empty_like(logits).exponential_(generator=self.generator).log()
        return (logits + gumbels).argmax(dim=-1).to(torch.int32)

    @torch.no_grad()
    def __call__(self, obs):
        logits = self.actor(obs)
        if isinstance(logits, (list, tuple)):  # MultiDiscrete: [B,N,2] = (movement index, communication symbol)
            return torch.stack([self._sample(x) for x in logits], dim=-1)
        return self._sample(logits)


class FusedActor(object):
    """The same ActorNetwork evaluated by libpworld instead of MIOpen's ~45-kernel RNN path.

    Weights are snapshotted from ``actor`` (call ``refresh()`` after the learner updates it).  Default: ONE launch
    (``pw_actor_fused``): dense1 + ReLU and the input projections of both LSTM directions on the matrix cores
    (exact float32 MFMA), the recurrence over the agent axis, the output head and the Gumbel-argmax sampling, with
    every intermediate in registers / LDS.  ``PW_ACTOR_NO_FUSE=1`` selects the same arithmetic as three launches
    (``pw_actor_front``, ``pw_bilstm_forward``, ``pw_actor_head`` -- identical results, also used when N > 96), and
    ``PW_ACTOR_NO_MFMA=1`` additionally replaces the front end by weight-stationary dense1 + a rocBLAS GEMM.
    """

    def __init__(self, actor, seed=0):
        import ctypes as C
        from . import _lib
        self._C, self._lib_mod, self.lib = C, _lib, _lib.load()
        import os
        self.actor, self.seed, self.calls = actor, int(seed), 0
        self.use_mfma_front = not os.environ.get('PW_ACTOR_NO_MFMA')
        self.use_fused = self.use_mfma_front and not os.environ.get('PW_ACTOR_NO_FUSE')
        self.refresh()
        self._step_dev = torch.zeros(1, dtype=torch.int64, device=self.device)  # Philox step (hipGraph mode)
        self.graph_mode = False
        self.defer_step_advance = False  # graph mode: the caller advances _step_dev (pw_rollout_tail)

    @torch.no_grad()
    def refresh(self):
        a = self.actor
        lin1, lstm = a.dense1.module, a.bilstm
        two = type(a.out_dim) is list  # MultiDiscrete actor: two heads on the same BiLSTM output
        heads = [a.dense2_1.module, a.dense2_2.module] if two else [a.dense2.module]
        self.heads = tuple(int(h.out_features) for h in heads)
        assert lstm.hidden_size == 32 and lstm.bidirectional and lstm.num_layers == 1 and sum(self.heads) <= 16
        dev = lin1.weight.device
        assert dev.type == 'cuda', 'FusedActor needs the actor on the GPU (no CPU fallback)'
        f = lambda t: t.detach().to(torch.float32).contiguous()  # noqa: E731
        self.w1, self.b1 = f(lin1.weight), f(lin1.bias)                                          # [64, D]
        self.wih = f(torch.cat([lstm.weight_ih_l0, lstm.weight_ih_l0_reverse], 0))             # [256, 64]
        self.wih_t = f(self.wih.t())                                                            # [64, 256]
        self.bih = f(torch.cat([lstm.bias_ih_l0 + lstm.bias_hh_l0,
                                lstm.bias_ih_l0_reverse + lstm.bias_hh_l0_reverse], 0))         # [256]
        self.whh_f, self.whh_r = f(lstm.weight_hh_l0), f(lstm.weight_hh_l0_reverse)             # [128, 32]
        self.w2 = f(torch.cat([h.weight for h in heads], 0))                                    # [sum(heads), 64]
        self.b2 = f(torch.cat([h.bias for h in heads], 0))
        self.device = dev
        D = self.w1.shape[1]
        self.frag = torch.empty(self.lib.pw_actor_front_pack_floats(D), dtype=torch.float32, device=dev)
        cp = lambda t: self._C.c_void_p(t.data_ptr())  # noqa: E731
        self._lib_mod.check(self.lib.pw_actor_front_pack(cp(self.w1), cp(self.wih), D, cp(self.frag),
                                                         self._C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))

    def _stream(self):
        return self._C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @torch.no_grad()
    def _fused(self, obs, want_h=False, want_logits=False, want_act=False):
        """One launch: obs [B,N,D] -> (H [B,N,64], logits [B,N,5], act [B,N] int32), each None unless wanted."""
        B, N, D = obs.shape
        x = obs.reshape(B * N, D).to(torch.float32).contiguous()
        h = torch.empty(B, N, 64, dtype=torch.float32, device=self.device) if want_h else None
        n0, n1 = self.heads[0], (self.heads[1] if len(self.heads) > 1 else 0)
        logits = torch.empty(B, N, n0 + n1, dtype=torch.float32, device=self.device) if want_logits else None
        act = torch.empty((B, N, 2) if n1 else (B, N), dtype=torch.int32, device=self.device) if want_act else None
        p = lambda t: None if t is None else self._C.c_void_p(t.data_ptr())  # noqa: E731
        step_dev = p(self._step_dev) if (self.graph_mode and want_act) else None
        self._lib_mod.check(self.lib.pw_actor_fused(p(x), p(self.frag), p(self.b1), p(self.bih), p(self.whh_f),
                                                    p(self.whh_r), p(self.w2), p(self.b2), n0, n1, B, N, D, 1,
                                                    self.seed, self.calls, step_dev, p(h), p(logits), p(act),
                                                    self._stream()))
        if step_dev is not None and not self.defer_step_advance:  # captured: the counter advances on the device
            self._lib_mod.check(self.lib.pw_counter_add(p(self._step_dev), 1, 0, self._stream()))
        return h, logits, act

    @torch.no_grad()
    def hidden(self, obs):
        """obs [B,N,D] -> relu(BiLSTM(relu(dense1(obs)))) [B,N,64]."""
        B, N, D = obs.shape
        if self.use_fused and N <= 96:
            return self._fused(obs, want_h=True)[0]
        x = obs.reshape(B * N, D).to(torch.float32).contiguous()
        p = lambda t: self._C.c_void_p(t.data_ptr())  # noqa: E731
        h = torch.empty(B, N, 64, dtype=torch.float32, device=self.device)
        if self.use_mfma_front:
            # dense1 + ReLU + both input projections in ONE launch on the matrix cores (exact f32 MFMA);
            # the hidden activations never leave registers
            g = torch.empty(B * N, 256, dtype=torch.float32, device=self.device)     # [B,N,2,128]
            self._lib_mod.check(self.lib.pw_actor_front(p(x), p(self.frag), p(self.b1), p(self.bih), B * N, D, p(g),
                                                        self._stream()))
        else:
            # the same in three launches: weight-stationary dense1 + ReLU, then a rocBLAS GEMM
            x1 = torch.empty(B * N, 64, dtype=torch.float32, device=self.device)
            self._lib_mod.check(self.lib.pw_dense(p(x), p(self.w1), p(self.b1), B * N, D, 64, 1, p(x1), self._stream()))
            g = torch.addmm(self.bih, x1, self.wih_t)        # [B*N, 256] = [B,N,2,128]
        self._lib_mod.check(self.lib.pw_bilstm_forward(p(g), p(self.whh_f), p(self.whh_r), B, N, 1, p(h),
                                                       self._stream()))
        return h

    @torch.no_grad()
    def _head(self, h, want_logits, want_act):
        B, N, _ = h.shape
        logits = torch.empty(B, N, 5, dtype=torch.float32, device=self.device) if want_logits else None
        act = torch.empty(B, N, dtype=torch.int32, device=self.device) if want_act else None
        p = lambda t: None if t is None else self._C.c_void_p(t.data_ptr())  # noqa: E731
        step_dev = p(self._step_dev) if (self.graph_mode and want_act) else None
        self._lib_mod.check(self.lib.pw_actor_head(p(h), p(self.w2), p(self.b2), B * N, self.seed, self.calls,
                                                   step_dev, p(logits), p(act), self._stream()))
        if step_dev is not None and not self.defer_step_advance:  # captured: the counter advances on the device
            self._lib_mod.check(self.lib.pw_counter_add(p(self._step_dev), 1, 0, self._stream()))
        return logits, act

    def _one_launch(self, obs):
        if self.use_fused and obs.shape[1] <= 96:
            return True
        if self.heads != (5,):
            raise NotImplementedError('the three-launch chain serves the single 5-logit head only; two-head '
                                      '(MultiDiscrete) actors need the one-launch kernel (N <= 96)')
        return False

    def logits(self, obs):
        """[B,N,5]; two-head actors: the list [logits_move [B,N,n0], logits_comm [B,N,n1]] like ActorNetwork."""
        if self._one_launch(obs):
            lg = self._fused(obs, want_logits=True)[1]
            return lg if len(self.heads) == 1 else [lg[..., :self.heads[0]], lg[..., self.heads[0]:]]
        return self._head(self.hidden(obs), True, False)[0]

    def __call__(self, obs):
        """-> Gumbel-sampled action index [B,N] int32 (one fresh Philox stream per call)."""
        if self._one_launch(obs):
            act = self._fused(obs, want_act=True)[2]
        else:
            act = self._head(self.hidden(obs), False, True)[1]
        self.calls += 1
        return act

    @torch.no_grad()
    def rollout(self, env, num_steps, out=None, memory=None, stats=None):
        """``num_steps`` x (this actor + Gumbel sampling + ``env`` step with auto-reset) as ONE launch
        (``pw_policy_rollout``): observations, actions and world state stay on the CU between steps.  Continues
        from the env's current state.  -> dict of [T, ...] outputs as ``BatchedParticleEnv.rollout`` plus
        ``act`` [T,B,N] int32.  Same results as ``num_steps`` iterations of ``act = self(obs); env.step(act)``.

        ``memory`` (a device ``ReplayBuffer``): the transitions go straight into the ring from the same launch;
        ``stats`` = (episode_return [B], finished_sum, finished_count) tensors: the episode bookkeeping too.  With a
        sink, ``out=False`` skips the step outputs altogether (only the ring / statistics are written)."""
        from ._lib import PwRolloutSink, PwStepIO
        T, B, N = int(num_steps), env.num_envs, env.n
        two = len(self.heads) == 2   # MultiDiscrete actor: simple_reference, act [T,B,N,2] = (movement, symbol); the sink is a two-head ring
        if two:
            assert env.scenario_name == 'simple_reference' and self.heads == (5, env.dim_c), \
                'two-head rollouts serve simple_reference (heads 5 | dim_c)'
            assert memory is None or tuple(memory.act_heads or ()) == self.heads, \
                'simple_reference: the ring sink is a ReplayBuffer built with act_heads=(5, dim_c)'
        else:
            assert self.heads == (5,), 'single 5-logit head only' 
        if out is False:
            assert memory is not None or stats is not None
            out = {}
        else:
            out = env.alloc_outputs(T, coll=False) if out is None else out
            if 'act' not in out:
                out['act'] = torch.empty((T, B, N, 2) if two else (T, B, N), dtype=torch.int32, device=self.device)
            # a caller-supplied action buffer of the single-head shape under a two-head actor would be overrun by the kernel
            # (it writes act[2 row], act[2 row + 1]): refuse it here, whoever built the dict
            want = (T, B, N, 2) if two else (T, B, N)
            if tuple(out['act'].shape) != want or out['act'].dtype != torch.int32 or not out['act'].is_contiguous():
                raise ValueError('rollout: out["act"] must be a contiguous int32 tensor of shape %r (%s actor), got %r %s'
                                 % (want, 'two-head' if two else 'one-head', tuple(out['act'].shape), out['act'].dtype))
        io = PwStepIO()
        for name in ('obs', 'final_obs', 'rew', 'rew_shared', 'done', 'terminal'):
            t = out.get(name)
            if t is not None:
                assert t.is_contiguous() and t.shape[0] == T
                setattr(io, name, t.data_ptr())
        p = lambda t: None if t is None else self._C.c_void_p(t.data_ptr())  # noqa: E731
        sink = None
        if memory is not None or stats is not None:
            sink = PwRolloutSink()
            if memory is not None:
                if memory._store is None:
                    memory._device = self.device if memory._device is None else memory._device
                    memory._allocate(N, env.obs_dim)
                assert T * B <= memory._maxsize
                sink.ring = self._C.addressof(memory._store)
                sink.ring_start = memory._next_idx
            if stats is not None:
                if getattr(self, '_sink_scratch_key', None) != (B, N):
                    nbytes = self.lib.pw_policy_rollout_scratch_bytes(env._h)
                    self._sink_scratch = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
                    self._sink_scratch_key = (B, N)
                sink.episode_return, sink.finished_sum, sink.finished_count = (t.data_ptr() for t in stats)
                sink.scratch = self._sink_scratch.data_ptr()
        step_dev = p(self._step_dev) if self.graph_mode else None
        self._lib_mod.check(self.lib.pw_policy_rollout(env._h, p(self.frag), p(self.b1), p(self.bih), p(self.whh_f),
                                                       p(self.whh_r), p(self.w2), p(self.b2), 1, self.seed, self.calls,
                                                       step_dev, self._C.byref(io), p(out.get('act')), T,
                                                       None if sink is None else self._C.byref(sink), self._stream()))
        if memory is not None:
            memory._next_idx = (memory._next_idx + T * B) % memory._maxsize
            memory._len = min(memory._len + T * B, memory._maxsize)
        if step_dev is not None and not self.defer_step_advance:
            self._lib_mod.check(self.lib.pw_counter_add(p(self._step_dev), T, 0, self._stream()))
        self.calls += T
        return out

    def begin_graph(self):
        """Switch to the device-side step counter (call before hipGraph capture)."""
        self._step_dev.fill_(self.calls)
        self.graph_mode = True


class FusedExploration(object):
    """``Trainer.get_exploration_action`` (rls/agent/multiagent/ddpg_gumbel_fix.py:86-107) on the one-launch actor:
    list of N observation arrays -> the hard Gumbel-softmax one-hot(s) the reference returns -- ``ndarray [1,N,5]``
    float32 (Discrete) or a list of two such arrays (MultiDiscrete, [1,N,5] and [1,N,dim_c]).  One HIP launch
    instead of MIOpen's ~45-kernel RNN path per environment step; weights are re-snapshotted by ``refresh()``."""

    def __init__(self, actor, action_type='Discrete', seed=0):
        self.action_type = action_type
        self.fused = FusedActor(actor, seed=seed)
        assert (len(self.fused.heads) == 2) == (action_type == 'MultiDiscrete')

    def refresh(self):
        self.fused.refresh()

    @torch.no_grad()
    def get_exploration_action(self, state):
        obs = torch.from_numpy(np.array([np.stack(state)], dtype='float32')).to(self.fused.device)   # process_obs
        idx = self.fused(obs).cpu().numpy()
        eye = [np.eye(n, dtype=np.float32) for n in self.fused.heads]
        if self.action_type == 'Discrete':
            return eye[0][idx]
        return [eye[0][idx[..., 0]], eye[1][idx[..., 1]]]


def accelerate_trainer(trainer, seed=0):
    """Patch an instance of the reference's ``Trainer`` in place: ``get_exploration_action`` runs on the one-launch
    HIP actor, and the weight snapshot is refreshed after every ``optimize()`` (and ``load_models``).  Everything
    else of the learner is untouched.  Returns the ``FusedExploration`` object."""
    fx = FusedExploration(trainer.actor, getattr(trainer, 'action_type', 'Discrete'), seed=seed)
    trainer.get_exploration_action = fx.get_exploration_action

    def _wrap(name):
        inner = getattr(trainer, name, None)
        if inner is None:
            return

        def wrapped(*a, **k):
            out = inner(*a, **k)
            fx.refresh()
            return out
        setattr(trainer, name, wrapped)
    _wrap('optimize')
    _wrap('load_models')
    return fx


class UniformRandomPolicy(object):
    """i.i.d. uniform action indices (the synthetic-action workload of bench.py).  ``num_actions`` may be a
    list with one entry per agent (simple_speaker_listener: [3, 5])."""

    def __init__(self, num_actions=5, generator=None):
        self.num_actions, self.generator = num_actions, generator

    def __call__(self, obs):
        if isinstance(self.num_actions, (list, tuple)):
            cols = [torch.randint(0, int(n), obs.shape[:1], device=obs.device, dtype=torch.int32, generator=self.generator)
                    for n in self.num_actions]
            return torch.stack(cols, 1)
        return torch.randint(0, self.num_actions, obs.shape[:2], device=obs.device, dtype=torch.int32,
                             generator=self.generator)
