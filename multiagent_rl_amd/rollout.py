"""The rollout loop: host mirror of ``experiments/run.py`` plus its batched, device-resident form.

``run(env, actor, critic, Trainer, scenario_name, action_type, cnt)`` keeps the reference's
signature and call sequence (``experiments/run.py:11-103``): one ``get_exploration_action`` and one
``env.step`` per iteration, ``rew_shared = np.sum(rew_n)`` (:46), ``terminal`` at
``arglist.max_episode_len`` (:50), ``memory.add(obs_n, action_n_env, rew_shared, new_obs_n,
float(done))`` (:52), ``env.reset()`` on done/terminal (:59-60), ``optimize()`` every
``update_rate`` steps after ``warmup_steps`` (:78-81), history pickle + ``save_models`` at the
end (:96-103).  ``env`` and ``Trainer`` are duck-typed exactly as there, so the reference's own
Trainer classes plug in unchanged.

``BatchedRollout`` is what the MI355X is for: B envs stepped by one fused kernel per step, the
policy evaluated on [B,N,D] in HBM, transitions appended to the device replay ring -- no host
synchronisation inside the loop.
"""
import os
import pickle
import time

import numpy as np
import torch

from . import arglist as _default_arglist


def run(env, actor, critic, Trainer, scenario_name=None, action_type='Discrete', cnt=0,
        arglist=None, memory=None, out_dir='Models', log=print, per_agent_transition=False):
    """Training rollout, one env (B = 1), host loop: experiments/run.py:11-103.
    ``per_agent_transition=True`` stores the BiCNet tuple instead (per-agent ``rew_n`` and float
    ``done_n``; experiments/run_BIC.py:46,50).  Returns the history dict it also pickles."""
    return _loop(env, actor, critic, Trainer, scenario_name, action_type, cnt, arglist, memory, out_dir, log,
                 evaluate=False, per_agent_transition=per_agent_transition)


def run_test(env, actor, critic, Trainer, scenario_name=None, action_type='Discrete', cnt=0,
             arglist=None, memory=None, out_dir='Models', log=print):
    """Evaluation rollout: experiments/run.py:106-200 -- ``load_models(appx + scenario + '_fin_' + cnt)``
    first, a progress line every 10 episodes, history pickled to ``test_history_<scenario>_<cnt>.pkl``
    with the replay memory inside, no ``save_models``.  (It still calls ``optimize()``: the reference
    leaves ``arglist.is_training`` True)"""
    return _loop(env, actor, critic, Trainer, scenario_name, action_type, cnt, arglist, memory, out_dir, log,
                 evaluate=True, per_agent_transition=False)


def _action_heads(env, action_type):
    """Head sizes of one agent's action, read the way main.py:51-58 reads them: ``space.n`` for Discrete,
    ``space.high - space.low + 1`` for MultiDiscrete.  (Agents whose spaces differ make ReplayBuffer.add raise.)"""
    space = env.action_space[0]
    if action_type == 'MultiDiscrete' and hasattr(space, 'high'):
        return tuple(int(h - l + 1) for h, l in zip(space.high, space.low))
    return (int(space.n),) if hasattr(space, 'n') else None


class EpisodeLedger(object):
    """Reward accounting of the rollout loops.  ``totals[k]`` = summed reward of episode k over all agents, ``by_agent[i][k]`` the
    share of agent i; the last entry is the episode in progress (which is why a finished run's history ends with a 0, as the
    reference's does: run.py:62-65,96).  ``report()`` closes a window of ``window`` episodes (run.py:84-93)."""

    def __init__(self, num_agents, window):
        self.window = int(window)
        self.totals = [0.0]
        self.by_agent = [[0.0] for _ in range(num_agents)]
        self.window_means, self.window_agent_means = [], []

    @property
    def episodes(self):
        return len(self.totals)

    def credit(self, rew_n):
        for i, r in enumerate(rew_n):
            self.totals[-1] += r
            self.by_agent[i][-1] += r

    def open_next(self):
        self.totals.append(0)
        for track in self.by_agent:
            track.append(0)

    def window_full(self):
        return self.episodes % self.window == 0

    def report(self):
        mean = np.mean(self.totals[-self.window:])
        self.window_means.append(mean)
        for track in self.by_agent:
            self.window_agent_means.append(np.mean(track[-self.window:]))
        return mean

    def history(self):
        """The dict the reference pickles (run.py:96-100); experiments/reward_plot.py:35-50 reads these two keys."""
        return {'reward_episodes': self.totals, 'reward_episodes_by_agents': self.by_agent}


class LearnGate(object):
    """When the learner runs: every ``update_rate`` env steps once more than ``warmup_steps`` have been taken, while
    ``is_training`` (run.py:78-81).  ``due(step)`` is the host loop's per-step test; ``due_between(a, b)`` counts the gate's
    openings inside a chunk of env steps (a, b] for the batched engine, which advances many steps per launch."""

    def __init__(self, cfg):
        self.warmup, self.rate, self.cfg = int(cfg.warmup_steps), int(cfg.update_rate), cfg

    def due(self, train_step):
        return train_step > self.warmup and train_step % self.rate == 0 and self.cfg.is_training

    def due_between(self, before, after):
        if not self.cfg.is_training:
            return 0
        lo = max(before, self.warmup)
        return max(0, after // self.rate - lo // self.rate)


def write_history(hist, out_dir, scenario_name, cnt, evaluate=False):
    """``Models/history_<scenario>_<cnt>.pkl`` (run.py:96-100) / ``test_history_...`` (run.py:193-195)."""
    if out_dir is None:
        return None
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, ('test_history_' if evaluate else 'history_') + scenario_name + '_' + str(cnt) + '.pkl')
    with open(path, 'wb') as fp:
        pickle.dump(hist, fp)
    return path


class _HostSession(object):
    """One run()/run_test() call on a B = 1 env: the order of calls on ``env`` and the learner is the contract
    (tests/golden/run_trace.json, run_test_trace.json, run_multidiscrete_trace.json were recorded from the reference's loop)."""

    def __init__(self, env, learner, cfg, action_type, per_agent_transition, evaluate, log):
        self.env, self.learner, self.cfg, self.log = env, learner, cfg, log
        self.multi = action_type == 'MultiDiscrete'
        self.per_agent, self.evaluate = per_agent_transition, evaluate
        self.ledger = EpisodeLedger(env.n, 10 if evaluate else cfg.save_rate)
        self.gate = LearnGate(cfg)
        self.steps = self.episode_step = 0

    def _act(self, obs_n):
        """Learner output -> what env.step takes: list[N] of float64 one-hot rows (run.py:37-41)."""
        out = self.learner.get_exploration_action(obs_n)
        if self.multi:      # one array per head; an agent's action is the concatenation of its heads' one-hots
            return [np.concatenate(pair, axis=-1) for pair in zip(out[0][0], out[1][0])]
        return [np.array(row) for row in out[0].tolist()]

    def _store(self, obs_n, action_n_env, rew_n, new_obs_n, done_n):
        if self.per_agent:  # BiCNet tuple, run_BIC.py:46,50
            self.learner.memory.add(obs_n, action_n_env, rew_n, new_obs_n, [float(d) for d in done_n])
        else:               # run.py:46,49,52
            self.learner.memory.add(obs_n, action_n_env, np.sum(rew_n), new_obs_n, float(all(done_n)))

    def play(self):
        env, cfg, led = self.env, self.cfg, self.ledger
        obs_n = env.reset()
        clock = time.time()
        self.log('Starting iterations...')
        while led.episodes <= cfg.num_episodes:
            action_n_env = self._act(obs_n)
            new_obs_n, rew_n, done_n, _ = env.step(action_n_env)
            self.episode_step += 1
            self.steps += 1
            over = self.episode_step >= cfg.max_episode_len     # run.py:50 "terminal"
            self._store(obs_n, action_n_env, rew_n, new_obs_n, done_n)
            led.credit(rew_n)
            if over or all(done_n):
                obs_n, self.episode_step = env.reset(), 0
                led.open_next()
            else:
                obs_n = new_obs_n
            if cfg.display:                                      # run.py:70-74: rendering skips learning and reports
                time.sleep(0.1)
                env.render()
                continue
            if self.gate.due(self.steps):
                self.learner.optimize()
            if over and led.window_full():
                mean = led.report()
                self.log('steps: {}, episodes: {}, mean episode reward: {}, time: {}'.format(
                    self.steps, led.episodes, mean, round(time.time() - clock, 3)))
                clock = time.time()
        self.log('...Finished total of {} episodes.'.format(led.episodes))
        return led.history()


def _loop(env, actor, critic, Trainer, scenario_name, action_type, cnt, arglist, memory, out_dir, log,
          evaluate, per_agent_transition):
    cfg = _default_arglist if arglist is None else arglist
    if action_type not in ('Discrete', 'MultiDiscrete'):
        raise ValueError('action_type must be Discrete or MultiDiscrete, got %r' % (action_type,))
    log('observation shape: ', env.observation_space)
    log('action shape: ', env.action_space)
    if memory is None:
        from .replay_buffer import ReplayBuffer
        memory = ReplayBuffer(size=1e+6, act_heads=_action_heads(env, action_type), per_agent=per_agent_transition)
    learner = Trainer(actor, critic, memory, action_type=action_type)
    model_name = scenario_name + '_fin_' + str(cnt)
    if evaluate:
        learner.load_models(getattr(cfg, 'appx', '') + model_name)
    hist = _HostSession(env, learner, cfg, action_type, per_agent_transition, evaluate, log).play()
    if evaluate:
        hist['memory'] = memory
    write_history(hist, out_dir, scenario_name, cnt, evaluate)
    if not evaluate:
        learner.save_models(model_name)
    return hist


class BatchedRollout(object):
    """Device-resident rollout of a ``BatchedParticleEnv`` (auto_reset=True).

    Per step: ``actions = policy(obs)`` -> ``env.step(actions)`` (one fused launch) ->
    ``memory.add_batch(...)`` with the reference's transition tuple (obs, action, shared reward,
    next obs BEFORE reset, done) -- the same bookkeeping as run.py:44-65, vectorised over B.
    (MultiDiscrete rollouts -- actions [B,N,2] -- need a ring built with ``act_heads=(5, dim_c)``; they take the
    per-step ``add_batch`` path.)
    Episode returns are accumulated on the device; nothing is read back inside ``collect``.
    """

    def __init__(self, env, policy, memory=None):
        assert env.cfg.auto_reset, 'BatchedRollout needs an auto-resetting env'
        self.env, self.policy, self.memory = env, policy, memory
        self.obs = env.reset()
        B = env.num_envs
        dev = self.obs.device
        self.env_steps = 0
        self.episode_return = torch.zeros(B, device=dev)
        self.finished_return_sum = torch.zeros((), dtype=torch.float64, device=dev)
        self.finished_episodes = torch.zeros((), dtype=torch.int64, device=dev)
        self._graph = None

    def _bookkeeping(self, rew_shared, terminal, counters=()):
        """Episode returns on the device; ``counters`` = up to two (tensor, delta, modulo) device counters that the
        same launch advances (captured steps: replay cursor, Philox step)."""
        import ctypes as C
        from ._lib import check
        p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        cs = list(counters) + [(None, 0, 0)] * (2 - len(counters))
        args = []
        for t, d, m in cs:
            args += [None if t is None else p(t), int(d), int(m)]
        check(self.env.lib.pw_rollout_tail(p(rew_shared), p(terminal), self.env.num_envs, p(self.episode_return),
                                           p(self.finished_return_sum), p(self.finished_episodes), *args,
                                           self.env._stream()))

    def _sink(self, obs, actions, out, parity=None, step_counter=None):
        """Replay append + episode-return bookkeeping: ONE launch when there is a device ring."""
        if self.memory is not None and actions.dtype == torch.int32 and actions.dim() == 2 and \
                out.get('final_obs') is not None:
            self.memory.add_batch_tail(obs, actions, out['rew_shared'], out['obs'], out['final_obs'], out['terminal'],
                                       self.episode_return, self.finished_return_sum, self.finished_episodes,
                                       step_counter=step_counter, parity=parity)
            return
        counters = [] if step_counter is None else [(step_counter, 1, 0)]
        if self.memory is not None:
            self.memory.add_batch(obs, actions, out['rew_shared'], out['obs'], out.get('final_obs'), out['terminal'],
                                  device_cursor=parity is not None, advance_cursor=False)
            if parity is not None:
                counters.append((self.memory._cursor, self.env.num_envs, self.memory._maxsize))
        self._bookkeeping(out['rew_shared'], out['terminal'], counters)

    def step(self):
        obs = self.obs
        actions = self.policy(obs)
        nxt, rew, done, info = self.env.step(actions)
        self._sink(obs, actions, dict(info, obs=nxt))
        self.obs = nxt
        self.env_steps += self.env.num_envs
        return nxt, rew, done, info

    def capture(self, steps_per_replay=2):
        """Capture ``steps_per_replay`` (even) rollout steps -- policy forward, sampling, the fused env
        step, the replay append, the return bookkeeping -- into ONE hipGraph.  Afterwards ``collect``
        replays it: one host call per ``steps_per_replay`` steps instead of ~20 launches per step.
        Values that change per step live in device memory (replay cursor, Philox step; see
        pw_counter_add).  Observations ping-pong between two static buffers."""
        assert steps_per_replay % 2 == 0 and self._graph is None
        env, dev = self.env, self.obs.device
        self._static = [dict(env.alloc_outputs(), obs_in=None) for _ in range(2)]
        self._obs_buf = [self.obs.clone(), torch.empty_like(self.obs)]
        if hasattr(self.policy, 'begin_graph'):
            self.policy.begin_graph()
        defer = hasattr(self.policy, 'defer_step_advance')
        if defer:
            self.policy.defer_step_advance = True
        if self.memory is not None:
            if self.memory._store is None:
                self.memory._allocate(env.n, env.obs_dim)
            self.memory.sync_cursor()

        def one(i):
            src, dst = self._obs_buf[i & 1], self._obs_buf[(i + 1) & 1]
            out = self._static[i & 1]
            out['obs'] = dst
            actions = self.policy(src)
            env.step(actions, out=out)
            self._sink(src, actions, out, parity=i & 1, step_counter=self.policy._step_dev if defer else None)

        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):  # warm-up outside capture (allocator, rocBLAS workspaces, lazy inits)
            for i in range(2):
                one(i)
        torch.cuda.current_stream(dev).wait_stream(side)
        if self.memory is not None:
            self.memory.note_graph_adds(2 * env.num_envs)
        if hasattr(self.policy, 'calls'):
            pass  # begin_graph() moved the step counter to the device
        self.env_steps += 2 * env.num_envs
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            for i in range(steps_per_replay):
                one(i)
        self._graph_steps = steps_per_replay
        self.obs = self._obs_buf[0]
        return self

    def collect(self, num_steps):
        if self._graph is not None:
            n = max(1, num_steps // self._graph_steps)
            for _ in range(n):
                self._graph.replay()
            steps = n * self._graph_steps
            if self.memory is not None:
                self.memory.note_graph_adds(steps * self.env.num_envs)
            self.env_steps += steps * self.env.num_envs
            return self.env_steps
        for _ in range(num_steps):
            self.step()
        return self.env_steps

    def collect_one_launch(self, num_steps, chunk=100, keep_outputs=False):
        """The same rollout with ONE launch per ``chunk`` steps instead of three per step: ``FusedActor.rollout``
        runs policy + sampling + env step for the whole chunk with observations / actions / world state resident on
        the CU, and writes the transitions straight into the replay ring and the episode statistics from the same
        kernel (``pw_rollout_sink``).  ``keep_outputs=True`` also materialises the chunk's [T, ...] step outputs
        (``self.last_chunk``).  Needs a FusedActor and a simple_spread fast-path, homogeneous-role simple_tag or simple_reference env (the latter
        with a ring built with ``act_heads=(5, dim_c)``); stores exactly what ``collect``
        stores (statistics up to float64 summation order)."""
        assert self._graph is None and hasattr(self.policy, 'rollout')
        stats = (self.episode_return, self.finished_return_sum, self.finished_episodes)
        done_steps = 0
        while done_steps < num_steps:
            T = min(chunk, num_steps - done_steps)
            if keep_outputs:
                if getattr(self, '_chunk_T', None) != T:
                    self._chunk_T, self.last_chunk = T, None
                self.last_chunk = self.policy.rollout(self.env, T, self.last_chunk, memory=self.memory, stats=stats)
            else:
                self.policy.rollout(self.env, T, False, memory=self.memory, stats=stats)
            done_steps += T
            self.env_steps += T * self.env.num_envs
        self.obs = self.env.observe()
        return self.env_steps

    def stats(self):
        """Host read-back (synchronises): mean return of the episodes finished so far."""
        n = float(self.finished_episodes.item())
        return dict(env_steps=self.env_steps, episodes=int(n),
                    mean_episode_reward=float(self.finished_return_sum.item()) / n if n else float('nan'))
