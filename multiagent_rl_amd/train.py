"""The batched counterpart of the reference's entry path: ``main.py:29-67`` (seed protocol, dims read from the env, the
nets, the call of ``run``) and ``experiments/run.py:11-103`` (rollout -> ``memory.add`` -> learner gate -> report -> history
pickle + ``save_models``), for B environments advanced together on the GPU.

What changes against the B = 1 loop of ``rollout.run`` is only WHERE the work happens:

* rollout: ``BatchedRollout.collect_one_launch`` -- ``chunk`` batched steps of actor + Gumbel sampling + env step + replay
  append + episode statistics as ONE launch (``pw_policy_rollout``), no host synchronisation inside a chunk;
* learner gate (run.py:78-81): ``optimize()`` runs once per ``update_rate`` ENV-steps once more than ``warmup_steps`` have been
  taken -- counted in env-steps exactly as the reference counts ``train_step`` (one per ``env.step`` of one world); a chunk
  advances ``chunk * B`` of them, so the gate opens ``LearnGate.due_between(before, after)`` times after it (bounded by
  ``max_updates_per_chunk``: at B = 4096 a 100-step chunk would otherwise owe 4096 updates; ``stats`` reports ``updates_owed``
  beside ``updates_run`` and the log says so when they differ);
* after the learner ran, the rollout's weight snapshot follows it (``FusedActor.refresh``) and, with several ranks, the
  learner rank's actor goes to every rollout rank as one flat broadcast (``dist.broadcast_actor``);
* multi-GPU: every rank rolls out its shard of the env batch; the transitions of all ranks reach the learner rank's ring
  through ``dist.FullTransitionGather`` (state-only wire blocks for simple_spread).

The Trainer is duck-typed exactly as in ``experiments/run.py:21``: ``Trainer(actor, critic, memory, action_type=...)`` with
``.actor``, ``.optimize()``, ``.save_models(name)`` -- the reference's own ``rls.agent.multiagent.ddpg_gumbel_fix.Trainer`` plugs
in unchanged (its ``process_batch`` reads the device ring through ``make_index`` / ``sample_index``).
"""
import time

import numpy as np
import torch

from . import arglist as _default_arglist
from .rollout import BatchedRollout, LearnGate, write_history


def seed_everything(seed):
    """main.py:41-49: ``np.random.seed``, ``torch.manual_seed``, ``torch.cuda.manual_seed_all`` (the env's own seed is the
    ``seed`` argument of ``make_batched_env``: the batched env draws its resets from Philox keyed by it, not from NumPy)."""
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    return seed


def dims_from_env(env):
    """main.py:51-58: observation length, per-agent action size(s), and 'Discrete' | 'MultiDiscrete'."""
    dim_obs = env.observation_space[0].shape[0]
    space = env.action_space[0]
    if hasattr(space, 'high'):
        return dim_obs, (space.high + 1).tolist(), 'MultiDiscrete'
    return dim_obs, space.n, 'Discrete'


class ChunkLedger(object):
    """Per-episode returns out of a chunk's [T, B] outputs, vectorised: ``reward_episodes`` / ``reward_episodes_by_agents``
    get one entry per FINISHED episode, in the order the B = 1 loop of ``experiments/run.py:55-65`` would have appended them had it
    stepped the B worlds side by side: by END STEP, then by env (``terminal.nonzero()`` order) -- the lists
    ``experiments/reward_plot.py:35-50`` reads.  ``history()`` closes them the way run.py:62-65 leaves its lists: behind the finished
    episodes stands the entry of the episode in progress -- here one per env, in env order (B = 1: the reference's single trailing
    entry); ``open_episodes`` says how many, so a reader that wants finished episodes only drops that many from the end."""

    def __init__(self, num_envs, num_agents, device):
        self.carry = torch.zeros(num_envs, num_agents, dtype=torch.float64, device=device)
        self.totals, self.by_agent = [], [[] for _ in range(num_agents)]

    @torch.no_grad()
    def absorb(self, rew, terminal):
        """rew [T,B,N] float32, terminal [T,B] bool."""
        T, B, N = rew.shape
        term = terminal.bool()
        ends_before = torch.cumsum(term.long(), 0) - term.long()            # episode index of each step inside the chunk
        n_seg = int(ends_before.max().item()) + 1
        sums = torch.zeros(n_seg, B, N, dtype=torch.float64, device=rew.device)
        sums.scatter_add_(0, ends_before[:, :, None].expand(T, B, N), rew.double())
        sums[0] += self.carry
        t_end, e_end = torch.nonzero(term, as_tuple=True)                   # row-major: by end step, then by env
        if t_end.numel():
            per_agent = sums[ends_before[t_end, e_end], e_end].cpu().numpy()  # [finished episodes, N]
            self.totals.extend(per_agent.sum(1).tolist())
            for i in range(N):
                self.by_agent[i].extend(per_agent[:, i].tolist())
        ends = torch.cumsum(term.long(), 0)[-1]                             # finished episodes per env
        last = torch.clamp(ends, max=n_seg - 1)
        open_ = sums[last, torch.arange(B, device=rew.device)]
        self.carry = torch.where((ends < n_seg)[:, None], open_, torch.zeros_like(open_))

    def mean_of_last(self, n):
        """run.py:86: ``np.mean(episode_rewards[-save_rate:])`` over finished episodes."""
        tail = self.totals[-int(n):]
        return float(np.mean(tail)) if tail else float('nan')

    def history(self, include_open=True):
        totals, by_agent, n_open = list(self.totals), [list(x) for x in self.by_agent], 0
        if include_open:
            open_ = self.carry.cpu().numpy()
            n_open = open_.shape[0]
            totals.extend(open_.sum(1).tolist())
            for i in range(open_.shape[1]):
                by_agent[i].extend(open_[:, i].tolist())
        return {'reward_episodes': totals, 'reward_episodes_by_agents': by_agent, 'open_episodes': n_open}


def merge_histories(parts):
    """Per-rank histories -> one, in RANK order (rank r owns global envs [r B, (r + 1) B)): what the learner rank pickles."""
    out = {'reward_episodes': [], 'reward_episodes_by_agents': [[] for _ in parts[0]['reward_episodes_by_agents']],
           'open_episodes': [p.get('open_episodes', 0) for p in parts], 'episodes_per_rank': [len(p['reward_episodes']) for p in parts]}
    for p in parts:
        out['reward_episodes'].extend(p['reward_episodes'])
        for i, lst in enumerate(p['reward_episodes_by_agents']):
            out['reward_episodes_by_agents'][i].extend(lst)
    return out


def train_batched(env, actor, critic, Trainer, scenario_name, action_type='Discrete', cnt=0, arglist=None, memory=None,
                  out_dir='Models', log=print, chunk=100, max_updates_per_chunk=None, policy_seed=None,
                  per_episode_history=True, gather=None, rank=0, world=1, make_rollout=None, ring='rows'):
    """``experiments/run.py:run`` for a ``BatchedParticleEnv`` (auto_reset=True).  Runs until ``arglist.num_episodes`` episodes
    have finished (over all envs of this rank), then pickles the history with the reference's keys and saves the models.
    ``gather`` (a ``dist.FullTransitionGather``) switches to the multi-GPU form: every rank rolls out into the gather's wire
    block, the learner lives on rank 0.  ``make_rollout(env, actor, memory, seed) -> (fused_actor, batched_rollout)`` replaces
    the HIP pair (tests drive the control flow without a GPU).  ``ring='state'`` (single-rank form; simple_spread with the local
    observation, simple_tag): the rollout's ring sink fills a STATE ring -- {vel, pos} + the episode's landmarks per transition, a
    third of the bytes -- and ``sample_index`` rebuilds the rows the learner trains on (bit-identical batches).  Returns the history
    dict (with ``stats``: env-steps, updates, wall time)."""
    from .replay_buffer import ReplayBuffer
    cfg = _default_arglist if arglist is None else arglist
    if action_type not in ('Discrete', 'MultiDiscrete'):
        raise ValueError('action_type must be Discrete or MultiDiscrete, got %r' % (action_type,))
    B, N = env.num_envs, env.n
    log('observation shape: ', env.observation_space)
    log('action shape: ', env.action_space)
    learns = rank == 0
    if memory is None and learns:
        if gather is not None:
            memory = gather.memory
        else:
            heads = tuple(int(h) for h in (env.action_space[0].high + 1)) if action_type == 'MultiDiscrete' else None
            kw = dict(act_heads=heads) if heads else {}
            if ring == 'state':
                if heads or getattr(env, 'scenario_name', None) not in ('simple_spread', 'simple_tag'):
                    raise ValueError("ring='state' serves simple_spread (local observation) and simple_tag")
                kw = dict(state_ring=dict(scenario=env.scenario_name, num_landmarks=env.num_landmarks,
                                          num_adversaries=env.cfg.num_adversaries if env.scenario_name == 'simple_tag' else 0))
            memory = ReplayBuffer(int(1e6), N, env.obs_dim, device_index=True, **kw)   # the batch's indices drawn on the device
    learner = Trainer(actor, critic, memory, action_type=action_type)
    seed = (cnt + 12345678 if policy_seed is None else policy_seed) + rank
    if make_rollout is None:
        from .policy import FusedActor
        fused = FusedActor(learner.actor, seed=seed)
        ro = BatchedRollout(env, fused, None if gather is not None else memory)
    else:
        fused, ro = make_rollout(env, learner.actor, None if gather is not None else memory, seed)
    gate = LearnGate(cfg)
    # one ledger per rank in EVERY mode: a gather's chunk leaves rew / terminal in its side buffers, a MultiDiscrete chunk has the
    # same [T, B, N] rewards as a Discrete one (run.py:96-100 pickles one entry per episode whatever the action type)
    ledger = ChunkLedger(B, N, ro.obs.device) if per_episode_history else None
    updates = updates_owed = reports = 0
    t_begin = clock = time.time()
    log('Starting iterations...')
    obs0 = ro.obs
    while int(ro.finished_episodes.item()) < cfg.num_episodes:      # one read-back per chunk
        before = ro.env_steps * world
        if gather is not None:
            out = gather.outputs()
            fused.rollout(env, chunk, out, stats=(ro.episode_return, ro.finished_return_sum, ro.finished_episodes))
            if ledger is not None:          # before the next launch reuses the side buffers (same stream: ordered)
                ledger.absorb(out['rew'], out['terminal'])
            gather(obs0)
            obs0 = out['obs'][chunk - 1]
            ro.env_steps += chunk * B
        else:
            ro.collect_one_launch(chunk, chunk=chunk, keep_outputs=ledger is not None)
            if ledger is not None:
                ledger.absorb(ro.last_chunk['rew'], ro.last_chunk['terminal'])
        due = gate.due_between(before, ro.env_steps * world)
        updates_owed += due                 # what run.py:78-81 would have run at one optimize() per update_rate env-steps
        if max_updates_per_chunk is not None:
            due = min(due, int(max_updates_per_chunk))
        if due and learns and len(memory) >= getattr(cfg, 'batch_size', 1):
            if gather is not None:
                gather.wait_ingested()          # the root appends on a side stream: order the learner's reads behind them
            for _ in range(due):
                learner.optimize()
            updates += due
            if gather is not None and hasattr(gather, 'mark_reads_done'):
                gather.mark_reads_done()        # ... and the next appends behind these reads (the ring wraps)
        if due:
            if world > 1:
                from .dist import broadcast_actor
                broadcast_actor(learner.actor, src=0, fused=fused)
            else:
                fused.refresh()
        st_eps = int(ro.finished_episodes.item())
        if st_eps // cfg.save_rate > reports:
            reports = st_eps // cfg.save_rate
            s = ro.stats()
            # run.py:86: the mean over the LAST save_rate episodes (the all-time mean only where no ledger is kept)
            mean = ledger.mean_of_last(cfg.save_rate) if ledger is not None else s['mean_episode_reward']
            log('steps: {}, episodes: {}, mean episode reward: {}, time: {}'.format(
                ro.env_steps, s['episodes'], mean, round(time.time() - clock, 3)))
            clock = time.time()
    if gather is not None:
        gather.finish()
    s = ro.stats()
    hist = ledger.history() if ledger is not None else {'reward_episodes': [], 'reward_episodes_by_agents': [[] for _ in range(N)],
                                                        'open_episodes': 0}
    if world > 1 and ledger is not None:    # every rank's episodes to the learner rank, concatenated in rank order
        import torch.distributed as tdist
        parts = [None] * world if learns else None
        tdist.gather_object(hist, parts, dst=0)
        if learns:
            hist = merge_histories(parts)
    hist['stats'] = dict(env_steps=ro.env_steps, episodes=s['episodes'], mean_episode_reward=s['mean_episode_reward'],
                         updates=updates, updates_run=updates, updates_owed=updates_owed,
                         updates_skipped=updates_owed - updates if learns else None,
                         wall_s=time.time() - t_begin, num_envs=B, chunk=chunk, world=world)
    if learns and updates < updates_owed:
        log('learner: {} of {} owed updates run (max_updates_per_chunk={}, or the ring was still empty)'.format(
            updates, updates_owed, max_updates_per_chunk))
    log('...Finished total of {} episodes.'.format(s['episodes']))
    if learns:
        write_history(hist, out_dir, scenario_name, cnt)
        learner.save_models(scenario_name + '_fin_' + str(cnt))
    return hist
