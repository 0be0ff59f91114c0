"""The batched entry path (multiagent_rl_amd/train.py, examples/train_batched.py) -- counterpart of main.py:29-67 +
experiments/run.py:11-103.  CPU: the control flow with a stub Trainer and a stub rollout (collect -> optimize() gate in env-steps
-> actor refresh -> report -> history + save_models); GPU: two chunks of the real thing."""
import os
import pickle
import sys

import numpy as np
import pytest
import torch

from multiagent_rl_amd.rollout import LearnGate
from multiagent_rl_amd.train import ChunkLedger, dims_from_env, train_batched

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Args(object):
    max_episode_len, num_episodes, is_training = 25, 64, True
    batch_size, warmup_steps, update_rate, save_rate, display = 8, 1024, 100, 32, False


class _Space(object):
    def __init__(self, n):
        self.n, self.shape = n, (10,)


class _StubEnv(object):
    num_envs, n, obs_dim = 16, 3, 10
    observation_space = [_Space(5)] * 3
    action_space = [_Space(5)] * 3


class _StubTrainer(object):
    trace = None

    def __init__(self, actor, critic, memory, action_type='Discrete'):
        self.actor, self.memory = actor, memory
        self.trace.append(('trainer_init', action_type))

    def optimize(self):
        self.trace.append(('optimize', len(self.memory)))

    def save_models(self, name):
        self.trace.append(('save_models', name))


class _StubMemory(object):
    def __init__(self):
        self.n = 0

    def __len__(self):
        return self.n


class _StubFused(object):
    def __init__(self, trace):
        self.trace = trace

    def refresh(self):
        self.trace.append(('refresh',))


class _StubRollout(object):
    """BatchedRollout's surface as train_batched uses it: every env finishes an episode every 25 steps."""

    def __init__(self, env, memory, trace):
        self.env, self.memory, self.trace = env, memory, trace
        self.obs = torch.zeros(env.num_envs, env.n, env.obs_dim)
        self.env_steps, self.steps = 0, 0
        self.episode_return = torch.zeros(env.num_envs)
        self.finished_return_sum = torch.zeros((), dtype=torch.float64)
        self.finished_episodes = torch.zeros((), dtype=torch.int64)

    def collect_one_launch(self, num_steps, chunk=100, keep_outputs=False):
        B, N = self.env.num_envs, self.env.n
        self.trace.append(('collect', num_steps, keep_outputs))
        t = torch.arange(self.steps, self.steps + num_steps)
        term = ((t + 1) % 25 == 0)[:, None].expand(num_steps, B)
        self.last_chunk = dict(rew=-torch.ones(num_steps, B, N), terminal=term)
        self.steps += num_steps
        self.env_steps += num_steps * B
        self.memory.n += num_steps * B
        self.finished_episodes += int(term[:, 0].sum()) * B
        self.finished_return_sum += -75.0 * int(term[:, 0].sum()) * B

    def stats(self):
        n = float(self.finished_episodes)
        return dict(env_steps=self.env_steps, episodes=int(n), mean_episode_reward=float(self.finished_return_sum) / n if n else float('nan'))


def test_learn_gate_counts_openings_in_env_steps():
    """run.py:78-81: optimize() when train_step > warmup_steps and train_step % update_rate == 0."""
    cfg = _Args()
    g = LearnGate(cfg)
    per_step = [s for s in range(1, 3001) if g.due(s)]
    assert per_step[0] == 1100 and per_step[-1] == 3000 and len(per_step) == 20
    for a, b in ((0, 3000), (0, 1024), (1024, 1100), (1099, 1100), (1100, 1100), (950, 2150), (2999, 3000), (0, 99)):
        assert g.due_between(a, b) == len([s for s in per_step if a < s <= b]), (a, b)
    cfg2 = _Args()
    cfg2.is_training = False
    assert LearnGate(cfg2).due_between(0, 10 ** 6) == 0


def test_chunk_ledger_splits_a_chunk_into_episode_returns():
    """Per-episode returns out of [T, B, N] rewards with desynchronised episode ends, carried across chunks."""
    torch.manual_seed(0)
    T, B, N = 40, 5, 3
    rew = torch.randn(2 * T, B, N)
    term = torch.zeros(2 * T, B, dtype=torch.bool)
    for e in range(B):
        term[(7 + 3 * e)::(11 + e), e] = True
    led = ChunkLedger(B, N, 'cpu')
    led.absorb(rew[:T], term[:T])
    led.absorb(rew[T:], term[T:])
    # what experiments/run.py:55-65 appends when it steps the B worlds side by side: at every step, for every env in order, a
    # finished episode's per-agent sums -- (END STEP, env) order, whatever the chunking
    want, run = [], np.zeros((B, N))
    for t in range(2 * T):
        for e in range(B):
            run[e] += rew[t, e].double().numpy()
            if term[t, e]:
                want.append(run[e].copy())
                run[e] = 0.0
    got = np.stack([np.array(a) for a in led.by_agent], 1)
    np.testing.assert_allclose(got, np.stack(want), rtol=0, atol=1e-9)
    np.testing.assert_allclose(led.totals, np.stack(want).sum(1), rtol=0, atol=1e-9)
    h = led.history()
    assert sorted(h.keys()) == ['open_episodes', 'reward_episodes', 'reward_episodes_by_agents'] and h['open_episodes'] == B
    # behind the finished episodes: the episode in progress of every env (run.py:62-65 leaves ONE such trailing entry at B = 1)
    assert len(h['reward_episodes']) == len(want) + B and len(h['reward_episodes_by_agents'][0]) == len(want) + B
    np.testing.assert_allclose(h['reward_episodes'][-B:], run.sum(1), rtol=0, atol=1e-9)
    assert len(led.history(include_open=False)['reward_episodes']) == len(want)
    assert abs(led.mean_of_last(3) - np.mean(np.stack(want).sum(1)[-3:])) < 1e-9


def test_train_batched_orders_collect_learn_refresh_and_writes_the_history(tmp_path):
    """The loop of train_batched with a stub Trainer and a stub rollout: chunks of 10 steps x 16 envs = 160 env-steps; the
    gate opens once per 100 env-steps after 1024; every batch of updates is followed by ONE refresh of the rollout's weight
    snapshot; it stops once num_episodes episodes finished, pickles the reference's history keys and saves the models."""
    trace = []
    _StubTrainer.trace = trace
    env, mem, cfg = _StubEnv(), _StubMemory(), _Args()
    logs = []
    hist = train_batched(env, 'actor', 'critic', _StubTrainer, 'simple_spread', 'Discrete', cnt=3, arglist=cfg, memory=mem,
                         out_dir=str(tmp_path), log=lambda *a: logs.append(a), chunk=10,
                         make_rollout=lambda e, actor, memory, seed: (_StubFused(trace), _StubRollout(e, memory, trace)))
    kinds = [e[0] for e in trace]
    assert kinds[0] == 'trainer_init' and kinds[-1] == 'save_models' and trace[-1] == ('save_models', 'simple_spread_fin_3')
    n_chunks = kinds.count('collect')
    assert n_chunks == 10                                   # 64 episodes = 4 per env: 100 steps = 10 chunks of 10
    # gate: env-steps after chunk i = 160 i; openings in (160 (i-1), 160 i] beyond 1024
    steps, want = 0, []
    g = LearnGate(cfg)
    for i in range(n_chunks):
        want.append(('collect', 10, True))
        due = g.due_between(steps, steps + 160)
        steps += 160
        want += [('optimize', steps)] * due + ([('refresh',)] if due else [])
    assert trace[1:-1] == want
    assert kinds.count('optimize') == g.due_between(0, 1600) == 6      # at 1100, 1200, ..., 1600 env-steps
    assert hist['stats']['env_steps'] == 1600 and hist['stats']['updates'] == 6 and hist['stats']['episodes'] == 64
    assert hist['stats']['updates_owed'] == hist['stats']['updates_run'] == 6 and hist['stats']['updates_skipped'] == 0
    # 64 finished episodes + the 16 envs' episodes in progress (just reset: 0), as run.py:62-65 leaves its trailing entry
    assert hist['open_episodes'] == 16 and len(hist['reward_episodes']) == 64 + 16
    assert all(abs(x + 75.0) < 1e-9 for x in hist['reward_episodes'][:64]) and all(x == 0 for x in hist['reward_episodes'][64:])
    assert len(hist['reward_episodes_by_agents']) == 3 and len(hist['reward_episodes_by_agents'][0]) == 64 + 16
    saved = pickle.load(open(tmp_path / 'history_simple_spread_3.pkl', 'rb'))
    assert saved['reward_episodes'] == hist['reward_episodes']
    assert any('mean episode reward' in str(a[0]) for a in logs)          # the report line at save_rate episodes
    # a cap on the updates owed per chunk
    trace.clear()
    logs.clear()
    capped = train_batched(env, 'actor', 'critic', _StubTrainer, 'simple_spread', 'Discrete', arglist=cfg, memory=_StubMemory(),
                           out_dir=None, log=lambda *a: logs.append(a), chunk=50, max_updates_per_chunk=2,
                           make_rollout=lambda e, actor, memory, seed: (_StubFused(trace), _StubRollout(e, memory, trace)))
    assert [e[0] for e in trace].count('optimize') == 2      # chunk 1: 800 env-steps (warm-up), chunk 2: 6 owed, 2 run
    st = capped['stats']                                       # ... and the run says so instead of dropping them silently
    assert (st['updates_owed'], st['updates_run'], st['updates_skipped']) == (6, 2, 4)
    assert any('2 of 6 owed updates' in str(a[0]) for a in logs)
    # MultiDiscrete runs keep the per-episode history too (run.py:96-100 pickles it whatever the action type)
    trace.clear()
    md = train_batched(env, 'actor', 'critic', _StubTrainer, 'simple_reference', 'MultiDiscrete', arglist=cfg, memory=_StubMemory(),
                       out_dir=None, log=lambda *a: None, chunk=10,
                       make_rollout=lambda e, actor, memory, seed: (_StubFused(trace), _StubRollout(e, memory, trace)))
    assert len(md['reward_episodes']) == 64 + 16 and trace[1] == ('collect', 10, True)
    assert dims_from_env(env) == (10, 5, 'Discrete')


@pytest.mark.gpu
def test_entry_script_two_chunks_on_the_gpu(tmp_path):
    """examples/train_batched.py end to end on cuda:0 with the stand-in learner: 256 envs, 50-step chunks, two chunks
    (4 episodes per env), learner updates between them, history + model files written, actor weights changed."""
    sys.path.insert(0, os.path.join(ROOT, 'examples'))
    import train_batched as entry
    from multiagent_rl_amd import arglist
    saved = (arglist.num_episodes, arglist.save_rate, arglist.warmup_steps, arglist.batch_size)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        arglist.warmup_steps, arglist.batch_size = 1024, 1024
        res = entry.main(['--scenario', 'simple_spread', '--envs', '256', '--agents', '3', '--episodes', '1024', '--chunk', '50',
                          '--save-rate', '512', '--max-updates-per-chunk', '3', '--out-dir', str(tmp_path / 'Models')])
    finally:
        os.chdir(cwd)
        arglist.num_episodes, arglist.save_rate, arglist.warmup_steps, arglist.batch_size = saved
    (name, cnt, st), = res
    assert name == 'simple_spread' and cnt == 0 and st['episodes'] == 1024 and st['env_steps'] == 100 * 256
    assert st['updates'] == 6 and np.isfinite(st['mean_episode_reward']) and st['mean_episode_reward'] < 0
    hist = pickle.load(open(tmp_path / 'Models' / 'history_simple_spread_0.pkl', 'rb'))
    assert hist['open_episodes'] == 256 and len(hist['reward_episodes']) == 1024 + 256 and len(hist['reward_episodes_by_agents']) == 3
    assert abs(np.mean(hist['reward_episodes'][:1024]) - st['mean_episode_reward']) < 1e-3 * abs(st['mean_episode_reward'])
    assert os.path.exists(tmp_path / 'Models' / 'simple_spread_fin_0_actor.pt')


@pytest.mark.gpu
def test_train_batched_through_the_full_gather_on_one_rank(tmp_path):
    """The multi-rank form of train_batched with world = 1 on cuda:0 (no process group needed: a one-rank gather posts no transfer):
    the rollout writes into the gather's wire block (state-only), the ring the learner samples is the gather's, filled one chunk
    late on the side stream -- and holds exactly the transitions the same rollout's own ring sink stores."""
    sys.path.insert(0, os.path.join(ROOT, 'examples'))
    from madr_learner import CriticNetwork, Trainer
    from multiagent_rl_amd import make_batched_env
    from multiagent_rl_amd.dist import FullTransitionGather
    from multiagent_rl_amd.policy import ActorNetwork, FusedActor
    from multiagent_rl_amd.replay_buffer import ReplayBuffer

    class Cfg(_Args):
        num_episodes, batch_size, warmup_steps, update_rate, save_rate = 512, 256, 10 ** 9, 100, 10 ** 9   # never learns: weights fixed

    torch.manual_seed(3)
    B, N, T = 128, 3, 50
    dev = torch.device('cuda', 0)
    mk = lambda: make_batched_env('simple_spread', B, n=N, auto_reset=True, max_episode_len=25, seed=12345678)  # noqa: E731
    env = mk()
    actor = ActorNetwork(env.obs_dim, 5)
    gather = FullTransitionGather(env, T, 0, 1, dev, capacity=4 * T * B)
    assert gather.state_wire
    hist = train_batched(env, actor, CriticNetwork(env.obs_dim + 5), Trainer, 'simple_spread', 'Discrete', cnt=0, arglist=Cfg(),
                         out_dir=str(tmp_path), log=lambda *a: None, chunk=T, gather=gather, rank=0, world=1)
    assert hist['stats']['episodes'] == 512 and hist['stats']['env_steps'] == 2 * T * B and hist['stats']['updates'] == 0
    assert len(gather.memory) == 2 * T * B
    # the gather mode keeps the per-episode history as well: 4 finished episodes per env, then the 128 open ones
    assert hist['open_episodes'] == B and len(hist['reward_episodes']) == 512 + B
    assert abs(np.mean(hist['reward_episodes'][:512]) - hist['stats']['mean_episode_reward']) < 1e-3 * abs(hist['stats']['mean_episode_reward'])
    # the same rollout with the kernel's own ring sink: same seeds, same weights -> the same ring
    env2 = mk()
    want = ReplayBuffer(4 * T * B, N, env2.obs_dim)
    env2.reset()
    fused = FusedActor(actor.to(dev), seed=12345678)
    for _ in range(2):
        fused.rollout(env2, T, False, memory=want)
    torch.cuda.synchronize()
    for name in ('obs', 'next_obs', 'act', 'rew', 'done'):
        assert torch.equal(getattr(gather.memory, name)[:2 * T * B], getattr(want, name)[:2 * T * B]), name


@pytest.mark.gpu
@pytest.mark.parametrize('scenario', ['simple_spread', 'simple_tag', 'simple_reference'])
def test_train_batched_learns_from_the_gathers_ring_in_every_scenario(tmp_path, scenario):
    """The multi-rank form of train_batched with world = 1 on cuda:0, the learner RUNNING: simple_spread and simple_tag through state-only
    wire blocks into a STATE ring (sample_index rebuilds the rows the learner trains on), simple_reference (MultiDiscrete, two action
    heads) through compact-row blocks into its two-head ring.  The history has one entry per finished episode (+ the open ones), the
    owed / run update counts are reported, the actor's weights moved."""
    sys.path.insert(0, os.path.join(ROOT, 'examples'))
    from madr_learner import CriticNetwork, Trainer
    from multiagent_rl_amd import make_batched_env
    from multiagent_rl_amd.dist import FullTransitionGather
    from multiagent_rl_amd.policy import ActorNetwork

    class Cfg(_Args):
        num_episodes, batch_size, warmup_steps, update_rate, save_rate = 1024, 256, 1024, 1000, 512

    torch.manual_seed(5)
    B, T = 128, 50
    dev = torch.device('cuda', 0)
    kw = dict(n=3) if scenario == 'simple_spread' else dict(num_adversaries=2, num_good=1) if scenario == 'simple_tag' else {}
    env = make_batched_env(scenario, B, auto_reset=True, max_episode_len=25, seed=12345678, **kw)
    dim_obs, dim_action, action_type = dims_from_env(env)
    assert action_type == ('MultiDiscrete' if scenario == 'simple_reference' else 'Discrete')
    actor = ActorNetwork(dim_obs, dim_action)
    before = actor.dense1.module.weight.detach().clone()
    n_act = sum(dim_action) if isinstance(dim_action, list) else dim_action
    gather = FullTransitionGather(env, T, 0, 1, dev, capacity=6 * T * B, ring='rows' if scenario == 'simple_reference' else 'state')
    assert gather.ref_wire == (scenario == 'simple_reference') and (gather.memory.state_ring is not None) == (scenario != 'simple_reference')
    hist = train_batched(env, actor, CriticNetwork(dim_obs + n_act), Trainer, scenario, action_type, cnt=0, arglist=Cfg(),
                         out_dir=str(tmp_path), log=lambda *a: None, chunk=T, gather=gather, rank=0, world=1, max_updates_per_chunk=2)
    st = hist['stats']
    assert st['episodes'] == 1024 and st['env_steps'] == 4 * T * B and len(gather.memory) == 4 * T * B
    assert st['updates_run'] >= 2 and st['updates_owed'] > st['updates_run'] and st['updates_skipped'] == st['updates_owed'] - st['updates_run']
    assert hist['open_episodes'] == B and len(hist['reward_episodes']) == 1024 + B and len(hist['reward_episodes_by_agents']) == env.n
    assert not torch.equal(before, actor.dense1.module.weight.detach().cpu())
    saved = pickle.load(open(tmp_path / ('history_%s_0.pkl' % scenario), 'rb'))
    assert saved['reward_episodes'] == hist['reward_episodes']
    # what the learner sampled from: finite rows of the right shape (state ring: rebuilt by pw_replay_gather)
    obs, act, rew, nxt, done = gather.memory.sample_index(list(range(0, 4 * T * B, 97)))
    assert obs.shape[1:] == (env.n, dim_obs) and act.shape[-1] == n_act and torch.isfinite(obs).all() and torch.isfinite(nxt).all()
    assert (act.sum(-1) == (2 if scenario == 'simple_reference' else 1)).all()


def test_learn_gate_and_ledger_properties():
    """Property checks (hypothesis): chunked gate openings add up to the per-step gate whatever the chunking; the ledger's episode
    returns do not depend on where the chunks are cut."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=60, deadline=None)
    @given(warm=st.integers(0, 500), rate=st.integers(1, 97), cuts=st.lists(st.integers(1, 400), min_size=1, max_size=12))
    def gate(warm, rate, cuts):
        cfg = _Args()
        cfg.warmup_steps, cfg.update_rate = warm, rate
        g = LearnGate(cfg)
        total, pos = 0, 0
        for c in cuts:
            total += g.due_between(pos, pos + c)
            pos += c
        assert total == sum(1 for s in range(1, pos + 1) if g.due(s))

    @settings(max_examples=25, deadline=None)
    @given(seed=st.integers(0, 10 ** 6), cut=st.integers(1, 59))
    def ledger(seed, cut):
        g = torch.Generator().manual_seed(seed)
        T, B, N = 60, 4, 2
        rew = torch.randn(T, B, N, generator=g)
        term = torch.rand(T, B, generator=g) < 0.15
        whole, parts = ChunkLedger(B, N, 'cpu'), ChunkLedger(B, N, 'cpu')
        whole.absorb(rew, term)
        parts.absorb(rew[:cut], term[:cut])
        parts.absorb(rew[cut:], term[cut:])
        assert len(whole.totals) == len(parts.totals) == int(term.sum())
        np.testing.assert_allclose(whole.totals, parts.totals, rtol=0, atol=1e-9)   # (end step, env) order: independent of the cuts
        np.testing.assert_allclose(whole.carry.numpy(), parts.carry.numpy(), rtol=0, atol=1e-9)

    gate()
    ledger()
