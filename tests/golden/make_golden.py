#!/usr/bin/env python3
"""Generates the golden fixtures in this directory by importing the REFERENCE (read-only at
/root/reference) in the build container.  The reference itself never travels; only these
inputs/outputs do.  Run from the repo root:  python tests/golden/make_golden.py

What the reference can pin for the hot path -- its physics is an absent
third-party package, so these fixtures pin the BOUNDARY, not the arithmetic:
  run_trace.json      experiments.run.run driven unmodified (reference loop, reference
                      ReplayBuffer) with the CPU oracle env + a recording stub Trainer: the exact
                      call sequence, container types/shapes/dtypes, per-episode rewards, history keys.
  run_test_trace.json experiments.run.run_test the same way (load_models first, test_history_*.pkl with the memory).
  run_multidiscrete_trace.json  experiments.run.run on simple_reference: the MultiDiscrete branch (run.py:39-41).
  replay_buffer.json  rls.replay_buffer.ReplayBuffer: ring semantics, make_index under random.seed,
                      encode shapes (NumPy-1 semantics; the literal call raises under NumPy >= 2).
  actor_forward.npz   rls.model.ac_network_multi_gumbel.ActorNetwork: state_dict, input, logits, and the
                      hard Gumbel one-hot of ddpg_gumbel_fix.Trainer.gumbel_softmax under torch.manual_seed.
  reference_scenarios.json  experiments.scenarios (R2 make_env, R3 local_obs_*) -- the REFERENCE's own factory and
                      observation functions, imported behind a two-module shell for the absent third-party `multiagent`
                      package (its `scenarios.load(..).Scenario` / `environment.MultiAgentEnv` resolve to the CPU
                      oracle's world classes; nothing of experiments/scenarios.py is restated): the flags make_env
                      leaves on the env (collaborative / shared_reward, force_discrete_action, the n= path), the spaces,
                      and the rows local_obs_simple_spread / _reference / _speaker_listener return on fixed states.
"""
import json
import os
import pickle
import random
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference')

from oracle import particle_oracle as po  # noqa: E402
from tests.trace_util import RecordingEnv, StubTrainer, fingerprint  # noqa: E402


def make_run_trace():
    from rls import arglist
    from experiments.run import run
    arglist.num_episodes, arglist.warmup_steps, arglist.update_rate, arglist.save_rate = 3, 10, 20, 2
    arglist.max_episode_len = 25
    np.random.seed(12345678)
    env = RecordingEnv(po.make_oracle_env('simple_spread'))
    np.random.seed(12345678)  # main.py:47
    trace = env.trace
    StubTrainer.trace = trace
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as d:
        os.chdir(d)
        os.makedirs('Models')
        try:
            run(env, None, None, StubTrainer, 'simple_spread', 'Discrete', cnt=0)
            hist = pickle.load(open('Models/history_simple_spread_0.pkl', 'rb'))
        finally:
            os.chdir(cwd)
    mem = StubTrainer.last.memory
    out = dict(arglist=dict(num_episodes=3, warmup_steps=10, update_rate=20, save_rate=2, max_episode_len=25),
               trace=trace, history_keys=sorted(hist.keys()),
               reward_episodes=[float(x) for x in hist['reward_episodes']],
               reward_episodes_by_agents=[[float(x) for x in a] for a in hist['reward_episodes_by_agents']],
               memory_len=len(mem), memory_class=type(mem).__module__ + '.' + type(mem).__name__,
               first_transition=fingerprint(mem._storage[0]), last_transition=fingerprint(mem._storage[-1]))
    json.dump(out, open(os.path.join(HERE, 'run_trace.json'), 'w'), indent=0)
    print('run_trace.json: %d events' % len(trace))


def make_run_test_trace():
    """experiments.run.run_test (evaluation loop) driven the same way."""
    from rls import arglist
    from experiments.run import run_test
    arglist.num_episodes, arglist.warmup_steps, arglist.update_rate, arglist.save_rate = 2, 10, 20, 2
    arglist.max_episode_len, arglist.appx = 25, 'pfx/'
    np.random.seed(12345679)
    env = RecordingEnv(po.make_oracle_env('simple_spread', n=4))
    np.random.seed(12345679)
    StubTrainer.trace = env.trace
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as d:
        os.chdir(d)
        os.makedirs('Models')
        try:
            run_test(env, None, None, StubTrainer, 'simple_spread', 'Discrete', cnt=1)
            files = sorted(os.listdir('Models'))
            hist = pickle.load(open('Models/test_history_simple_spread_1.pkl', 'rb'))
        finally:
            os.chdir(cwd)
    out = dict(arglist=dict(num_episodes=2, warmup_steps=10, update_rate=20, save_rate=2, max_episode_len=25, appx='pfx/'),
               trace=env.trace, files=files, history_keys=sorted(hist.keys()),
               reward_episodes=[float(x) for x in hist['reward_episodes']], memory_len=len(hist['memory']))
    json.dump(out, open(os.path.join(HERE, 'run_test_trace.json'), 'w'), indent=0)
    print('run_test_trace.json: %d events, files %s' % (len(env.trace), files))


def make_multidiscrete_trace():
    """experiments.run.run on simple_reference (MultiDiscrete branch, run.py:39-41)."""
    from rls import arglist
    from experiments.run import run
    arglist.num_episodes, arglist.warmup_steps, arglist.update_rate, arglist.save_rate = 2, 10, 20, 2
    arglist.max_episode_len = 25
    np.random.seed(12345680)
    env = RecordingEnv(po.make_oracle_env('simple_reference'))
    env.env.force_discrete_action = True
    np.random.seed(12345680)
    StubTrainer.trace = env.trace
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as d:
        os.chdir(d)
        os.makedirs('Models')
        try:
            run(env, None, None, StubTrainer, 'simple_reference', 'MultiDiscrete', cnt=2)
            hist = pickle.load(open('Models/history_simple_reference_2.pkl', 'rb'))
        finally:
            os.chdir(cwd)
    out = dict(arglist=dict(num_episodes=2, warmup_steps=10, update_rate=20, save_rate=2, max_episode_len=25),
               trace=env.trace, reward_episodes=[float(x) for x in hist['reward_episodes']])
    json.dump(out, open(os.path.join(HERE, 'run_multidiscrete_trace.json'), 'w'), indent=0)
    print('run_multidiscrete_trace.json: %d events' % len(env.trace))


def make_replay():
    from rls.replay_buffer import ReplayBuffer
    rb = ReplayBuffer(5)
    N, D = 3, 10
    rng = np.random.RandomState(0)
    added = []
    for i in range(8):
        obs = [rng.randn(D) for _ in range(N)]
        act = [np.eye(5)[rng.randint(5)] for _ in range(N)]
        nxt = [rng.randn(D) for _ in range(N)]
        rew, done = float(rng.randn()), float(i % 4 == 3)
        rb.add(obs, act, rew, nxt, done)
        added.append(dict(obs=np.stack(obs).tolist(), act=np.stack(act).tolist(), rew=rew,
                          next_obs=np.stack(nxt).tolist(), done=done))
    random.seed(0)
    idx = rb.make_index(4)
    raises = False
    try:
        rb.sample_index(idx)
    except ValueError:
        raises = True  # np.array(list, copy=False) under NumPy >= 2
    enc = [np.asarray([np.asarray(rb._storage[i][k]) for i in idx]) for k in range(5)]  # NumPy-1 semantics
    random.seed(0)
    big = ReplayBuffer(1e6)
    for _ in range(8):
        big.add(0, 0, 0, 0, 0)
    out = dict(size=5, transitions=added, len=len(rb), next_idx=rb._next_idx,
               storage_rewards=[float(t[2]) for t in rb._storage], make_index_seed0=idx,
               make_index_seed0_len8_batch4=big.make_index(4),
               encode_raises_on_numpy2=raises, encode_shapes=[list(e.shape) for e in enc],
               encode_dtypes=[str(e.dtype) for e in enc], encode=[e.tolist() for e in enc])
    json.dump(out, open(os.path.join(HERE, 'replay_buffer.json'), 'w'), indent=0)
    print('replay_buffer.json: len %d next %d idx %s raises %s' % (len(rb), rb._next_idx, idx, raises))


def make_actor():
    from rls.model.ac_network_multi_gumbel import ActorNetwork
    from rls.agent.multiagent.ddpg_gumbel_fix import Trainer
    torch.manual_seed(12345678)
    D, N, B = 16, 6, 7
    actor = ActorNetwork(input_dim=D, out_dim=5)
    obs = torch.randn(B, N, D)
    with torch.no_grad():
        logits = actor(obs)
    torch.manual_seed(4321)
    onehot = Trainer.gumbel_softmax(None, logits, hard=True)  # the method does not touch self
    arrays = {'sd/' + k: v.numpy() for k, v in actor.state_dict().items()}
    arrays.update(obs=obs.numpy(), logits=logits.numpy(), gumbel_seed=np.array(4321), onehot=onehot.numpy())
    np.savez_compressed(os.path.join(HERE, 'actor_forward.npz'), **arrays)
    print('actor_forward.npz: %s' % sorted(actor.state_dict().keys()))


def _import_reference_scenarios():
    """experiments/scenarios.py:2-3 imports `multiagent.scenarios` and `multiagent.environment.MultiAgentEnv`; the package
    is absent.  A shell of two modules supplies exactly those two names from the CPU oracle, so
    that the reference's OWN make_env / local_obs_* code runs (on the oracle's world objects)."""
    import types

    def with_upstream_observation(cls):
        # upstream scenarios call their (full) observation `observation`; make_env patches the local one over it
        return type(cls.__name__, (cls,), dict(observation=cls.observation_full))

    table = {'simple_spread.py': with_upstream_observation(po.SimpleSpread),
             'simple_reference.py': with_upstream_observation(po.SimpleReference),
             'simple_speaker_listener.py': with_upstream_observation(po.SimpleSpeakerListener)}
    pkg = types.ModuleType('multiagent')
    scen = types.ModuleType('multiagent.scenarios')
    scen.load = lambda name: types.SimpleNamespace(Scenario=table[name])
    envm = types.ModuleType('multiagent.environment')
    envm.MultiAgentEnv = po.OracleMultiAgentEnv
    pkg.scenarios, pkg.environment = scen, envm
    sys.modules.update({'multiagent': pkg, 'multiagent.scenarios': scen, 'multiagent.environment': envm})
    import experiments.scenarios as ref
    return ref


def make_reference_scenarios():
    """R2 / R3 from the reference's own code (experiments/scenarios.py:6-64,124-192)."""
    ref = _import_reference_scenarios()
    rng = np.random.RandomState(20241004)

    def grid(*shape):  # multiples of 2^-12 in [-1, 1]: every difference below is exact in float32 AND float64
        return rng.randint(-4096, 4097, shape) / 4096.0

    out = {}
    for key, name, n in (('simple_spread', 'simple_spread', None), ('simple_spread_n6', 'simple_spread', 6),
                         ('simple_reference', 'simple_reference', None),
                         ('simple_speaker_listener', 'simple_speaker_listener', None)):
        np.random.seed(12345678)
        env = ref.make_env(name, n=n, benchmark=False, discrete_action=True, local_observation=True)  # main.py:39
        world = env.world
        rec = dict(scenario=name, n_arg=n, n=env.n, num_landmarks=len(world.landmarks),
                   world_collaborative=bool(world.collaborative), shared_reward=bool(env.shared_reward),
                   force_discrete_action=bool(env.force_discrete_action),
                   discrete_action_space=bool(env.discrete_action_space),
                   discrete_action_input=bool(env.discrete_action_input),
                   observation_is_reference_function=env.observation_callback.__func__.__name__,
                   observation_space=[list(s.shape) for s in env.observation_space],
                   action_space=[repr(s) for s in env.action_space],
                   action_has_high=[hasattr(s, 'high') for s in env.action_space], states=[])
        fn = {'simple_spread': ref.local_obs_simple_spread, 'simple_reference': ref.local_obs_simple_reference,
              'simple_speaker_listener': ref.local_obs_simple_speaker_listener}[name]
        N, L = len(world.agents), len(world.landmarks)
        for case in range(6):
            pos, vel, lm = grid(N, 2), grid(N, 2) * 1.5, grid(L, 2)
            po.set_world_state(world, pos, vel, lm)
            st = dict(pos=pos.tolist(), vel=vel.tolist(), landmarks=lm.tolist())
            if name != 'simple_spread':
                comm = grid(N, world.dim_c) * 0.5 + 0.5
                goal = rng.randint(0, L, N)
                for i, a in enumerate(world.agents):
                    a.state.c = comm[i].copy()
                    if a.goal_b is not None:  # the speaker_listener listener has none (zeros in its row)
                        a.goal_b = world.landmarks[int(goal[i])]
                st.update(comm=comm.tolist(), goal=[int(g) for g in goal],
                          landmark_colors=[[float(c) for c in l.color] for l in world.landmarks])
            # the reference's functions, called the way make_env bound them (scenario.observation) AND directly
            rows = [np.asarray(env.observation_callback(a, world)) for a in world.agents]
            direct = [np.asarray(fn(env.observation_callback.__self__, a, world)) for a in world.agents]
            assert all(np.array_equal(r, d) for r, d in zip(rows, direct))
            st['obs'] = [r.tolist() for r in rows]
            rec['states'].append(st)
        # one env.step through the reference-built env: per-agent rewards (collaborative False), done flags, and that a soft
        # action is arg-maxed (force_discrete_action True) -- positions after the step depend on it
        np.random.seed(7)
        obs0 = env.reset()
        width = [s.n if not hasattr(s, 'high') else int(np.sum(s.high - s.low + 1)) for s in env.action_space]
        soft = [rng.uniform(0.05, 1.0, w) for w in width]
        st0 = po.get_world_state(world)
        obs1, rew, done, info = env.step([a.copy() for a in soft])
        st1 = po.get_world_state(world)
        rec['step'] = dict(numpy_seed=7, reset_obs=[np.asarray(o).tolist() for o in obs0], soft_actions=[a.tolist() for a in soft],
                           pos0=st0[0].tolist(), vel0=st0[1].tolist(), landmarks=st0[2].tolist(),
                           goal=[world.landmarks.index(a.goal_b) if getattr(a, 'goal_b', None) is not None else -1
                                 for a in world.agents],
                           pos1=st1[0].tolist(), vel1=st1[1].tolist(), obs=[np.asarray(o).tolist() for o in obs1],
                           rew=[float(r) for r in rew], done=[bool(d) for d in done], info_keys=sorted(info.keys()))
        out[key] = rec
    json.dump(out, open(os.path.join(HERE, 'reference_scenarios.json'), 'w'), indent=0)
    print('reference_scenarios.json: %s' % {k: (v['n'], v['observation_space'][0]) for k, v in out.items()})


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'scenarios':
        make_reference_scenarios()
        sys.exit(0)
    make_run_trace()
    make_run_test_trace()
    make_multidiscrete_trace()
    make_replay()
    make_actor()
    make_reference_scenarios()
