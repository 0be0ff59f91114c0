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


if __name__ == '__main__':
    make_run_trace()
    make_run_test_trace()
    make_multidiscrete_trace()
    make_replay()
    make_actor()
