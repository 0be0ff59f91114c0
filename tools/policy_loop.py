#!/usr/bin/env python3
"""Measurement helper: policy-in-the-loop rollout throughput (actor forward + Gumbel argmax + pw_step +
replay add per step, nothing read back).  python tools/policy_loop.py [--envs 4096] [--steps 300] [--graph]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from multiagent_rl_amd import make_batched_env  # noqa: E402
from multiagent_rl_amd.policy import ActorNetwork, FusedActor, GumbelPolicy, UniformRandomPolicy  # noqa: E402
from multiagent_rl_amd.replay_buffer import ReplayBuffer  # noqa: E402
from multiagent_rl_amd.rollout import BatchedRollout  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--envs', type=int, default=4096)
ap.add_argument('--agents', type=int, default=6)
ap.add_argument('--steps', type=int, default=300)
ap.add_argument('--policy', default='actor', choices=['actor', 'fused', 'uniform'])
ap.add_argument('--no-replay', action='store_true')
ap.add_argument('--graph', action='store_true')
ap.add_argument('--scenario', default='simple_spread', choices=['simple_spread', 'simple_reference'])
a = ap.parse_args()
torch.manual_seed(0)
ref = a.scenario == 'simple_reference'
env = make_batched_env(a.scenario, a.envs, n=None if ref else a.agents, auto_reset=True)
out_dim = [5, 10] if ref else 5          # main.py:52-54: MultiDiscrete -> two heads
pol = {'actor': lambda: GumbelPolicy(ActorNetwork(env.obs_dim, out_dim).cuda()),
       'fused': lambda: FusedActor(ActorNetwork(env.obs_dim, out_dim).cuda()),
       'uniform': UniformRandomPolicy}[a.policy]()
mem = None if (a.no_replay or ref) else ReplayBuffer(1e6, a.agents, env.obs_dim)  # the ring stores single-head actions
ro = BatchedRollout(env, pol, mem)
if a.graph:
    ro.capture()
ro.collect(30)
torch.cuda.synchronize()
t0 = time.perf_counter()
ro.collect(a.steps)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print('%s B=%d N=%d policy=%s replay=%s graph=%s: %.1f us/step  %.3e env-steps/s  %s' % (
    a.scenario, a.envs, env.n, a.policy, mem is not None, a.graph, dt / a.steps * 1e6, a.envs * a.steps / dt, ro.stats()))
