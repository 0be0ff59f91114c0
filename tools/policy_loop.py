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
a = ap.parse_args()
torch.manual_seed(0)
env = make_batched_env('simple_spread', a.envs, n=a.agents, auto_reset=True)
pol = {'actor': lambda: GumbelPolicy(ActorNetwork(env.obs_dim, 5).cuda()),
       'fused': lambda: FusedActor(ActorNetwork(env.obs_dim, 5).cuda()),
       'uniform': UniformRandomPolicy}[a.policy]()
mem = None if a.no_replay else ReplayBuffer(1e6, a.agents, env.obs_dim)
ro = BatchedRollout(env, pol, mem)
if a.graph:
    ro.capture()
ro.collect(30)
torch.cuda.synchronize()
t0 = time.perf_counter()
ro.collect(a.steps)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print('B=%d N=%d policy=%s replay=%s graph=%s: %.1f us/step  %.3e env-steps/s  %s' % (
    a.envs, a.agents, a.policy, mem is not None, a.graph, dt / a.steps * 1e6, a.envs * a.steps / dt, ro.stats()))
