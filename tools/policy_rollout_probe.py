#!/usr/bin/env python3
"""Measurement helper: one-launch policy rollout (pw_policy_rollout) throughput.
python tools/policy_rollout_probe.py [--envs 4096] [--chunk 100] [--steps 1000]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from multiagent_rl_amd import make_batched_env  # noqa: E402
from multiagent_rl_amd.policy import ActorNetwork, FusedActor  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--envs', type=int, default=4096)
ap.add_argument('--agents', type=int, default=6)
ap.add_argument('--chunk', type=int, default=100)
ap.add_argument('--steps', type=int, default=1000)
ap.add_argument('--form', type=int, default=0, help='pw_dispatch.policy_form: 0 auto, 1/2/3 the kernel forms')
ap.add_argument('--replay', action='store_true', help='BatchedRollout.collect_one_launch: + ring append and bookkeeping')
a = ap.parse_args()
torch.manual_seed(0)
env = make_batched_env('simple_spread', a.envs, n=a.agents, auto_reset=True)
if a.form:
    env.set_dispatch(policy_form=a.form)
pol = FusedActor(ActorNetwork(env.obs_dim, 5).cuda())
if a.replay:
    from multiagent_rl_amd.replay_buffer import ReplayBuffer
    from multiagent_rl_amd.rollout import BatchedRollout
    ro = BatchedRollout(env, pol, ReplayBuffer(1e6, a.agents, env.obs_dim))
    ro.collect_one_launch(a.chunk, a.chunk)
    torch.cuda.synchronize()
    n = max(1, a.steps // a.chunk)
    t0 = time.perf_counter()
    ro.collect_one_launch(n * a.chunk, a.chunk)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('B=%d N=%d chunk=%d with ring append + bookkeeping: %.2f us/step  %.3e env-steps/s  %s' % (
        a.envs, a.agents, a.chunk, dt / (n * a.chunk) * 1e6, a.envs * n * a.chunk / dt, ro.stats()))
    sys.exit(0)
env.reset()
out = pol.rollout(env, a.chunk)
torch.cuda.synchronize()
n = max(1, a.steps // a.chunk)
t0 = time.perf_counter()
for _ in range(n):
    pol.rollout(env, a.chunk, out)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print('form %d B=%d N=%d chunk=%d: %.2f us/step  %.3e env-steps/s  mean shared reward %.3f' % (
    a.form, a.envs, a.agents, a.chunk, dt / (n * a.chunk) * 1e6, a.envs * n * a.chunk / dt, out['rew_shared'].mean().item()))
