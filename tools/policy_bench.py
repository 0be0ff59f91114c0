#!/usr/bin/env python3
"""Policy-in-the-loop rollout timing (pw_policy_rollout, one launch per chunk): us per batched step.
Usage: python tools/policy_bench.py [--envs 4096] [--agents 6] [--chunk 100] [--steps 1000] [--no-sink]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from multiagent_rl_amd.env import BatchedParticleEnv  # noqa: E402
from multiagent_rl_amd.policy import ActorNetwork, FusedActor  # noqa: E402
from multiagent_rl_amd.replay_buffer import ReplayBuffer  # noqa: E402
from multiagent_rl_amd.rollout import BatchedRollout  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--envs', type=int, default=4096)
ap.add_argument('--agents', default='6')
ap.add_argument('--chunk', type=int, default=100)
ap.add_argument('--steps', type=int, default=1000)
ap.add_argument('--no-sink', action='store_true')
ap.add_argument('--scenario', default='simple_spread', help="simple_spread, or simple_tag (4 adversaries + 2 good; --agents ignored)")
a = ap.parse_args()
for N in [int(x) for x in a.agents.split(',')]:
    torch.manual_seed(0)
    if a.scenario == 'simple_tag':
        env = BatchedParticleEnv('simple_tag', a.envs, num_adversaries=4, num_good=2, max_episode_len=25, auto_reset=True, seed=1)
        N = env.n
    else:
        env = BatchedParticleEnv('simple_spread', a.envs, num_agents=N, max_episode_len=25, auto_reset=True, seed=1)
    actor = FusedActor(ActorNetwork(env.obs_dim, 5).cuda().eval(), seed=2)
    if a.no_sink:
        env.reset()
        out = actor.rollout(env, a.chunk)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps // a.chunk):
            actor.rollout(env, a.chunk, out)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    else:
        ro = BatchedRollout(env, actor, ReplayBuffer(int(1e6), N, env.obs_dim))
        ro.collect_one_launch(a.chunk, chunk=a.chunk)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ro.collect_one_launch(a.steps, chunk=a.chunk)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    n = a.steps // a.chunk * a.chunk
    print(a.scenario + ' B=%d N=%d D=%d %s: %.2f us/step  %.3e env-steps/s' % (a.envs, N, env.obs_dim, 'no sink' if a.no_sink else 'ring sink',
                                                              dt / n * 1e6, a.envs * n / dt), flush=True)
