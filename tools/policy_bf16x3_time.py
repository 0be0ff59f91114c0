#!/usr/bin/env python3
"""Measurement helper: the one-launch policy rollout in the exact f32 form and in the opt-in bf16x3 input-projection mode."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from multiagent_rl_amd import make_batched_env  # noqa: E402
from multiagent_rl_amd.policy import ActorNetwork, FusedActor  # noqa: E402

for N in (3, 6, 12):
    for mode in ('f32', 'bf16x3'):
        torch.manual_seed(0)
        env = make_batched_env('simple_spread', 4096, n=N, auto_reset=True)
        env.set_actor_precision(mode)
        pol = FusedActor(ActorNetwork(env.obs_dim, 5).cuda())
        env.reset()
        out = pol.rollout(env, 100)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            pol.rollout(env, 100, out)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print('N=%d %s: %.2f us/step  %.3e env-steps/s  mean shared reward %.3f' % (
            N, mode, dt / 1000 * 1e6, 4096 * 1000 / dt, out['rew_shared'].mean().item()))
