#!/usr/bin/env python3
"""Measurement helper: the one-launch actor (pw_actor_fused) per call, by N: python tools/fused_actor_time.py [--envs 4096]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from multiagent_rl_amd.policy import ActorNetwork, FusedActor  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--envs', type=int, default=4096)
a = ap.parse_args()
torch.manual_seed(0)
for N in (2, 3, 6, 12, 16, 24):
    D = 4 + 2 * N if N <= 30 else 64
    pol = FusedActor(ActorNetwork(D, 5).cuda().eval())
    obs = torch.randn(a.envs, N, D, device='cuda')
    for _ in range(5):
        pol(obs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        pol(obs)
    torch.cuda.synchronize()
    print('pw_actor_fused B=%d N=%d D=%d: %.2f us per call' % (a.envs, N, D, (time.perf_counter() - t0) / 200 * 1e6))
