#!/usr/bin/env python3
"""B-sweep at simple_spread N = L = 6 ("a B-sweep {4096 ... 2^22} at N = 6 to show the latency -> bandwidth transition"),
measured the way the bench headline is (bench.measure_rollout: HIP-event bracket over K launches after W, clock ramp, output ring,
every output written): env-steps/s, us per batched step, the fraction of the 8 TB/s roofline by algorithmic bytes (678 B per
env-step) and the kernel the dispatcher chose.   python3 tools/sweep_B.py > profiles/r4_sweep_B.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from multiagent_rl_amd.env import BatchedParticleEnv

dev = torch.device('cuda', 0)
print('# simple_spread N = L = 6, local observation, episode 25 with auto-reset, every output written; 1 launch = T batched steps')
print('# %9s %6s %12s %14s %8s  %s' % ('B', 'T', 'us/step', 'env-steps/s', 'frac', 'kernel'))
for B in (1024, 4096, 8192, 16384, 65536, 262144, 1048576, 4194304):
    per_step = B * 6 * (2 * 16 * 4 + 4 + 4 + 1) + B * 5
    T = max(2, min(1000, int(8e9 // per_step)))            # one output slot <= 8 GB
    env = BatchedParticleEnv('simple_spread', B, num_agents=6, max_episode_len=25, auto_reset=True, seed=12345678)
    m = bench.measure_rollout(env, dev, T, K=6, W=2, ring_cap_bytes=40e9)
    rate = B * T / (m['launch_ms'] * 1e-3)
    print('  %9d %6d %12.3f %14.4g %8.3f  %s' % (B, T, m['launch_ms'] * 1e3 / T, rate, rate * env.bytes_per_env_step / 8e12, env.last_kernel()),
          flush=True)
    del env
    torch.cuda.empty_cache()
