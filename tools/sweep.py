#!/usr/bin/env python3
"""Measurement helper (not part of the product): time pw_rollout over B / N / T sweeps.
Usage: python tools/sweep.py [--envs 4096,65536] [--agents 6] [--chunk 25] [--steps 2000] [--ring 200]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from multiagent_rl_amd.env import BatchedParticleEnv  # noqa: E402


def measure(B, N, T, steps, ring, scenario='simple_spread', reps=3, min_outputs=False, **extra):
    kw = dict(num_agents=N) if scenario == 'simple_spread' else dict(num_adversaries=4, num_good=2)
    kw.update(extra)
    env = BatchedParticleEnv(scenario, B, max_episode_len=25, auto_reset=True, **kw)
    N = env.n
    ring = max(T, min(ring, steps) // T * T)
    acts = torch.randint(0, 5, (ring, B, N), device='cuda', dtype=torch.int32)
    outs = env.alloc_outputs(ring, coll=False)
    if min_outputs:
        keep = min_outputs.split(',') if isinstance(min_outputs, str) else ('obs', 'rew', 'done')
        outs = {k: v for k, v in outs.items() if k in keep}
    env.reset()
    nl = steps // T

    plans = [env.plan_rollout(acts[s:s + T], {k: v[s:s + T] for k, v in outs.items()}) for s in range(0, ring, T)]

    def run():
        for i in range(nl):
            plans[i % len(plans)]()
    run()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    us_step = best / (nl * T) * 1e6
    rate = B * nl * T / best
    gbs = rate * env.bytes_per_env_step / 1e9
    print('B=%8d N=%2d T=%4d steps=%6d  %8.3f us/step  %.3e env-steps/s  %7.1f GB/s (algorithmic %d B)  %.1f%% of 8 TB/s'
          % (B, N, T, nl * T, us_step, rate, gbs, env.bytes_per_env_step, gbs / 80.0), flush=True)
    return rate


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', default='4096')
    ap.add_argument('--agents', default='6')
    ap.add_argument('--chunk', default='25')
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--ring', type=int, default=200)
    ap.add_argument('--scenario', default='simple_spread')
    ap.add_argument('--min-outputs', default='')
    ap.add_argument('--contact-margin', type=float, default=None)
    a = ap.parse_args()
    for B in [int(x) for x in a.envs.split(',')]:
        for N in [int(x) for x in a.agents.split(',')]:
            for T in [int(x) for x in a.chunk.split(',')]:
                extra = {} if a.contact_margin is None else dict(contact_margin=a.contact_margin)
                measure(B, N, T, a.steps, a.ring, a.scenario, min_outputs=a.min_outputs, **extra)
