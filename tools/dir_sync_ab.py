#!/usr/bin/env python3
"""A/B of the per-direction hand-off (pw_dispatch.policy_form 5: the four waves of an LSTM direction meet through an LDS counter inside
the timestep loop) against form 3 (one workgroup barrier per timestep): bit-identity of every output of a 50-step launch, then us per
batched step with the ring sink (policy_profile_run's method), interleaved repeats.  Needs the experiment applied first (it was measured and
removed: `git apply tools/experiments/r5_dir_sync.patch`, rebuild), then   python3 tools/dir_sync_ab.py   (profiles/r5_policy_dir_sync.txt)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multiagent_rl_amd.env import BatchedParticleEnv
from multiagent_rl_amd.policy import ActorNetwork, FusedActor
from multiagent_rl_amd.replay_buffer import ReplayBuffer
from multiagent_rl_amd.rollout import BatchedRollout

dev = torch.device('cuda', 0)


def outputs(N, form, T=50, B=4096):
    torch.manual_seed(1)
    env = BatchedParticleEnv('simple_spread', B, num_agents=N, max_episode_len=25, auto_reset=True, seed=11)
    env.set_dispatch(policy_form=form)
    env.reset()
    actor = FusedActor(ActorNetwork(env.obs_dim, 5).to(dev).eval(), seed=5)
    out = actor.rollout(env, T)
    torch.cuda.synchronize()
    return out, env.last_kernel()


def timing(N, form, T=100, B=4096, chunks=10):
    torch.manual_seed(1)
    env = BatchedParticleEnv('simple_spread', B, num_agents=N, max_episode_len=25, auto_reset=True, seed=12345678)
    env.set_dispatch(policy_form=form)
    actor = FusedActor(ActorNetwork(env.obs_dim, 5).to(dev).eval(), seed=12345678)
    ro = BatchedRollout(env, actor, ReplayBuffer(int(1e6), env.n, env.obs_dim))
    t_r = time.perf_counter()
    while time.perf_counter() - t_r < 0.06:
        ro.collect_one_launch(T, chunk=T)
        torch.cuda.synchronize()
    ro.collect_one_launch(3 * T, chunk=T)
    torch.cuda.synchronize()
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    ev[0].record()
    ro.collect_one_launch(chunks * T, chunk=T)
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) * 1e3 / (chunks * T)


for N in (3, 6, 9, 12):
    a, ka = outputs(N, 3)
    b, kb = outputs(N, 5)
    same = all(torch.equal(a[k], b[k]) for k in ('obs', 'rew', 'rew_shared', 'terminal', 'act', 'final_obs'))
    print('N = %2d: form 5 outputs %s form 3 (%s)' % (N, 'bit-identical to' if same else 'DIFFER from', ka), flush=True)
print('# us per batched step at B = 4096, 100-step chunks with the ring sink; three interleaved repeats: form 3 | form 5 (per-direction hand-off)')
for N in (3, 6, 9, 12):
    t3, t5 = [], []
    for rep in range(3):
        t3.append(timing(N, 3))
        t5.append(timing(N, 5))
    print('N = %2d:  %s  |  %s   median %.2f -> %.2f (%+.1f %%)' % (N, ' '.join('%.2f' % x for x in t3), ' '.join('%.2f' % x for x in t5),
                                                                  sorted(t3)[1], sorted(t5)[1], 100.0 * (sorted(t5)[1] / sorted(t3)[1] - 1.0)), flush=True)
