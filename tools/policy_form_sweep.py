#!/usr/bin/env python3
"""us per batched step of the one-launch policy rollout by N and kernel form (pw_dispatch.policy_form), B = 4096, 100-step
launches with the ring sink: python3 tools/policy_form_sweep.py [N ...]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multiagent_rl_amd.env import BatchedParticleEnv
from multiagent_rl_amd.policy import ActorNetwork, FusedActor
from multiagent_rl_amd.replay_buffer import ReplayBuffer
from multiagent_rl_amd.rollout import BatchedRollout

Ns = [int(x) for x in sys.argv[1:]] or [6, 12, 14, 16, 20, 24, 30]
B, T = 4096, 100
for N in Ns:
    row = []
    for form in (0, 3, 4):
        try:
            env = BatchedParticleEnv('simple_spread', B, num_agents=N, max_episode_len=25, auto_reset=True, seed=1)
            env.set_dispatch(policy_form=form)
            ro = BatchedRollout(env, FusedActor(ActorNetwork(env.obs_dim, 5).cuda().eval(), seed=2), ReplayBuffer(int(1e6), N, env.obs_dim))
            K = 3 if N >= 20 and form == 3 else 6
            ro.collect_one_launch(T, chunk=T)
            torch.cuda.synchronize()
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
            ro.collect_one_launch(K * T, chunk=T)
            ev[1].record()
            torch.cuda.synchronize()
            row.append('form %d %-28s %8.1f us' % (form, env.last_kernel(), ev[0].elapsed_time(ev[1]) * 1e3 / (K * T)))
            del ro, env
        except Exception as e:
            row.append('form %d: %s' % (form, str(e)[:60]))
    print('N = %2d | ' % N + ' | '.join(row), flush=True)
