#!/usr/bin/env python3
"""simple_reference with the two-head actor in the loop: the one-launch form (pw_policy_rollout, 100-step chunks, + the
two-head ring append) against the per-step form (FusedActor + env.step + ring append, hipGraph).  us per batched step."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multiagent_rl_amd import make_batched_env
from multiagent_rl_amd.policy import ActorNetwork, FusedActor
from multiagent_rl_amd.replay_buffer import ReplayBuffer
from multiagent_rl_amd.rollout import BatchedRollout

for B in (4096, 65536):
    torch.manual_seed(0)
    env = make_batched_env('simple_reference', B, auto_reset=True, max_episode_len=25, seed=3)
    actor = ActorNetwork(env.obs_dim, [5, 10]).cuda().eval()
    mem = ReplayBuffer(int(8e6), 2, env.obs_dim, act_heads=(5, 10))
    ro = BatchedRollout(env, FusedActor(actor, seed=7), mem)
    ro.collect_one_launch(200, chunk=100)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ro.collect_one_launch(1000, chunk=100)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('B=%d one launch per 100 steps (+ ring append): %.2f us per step, %.3e env-steps/s' % (B, dt / 1000 * 1e6, B * 1000 / dt))
    env2 = make_batched_env('simple_reference', B, auto_reset=True, max_episode_len=25, seed=3)
    ro2 = BatchedRollout(env2, FusedActor(actor, seed=7), ReplayBuffer(int(8e6), 2, env.obs_dim, act_heads=(5, 10)))
    ro2.capture(2); ro2.collect(50)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ro2.collect(500)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('B=%d per-step form, hipGraph: %.2f us per step, %.3e env-steps/s' % (B, dt / 500 * 1e6, B * 500 / dt))
