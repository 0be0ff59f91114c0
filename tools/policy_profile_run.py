#!/usr/bin/env python3
"""One policy-in-the-loop workload for the profiler (tools/collect_profiles_r4.sh): W warm-up + K timed chunks of
BatchedRollout.collect_one_launch (pw_policy_rollout: actor + Gumbel sampling + env step + ring append per chunk), HIP-event
bracket on the launch stream, ONE JSON line: env-steps/s, us per batched step, launch_ms, algorithmic flops (the actor's dense
products, bench.py actor_flops_per_env_step) and the f32-MFMA roofline fraction.
  python3 tools/policy_profile_run.py --workload c2|tag|ref|n12|n24 [--chunk 100] [--steps 10] [--warmup 3]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

WORKLOADS = {
    'c2': ('simple_spread', 4096, dict(num_agents=6), 5),
    'n3': ('simple_spread', 4096, dict(num_agents=3), 5),
    'n12': ('simple_spread', 4096, dict(num_agents=12), 5),
    'n24': ('simple_spread', 4096, dict(num_agents=24), 5),
    'n16': ('simple_spread', 4096, dict(num_agents=16), 5),
    'n30': ('simple_spread', 4096, dict(num_agents=30), 5),
    'n48': ('simple_spread', 4096, dict(num_agents=48), 5),
    'tag': ('simple_tag', 8192, dict(num_adversaries=4, num_good=2), 5),
    'ref': ('simple_reference', 4096, {}, [5, 10]),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='c2', choices=sorted(WORKLOADS))
    ap.add_argument('--chunk', type=int, default=100)
    ap.add_argument('--steps', type=int, default=10, help='timed chunks (= launches of the rollout kernel)')
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--form', type=int, default=0, help='pw_dispatch.policy_form (0 = automatic)')
    a = ap.parse_args()
    import torch
    from bench import F32_MFMA_PEAK_TFLOPS, actor_flops_per_env_step
    from multiagent_rl_amd.env import BatchedParticleEnv
    from multiagent_rl_amd.policy import ActorNetwork, FusedActor
    from multiagent_rl_amd.replay_buffer import ReplayBuffer
    from multiagent_rl_amd.rollout import BatchedRollout
    scen, B, kw, heads = WORKLOADS[a.workload]
    torch.manual_seed(12345678)
    dev = torch.device('cuda', 0)
    env = BatchedParticleEnv(scen, B, max_episode_len=25, auto_reset=True, seed=12345678, **kw)
    if a.form:
        env.set_dispatch(policy_form=a.form)
    two = isinstance(heads, list)
    actor = FusedActor(ActorNetwork(env.obs_dim, heads).to(dev).eval(), seed=12345678)
    mem = ReplayBuffer(int(8e6 if two else 5e5 if env.n > 32 else 1e6), env.n, env.obs_dim, **(dict(act_heads=(5, 10)) if two else {}))
    ro = BatchedRollout(env, actor, mem)
    T = a.chunk
    t_r = time.perf_counter()
    while time.perf_counter() - t_r < 0.06:           # clock ramp, as the bench does
        ro.collect_one_launch(T, chunk=T)
        torch.cuda.synchronize()
    for _ in range(a.warmup):
        ro.collect_one_launch(T, chunk=T)
    torch.cuda.synchronize()
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    ev[0].record()
    ro.collect_one_launch(a.steps * T, chunk=T)
    ev[1].record()
    torch.cuda.synchronize()
    sec = ev[0].elapsed_time(ev[1]) * 1e-3
    n_out = sum(heads) if two else heads
    fl = actor_flops_per_env_step(env.n, env.obs_dim, n_out)
    launch_s = sec / a.steps
    ach = fl * B * T / launch_s / 1e12
    print(json.dumps(dict(workload=a.workload, scenario=scen, B=B, N=env.n, D=env.obs_dim, n_out=n_out, chunk=T, steps=a.steps,
                          warmup=a.warmup, value=B * T * a.steps / sec, unit='env-steps/s', us_per_step=sec / (a.steps * T) * 1e6,
                          launch_ms=launch_s * 1e3, kernel=env.last_kernel(), launches_per_chunk=1,
                          flops_per_env_step=fl,
                          roofline=dict(bound='mfma_f32', achieved=ach, peak=F32_MFMA_PEAK_TFLOPS, unit='TFLOP/s',
                                        frac=ach / F32_MFMA_PEAK_TFLOPS))))


if __name__ == '__main__':
    main()
