import sys, os
sys.path.insert(0, '/root/repo')
import torch
from multiagent_rl_amd.env import BatchedParticleEnv
from multiagent_rl_amd.policy import ActorNetwork, FusedActor
from multiagent_rl_amd.replay_buffer import ReplayBuffer
B, T = 4096, 100
def run(mode):
    torch.manual_seed(0)
    env = BatchedParticleEnv('simple_reference', B, max_episode_len=25, auto_reset=True, seed=3)
    env.reset()
    actor = FusedActor(ActorNetwork(env.obs_dim, [5, 10]).cuda().eval(), seed=7)
    mem = ReplayBuffer(int(8e6), 2, env.obs_dim, act_heads=(5, 10))
    mem._allocate(2, env.obs_dim) if mem._store is None else None
    ret = torch.zeros(B, device='cuda'); fs = torch.zeros((), dtype=torch.float64, device='cuda'); fc = torch.zeros((), dtype=torch.int64, device='cuda')
    out = None
    def one():
        nonlocal out
        if mode == 'outputs': out = actor.rollout(env, T, out)
        elif mode == 'ring': actor.rollout(env, T, False, memory=mem)
        elif mode == 'ring+stats': actor.rollout(env, T, False, memory=mem, stats=(ret, fs, fc))
        elif mode == 'stats': out = actor.rollout(env, T, out, stats=(ret, fs, fc))
    for _ in range(3): one()
    torch.cuda.synchronize()
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    ev[0].record()
    for _ in range(10): one()
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) * 1e3 / (10 * T)
for m in ('outputs', 'stats', 'ring', 'ring+stats', 'outputs', 'ring+stats'):
    print('%-12s %.2f us/step' % (m, run(m)), flush=True)
