import os, sys
sys.path.insert(0, '/root/repo')
import torch, bench
from multiagent_rl_amd.env import BatchedParticleEnv
dev = torch.device('cuda', 0)
for B in (24576, 32768, 49152, 65536, 98304, 131072):
    for name, disp in (('auto', {}), ('duo', dict(duo=1)), ('stream', dict(duo=0))):
        per_step = B * 6 * (2 * 16 * 4 + 4 + 4 + 1) + B * 5
        T = max(2, min(1000, int(8e9 // per_step)))
        env = BatchedParticleEnv('simple_spread', B, num_agents=6, max_episode_len=25, auto_reset=True, seed=12345678, dispatch=disp or None)
        res = []
        for rep in range(2):
            m = bench.measure_rollout(env, dev, T, K=6, W=2, ring_cap_bytes=40e9)
            res.append(B * T / (m['launch_ms'] * 1e-3))
        print('B %7d %-6s T %4d  %.4g %.4g env-steps/s  %s' % (B, name, T, res[0], res[1], env.last_kernel()), flush=True)
        del env; torch.cuda.empty_cache()
