"""Host-side cost of one SampledTransitionGather exchange (1-rank RCCL rehearsal)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
import torch, torch.distributed as dist
from multiagent_rl_amd.env import BatchedParticleEnv
from multiagent_rl_amd.dist import SampledTransitionGather
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
env = BatchedParticleEnv('simple_spread', 4096, num_agents=6, auto_reset=True)
env.reset()
acts = torch.randint(0, 5, (100, 4096, 6), device='cuda', dtype=torch.int32)
out = env.rollout(acts)
g = SampledTransitionGather(env, 128, 0, 1, 'cuda:0', every=1)
for _ in range(5): g(out, acts)
torch.cuda.synchronize()
import cProfile, pstats
t0 = time.perf_counter()
for _ in range(50): g(out, acts)
host = (time.perf_counter() - t0) / 50
torch.cuda.synchronize()
print('host time per exchange: %.1f us' % (host * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(50): g(out, acts)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(12)
dist.destroy_process_group()
