import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multiagent_rl_amd import make_batched_env
from multiagent_rl_amd.dist import FullTransitionGather
from multiagent_rl_amd.replay_buffer import ReplayBuffer
A, G, L, B, T, ep = 1, 1, 1, 40, 9, 1
env = make_batched_env('simple_tag', B, num_adversaries=A, num_good=G, num_landmarks=L, auto_reset=True, max_episode_len=ep, seed=9)
N, D = A + G, env.obs_dim
dev = torch.device('cuda', 0)
cap = T * B
g = FullTransitionGather(env, T, 0, 1, dev, capacity=cap, overlap_ingest=False)
want = ReplayBuffer(cap, N, D)
obs0 = env.reset()
torch.manual_seed(0)
acts = torch.randint(0, 5, (T, B, N), device='cuda', dtype=torch.int32)
out = g.outputs()
out['act'].copy_(acts)
env.rollout(acts, out={n_: v for n_, v in out.items() if n_ != 'act'})
dense = {n_: v.clone() for n_, v in out.items()}
g(obs0)
g.finish()
want.add_rollout(obs0, dense)
torch.cuda.synchronize()
print('kernel', env.last_kernel())
for name in ('obs', 'next_obs'):
    x, y = getattr(g.memory, name)[:T * B], getattr(want, name)[:T * B]
    bad = torch.nonzero(~((x == y) | (x.isnan() & y.isnan())))
    print(name, 'mismatches', bad.shape[0], 'nan got', int(x.isnan().sum()), 'nan want', int(y.isnan().sum()))
    for r in bad[:12].tolist():
        s, a, c = r
        print('  slot %d (t %d, e %d) agent %d col %d: got %r want %r' % (s, s // B, s % B, a, c, x[s, a, c].item(), y[s, a, c].item()))
v = g.views(g.wire[0])
print('epi[:, 0]', v['epi'][:, 0].tolist())
