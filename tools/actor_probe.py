import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multiagent_rl_amd.policy import ActorNetwork, GumbelPolicy
torch.manual_seed(0)
B, N, D = 4096, 6, 16
actor = ActorNetwork(D, 5).cuda().eval()
obs = torch.randn(B, N, D, device='cuda')
pol = GumbelPolicy(actor)
def timeit(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
with torch.no_grad():
    print('actor forward eager: %.1f us' % timeit(lambda: actor(obs)))
    print('policy (actor+gumbel+argmax) eager: %.1f us' % timeit(lambda: pol(obs)))
    x = torch.relu(actor.dense1(obs))
    print('dense1+relu: %.1f us' % timeit(lambda: torch.relu(actor.dense1(obs))))
    print('bilstm: %.1f us' % timeit(lambda: actor.bilstm(x, None)))
    h, _ = actor.bilstm(x, None)
    print('dense2: %.1f us' % timeit(lambda: actor.dense2(torch.relu(h))))
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): pol(obs)
    torch.cuda.current_stream().wait_stream(s)
    try:
        with torch.cuda.graph(g):
            out = pol(obs)
        print('policy graph replay: %.1f us' % timeit(lambda: g.replay()))
    except Exception as e:
        print('graph capture failed:', repr(e)[:200])
from torch.profiler import profile, ProfilerActivity
with torch.no_grad(), profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for _ in range(10): actor(obs)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=12, max_name_column_width=60))

# ---- fused HIP actor
from multiagent_rl_amd.policy import FusedActor
fa = FusedActor(actor)
with torch.no_grad():
    print('fused actor (2 GEMM + bilstm + head, sampled action) eager: %.1f us' % timeit(lambda: fa(obs)))
    print('  hidden only: %.1f us' % timeit(lambda: fa.hidden(obs)))
    g2 = torch.cuda.CUDAGraph()
    s2 = torch.cuda.Stream()
    with torch.cuda.stream(s2):
        for _ in range(3): fa(obs)
    torch.cuda.current_stream().wait_stream(s2)
    with torch.cuda.graph(g2):
        out2 = fa(obs)
    print('fused actor graph replay: %.1f us' % timeit(lambda: g2.replay()))
with torch.no_grad(), profile(activities=[ProfilerActivity.CUDA]) as prof2:
    for _ in range(10): fa(obs)
    torch.cuda.synchronize()
print(prof2.key_averages().table(sort_by='cuda_time_total', row_limit=8, max_name_column_width=70))
