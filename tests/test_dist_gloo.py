"""CPU, world_size 2, gloo: the multi-GPU exchange choreography of multiagent_rl_amd.dist
(double-buffered async gather of sampled transition rows into the root's ring, rank-ordered
ingest, one-exchange-late completion) and the env-id sharding arithmetic.  The two HIP launches
(pack / ring append) are replaced by torch stand-ins here; their GPU parity is in test_gpu_engine.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from multiagent_rl_amd.dist import SampledTransitionGather, broadcast_actor, row_width, shard_env_ids

B, N, D, T = 8, 3, 10, 5


class _Env(object):
    num_envs, n, obs_dim = B, N, D


class _HostRing(object):
    def __init__(self):
        self.rows = []


class _CpuGather(SampledTransitionGather):
    def _make_memory(self):
        return _HostRing()

    def _pack(self, out, actions, sel_t, sel_e, rows):
        rows.copy_(pack_reference(out, actions, sel_t, sel_e))

    def _ingest(self, rows):
        self.memory.rows.append(rows.clone())


def pack_reference(out, actions, sel_t, sel_e):
    """Row layout of include/pworld.h: [obs ND | next_obs ND | act N | rew | done]."""
    t, e = sel_t.long(), sel_e.long()
    obs = out['obs'][t - 1, e].reshape(len(t), -1)
    nxt = torch.where(out['terminal'][t, e].bool()[:, None, None], out['final_obs'][t, e], out['obs'][t, e])
    return torch.cat([obs, nxt.reshape(len(t), -1), actions[t, e].float(), out['rew_shared'][t, e][:, None],
                      torch.zeros(len(t), 1)], dim=1)


def chunk(rank, k):
    g = torch.Generator()
    g.manual_seed(1000 * rank + k)
    out = dict(obs=torch.randn(T, B, N, D, generator=g), final_obs=torch.randn(T, B, N, D, generator=g),
               rew_shared=torch.randn(T, B, generator=g), terminal=torch.zeros(T, B, dtype=torch.bool))
    out['terminal'][T - 1] = True
    acts = torch.randint(0, 5, (T, B, N), generator=g, dtype=torch.int32)
    return out, acts


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    gat = _CpuGather(_Env(), batch_size=8, rank=rank, world=world, device='cpu', every=2, seed=3)
    assert gat.R == 4 and gat.W == row_width(N, D)
    for k in range(6):
        out, acts = chunk(rank, k)
        gat(out, acts)
    gat.finish()
    dist.barrier()
    # learner -> rollout ranks: the actor's parameters as one flat broadcast
    from multiagent_rl_amd.policy import ActorNetwork
    torch.manual_seed(100 + rank)                       # every rank starts from different weights
    actor = ActorNetwork(D, 5)
    n_moved = broadcast_actor(actor, src=0)
    torch.manual_seed(100)
    want = ActorNetwork(D, 5)
    same = all(torch.equal(a, b) for a, b in zip(actor.state_dict().values(), want.state_dict().values()))
    assert same and n_moved == sum(p.numel() for p in want.parameters())
    if rank == 0:
        q.put((gat.exchanges, gat.rows_ingested, [r.numpy() for r in gat.memory.rows]))
    else:
        q.put((gat.exchanges, gat.rows_ingested, None))
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(120)
@pytest.mark.skipif(torch.cuda.device_count() > 0, reason='CPU-container test: it spawns (execs) worker processes, '
                    'which a process that may have initialised the GPU must not do')
def test_sampled_transition_gather_world2():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in range(world)]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    root = [r for r in res if r[2] is not None][0]
    exchanges, ingested, rows = root
    assert exchanges == 3 and ingested == 3 * world * 4 and len(rows) == 3
    rows = [part for r in rows for part in np.split(r, world)]
    # expected: exchange x happens on chunks 1, 3, 5 (every=2); rank order inside an exchange
    want = []
    for k in (1, 3, 5):
        for r in range(world):
            g = torch.Generator()
            g.manual_seed(3 * 7919 + r)
            sel_t = torch.randint(1, T, (4,), generator=g, dtype=torch.int32)
            sel_e = torch.randint(0, B, (4,), generator=g, dtype=torch.int32)
            out, acts = chunk(r, k)
            want.append(pack_reference(out, acts, sel_t, sel_e).numpy())
    for got, w in zip(rows, want):
        np.testing.assert_array_equal(got, w)
    # terminal rows take the pre-reset observation
    assert all((w[:, :N * D] != w[:, N * D:2 * N * D]).any() for w in want)


def test_shard_env_ids():
    assert shard_env_ids(0, 8, 4096) == (0, 32768)
    assert shard_env_ids(7, 8, 4096) == (28672, 32768)
    bases = [shard_env_ids(r, 4, 100)[0] for r in range(4)]
    assert bases == [0, 100, 200, 300]
