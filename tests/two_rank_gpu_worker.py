"""Worker of tests/test_gpu_two_ranks.py: ONE rank of a world of two (or three) that share GPU 0, process group "gloo".

RCCL refuses two ranks on one device ("Duplicate GPU detected"), so on a one-GPU box the full gather's blocks travel through
pinned host buffers (FullTransitionGather(transport='host'), chosen by itself for a gloo group) -- everything either side of the
transfer is the product's HIP path in every process: the policy-in-the-loop rollout launch writing into the wire block,
pw_state_wire_begin / _finalize (pw_chunk_wire_finalize, pw_ref_wire_finalize), the root's appends on its side stream
(pw_replay_add_state_wire into a STATE ring / row ring, pw_replay_add_wire, pw_replay_add_ref_wire), pw_replay_gather.

Every rank also runs a twin of its shard (same seeds, same env_id_base) whose rollout launch fills a local ROW ring through its
own sink; the twins' rings go to rank 0 as CPU tensors, and rank 0 checks that its gathered ring holds, for exchange x and rank r,
exactly rank r's chunk x at slots [(x*world + r)*T*B, +T*B) -- bit for bit.  One JSON line per case on stdout (rank 0)."""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CASES = [
    # name, scenario, env kwargs, B, T, episode length, ring, wire, chunks
    ('spread_state_ring', 'simple_spread', dict(n=6), 512, 40, 25, 'state', 'auto', 5),
    ('spread_n3_rows_ring', 'simple_spread', dict(n=3), 100, 31, 7, 'rows', 'auto', 4),
    ('spread_row_blocks', 'simple_spread', dict(n=6), 77, 26, 25, 'rows', 'rows', 4),
    ('spread_n24_state_ring', 'simple_spread', dict(n=24), 70, 30, 25, 'state', 'auto', 4),
    ('tag_state_ring', 'simple_tag', dict(num_adversaries=4, num_good=2), 300, 50, 25, 'state', 'auto', 4),
    ('tag_rows_ring', 'simple_tag', dict(num_adversaries=2, num_good=3), 37, 30, 11, 'rows', 'auto', 4),
    ('reference_two_head', 'simple_reference', dict(), 256, 40, 25, 'rows', 'auto', 4),
]


def run_case(rank, world, dev, name, scenario, kw, B, T, ep, ring, wire, chunks):
    chunks = int(os.environ.get('PW_TWO_RANK_CHUNKS', chunks))   # stress runs: more chunks than block slots, many times over
    from multiagent_rl_amd import make_batched_env
    from multiagent_rl_amd.dist import FullTransitionGather
    from multiagent_rl_amd.policy import ActorNetwork, FusedActor
    from multiagent_rl_amd.replay_buffer import ReplayBuffer
    torch.manual_seed(0)                                             # the same weights on every rank
    mk = lambda: make_batched_env(scenario, B, auto_reset=True, max_episode_len=ep, seed=11, env_id_base=rank * B, **kw)  # noqa: E731
    env_a, env_b = mk(), mk()
    N, D = env_a.n, env_a.obs_dim
    two = scenario == 'simple_reference'
    net = ActorNetwork(D, [5, 10] if two else 5).to(dev).eval()
    wire_actor, sink_actor = FusedActor(net, seed=3 + rank), FusedActor(net, seed=3 + rank)
    full = FullTransitionGather(env_a, T, rank, world, dev, capacity=chunks * world * T * B, ring=ring, wire=wire)
    assert full.transport == 'host', full.transport
    full.prime()
    want = ReplayBuffer(chunks * T * B, N, D, **(dict(act_heads=(5, 10)) if two else {}))
    env_a.reset()
    env_b.reset()
    obs0 = env_a.observe()
    for k in range(chunks):
        out = full.outputs()
        wire_actor.rollout(env_a, T, out)
        full(obs0)
        obs0 = out['obs'][T - 1].clone()
        sink_actor.rollout(env_b, T, False, memory=want)
    full.finish()
    torch.cuda.synchronize()
    n = chunks * T * B
    mine = [x.cpu().contiguous() for x in want.sample_index(list(range(n)))]
    if rank:
        for x in mine:
            dist.send(x, 0)
        return None
    parts = [mine]
    for r in range(1, world):
        theirs = [torch.empty_like(x) for x in mine]
        for x in theirs:
            dist.recv(x, r)
        parts.append(theirs)
    assert full.rows_ingested == world * n == len(full.memory), (full.rows_ingested, len(full.memory))
    names = ('obs', 'act', 'rew', 'next_obs', 'done')
    for x in range(chunks):
        for r in range(world):
            lo = (x * world + r) * T * B
            got = full.memory.sample_index(list(range(lo, lo + T * B)))
            for nm, g, w in zip(names, got, parts[r]):
                assert torch.equal(g.cpu(), w[x * T * B:(x + 1) * T * B]), (name, 'exchange', x, 'rank', r, nm)
    distinct = all(not torch.equal(parts[0][0], parts[r][0]) for r in range(1, world))    # the shards are different worlds
    assert distinct
    state_ring = ring == 'state'
    if state_ring:
        assert tuple(full.memory.obs.shape[1:]) == (N, 4)
    return dict(case=name, world=world, transitions=world * n, bytes_per_env_step=round(full.bytes_per_env_step, 1),
                wire='ref' if full.ref_wire else 'state' if full.state_wire else 'rows', ring=ring, ok=True)


def main():
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dist.init_process_group('gloo', rank=rank, world_size=world)
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    only = sys.argv[1:]
    try:
        for case in CASES:
            if only and case[0] not in only:
                continue
            res = run_case(rank, world, dev, *case)
            if rank == 0:
                print(json.dumps(res), flush=True)
            dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
