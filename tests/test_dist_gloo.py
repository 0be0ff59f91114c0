"""CPU, world_size 2, gloo: the multi-GPU exchange choreography of multiagent_rl_amd.dist and the env-id
sharding arithmetic.
  * FullTransitionGather -- north_star's collective: every transition of both ranks lands in the root's ring in
    rank order with the PRE-reset next observation (experiments/run.py:52 vs :60);
  * SampledTransitionGather -- the decimated form the synthetic-action headline uses (double-buffered async
    gather of freshly sampled rows, rank-ordered ingest, one-exchange-late completion);
  * broadcast_actor -- the learner's weights back to the rollout ranks.
The HIP launches (pack / finalize / ring append) are replaced by the torch restatements of tests/dist_standins.py
here; their GPU parity against the same restatements is in test_gpu_engine.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from multiagent_rl_amd.dist import broadcast_actor, row_width, shard_env_ids
from tests.dist_standins import CpuFullGather, CpuSampledGather, pack_reference, ref_rows, rows_from_state

B, N, D, T = 8, 3, 10, 5
EP = 3  # episode length of the full-gather test: chunks of T = 5 steps see 1 or 2 episode ends per env


L = (D - 4) // 2


class _Env(object):
    num_envs, n, obs_dim, max_episode_len, num_landmarks = B, N, D, EP, L


class _SpreadEnv(_Env):
    """What FullTransitionGather's state-only wire asks of an env, as a stub: simple_spread with the local observation, the
    chunk's start (state, landmarks, episode numbers) and the landmarks a reset draws for (env, episode number)."""
    scenario_name, local_observation = 'simple_spread', True

    def __init__(self, rank):
        self.rank, self.start = rank, None

    def wire_start(self):
        return self.start

    def landmarks_of_episode(self, e, epn):
        return landmarks_of(self.rank, e, epn)


class _TagEnv(_SpreadEnv):
    """simple_tag 2 + 1 (A = 2 adversaries, one good agent) with L = 3 landmarks: rows of 4 + 2L + 2(N - 1) + 2G = 16 numbers."""
    scenario_name, num_adversaries, obs_dim = 'simple_tag', 2, 4 + 2 * L + 2 * (N - 1) + 2 * (N - 2)


class _RefEnv(object):
    """simple_reference: two agents, 21-number rows, two-head actions (movement, symbol)."""
    num_envs, n, obs_dim, max_episode_len, num_landmarks, scenario_name = B, 2, 21, EP, 3, 'simple_reference'


def ref_goal(rank, e, a, epn):
    return (torch.as_tensor(e) + 2 * a + torch.as_tensor(epn) + rank) % 3


def ref_chunk(rank, k, step0):
    """A simple_reference rollout chunk whose rows are what the env writes: head (vel, landmark - pos) arbitrary, the goal colour of
    the episode in progress, the one-hot of the symbol the other agent emitted in this step (zeros on a post-reset row)."""
    g = torch.Generator()
    g.manual_seed(555 + 1000 * rank + k)
    head, fhead = torch.randn(T, B, 2, 8, generator=g), torch.randn(T, B, 2, 8, generator=g)
    act = torch.stack([torch.randint(0, 5, (T, B, 2), generator=g), torch.randint(0, 10, (T, B, 2), generator=g)], -1).int()
    t = torch.arange(T)[:, None] + step0
    e = torch.arange(B)[None, :].expand(T, B)
    term = ((t + e) % EP) == EP - 1
    a = torch.arange(2)[None, None, :]
    goal_after = ref_goal(rank, e[..., None], a, episode_number(e, t + 1)[..., None])
    goal_before = ref_goal(rank, e[..., None], a, episode_number(e, t)[..., None])
    sym = act[..., 1].flip(-1).long()                                     # the OTHER agent's symbol
    obs = ref_rows(head, goal_after, torch.where(term[..., None], torch.full_like(sym, 255), sym))
    return dict(obs=obs, final_obs=ref_rows(fhead, goal_before, sym), terminal=term, act=act,
                rew_shared=torch.randn(T, B, generator=g))


def ref_obs0(rank):
    g = torch.Generator()
    g.manual_seed(31 + rank)
    e = torch.arange(B)
    goal = ref_goal(rank, e[:, None], torch.arange(2)[None, :], episode_number(e, 0)[:, None])
    return ref_rows(torch.randn(B, 2, 8, generator=g), goal, torch.full((B, 2), 255))


def landmarks_of(rank, e, epn):
    """Stand-in for the reset's Philox draw: landmarks [n, L, 2] of env e's episode number epn on this rank."""
    e, epn = torch.as_tensor(e).double()[:, None, None], torch.as_tensor(epn).double()[:, None, None]
    l, c = torch.arange(L).double()[None, :, None], torch.arange(2).double()[None, None, :]
    return torch.sin(1.3 * rank + 0.7 * e + 2.1 * epn + 0.37 * l + 1.9 * c).float()


def episode_number(e, s):
    """Episode number of env e before global step s: its clock starts at e % EP, an episode ends every EP steps."""
    return 1 + (s + e % EP) // EP


def spread_chunk(rank, k, step0, scenario='simple_spread', num_adversaries=0):
    """A rollout chunk of a simple_spread (or simple_tag) env: random states, rows that ARE the local observation of those states among the
    landmarks of the episode in progress, pre-reset rows among the landmarks of the episode that just ended."""
    g = torch.Generator()
    g.manual_seed(4242 + 1000 * rank + k)
    state = torch.randn(T, B, N, 4, generator=g)
    final_state = torch.randn(T, B, N, 4, generator=g)
    out = dict(rew_shared=torch.randn(T, B, generator=g), act=torch.randint(0, 5, (T, B, N), generator=g, dtype=torch.int32))
    gr = torch.Generator()
    gr.manual_seed(777 + 1000 * rank + k)                               # (a generator of its own: the draws above keep their values)
    out['rew'] = torch.randn(T, B, N, generator=gr)                     # per-agent rewards (the ledger's input)
    t = torch.arange(T)[:, None] + step0
    e = torch.arange(B)[None, :].expand(T, B)
    out['terminal'] = ((t + e) % EP) == EP - 1
    lm_after = landmarks_of(rank, e.reshape(-1), episode_number(e, t + 1).reshape(-1)).reshape(T, B, L, 2)
    lm_before = landmarks_of(rank, e.reshape(-1), episode_number(e, t).reshape(-1)).reshape(T, B, L, 2)
    out['obs'] = rows_from_state(state, lm_after, scenario, num_adversaries)
    out['final_obs'] = rows_from_state(final_state, lm_before, scenario, num_adversaries)
    return out


def spread_start(rank):
    """(state, landmarks, episode numbers) before global step 0."""
    g = torch.Generator()
    g.manual_seed(999 + rank)
    e = torch.arange(B)
    return torch.randn(B, N, 4, generator=g), landmarks_of(rank, e, episode_number(e, 0)), episode_number(e, 0)


def chunk(rank, k):
    g = torch.Generator()
    g.manual_seed(1000 * rank + k)
    out = dict(obs=torch.randn(T, B, N, D, generator=g), final_obs=torch.randn(T, B, N, D, generator=g),
               rew_shared=torch.randn(T, B, generator=g), terminal=torch.zeros(T, B, dtype=torch.bool))
    out['terminal'][T - 1] = True
    acts = torch.randint(0, 5, (T, B, N), generator=g, dtype=torch.int32)
    return out, acts


def full_chunk(rank, k, step0):
    """A rollout chunk as the env would produce it: env e's episode clock starts at (e % EP) so that episode ends
    are NOT in lockstep; terminal every EP steps."""
    g = torch.Generator()
    g.manual_seed(77 + 1000 * rank + k)
    out = dict(obs=torch.randn(T, B, N, D, generator=g), final_obs=torch.randn(T, B, N, D, generator=g),
               rew_shared=torch.randn(T, B, generator=g),
               act=torch.randint(0, 5, (T, B, N), generator=g, dtype=torch.int32))
    t = torch.arange(T)[:, None] + step0
    e = torch.arange(B)[None, :]
    out['terminal'] = ((t + e) % EP) == EP - 1
    return out


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)

    # ---- full gather: 3 chunks per rank
    full = CpuFullGather(_Env(), T, rank, world, 'cpu')
    assert full.lay.F == 2 and full.lay.total_bytes % 256 == 0
    obs0 = torch.full((B, N, D), float(rank))
    for k in range(3):
        src = full_chunk(rank, k, k * T)
        out = full.outputs()
        for name in ('obs', 'rew_shared', 'terminal', 'act', 'final_obs'):
            out[name].copy_(src[name])          # "the rollout kernel wrote its outputs"
        full(obs0)
        obs0 = src['obs'][T - 1].clone()
    full.finish()
    assert not full.state_wire and full.bytes_per_env_step > 4 * N * D
    dist.barrier()

    # ---- the same gather on state-only wire blocks (simple_spread): the root REBUILDS the rows
    senv = _SpreadEnv(rank)
    sfull = CpuFullGather(senv, T, rank, world, 'cpu')
    assert sfull.state_wire and sfull.lay.F == 2 and sfull.bytes_per_env_step < 0.6 * full.bytes_per_env_step
    state0, lm0, ep0 = spread_start(rank)
    for k in range(3):
        src = spread_chunk(rank, k, k * T)
        senv.start = (state0, lm0, ep0)
        out = sfull.outputs()                    # snapshots the chunk's start into the block
        for name in ('obs', 'rew_shared', 'terminal', 'act', 'final_obs'):
            out[name].copy_(src[name])           # "the rollout kernel wrote its outputs"
        sfull(None)
        e = torch.arange(B)
        state0, ep0 = src['obs'][T - 1][..., :4].clone(), episode_number(e, (k + 1) * T)
        lm0 = landmarks_of(rank, e, ep0)
    sfull.finish()
    dist.barrier()

    # ---- the same blocks into the learner rank's STATE ring (rows rebuilt when sampled), and simple_tag on the same layout
    extra = {}
    for name, mk_env, kw in (('state_ring', lambda: _SpreadEnv(rank), dict(ring='state', transport='host')),
                             ('tag', lambda: _TagEnv(rank), {})):
        xenv = mk_env()
        xg = CpuFullGather(xenv, T, rank, world, 'cpu', **kw)
        assert xg.state_wire and xg.scenario == xenv.scenario_name and xg.A == getattr(xenv, 'num_adversaries', 0)
        assert xg.transport == kw.get('transport', 'direct')    # 'host': the blocks staged through host buffers (the one-GPU multi-rank path)
        scen, A = xg.scenario, xg.A
        state0, lm0, ep0 = spread_start(rank)
        for k in range(3):
            src = spread_chunk(rank, k, k * T, scen, A)
            xenv.start = (state0, lm0, ep0)
            out = xg.outputs()
            for nm in ('obs', 'rew_shared', 'terminal', 'act', 'final_obs'):
                out[nm].copy_(src[nm])
            xg(None)
            e = torch.arange(B)
            state0, ep0 = src['obs'][T - 1][..., :4].clone(), episode_number(e, (k + 1) * T)
            lm0 = landmarks_of(rank, e, ep0)
        xg.finish()
        dist.barrier()
        if rank == 0:
            if name == 'state_ring':
                n = len(xg.memory)
                extra[name] = (xg.rows_ingested, [x.numpy() for x in xg.memory.sample_index(list(range(n)))],
                               sum(t['state'].numel() + t['next_state'].numel() + t['lm'].numel() for t in xg.memory.transitions) * 4 // n)
            else:
                extra[name] = (xg.rows_ingested, [{kk: vv.numpy() for kk, vv in tr.items() if kk in ('obs', 'next_obs', 'act', 'rew')}
                                                  for tr in xg.memory.transitions], xg.bytes_per_env_step)
    # ---- simple_reference on compact-row blocks: two action heads, the 21-number rows rebuilt at the root
    rg = CpuFullGather(_RefEnv(), T, rank, world, 'cpu')
    assert rg.ref_wire and not rg.state_wire and tuple(rg.side['act'].shape) == (T, B, 2, 2)
    obs0 = ref_obs0(rank)
    for k in range(3):
        src = ref_chunk(rank, k, k * T)
        out = rg.outputs()
        for nm in ('obs', 'rew_shared', 'terminal', 'act', 'final_obs'):
            out[nm].copy_(src[nm])
        rg(obs0)
        obs0 = src['obs'][T - 1].clone()
    rg.finish()
    dist.barrier()
    if rank == 0:
        extra['ref'] = (rg.rows_ingested, [{kk: vv.numpy() for kk, vv in tr.items()} for tr in rg.memory.transitions], rg.bytes_per_env_step)

    # ---- sampled gather
    gat = CpuSampledGather(_Env(), batch_size=4 * world, rank=rank, world=world, device='cpu', every=2, seed=3)
    assert gat.R == 4 and gat.W == row_width(N, D)
    for k in range(6):
        out, acts = chunk(rank, k)
        gat(out, acts)
    gat.finish()
    with pytest.raises(ValueError):
        gat._selection(1)       # a one-step chunk does not hold the observation acted on
    dist.barrier()

    # ---- learner -> rollout ranks: the actor's parameters as one flat broadcast
    from multiagent_rl_amd.policy import ActorNetwork
    torch.manual_seed(100 + rank)                       # every rank starts from different weights
    actor = ActorNetwork(D, 5)
    n_moved = broadcast_actor(actor, src=0)
    torch.manual_seed(100)
    want = ActorNetwork(D, 5)
    same = all(torch.equal(a, b) for a, b in zip(actor.state_dict().values(), want.state_dict().values()))
    assert same and n_moved == sum(p.numel() for p in want.parameters())
    if rank == 0:
        ring = [{k: v.numpy() for k, v in tr.items()} for tr in full.memory.transitions]
        sring = [{k: v.numpy() for k, v in tr.items()} for tr in sfull.memory.transitions]
        q.put((gat.exchanges, gat.rows_ingested, [r.numpy() for r in gat.memory.rows], full.rows_ingested, ring,
               sfull.rows_ingested, sring, extra))
    else:
        q.put((gat.exchanges, gat.rows_ingested, None, full.rows_ingested, None, sfull.rows_ingested, None, None))
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_world(world):
    if torch.cuda.device_count() > 0:
        pytest.skip('CPU-container test: it spawns (execs) worker processes, which a process that may have '
                    'initialised the GPU must not do')
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=60 + 15 * world) for _ in range(world)]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    return world, [r for r in res if r[2] is not None][0]


_world_cache = {}


def _cached_world(world):
    if world not in _world_cache:
        _world_cache[world] = _run_world(world)
    return _world_cache[world]


@pytest.fixture(scope='module')
def world2():
    return _cached_world(2)


@pytest.fixture(scope='module', params=[2, 4, 8])
def worldn(request):
    return _cached_world(request.param)


@pytest.mark.timeout(240)
def test_full_transition_gather_world2(worldn):
    """Every transition of every rank (2, 4 and 8 ranks: the 8-rank case is BASELINE configs[3]'s fan-in -- seven
    receives in one batch_isend_irecv group at the root, rank-ordered ingest), in (exchange, rank, step, env) order,
    next_obs = the pre-reset row."""
    world, root = worldn
    ingested, ring = root[3], root[4]
    assert ingested == 3 * world * T * B and len(ring) == 3 * world
    i = 0
    n_final = 0
    for k in range(3):
        for r in range(world):
            src = full_chunk(r, k, k * T)
            obs0 = torch.full((B, N, D), float(r)) if k == 0 else full_chunk(r, k - 1, (k - 1) * T)['obs'][T - 1]
            want_obs = torch.cat([obs0[None], src['obs'][:-1]], 0).reshape(T * B, N, D)
            want_next = torch.where(src['terminal'][:, :, None, None], src['final_obs'], src['obs']).reshape(T * B, N, D)
            got = ring[i]
            np.testing.assert_array_equal(got['obs'], want_obs.numpy())
            np.testing.assert_array_equal(got['next_obs'], want_next.numpy())
            np.testing.assert_array_equal(got['act'], src['act'].reshape(T * B, N).numpy().astype(np.uint8))
            np.testing.assert_array_equal(got['rew'], src['rew_shared'].reshape(T * B).numpy())
            assert not got['done'].any()
            n_final += int(src['terminal'].sum())
            i += 1
    assert n_final > 0 and n_final < 3 * world * T * B  # both kinds of rows were exercised


@pytest.mark.timeout(240)
def test_full_transition_gather_state_only_wire(worldn):
    """The same fan-in on STATE-ONLY wire blocks (2, 4, 8 ranks): what reaches the root's ring are the rows rebuilt from
    {vel, pos} + the episode's landmarks, and they equal the senders' dense rows exactly -- obs_t, the pre-reset next_obs,
    actions, rewards, in (exchange, rank, step, env) order."""
    world, root = worldn
    ingested, ring = root[5], root[6]
    assert ingested == 3 * world * T * B and len(ring) == 3 * world
    i = n_final = 0
    for k in range(3):
        for r in range(world):
            src = spread_chunk(r, k, k * T)
            if k == 0:
                st, lm, _ = spread_start(r)
                obs0 = rows_from_state(st, lm)
            else:
                obs0 = spread_chunk(r, k - 1, (k - 1) * T)['obs'][T - 1]
            want_obs = torch.cat([obs0[None], src['obs'][:-1]], 0).reshape(T * B, N, D)
            want_next = torch.where(src['terminal'][:, :, None, None], src['final_obs'], src['obs']).reshape(T * B, N, D)
            got = ring[i]
            np.testing.assert_array_equal(got['obs'], want_obs.numpy())
            np.testing.assert_array_equal(got['next_obs'], want_next.numpy())
            np.testing.assert_array_equal(got['act'], src['act'].reshape(T * B, N).numpy().astype(np.uint8))
            np.testing.assert_array_equal(got['rew'], src['rew_shared'].reshape(T * B).numpy())
            assert not got['done'].any()
            n_final += int(src['terminal'].sum())
            i += 1
    assert 0 < n_final < 3 * world * T * B


def _expected_transitions(world, chunk_fn, obs0_fn):
    """(exchange, rank, step, env) order of the root's ring: per block obs_t = the previous step's row, next_obs_t = the pre-reset row."""
    out = []
    for k in range(3):
        for r in range(world):
            src = chunk_fn(r, k, k * T)
            obs0 = obs0_fn(r) if k == 0 else chunk_fn(r, k - 1, (k - 1) * T)['obs'][T - 1]
            rows = src['obs'].shape[2:]
            out.append(dict(obs=torch.cat([obs0[None], src['obs'][:-1]], 0).reshape(T * B, *rows).numpy(),
                            next_obs=torch.where(src['terminal'][:, :, None, None], src['final_obs'], src['obs']).reshape(T * B, *rows).numpy(),
                            act=src['act'].reshape(T * B, *src['act'].shape[2:]).numpy().astype(np.uint8),
                            rew=src['rew_shared'].reshape(T * B).numpy()))
    return out


@pytest.mark.timeout(240)
def test_full_gather_into_the_state_ring(worldn):
    """The learner rank's STATE ring (2, 4, 8 ranks): it keeps {vel, pos} before / after + the episode's landmarks per transition
    (32 N + 8 L bytes instead of 8 N D) and sample_index rebuilds the rows -- every transition of every rank, in ring order, equals
    the senders' dense rows exactly (and hence the row ring of test_full_transition_gather_state_only_wire)."""
    world, root = worldn
    ingested, sampled, bytes_per = root[7]['state_ring']
    assert ingested == 3 * world * T * B and bytes_per == 32 * N + 8 * L < 8 * N * D
    want = _expected_transitions(world, spread_chunk, lambda r: rows_from_state(*spread_start(r)[:2]))
    obs, act, rew, nxt, done = sampled
    np.testing.assert_array_equal(obs, np.concatenate([w['obs'] for w in want]))
    np.testing.assert_array_equal(nxt, np.concatenate([w['next_obs'] for w in want]))
    np.testing.assert_array_equal(act.argmax(-1), np.concatenate([w['act'] for w in want]))
    np.testing.assert_array_equal(rew, np.concatenate([w['rew'] for w in want]))
    assert not done.any()


@pytest.mark.timeout(240)
def test_full_gather_simple_tag_state_only_wire(worldn):
    """simple_tag (2 adversaries + 1 good agent, L = 3) on state-only blocks at 2, 4, 8 ranks: the 16-number rows (other agents'
    relative positions, the good agent's velocity, zero padding) rebuilt at the root equal the senders' dense rows exactly."""
    world, root = worldn
    ingested, ring, per_step = root[7]['tag']
    Dt = _TagEnv.obs_dim
    assert ingested == 3 * world * T * B and len(ring) == 3 * world and per_step > 0
    tag = lambda r, k, s0: spread_chunk(r, k, s0, 'simple_tag', 2)   # noqa: E731
    want = _expected_transitions(world, tag, lambda r: rows_from_state(*spread_start(r)[:2], 'simple_tag', 2))
    assert want[0]['obs'].shape[-1] == Dt == 16
    for got, w in zip(ring, want):
        for name in ('obs', 'next_obs', 'act', 'rew'):
            np.testing.assert_array_equal(got[name], w[name], err_msg=name)


@pytest.mark.timeout(240)
def test_full_gather_simple_reference_compact_rows(worldn):
    """simple_reference (MultiDiscrete: two action heads) on compact-row blocks at 2, 4, 8 ranks: 8 of the 21 numbers of a row, one
    goal byte per agent and episode and both heads as bytes travel; the rows rebuilt at the root (goal colour, the other agent's one-hot
    symbol, zeros after a reset) equal the senders' dense rows exactly, and BOTH heads arrive."""
    world, root = worldn
    ingested, ring, per_step = root[7]['ref']
    assert ingested == 3 * world * T * B and len(ring) == 3 * world and per_step > 0   # (bytes per env-step at real sizes: the GPU test)
    want = _expected_transitions(world, ref_chunk, ref_obs0)
    ends = 0
    for got, w in zip(ring, want):
        for name in ('obs', 'next_obs', 'rew'):
            np.testing.assert_array_equal(got[name], w[name], err_msg=name)
        np.testing.assert_array_equal(got['act'], w['act'])
        assert got['act'].shape == (T * B, 2, 2) and got['act'][..., 1].max() > 4      # the symbol head is there (0..9)
        ends += int((got['obs'][1:, :, 11:].sum(-1) == 0).all(-1).sum())
    assert ends > 0                                                                    # post-reset rows (no symbol visible) occurred


@pytest.mark.timeout(120)
def test_sampled_transition_gather_world2(world2):
    world, root = world2
    exchanges, ingested, rows = root[:3]
    assert exchanges == 3 and ingested == 3 * world * 4 and len(rows) == 3
    rows = [part for r in rows for part in np.split(r, world)]
    # expected: exchange x happens on chunks 1, 3, 5 (every=2); rank order inside an exchange; every exchange
    # draws a FRESH selection from the rank's generator (consecutive draws of one seeded stream)
    gens = []
    for r in range(world):
        g = torch.Generator()
        g.manual_seed(3 * 7919 + r)
        gens.append(g)
    want, sels = [], []
    for k in (1, 3, 5):
        for r in range(world):
            sel_t = torch.randint(1, T, (4,), generator=gens[r], dtype=torch.int32)
            sel_e = torch.randint(0, B, (4,), generator=gens[r], dtype=torch.int32)
            sels.append((r, sel_t.tolist(), sel_e.tolist()))
            out, acts = chunk(r, k)
            want.append(pack_reference(out, acts, sel_t, sel_e).numpy())
    for got, w in zip(rows, want):
        np.testing.assert_array_equal(got, w)
    # terminal rows take the pre-reset observation
    assert all((w[:, :N * D] != w[:, N * D:2 * N * D]).any() for w in want)
    # the selection changes from exchange to exchange
    r0 = [s[1:] for s in sels if s[0] == 0]
    assert r0[0] != r0[1] and r0[1] != r0[2]


def test_root_receive_memory_at_c4():
    """BASELINE configs[3] (C4): 8 ranks x B = 4096, N = 6, D = 16, 100-step chunks.  The root's receive side is three slots
    of seven blocks; the block size comes from libpworld's own layout arithmetic (host code, no GPU): 414 B per env-step."""
    import ctypes as C
    from multiagent_rl_amd import _lib
    from multiagent_rl_amd.dist import FullTransitionGather
    lay = _lib.PwChunkWire()
    assert _lib.load().pw_chunk_wire_layout(100, 4096, 6, 16, 25, C.byref(lay)) == 0
    assert lay.F == 4 and lay.total_bytes % 256 == 0
    per_env_step = lay.total_bytes / (100 * 4096.0)
    assert 413.0 < per_env_step < 416.0                       # 395 B + (1 + F) / T observation batches
    got = FullTransitionGather.root_receive_bytes(8, lay.total_bytes)
    assert got == 3 * 7 * lay.total_bytes and 3.4e9 < got < 3.7e9       # three slots: ~3.6 GB of the root's 288 GB
    # the state-only block of the same chunk (what simple_spread ships): 17 N + 5 per step + 5 state / landmark batches
    slay = _lib.PwStateWire()
    assert _lib.load().pw_state_wire_layout(100, 4096, 6, 6, 25, C.byref(slay)) == 0
    assert slay.F == 4 and slay.D == 16 and slay.total_bytes % 256 == 0
    per_env_step = slay.total_bytes / (100 * 4096.0)
    assert 113.0 < per_env_step < 116.0 and per_env_step <= 130.0
    assert 9.0e8 < FullTransitionGather.root_receive_bytes(8, slay.total_bytes) < 1.05e9
    assert _lib.load().pw_state_wire_layout(1000, 8, 3, 3, 2, C.byref(slay)) < 0    # 500 episode ends per env and chunk
    # the constructor allocates exactly that (checked on the CPU stand-in at a small shape, every rank count)
    class E(object):
        num_envs, n, obs_dim, max_episode_len = 8, 3, 10, 3
    for world in (2, 8):
        g = CpuFullGather.__new__(CpuFullGather)
        g.rank, g.world, g.device, g.group = 0, world, torch.device('cpu'), None
        g.B, g.N, g.D, g.T, g.max_episode_len = 8, 3, 10, 5, 3
        small = g._layout(_lib.PwChunkWire)
        full = CpuFullGather(E(), 5, 0, world, 'cpu')
        held = sum(b.numel() for slot in full.recv for b in slot if b is not None)
        assert held == FullTransitionGather.root_receive_bytes(world, small.total_bytes)


def test_shard_env_ids():
    assert shard_env_ids(0, 8, 4096) == (0, 32768)
    assert shard_env_ids(7, 8, 4096) == (28672, 32768)
    bases = [shard_env_ids(r, 4, 100)[0] for r in range(4)]
    assert bases == [0, 100, 200, 300]
