"""Multi-GPU: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI).

The env batch shards trivially (no cross-env term anywhere in the step): rank r owns the global envs
``[r*B, (r+1)*B)`` via ``env_id_base`` -- the Philox reset key uses the GLOBAL env id, so an 8-way
split draws the same initial states as the unsplit batch (tests/test_gpu_parity.py).  The data path
has no collective.  The only exchange is the consumer's: the reference keeps ONE replay buffer and
ONE learner (experiments/run.py:20-21) that draws ``batch_size`` (1024) transitions every
``update_rate`` (100) steps (rls/arglist.py:15,18; run.py:78-81).

``SampledTransitionGather`` is that exchange, sized for xGMI (point-to-point links, ~153 GB/s
each) instead of copied from a single-process design: every ``update_rate`` steps each rank packs
``batch_size // world`` uniformly sampled transitions of its latest rollout chunk into rows
(``pw_pack_transitions``, one launch) and a single ``gather`` moves them peer -> root, where
``pw_replay_add_packed`` appends them to the device replay ring.  Gathering EVERY transition instead
would need 395 B x 2e9 env-steps/s = ~0.8 TB/s into one GPU -- more than all seven inbound links
together -- to refill a 1e6-slot ring thousands of times per second, of which the learner reads
1024 rows per update; the ring's capacity and the learner's appetite bound the useful ingest, so the
decimation happens at the source.  The collective is double-buffered and waited for one exchange
late, so it overlaps the next chunk's rollout.
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import _lib
from ._lib import PwStepIO, check


def shard_env_ids(rank, world, envs_per_rank):
    """-> (env_id_base, global_num_envs) of a contiguous split."""
    return rank * envs_per_rank, world * envs_per_rank


def row_width(num_agents, obs_dim):
    return 2 * num_agents * obs_dim + num_agents + 2


class SampledTransitionGather(object):
    """Call once per rollout chunk: ``gather(chunk_outputs, chunk_actions)``.

    chunk_outputs: dict of [T,B,...] tensors as ``BatchedParticleEnv.rollout`` fills them
    (needs obs, rew_shared, terminal and, with auto-reset, final_obs); chunk_actions [T,B,N] int32.
    """

    def __init__(self, env, batch_size, rank, world, device, memory=None, every=4, group=None, seed=0):
        self.rank, self.world, self.device, self.group = rank, world, torch.device(device), group
        self.B, self.N, self.D = env.num_envs, env.n, env.obs_dim
        self.R = max(1, batch_size // world)
        self.W = row_width(self.N, self.D)
        self.every = max(1, every)
        self.memory = memory
        self.calls = 0
        self.exchanges = 0
        self.rows_ingested = 0
        self._sel = {}
        self._seed = seed
        self.send = [torch.zeros(self.R, self.W, dtype=torch.float32, device=self.device) for _ in range(2)]
        self.recv = None
        if rank == 0:
            self.recv = [[torch.zeros(self.R, self.W, dtype=torch.float32, device=self.device)
                          for _ in range(world)] for _ in range(2)]
            if self.memory is None:
                self.memory = self._make_memory()
        self._pending = None

    # -- overridable pieces (the CPU gloo test substitutes torch stand-ins for the two HIP launches)
    def _make_memory(self):
        from .replay_buffer import ReplayBuffer
        return ReplayBuffer(int(1e6), self.N, self.D, device=self.device)

    def _selection(self, T):
        if T not in self._sel:
            g = torch.Generator()
            g.manual_seed(self._seed * 7919 + self.rank)
            lo = 1 if T > 1 else 0
            sel_t = torch.randint(lo, max(T, lo + 1), (self.R,), generator=g, dtype=torch.int32)
            sel_e = torch.randint(0, self.B, (self.R,), generator=g, dtype=torch.int32)
            self._sel[T] = (sel_t.to(self.device), sel_e.to(self.device))
        return self._sel[T]

    def _pack(self, out, actions, sel_t, sel_e, rows):
        lib = _lib.load()
        io = PwStepIO()
        io.act_idx = actions.data_ptr()
        io.obs = out['obs'].data_ptr()
        io.rew_shared = out['rew_shared'].data_ptr()
        if out.get('final_obs') is not None:
            io.final_obs = out['final_obs'].data_ptr()
        if out.get('terminal') is not None:
            io.terminal = out['terminal'].data_ptr()
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        check(lib.pw_pack_transitions(C.byref(io), self.B, self.N, self.D, C.c_void_p(sel_t.data_ptr()),
                                      C.c_void_p(sel_e.data_ptr()), self.R, C.c_void_p(rows.data_ptr()), stream))

    def _ingest(self, rows):
        m = self.memory
        lib = _lib.load()
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        n = rows.shape[0]
        check(lib.pw_replay_add_packed(C.byref(m._store), m._next_idx, n, C.c_void_p(rows.data_ptr()), stream))
        m._next_idx = (m._next_idx + n) % m._maxsize
        m._len = min(m._len + n, m._maxsize)

    # -- the exchange
    def _complete(self):
        if self._pending is None:
            return
        work, slot = self._pending
        work.wait()  # NCCL: orders the current stream after the collective; does not block the host
        if self.rank == 0:
            for rows in self.recv[slot]:  # rank order => deterministic ring layout
                self._ingest(rows)
                self.rows_ingested += rows.shape[0]
        self._pending = None

    def __call__(self, out, actions):
        self.calls += 1
        if self.calls % self.every:
            return
        T = int(actions.shape[0])
        sel_t, sel_e = self._selection(T)
        slot = self.exchanges & 1
        self._complete()  # the previous exchange had a whole chunk of rollout to finish behind
        self._pack(out, actions, sel_t, sel_e, self.send[slot])
        work = dist.gather(self.send[slot], self.recv[slot] if self.rank == 0 else None, dst=0,
                           group=self.group, async_op=True)
        self._pending = (work, slot)
        self.exchanges += 1

    def finish(self):
        self._complete()
