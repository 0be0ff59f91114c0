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
(``pw_pack_transitions``, one launch) and a single ``all_gather`` (world x ~100 KB: latency-, not
bandwidth-bound, so the collective with the least launch overhead wins) delivers them to the root, where
one ``pw_replay_add_packed`` launch appends them to the device replay ring.  Gathering EVERY transition instead
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


@torch.no_grad()
def broadcast_actor(actor, src=0, group=None, fused=None):
    """After the learner rank's ``optimize()``: ship the actor's parameters (~27 k floats, 108 KB) to every rollout
    rank as ONE flat RCCL broadcast -- a per-tensor broadcast would be 13 latency-bound
    collectives -- and, if given, refresh the ``FusedActor`` weight snapshot / MFMA fragments on this rank.
    Returns the number of floats moved."""
    tensors = [p.data for p in actor.parameters()] + [b.data for b in actor.buffers() if b.is_floating_point()]
    flat = torch.cat([t.reshape(-1).to(torch.float32) for t in tensors])
    if flat.is_cuda and dist.get_backend(group) == 'gloo':   # no device transport: through the host
        host = flat.cpu()
        dist.broadcast(host, src=src, group=group)
        flat.copy_(host)
    else:
        dist.broadcast(flat, src=src, group=group)
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t))
        off += n
    if fused is not None:
        fused.refresh()
    return off


class SampledTransitionGather(object):
    """Call once per rollout chunk: ``gather(chunk_outputs, chunk_actions)``.

    chunk_outputs: dict of [T,B,...] tensors as ``BatchedParticleEnv.rollout`` fills them
    (needs obs, rew_shared, terminal and, with auto-reset, final_obs); chunk_actions [T,B,N] int32.
    """

    def __init__(self, env, batch_size, rank, world, device, memory=None, every=4, group=None, seed=0,
                 side_stream=False):
        self.rank, self.world, self.device, self.group = rank, world, torch.device(device), group
        self.B, self.N, self.D = env.num_envs, env.n, env.obs_dim
        self.R = max(1, batch_size // world)
        self.W = row_width(self.N, self.D)
        self.every = max(1, every)
        self.memory = memory
        self.calls = 0
        self.exchanges = 0
        self.rows_ingested = 0
        self._gen = None
        self._seed = seed
        self.send = [torch.zeros(self.R, self.W, dtype=torch.float32, device=self.device) for _ in range(2)]
        # every rank receives the (small) batch: one all_gather kernel has far less host and launch
        # overhead than a rooted gather's grouped send/recv, and replicated learners get the rows for free
        self.recv = [torch.zeros(world * self.R, self.W, dtype=torch.float32, device=self.device) for _ in range(2)]
        if rank == 0 and self.memory is None:
            self.memory = self._make_memory()
        self._pending = None
        # Optional: run pack / ingest on their own HIP stream behind an event recorded after the chunk's
        # rollout kernel.  Off by default: on this stack the extra stream usually shares the main stream's
        # hardware queue, so it buys nothing; the default path instead fuses ingest + pack into ONE launch
        # (pw_exchange) on the main stream (~6 us of kernel per exchange) while the collective itself
        # runs on RCCL's stream.  With side_stream=True the caller must not overwrite a chunk's output
        # buffers before the following exchange (or finish()) has run.
        # The side stream is a HIGH-PRIORITY stream: those live in their own hardware queue, so pack / ingest (and,
        # with TORCH_NCCL_HIGH_PRIORITY=1, RCCL's stream) overlap the next rollout launch instead of queueing
        # in front of it (rocprofv3 timeline: 36 us between rollout launches -> 19 us -> launch gap only).
        self.side = torch.cuda.Stream(self.device, priority=-1) if (side_stream and self.device.type == 'cuda') else None

    # -- overridable pieces (the CPU gloo test substitutes torch stand-ins for the two HIP launches)
    def _make_memory(self):
        from .replay_buffer import ReplayBuffer
        return ReplayBuffer(int(1e6), self.N, self.D, device=self.device)

    def _selection(self, T):
        """A FRESH uniform draw of R (t, e) cells per exchange (t >= 1: the observation acted on at t is the
        chunk's obs[t-1]), from a generator on the exchange's device seeded once with (seed, rank) -- no host
        round trip, and consecutive exchanges ship different episode phases / env indices."""
        if T < 2:
            raise ValueError('SampledTransitionGather needs chunks of T >= 2 steps (the observation acted on at '
                             't = 0 is not part of the chunk)')
        if self._gen is None:
            self._gen = torch.Generator(device=self.device)
            self._gen.manual_seed(self._seed * 7919 + self.rank)
        sel_t = torch.randint(1, T, (self.R,), generator=self._gen, dtype=torch.int32, device=self.device)
        sel_e = torch.randint(0, self.B, (self.R,), generator=self._gen, dtype=torch.int32, device=self.device)
        return sel_t, sel_e

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
        if not work.is_completed():
            work.wait()  # NCCL: orders the current stream after the collective; does not block the host
        if self.rank == 0:
            rows = self.recv[slot]  # [world * R, W] in rank order => deterministic ring layout
            self._ingest(rows)
            self.rows_ingested += rows.shape[0]
        self._pending = None

    def prime(self, out, actions, n=2):
        """n untimed exchanges: RCCL sets up its channels lazily on first use (milliseconds)."""
        calls, ex, ing = self.calls, self.exchanges, self.rows_ingested
        for _ in range(n):
            self.calls = self.every - 1
            self(out, actions)
        self.finish()
        self.calls, self.exchanges, self.rows_ingested = calls, ex, ing
        if self.rank == 0 and self.memory is not None and hasattr(self.memory, 'clear'):
            self.memory.clear()

    def __call__(self, out, actions):
        self.calls += 1
        if self.calls % self.every:
            return
        T = int(actions.shape[0])
        slot = self.exchanges & 1
        if self.side is None:
            self._exchange(out, actions, *self._selection(T), slot)
        else:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(self.device))  # after this chunk's rollout kernel
            with torch.cuda.stream(self.side):
                self.side.wait_event(ready)
                self._exchange(out, actions, *self._selection(T), slot)
        self.exchanges += 1

    def _exchange(self, out, actions, sel_t, sel_e, slot):
        rows_in = None
        if self._pending is not None:  # the previous exchange had a whole chunk of rollout to finish behind
            work, prev = self._pending
            # A cross-queue event wait stalls the main stream ~15-20 us on this stack even when the
            # collective finished long ago; a host-side completion query costs ~1 us.  Only if the
            # collective is genuinely still running do we order the stream behind it.
            if not work.is_completed():
                work.wait()
            self._pending = None
            if self.rank == 0:
                rows_in = self.recv[prev]
        self._ingest_and_pack(rows_in, out, actions, sel_t, sel_e, self.send[slot])
        if rows_in is not None:
            self.rows_ingested += rows_in.shape[0]
        work = dist.all_gather_into_tensor(self.recv[slot], self.send[slot], group=self.group, async_op=True)
        self._pending = (work, slot)

    def _ingest_and_pack(self, rows_in, out, actions, sel_t, sel_e, rows_out):
        """One fused launch (pw_exchange).  Subclasses that override _pack/_ingest get them called instead."""
        if type(self)._pack is not SampledTransitionGather._pack or type(self)._ingest is not SampledTransitionGather._ingest:
            if rows_in is not None:
                self._ingest(rows_in)
            self._pack(out, actions, sel_t, sel_e, rows_out)
            return
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
        m = self.memory
        n_in = 0 if rows_in is None else rows_in.shape[0]
        check(lib.pw_exchange(C.byref(m._store) if n_in else None, m._next_idx if n_in else 0, n_in,
                              C.c_void_p(rows_in.data_ptr()) if n_in else None, C.byref(io), self.B, self.N, self.D,
                              C.c_void_p(sel_t.data_ptr()), C.c_void_p(sel_e.data_ptr()), self.R,
                              C.c_void_p(rows_out.data_ptr()), stream))
        if n_in:
            m._next_idx = (m._next_idx + n_in) % m._maxsize
            m._len = min(m._len + n_in, m._maxsize)

    def finish(self):
        if self.side is None:
            self._complete()
        else:
            with torch.cuda.stream(self.side):
                self._complete()
            torch.cuda.current_stream(self.device).wait_stream(self.side)


class _HostStagedRecv(object):
    """One inbound block of ``FullTransitionGather(transport='host')``: a CPU receive into a pinned buffer; ``wait()`` completes
    it and queues the H2D copy into the root's device-side receive slot on the CURRENT stream."""

    def __init__(self, work, host, dst):
        self.work, self.host, self.dst = work, host, dst

    def is_completed(self):
        return False

    def wait(self):
        self.work.wait()
        self.dst.copy_(self.host, non_blocking=True)


class FullTransitionGather(object):
    """north_star's collective: EVERY transition of every rank's rollout chunk lands in the learner rank's replay
    ring (the reference keeps one buffer that sees every env-step: experiments/run.py:20-21,52).

    Sized for the policy-in-the-loop rollout (``pw_policy_rollout``: 2.8e8 env-steps/s per GPU since round 3, 1.9e8
    before), where a full gather fits xGMI: 395 B per env-step (+ (1 + F)/T observation batches per chunk) is
    ~110 GB/s per peer (75 before) against ~153 GB/s per link -- the root's seven inbound links are the bound now; the synthetic-action headline (4e9 env-steps/s per GPU = 1.6 TB/s per peer) keeps
    ``SampledTransitionGather``.

    Per chunk and rank ONE wire block (``pw_chunk_wire``): the rollout kernel writes ``obs`` / ``rew_shared``
    straight into the block (``outputs(slot)``: zero-copy), ``pw_chunk_wire_finalize`` (one launch) adds the
    observation acted on at step 0, the pre-reset rows of the steps that ended an episode, the byte-wide actions and
    the episode-end map; the blocks travel as direct peer -> root sends (grouped isend/irecv = RCCL send/recv in one
    group: all seven inbound xGMI links of the root at once, not a ring), triple-buffered (the root appends on a side stream beside the next rollout), completed one chunk late
    so the transfer overlaps the next chunk's rollout; the root appends block after block IN RANK ORDER with
    ``pw_replay_add_wire`` (one launch per block): exchange x, rank r, step t, env e -> ring slot
    ``(x*world + r)*T*B + t*B + e``.

    Usage per chunk:  ``out = g.outputs()``; ``policy.rollout(env, T, out)``; ``g(obs0)``; at the end ``g.finish()``.

    ``wire='state'`` (the default wherever it applies: simple_spread with the local observation) ships STATE-ONLY
    blocks (``pw_state_wire``): 16 B per agent and step instead of the 4 + 2L floats of its observation row, the
    landmarks once per episode, the pre-reset state only for the steps that ended an episode; the root REBUILDS the
    ``obs`` / ``next_obs`` rows (``pw_replay_add_state_wire``; every entry of a local row is the state itself or one float32
    subtraction, so the ring is bit-identical).  C2, T = 100: 114 B per env-step instead of 414 -- 32 GB/s per inbound xGMI
    link at 2.8e8 env-steps/s per rank instead of 115 (of ~153).  In this mode ``outputs()`` also snapshots the chunk's
    start (state, landmarks, episode numbers) into the block: call it right before the chunk's rollout launch.
    ``wire='rows'`` keeps the row block (any scenario).
    """

    SLOTS = 3

    def __init__(self, env, T, rank, world, device, memory=None, group=None, capacity=int(1e6), wire='auto', overlap_ingest=True,
                 ring='rows', transport='auto'):
        from ._lib import PwChunkWire, PwStateWire
        self.rank, self.world, self.device, self.group = rank, world, torch.device(device), group
        if transport not in ('auto', 'direct', 'host'):
            raise ValueError("transport must be 'auto', 'direct' or 'host'")
        if transport == 'auto':   # a process group without a device transport (gloo) gets the blocks through pinned host buffers
            transport = 'host' if (world > 1 and self.device.type == 'cuda' and dist.get_backend(group) == 'gloo') else 'direct'
        self.transport = transport
        self.B, self.N, self.D, self.T = env.num_envs, env.n, env.obs_dim, int(T)
        self.L = int(getattr(env, 'num_landmarks', 0))
        self.scenario = getattr(env, 'scenario_name', 'simple_spread')
        self.A = 0                                        # adversaries: agents [0, A) of simple_tag
        if self.scenario == 'simple_tag':
            self.A = int(env.cfg.num_adversaries if hasattr(env, 'cfg') else (getattr(env, 'num_adversaries', 0) or 0))
        self.env = env
        if ring not in ('rows', 'state'):
            raise ValueError("ring must be 'rows' or 'state'")
        # simple_reference (MultiDiscrete: act [T,B,N,2]) travels on its own compact-row blocks (pw_ref_wire: both heads as bytes, the
        # first 8 numbers of every row, goal bytes per episode; the root rebuilds the 21-number rows into the two-head ring).  The
        # row / state blocks carry ONE action index per agent (a two-head rollout overran their single-head side buffer).
        self.ref_wire = self.scenario == 'simple_reference'
        if self.ref_wire and (wire == 'rows' or ring == 'state'):
            raise ValueError("FullTransitionGather: simple_reference has a two-head (MultiDiscrete) action; it is served by the "
                             "compact-row wire (wire='auto') into a two-head row ring only")
        if self.scenario == 'simple_speaker_listener':
            raise ValueError('FullTransitionGather: simple_speaker_listener (per-agent action spaces) is not served')
        self.max_episode_len = int(env.cfg.max_episode_len) if hasattr(env, 'cfg') else int(env.max_episode_len)
        if wire not in ('auto', 'state', 'rows'):
            raise ValueError("wire must be 'auto', 'state' or 'rows'")
        if hasattr(env, 'cfg') and self.max_episode_len > 0 and not env.cfg.auto_reset:
            raise ValueError('FullTransitionGather needs an auto-resetting env: the blocks carry the pre-reset observation (state) of '
                             'every episode end, which only the in-kernel reset writes')
        fits = self._state_wire_applies(env) and not self.ref_wire
        if wire == 'state' and not fits:
            raise ValueError('state-only wire blocks serve simple_spread with the local observation (D = 4 + 2L) and simple_tag')
        self.state_wire = fits and wire != 'rows'
        if ring == 'state' and not self.state_wire:
            raise ValueError("ring='state' needs state-only wire blocks (a STATE ring is filled from them only)")
        self.ring_kind = ring
        self.lay = self._layout(_lib.PwRefWire if self.ref_wire else PwStateWire if self.state_wire else PwChunkWire)
        nbytes = self.lay.total_bytes
        dev = self.device
        # THREE slots: chunk k's blocks travel (and, at the root, are appended) while chunk k + 1 rolls out into the next slot and
        # chunk k + 2 may already start in the third -- the root's ring append runs on its own stream beside the NEXT rollout
        # launch instead of in front of it (profiles/r4_root_ingest.txt: at 8 blocks per chunk 22-23 -> 19-19.6 us per step)
        self.wire = [torch.zeros(nbytes, dtype=torch.uint8, device=dev) for _ in range(self.SLOTS)]
        self.recv = None
        if rank == 0:
            self.recv = [[None] + [torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(1, world)]
                         for _ in range(self.SLOTS)]
        # root: ring appends on a high-priority side stream (its own hardware queue); _slot_free[s]: slot s's blocks have been
        # appended (they may be overwritten), _finalized[s]: this rank's own block of slot s is complete on the main stream
        self._ingest_stream = torch.cuda.Stream(dev, priority=-1) if (dev.type == 'cuda' and rank == 0 and overlap_ingest) else None
        self._slot_free = [None] * self.SLOTS
        self._finalized = [None] * self.SLOTS
        self._reads_done = None     # event behind the learner's last reads of the ring (mark_reads_done)
        B, N, D = self.B, self.N, self.D
        # the rollout's outputs that do NOT travel as they are (finalize condenses them into the block)
        self.side = dict(final_obs=torch.empty(self.T, B, N, D, dtype=torch.float32, device=dev) if self.lay.F else None,
                         terminal=torch.zeros(self.T, B, dtype=torch.bool, device=dev),
                         act=torch.zeros((self.T, B, N, 2) if self.ref_wire else (self.T, B, N), dtype=torch.int32, device=dev),
                         rew=torch.empty(self.T, B, N, dtype=torch.float32, device=dev),
                         done=torch.zeros(self.T, B, N, dtype=torch.bool, device=dev))
        if self.state_wire or self.ref_wire:   # the rows stay on the sender: only their first four (eight) columns travel
            self.side['obs'] = torch.empty(self.T, B, N, D, dtype=torch.float32, device=dev)
        self._host = self._host_free = None
        if self.transport == 'host' and world > 1:
            pin = lambda: torch.empty(nbytes, dtype=torch.uint8, pin_memory=self.device.type == 'cuda')  # noqa: E731
            self._host = [[None] + [pin() for _ in range(1, world)] if rank == 0 else pin() for _ in range(self.SLOTS)]
            self._host_free = [None] * self.SLOTS   # root: the H2D copies out of slot s's host buffers have run
        self.memory = memory
        self.capacity = int(capacity)
        if rank == 0 and self.memory is None:
            self.memory = self._make_memory()
        self.exchanges = 0
        self.rows_ingested = 0
        self._pending = None

    # -- layout / views
    def _state_wire_applies(self, env):
        """Rows that are a function of {vel, pos} and the landmarks: simple_spread with the local observation, simple_tag."""
        name = getattr(env, 'scenario_name', None)
        local = env.cfg.obs_mode == _lib.PW_OBS_LOCAL if hasattr(env, 'cfg') else getattr(env, 'local_observation', True)
        if name == 'simple_tag':
            return self.D == 4 + 2 * self.L + 2 * (self.N - 1) + 2 * (self.N - self.A)
        return name == 'simple_spread' and local and self.D == 4 + 2 * self.L

    def _layout(self, Struct):
        lay = Struct()
        if Struct is _lib.PwRefWire:
            if (self.N, self.D) != (2, 21):
                raise ValueError('simple_reference: N = 2, D = 21 expected')
            check(_lib.load().pw_ref_wire_layout(self.T, self.B, self.max_episode_len, C.byref(lay)))
        elif Struct is _lib.PwStateWire:
            check(_lib.load().pw_state_wire_layout_scn(_lib.SCENARIOS[self.scenario], self.T, self.B, self.N, self.L, self.A,
                                                       self.max_episode_len, C.byref(lay)))
        else:
            check(_lib.load().pw_chunk_wire_layout(self.T, self.B, self.N, self.D, self.max_episode_len, C.byref(lay)))
        return lay

    def _view(self, block, off, shape, dtype):
        n = 1
        for x in shape:
            n *= x
        itemsize = torch.empty((), dtype=dtype).element_size()
        return block[off:off + n * itemsize].view(dtype).view(*shape)

    def views(self, block):
        """Typed views of one wire block (a uint8 tensor of ``lay.total_bytes``)."""
        lay, T, B, N, D = self.lay, self.T, self.B, self.N, self.D
        if getattr(self, 'ref_wire', False):
            F = max(lay.F, 0)
            return dict(head0=self._view(block, lay.head0, (B, N, 8), torch.float32),
                        head=self._view(block, lay.head, (T, B, N, 8), torch.float32),
                        final_head=self._view(block, lay.final_head, (F, B, N, 8), torch.float32),
                        goal=self._view(block, lay.goal, (F + 1, B, N), torch.uint8),
                        comm0=self._view(block, lay.comm0, (B, N), torch.uint8),
                        rew_shared=self._view(block, lay.rew_shared, (T, B), torch.float32),
                        act=self._view(block, lay.act, (T, B, N, 2), torch.uint8),
                        epi=self._view(block, lay.epi, (T, B), torch.uint8))
        if self.state_wire:
            F, L = max(lay.F, 0), self.L
            return dict(state0=self._view(block, lay.state0, (B, N, 4), torch.float32),
                        state=self._view(block, lay.state, (T, B, N, 4), torch.float32),
                        final_state=self._view(block, lay.final_state, (F, B, N, 4), torch.float32),
                        lm=self._view(block, lay.lm, (F + 1, B, L, 2), torch.float32),
                        ep0=self._view(block, lay.ep0, (B,), torch.int32),
                        rew_shared=self._view(block, lay.rew_shared, (T, B), torch.float32),
                        act=self._view(block, lay.act, (T, B, N), torch.uint8),
                        epi=self._view(block, lay.epi, (T, B), torch.uint8))
        return dict(obs0=self._view(block, lay.obs0, (B, N, D), torch.float32),
                    obs=self._view(block, lay.obs, (T, B, N, D), torch.float32),
                    final_rows=self._view(block, lay.final_rows, (max(lay.F, 0), B, N, D), torch.float32),
                    rew_shared=self._view(block, lay.rew_shared, (T, B), torch.float32),
                    act=self._view(block, lay.act, (T, B, N), torch.uint8),
                    fin_slot=self._view(block, lay.fin_slot, (T, B), torch.uint8))

    def outputs(self, slot=None):
        """The [T, ...] output dict for this chunk's rollout launch (``FusedActor.rollout(env, T, out)`` /
        ``BatchedParticleEnv.rollout``): obs and rew_shared are views INTO the wire block."""
        slot = self.exchanges % self.SLOTS if slot is None else slot
        if self._slot_free[slot] is not None:       # the side stream's appends of this slot's previous blocks have been issued:
            torch.cuda.current_stream(self.device).wait_event(self._slot_free[slot])   # nothing overwrites them before they ran
            self._slot_free[slot] = None
        v = self.views(self.wire[slot])
        if self.state_wire:
            self._begin(self.wire[slot])
        out = dict(obs=self.side['obs'] if (self.state_wire or self.ref_wire) else v['obs'], rew_shared=v['rew_shared'],
                   terminal=self.side['terminal'], act=self.side['act'], rew=self.side['rew'], done=self.side['done'])
        if self.side['final_obs'] is not None:
            out['final_obs'] = self.side['final_obs']
        return out

    @property
    def bytes_per_env_step(self):
        return self.lay.total_bytes / float(self.T * self.B)

    @classmethod
    def root_receive_bytes(cls, world, block_bytes):
        """HBM the learner rank holds for incoming blocks: SLOTS slots x one block per peer."""
        return cls.SLOTS * (int(world) - 1) * int(block_bytes)

    # -- overridable pieces (the CPU gloo test substitutes torch stand-ins for the two HIP launches)
    def _make_memory(self):
        from .replay_buffer import ReplayBuffer
        if self.ref_wire:
            return ReplayBuffer(self.capacity, self.N, self.D, device=self.device, act_heads=(5, _lib.PW_DIM_C), device_index=True)
        if self.ring_kind == 'state':   # the learner rank writes 32 N + 8 L bytes per transition instead of 8 N D; rows rebuilt when sampled
            return ReplayBuffer(self.capacity, self.N, self.D, device=self.device, device_index=True,
                                state_ring=dict(scenario=self.scenario, num_landmarks=self.L, num_adversaries=self.A))
        return ReplayBuffer(self.capacity, self.N, self.D, device=self.device, device_index=True)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _begin(self, block):
        """State-only wire: the chunk's start (state, landmarks, episode numbers) from the env's bound state planes."""
        check(_lib.load().pw_state_wire_begin(self.env._h, C.byref(self.lay), C.c_void_p(block.data_ptr()), self._stream()))

    def _finalize(self, block, obs0):
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())  # noqa: E731
        if self.ref_wire:
            assert obs0.is_contiguous() and obs0.dtype == torch.float32 and tuple(obs0.shape) == (self.B, self.N, self.D)
            check(_lib.load().pw_ref_wire_finalize(C.byref(self.lay), p(block), p(obs0), p(self.side['obs']), p(self.side['final_obs']),
                                                   p(self.side['terminal']), p(self.side['act']), self._stream()))
            return
        if self.state_wire:   # obs0 is already in the block as state0 (pw_state_wire_begin)
            check(_lib.load().pw_state_wire_finalize(self.env._h, C.byref(self.lay), p(block), p(self.side['obs']),
                                                     p(self.side['final_obs']), p(self.side['terminal']), p(self.side['act']),
                                                     self._stream()))
            return
        assert obs0.is_contiguous() and obs0.dtype == torch.float32 and tuple(obs0.shape) == (self.B, self.N, self.D)
        check(_lib.load().pw_chunk_wire_finalize(C.byref(self.lay), p(block), p(obs0), p(self.side['final_obs']),
                                                 p(self.side['terminal']), p(self.side['act']), self._stream()))

    def _ingest(self, block):
        m = self.memory
        lib = _lib.load()
        add = lib.pw_replay_add_ref_wire if self.ref_wire else lib.pw_replay_add_state_wire if self.state_wire else lib.pw_replay_add_wire
        check(add(C.byref(m._store), m._next_idx, C.byref(self.lay), C.c_void_p(block.data_ptr()), self._stream()))
        n = self.T * self.B
        m._next_idx = (m._next_idx + n) % m._maxsize
        m._len = min(m._len + n, m._maxsize)

    # -- the exchange
    def _post(self, slot):
        """Direct peer -> root transfers of this chunk's blocks, asynchronous."""
        if self.world == 1:
            return []
        if self.transport == 'host':
            return self._post_host(slot)
        if self.rank == 0:
            ops = [dist.P2POp(dist.irecv, self.recv[slot][r], r, group=self.group) for r in range(1, self.world)]
        else:
            ops = [dist.P2POp(dist.isend, self.wire[slot], 0, group=self.group)]
        return dist.batch_isend_irecv(ops)

    def _post_host(self, slot):
        """``transport='host'``: the same peer -> root sends staged through pinned host buffers (D2H, the group's CPU send / recv,
        H2D) -- for process groups without a device transport (gloo: ranks sharing one GPU, boxes without peer access).  The
        kernels either side (finalize, ring append) and the slot choreography are the ones of the direct path."""
        if self.rank == 0:
            if self._host_free[slot] is not None:
                self._host_free[slot].synchronize()      # the copies out of this slot's host buffers (three exchanges ago) have run
                self._host_free[slot] = None
            return [_HostStagedRecv(dist.irecv(self._host[slot][r], r, group=self.group), self._host[slot][r], self.recv[slot][r])
                    for r in range(1, self.world)]
        self._host[slot].copy_(self.wire[slot], non_blocking=True)
        if self.device.type == 'cuda':
            torch.cuda.current_stream(self.device).synchronize()     # the block is complete in host memory before the CPU send reads it
        return [dist.isend(self._host[slot], 0, group=self.group)]

    def _complete(self):
        if self._pending is None:
            return
        works, slot = self._pending
        self._pending = None
        side = self._ingest_stream
        if side is None:
            for w in works:
                if not w.is_completed():
                    w.wait()  # NCCL: orders the current stream after the transfer; does not block the host
            if self.rank == 0:
                if self._host_free is not None and self.device.type == 'cuda':
                    self._host_free[slot] = torch.cuda.Event()
                    self._host_free[slot].record(torch.cuda.current_stream(self.device))
                for r in range(self.world):  # rank order => deterministic ring layout
                    self._ingest(self.wire[slot] if r == 0 else self.recv[slot][r])
                    self.rows_ingested += self.T * self.B
            return
        # root, on the device: the appends of the PREVIOUS chunk's blocks run on the side stream, behind the transfers and behind
        # this rank's own finalize, while the main stream goes on with the next rollout launch
        with torch.cuda.stream(side):
            for w in works:
                if not w.is_completed():
                    w.wait()
            if self._finalized[slot] is not None:
                side.wait_event(self._finalized[slot])
            if self._reads_done is not None:     # the ring wraps: never overwrite rows an earlier optimize() is still sampling
                side.wait_event(self._reads_done)
            if self._host_free is not None:
                self._host_free[slot] = torch.cuda.Event()
                self._host_free[slot].record(side)
            for r in range(self.world):
                self._ingest(self.wire[slot] if r == 0 else self.recv[slot][r])
                self.rows_ingested += self.T * self.B
            ev = torch.cuda.Event()
            ev.record(side)
            self._slot_free[slot] = ev

    def wait_ingested(self):
        """Order the current stream behind every append issued so far (call before the learner samples the ring)."""
        if self._ingest_stream is not None:
            torch.cuda.current_stream(self.device).wait_stream(self._ingest_stream)

    def mark_reads_done(self):
        """The other direction of the same hand-off: call on the learner's stream right after its last read of the ring
        (``sample_index``) of an iteration.  The side stream's NEXT appends wait for this point -- at 8 ranks x T B = 409 600
        transitions per chunk a 1e6-slot ring wraps every other chunk, and an append that overtook a still-running
        ``optimize()`` would tear the (obs, next_obs) pairs it samples.  (``train_batched`` calls both; a learner that
        synchronises the host after every update -- ``float(loss)`` -- is ordered anyway.)"""
        if self._ingest_stream is not None:
            self._reads_done = torch.cuda.Event()
            self._reads_done.record(torch.cuda.current_stream(self.device))

    def __call__(self, obs0):
        """After the chunk's rollout launch (same stream): condense, complete the PREVIOUS chunk's transfer and
        append it at the root, then start this chunk's transfer."""
        slot = self.exchanges % self.SLOTS
        self._finalize(self.wire[slot], obs0)
        if self._ingest_stream is not None:
            self._finalized[slot] = torch.cuda.Event()
            self._finalized[slot].record(torch.cuda.current_stream(self.device))
        self._complete()
        self._pending = (self._post(slot), slot)
        self.exchanges += 1

    def prime(self):
        """One untimed exchange of whatever the blocks hold: RCCL sets up its peer channels lazily (milliseconds)."""
        ex, ing = self.exchanges, self.rows_ingested
        self._pending = (self._post(0), 0)
        works, _ = self._pending
        for w in works:
            w.wait()
        self._pending = None
        self.exchanges, self.rows_ingested = ex, ing

    def finish(self):
        self._complete()
        self.wait_ingested()
