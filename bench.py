#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched particle world on MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W`` prints ONE JSON line (rank 0).  For N > 1 the
driver may start it under ``python -m torch.distributed.run --nproc-per-node N ...`` (one rank per GPU, RCCL); a
plain ``python bench.py --gpus N`` starts those N rank processes ITSELF, as fresh children, before this process
has imported torch or touched a GPU, and exits with their return code.

Workload (BASELINE.json configs[1]): simple_spread, N = 6 agents, L = 6 landmarks, B = 4096 envs per GPU, local
observation (D = 16), episode length 25 with in-kernel auto-reset, synthetic uniform action indices
pre-generated on the device, seed 12345678.

ONE BENCH STEP = one pass of the hot path over one batch of synthetic input = ONE ``pw_rollout`` launch over an
action batch [T, B, N] with T = ``--chunk`` (1000) batched MultiAgentEnv.step's (state update + obs + reward +
done/terminal + auto-reset for all B envs; every step's outputs written to their own HBM buffers).  EXACTLY K such
launches are timed (after W untimed ones), so the driver's ``--steps 20 --warmup 5`` times 20 000 batched env
steps (~20 ms): a steady-state region, not one cold launch.  ``value`` = n_gpus * B * T * K / max-over-ranks wall
time; ``ms_per_step`` is per launch; ``config`` states T, K and the env-steps per bench step.

N > 1 (weak scaling, B per GPU fixed): envs are sharded by env_id_base.  Headline: the only exchange is the RCCL
gather of replay rows sampled from each rank's chunk (``SampledTransitionGather``); the policy-in-the-loop extra
measures north_star's collective, the FULL gather of every transition to the root's ring
(``FullTransitionGather``: state-only wire blocks, 114 B per env-step at C2 -- the root rebuilds the observation
rows), and reports bytes per env-step and GB/s per xGMI link.

Exit code: 0 only if every rank finished every part it started.  A run whose policy-in-the-loop extra timed out, whose
peer died, or whose closing barrier never completed still prints the (already measured) headline line, then exits 3; a
rank on which the extra raised exits 4 (it never enters a collective its peers are not in).

Extra objects: ``roofline`` (HIP-event timed launches vs the 8 TB/s HBM peak; algorithmic bytes 678 B/env-step;
counter traffic and the VALU issue share from the committed rocprofv3 summaries) and ``cpu_baseline`` (the
upstream-structured scalar NumPy oracle on one host core, bounded sample; rank 0, N = 1 only).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s copy ceiling)
STORE_CEILING_GBPS = 5500.0  # what a pure store stream reaches on this chip: 5.0-6.0 TB/s (tools/write_bw_probe.hip)
F32_MFMA_PEAK_TFLOPS = 157.3  # same guide: exact-f32 MFMA (16x16x4 / 32x32x2) = 64 flop/clk/SIMD; 155 TF measured


def actor_flops_per_env_step(N, D, n_out):
    """The reference actor (rls/model/ac_network_multi_gumbel.py:52-67) on one env's N observation rows: dense1 D->64, the BiLSTM's
    input projection 64->256 (2 directions x 4 gates x 32), its recurrence 2 x (32->128), the head 64->n_out; 2 flop per MAC."""
    return 2 * N * (64 * D + 64 * 256 + 2 * 32 * 128 + 64 * n_out)
STUB = bool(os.environ.get('PW_BENCH_STUB'))  # tests only: gloo + a no-op "kernel" (tests/test_bench_launcher.py)


def _profiler_preload():
    """rocprofv3 (and friends) preload a tool library that initialises the GPU before main() runs.  A process in that
    state must neither exec a launcher (self_launch -> torch.distributed.run -> rank processes) nor fork worker pools
    (on this pool an exec / fork hop from a GPU-initialised process can take the machine down)."""
    for var in ('LD_PRELOAD', 'ROCP_TOOL_LIBRARIES', 'ROCPROFILER_REGISTER_LIBRARY', 'HSA_TOOLS_LIB'):
        v = os.environ.get(var, '')
        if any(t in v for t in ('rocprof', 'roctracer', 'rocprofiler', 'librocprof')):
            return '%s=%s' % (var, v)
    return None


# ------------------------------------------------------------------------------------------------------------
# CPU baseline legs (rank 0, N = 1 only; run BEFORE this process touches the GPU -- they fork workers)
# ------------------------------------------------------------------------------------------------------------
def cpu_baseline(seconds, n_agents):
    """Reference-style CPU step: the scalar NumPy float64 oracle (upstream loop structure) driven
    through the MultiAgentEnv list API exactly as experiments/run.py drives it, 1 core."""
    import numpy as np
    from oracle import particle_oracle as po  # cpu_baseline leg: the oracle as the measured CPU port
    np.random.seed(12345678)
    env = po.make_oracle_env('simple_spread', n=n_agents)
    env.reset()
    rng = np.random.RandomState(12345678)
    eye = np.eye(5)
    steps, t0 = 0, time.perf_counter()
    while True:
        for _ in range(25):  # rls/arglist.py:5 episode
            env.step([eye[a].copy() for a in rng.randint(0, 5, n_agents)])
        env.reset()
        steps += 25
        el = time.perf_counter() - t0
        if el >= seconds:
            break
    return dict(value=steps / el, unit='env-steps/s', cores=1, kind='port',
                sample='%d env-steps (%d episodes of 25) of simple_spread N=%d, B=1, scalar NumPy float64 oracle '
                       '(restatement of the reference semantics, not the reference\'s code) via the MultiAgentEnv '
                       'list API, %.1f s' % (steps, steps // 25, n_agents, el))


def _cpu_worker(args):
    seconds, n_agents = args
    return cpu_baseline(seconds, n_agents)['value']


def _cgroup_cpu_quota():
    """CPUs this process's cgroup may use at once (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited / unknown."""
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()
        return None if q == 'max' else max(1, int(float(q) / float(per) + 0.999))
    except Exception:
        pass
    try:
        q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
        per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
        return None if q <= 0 else max(1, (q + per - 1) // per)
    except Exception:
        return None


def _host_workers():
    """Worker count for the all-core legs = the cores this process can really run on at once: its affinity mask, cut to the
    cgroup's CPU quota where there is one, and to one GPU's share of the host (os.cpu_count() / 8 GPUs: 32 of 256 on the
    MI355X box; the N = 1 line must not claim the cores of the other seven GPUs' jobs).  PW_BENCH_CPU_WORKERS overrides.
    cores, os_cpu_count, the affinity and the quota are all reported."""
    forced = os.environ.get('PW_BENCH_CPU_WORKERS')
    if forced:
        return max(1, int(forced))
    n = len(os.sched_getaffinity(0))
    quota = _cgroup_cpu_quota()
    if quota:
        n = min(n, quota)
    return max(1, min(n, max(8, (os.cpu_count() or 8) // 8)))


def _host_workers_note():
    return dict(affinity=len(os.sched_getaffinity(0)), cgroup_quota=_cgroup_cpu_quota(), os_cpu_count=os.cpu_count(),
                rule='min(affinity, cgroup CPU quota, os.cpu_count() / 8 GPUs)')


def cpu_baseline_all_cores(seconds, n_agents):
    """the same scalar oracle replicated over the host cores (independent envs)."""
    import multiprocessing as mp
    n = _host_workers()
    with mp.get_context('fork').Pool(n) as pool:
        rates = pool.map(_cpu_worker, [(seconds, n_agents)] * n)
    return dict(value=float(sum(rates)), unit='env-steps/s', cores=n, os_cpu_count=os.cpu_count(),
                sample='%d processes x %.0f s of the same workload' % (n, seconds))


def c_oracle_rate(B, n_agents, steps=50):
    import numpy as np
    from oracle import c_oracle as co
    cfg = co.make_config('simple_spread', n_agents, max_episode_len=25, auto_reset=True)
    o = co.COracle(cfg, B, np.float32)
    o.reset()
    act = np.random.RandomState(0).randint(0, 5, (B, n_agents)).astype(np.int32)
    o.step(act_idx=act)
    t0 = time.perf_counter()
    for _ in range(steps):
        o.step(act_idx=act)
    return B * steps / (time.perf_counter() - t0)


def _c_worker(args):
    B, n_agents = args
    return c_oracle_rate(B, n_agents, steps=25)


def c_oracle_all_cores(B, n_agents):
    """The plain-C float32 restatement (batched, compiled, no Python in the loop; it stands in for BASELINE.md's
    "vectorised NumPy [B,N] oracle" B2 and is stronger than it) on the host cores."""
    import multiprocessing as mp
    n = _host_workers()
    with mp.get_context('fork').Pool(n) as pool:
        rates = pool.map(_c_worker, [(B, n_agents)] * n)
    return dict(value=float(sum(rates)), unit='env-steps/s', cores=n, os_cpu_count=os.cpu_count(),
                sample='%d processes x 25 batched steps of B=%d' % (n, B))


# ------------------------------------------------------------------------------------------------------------
# self-launch of the N rank processes
# ------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n_ranks, argv):
    """``python bench.py --gpus N`` from a plain shell: start N fresh rank processes under torch.distributed.run
    (children of this process, which has not imported torch nor made any GPU call) and return their exit code."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n_ranks),
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')  # dmabuf IPC: RCCL needs it on this host driver
    return subprocess.call(cmd, env=env)


# ------------------------------------------------------------------------------------------------------------
# tests only: a rollout "env" that launches nothing (PW_BENCH_STUB=1), so that the launcher, the rank plumbing,
# both exchanges and the JSON line run on CPU with gloo.  Its line says data = "stub".
# ------------------------------------------------------------------------------------------------------------
class _StubEnv(object):
    def __init__(self, B, N, rank):
        import torch
        self.num_envs, self.n, self.num_landmarks, self.obs_dim = B, N, N, 4 + 2 * N
        self.max_episode_len, self.bytes_per_env_step, self.rank = 25, 57 * N + 8 * N + 8 * N * N, rank
        self._torch = torch

    def reset(self):
        return self._torch.zeros(self.num_envs, self.n, self.obs_dim)

    def alloc_outputs(self, T, coll=False):
        t, B, N, D = self._torch, self.num_envs, self.n, self.obs_dim
        out = dict(obs=t.zeros(T, B, N, D), final_obs=t.zeros(T, B, N, D), rew=t.zeros(T, B, N),
                   rew_shared=t.zeros(T, B), done=t.zeros(T, B, N, dtype=t.bool), terminal=t.zeros(T, B, dtype=t.bool))
        out['terminal'][24::25] = True
        return out

    def plan_rollout(self, actions, out):
        def launch():
            time.sleep(2e-4)
        return launch

    def stub_policy_chunk(self, out, k):
        """Stands in for pw_policy_rollout writing a chunk's outputs (recognisable values)."""
        out['obs'].fill_(float(1000 * self.rank + k))
        out['rew_shared'].fill_(float(k))
        out['act'].fill_(k % 5)
        out['terminal'].zero_()
        out['terminal'][24::25] = True
        if out.get('final_obs') is not None:
            out['final_obs'].fill_(-float(1000 * self.rank + k))


_PROFILE_REFUSED = []   # (file, reason) of summaries that matched a workload but were collected from other kernel code


def _profile_lookup(scenario, N, B, kernel=None, policy=False):
    """Counter-side figures of the dominant kernel from the committed rocprofv3 summaries (profiles/): HBM traffic
    per ENV-STEP (FETCH_SIZE + WRITE_SIZE passes, per the MI355X guide) and the VALU issue share (SQ counters).
    PMC counters cannot be read from inside this process; bench scales the per-env-step figure to its launch.
    A summary is quoted only if it was collected from the kernel code this tree holds: it records the hash of the kernel
    family's sources (tools/summarize_prof.py -> kernel_source_sha16; multiagent_rl_amd/build_native.py kernel_source_hash) and
    one whose hash differs -- or that predates the hash (rounds 1-4) -- is REFUSED, never replayed (the r3 summaries
    were quoted in the r4 line after the kernels had changed)."""
    import glob
    from multiagent_rl_amd import build_native
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_summary.json')), reverse=True):
        try:
            sm = json.load(open(f))
        except Exception:
            continue
        w = sm.get('workload', {})
        if w.get('scenario') == scenario and w.get('N') == N and w.get('B') == B and bool(w.get('policy')) == policy \
                and 'traffic_bytes_per_env_step' in sm:
            # counters belong to a kernel: only a profile of the kernel that actually ran is quoted
            fam = (kernel or '').split('<')[0]
            if fam and fam not in sm.get('kernel', ''):
                continue
            rel = os.path.relpath(f, ROOT)
            have = sm.get('kernel_source_sha16')
            want = build_native.kernel_source_hash(sm.get('kernel_family') or build_native.kernel_family(sm.get('kernel', '')))
            if have != want:
                _PROFILE_REFUSED.append((rel, 'no kernel-source hash recorded (collected before round 5)' if have is None else
                                         'kernel sources changed since it was collected (%s then, %s now)' % (have, want)))
                continue
            best = dict(sm, _file=rel)
            break
    return best


def _profile_refusals(prefix_filter=None):
    """Why no counter traffic is quoted, if a summary exists but was refused."""
    r = [x for x in _PROFILE_REFUSED if prefix_filter is None or prefix_filter in x[0]]
    return None if not r else 'refused: %s -- %s' % r[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=40,
                    help='timed bench steps = pw_rollout launches of --chunk batched env steps each')
    ap.add_argument('--warmup', type=int, default=4, help='untimed launches of the same size')
    ap.add_argument('--ramp-ms', type=float, default=60.0, help='untimed launches for this long before the warm-up (clock ramp)')
    ap.add_argument('--envs', type=int, default=4096, help='B per GPU')
    ap.add_argument('--agents', type=int, default=6)
    ap.add_argument('--chunk', type=int, default=1000,
                    help='batched env steps per pw_rollout launch (40 episodes; the per-launch cost, ~7 us of launch gap '
                         '+ tail, is 0.7 %% of a 1000-step launch)')
    ap.add_argument('--scenario', default='simple_spread')
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--batch-size', type=int, default=1024, help='replay rows gathered per sampled exchange (N > 1)')
    ap.add_argument('--policy-steps', type=int, default=1000, help='batched env steps of the policy-in-the-loop extra')
    ap.add_argument('--policy-chunk', type=int, default=100, help='steps per pw_policy_rollout launch')
    ap.add_argument('--exit-timeout', type=int, default=30, help='seconds the closing barrier may take (N > 1)')
    ap.add_argument('--policy-timeout', type=int, default=240, help='seconds after which the policy-in-the-loop extra is '
                                                                    'abandoned (the line is printed without it)')
    args = ap.parse_args()

    preload = _profiler_preload()
    if args.gpus > 1 and 'RANK' not in os.environ:
        if preload:
            raise SystemExit('bench.py --gpus %d under a profiler preload (%s): this process is already GPU-initialised and '
                             'must not start the rank launcher.  Profile ONE rank directly instead, e.g.\n'
                             '  PW_BENCH_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 rocprofv3 --kernel-trace --stats -- '
                             'python3 bench.py --no-cpu-baseline   (profiles/README.md)' % (args.gpus, preload))
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    # multi-process GPU work on this pool needs dmabuf IPC (the image exports it; kept here for an env that was built without it) --
    # read by the HSA runtime when the first process touches the GPU, so before torch does
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))

    # CPU baseline first, on rank 0 at N = 1 only, BEFORE this process touches the GPU (it forks workers)
    cpu_line = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline and args.scenario == 'simple_spread' and not STUB:
        cpu_line = cpu_baseline(args.cpu_seconds, args.agents)
        cpu_line['c_oracle_f32_1core_env_steps_per_s'] = c_oracle_rate(args.envs, args.agents)
        if preload:  # the all-core legs fork worker pools: not from a GPU-initialised (profiled) process
            cpu_line['all_cores'] = cpu_line['c_oracle_f32_all_cores'] = dict(skipped='profiler preload detected: %s' % preload)
        else:
            cpu_line['all_cores'] = cpu_baseline_all_cores(min(6.0, args.cpu_seconds), args.agents)
            cpu_line['c_oracle_f32_all_cores'] = c_oracle_all_cores(args.envs, args.agents)
        cpu_line['os_cpu_count'] = os.cpu_count()
        cpu_line['all_cores_workers'] = _host_workers_note()

    import torch
    import torch.distributed as dist

    if STUB:
        dev = torch.device('cpu')
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device('cuda', local_rank)

    def sync():
        if not STUB:
            torch.cuda.synchronize()

    use_dist = world > 1 or bool(os.environ.get('PW_BENCH_FORCE_DIST'))  # the latter: 1-rank RCCL rehearsal
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29531')
        # RCCL's streams in their own (high-priority) hardware queue: the exchange then overlaps the next rollout
        # launch instead of sitting in front of it in the main stream's queue (profiles/README.md, timeline)
        os.environ.setdefault('TORCH_NCCL_HIGH_PRIORITY', '1')
        if STUB:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    B, N, K, W, T = args.envs, args.agents, max(1, args.steps), max(0, args.warmup), max(2, args.chunk)
    if STUB:
        env = _StubEnv(B, N, rank)
    else:
        from multiagent_rl_amd.env import BatchedParticleEnv
        kw = dict(num_agents=N) if args.scenario == 'simple_spread' else dict(num_adversaries=4, num_good=2)
        env = BatchedParticleEnv(args.scenario, B, max_episode_len=25, auto_reset=True, seed=12345678,
                                 env_id_base=rank * B, **kw)
    N, D = env.n, env.obs_dim
    gen = torch.Generator(device=dev)
    gen.manual_seed(12345678 + rank)

    # Output buffers of RING launches, reused round-robin (a launch's outputs are consumed -- here: sampled by the
    # exchange right after it -- long before RING - 1 more launches have run).  ~1 GB per 1000-step launch at C2.
    slot_bytes = T * B * N * (2 * D * 4 + 4 + 4 + 1) + T * B * 5       # obs + final_obs + rew + action + done, + per-env planes
    RING = max(1, min(4, K, int(96e9 // max(1, slot_bytes))))
    acts = torch.randint(0, 5, (RING * T, B, N), generator=gen, device=dev, dtype=torch.int32)
    outs = env.alloc_outputs(RING * T, coll=False)
    slots = []
    for i in range(RING):
        view = {k: v[i * T:(i + 1) * T] for k, v in outs.items()}
        a = acts[i * T:(i + 1) * T]
        slots.append((env.plan_rollout(a, view), view, a))  # pw_step_io bound once per slot, as a C host would

    exchange_state = {'error': None}
    shard = None
    if use_dist:
        if STUB:
            from tests.dist_standins import CpuSampledGather as Gather
        else:
            from multiagent_rl_amd.dist import SampledTransitionGather as Gather
        # one exchange per launch: --batch-size rows in all, drawn afresh from each rank's latest chunk
        shard = Gather(env, args.batch_size, rank, world, dev, every=1,
                       **({} if STUB else dict(side_stream=not os.environ.get('PW_BENCH_NO_SIDE_STREAM'))))

    def run(n_launches, first, events=None, exchange=True):
        # ONE HIP-event pair brackets all launches of the timed region on the launch stream (a pair per launch
        # would put two extra packets between dependent kernels and slow what it measures)
        if events is not None:
            events[0].record()
        for i in range(n_launches):
            launch, view, a = slots[(first + i) % RING]
            launch()
            if exchange and shard is not None and exchange_state['error'] is None:
                try:
                    shard(view, a)
                except Exception as e:  # keep the sharded rollout measurable; the JSON line reports this
                    exchange_state['error'] = repr(e)[:200]
        if events is not None:
            events[1].record()
        if exchange and shard is not None and exchange_state['error'] is None:
            try:
                shard.finish()
            except Exception as e:
                exchange_state['error'] = repr(e)[:200]

    # Clock ramp.  A chip that has been idle (the CPU baseline above runs for ~20 s) reaches its sustained clock only
    # after tens of milliseconds of load: with the driver's `--warmup 5` (3 ms of launches) the timed launches ran
    # 6 % slower than after 50 (profiles/README.md).  So the SAME launch runs untimed for --ramp-ms first; the env is
    # then reset again, so the W warm-up and K timed launches below start from the state they always started from.
    ramp = dict(launches=0, ms=0.0)
    if not STUB and args.ramp_ms > 0:
        env.reset()
        t_r = time.perf_counter()
        while (time.perf_counter() - t_r) * 1e3 < args.ramp_ms:
            for i in range(8):
                slots[i % RING][0]()
            sync()
            ramp['launches'] += 8
        ramp['ms'] = (time.perf_counter() - t_r) * 1e3
    env.reset()
    run(W, 0)
    if shard is not None:
        try:
            shard.prime(slots[0][1], slots[0][2])
        except Exception as e:
            exchange_state['error'] = repr(e)[:200]
        shard.exchanges = shard.rows_ingested = 0  # report the timed region's exchanges only (nothing is pending)
    events = None if STUB else (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))

    sync()
    if use_dist:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    run(K, W, events)
    sync()
    if use_dist:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0

    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # N > 1: what the sampled exchange costs the headline -- the same K launches once more WITHOUT it (same bracket);
    # reported beside `value` in multi_gpu.sampled_exchange, never instead of it
    elapsed_plain = None
    if use_dist:
        sync()
        dist.barrier()
        sync()
        t1 = time.perf_counter()
        run(K, W + K, None, exchange=False)
        sync()
        dist.barrier()
        sync()
        t = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed_plain = float(t.item())

    # dominant kernel; HIP events on the launch stream.  Average launch duration = event-bracketed time of the K
    # launches / K (includes the inter-launch gaps, so it is an upper bound of the kernel's own duration; the
    # committed rocprofv3 kernel trace gives that one and must agree)
    launch_ms = (events[0].elapsed_time(events[1]) if events else elapsed * 1e3) / K
    env_steps_per_launch = B * T
    bytes_per_launch = float(env.bytes_per_env_step) * env_steps_per_launch
    achieved = bytes_per_launch / (launch_ms * 1e-3) / 1e9

    prof = _profile_lookup(args.scenario, N, B, env.last_kernel() if hasattr(env, 'last_kernel') else None)
    traffic = frac_by_traffic = None
    issue = None
    if prof is not None:
        traffic = prof['traffic_bytes_per_env_step'] * env_steps_per_launch
        frac_by_traffic = traffic / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS
        if 'valu_issue_share' in prof:
            issue = dict(valu_issue_share=prof['valu_issue_share'], valu_insts_per_wave_step=prof.get('valu_insts_per_wave_step'),
                         clocks_per_wave_step=prof.get('clocks_per_wave_step'), source=prof.get('sq_source'))

    # the last timed launch went to slot (W + K - 1) % RING; an env terminates at its 25th, 50th, ... step, and the
    # W + K launches ran T steps each from a fresh reset: step index s of that launch is global step
    # (W + K - 1) * T + s
    n_done = W + K + (K if use_dist else 0)   # launches since the reset (the exchange-free repeat included)
    last_view = slots[(n_done - 1) % RING][1]
    s_term = (-((n_done - 1) * T + 1)) % 25
    finite = bool(torch.isfinite(last_view['obs']).all().item())
    if s_term < T and not STUB:
        finite = finite and bool(last_view['terminal'][s_term].all().item())

    # ---- policy in the loop ("report policy-in-the-loop separately"; never part of `value`) ----
    # The headline is measured at this point.  The extra below must never cost the run its ONE JSON line: a watchdog
    # thread prints the line without the extra and ends the process if the extra does not return in time (a peer that
    # died inside a point-to-point exchange would otherwise leave the root waiting in a device synchronisation, which no
    # Python exception or signal handler can interrupt).
    # A rank on which the extra RAISES tells the others through the process group's key-value store (a TCP side channel,
    # not a collective): the root's watchdog sees the flag within a fraction of a second, prints the line with that error
    # and leaves; the failing rank leaves only after the line is out (a launcher that reaps every worker on the first
    # non-zero exit must not take the root's line with it).
    line_holder = {}
    policy_line = None
    watchdog = None
    store = None
    if use_dist:
        try:
            store = dist.distributed_c10d._get_default_store()
        except Exception:
            store = None
    FAIL_KEY, LINE_KEY = 'pw_bench_failed', 'pw_bench_line_out'
    if args.scenario == 'simple_spread' and not os.environ.get('PW_BENCH_NO_POLICY'):
        import threading

        def _give_up(why):
            if rank == 0 and 'line' in line_holder:
                ln = dict(line_holder['line'])
                ln['policy_in_loop'] = dict(error=why)
                print(json.dumps(ln, allow_nan=False), flush=True)
                if store is not None:
                    try:
                        store.set(LINE_KEY, '1')
                    except Exception:
                        pass
            os._exit(3)   # the line is out, but this run did NOT finish: never report success (torchrun / the driver see it)

        class _Watchdog(threading.Thread):
            daemon = True

            def __init__(self):
                threading.Thread.__init__(self)
                self.stop = threading.Event()

            def cancel(self):
                self.stop.set()

            def run(self):
                deadline = time.monotonic() + args.policy_timeout + (0 if rank == 0 else 10)
                while not self.stop.wait(0.25):
                    if time.monotonic() >= deadline:
                        _give_up('the policy-in-the-loop extra did not finish within %d s; headline unaffected'
                                 % args.policy_timeout)
                    if store is not None:
                        try:
                            if store.check([FAIL_KEY]):
                                _give_up('the policy-in-the-loop extra did not finish: %s; headline unaffected'
                                         % store.get(FAIL_KEY).decode(errors='replace'))
                        except Exception:
                            pass
        watchdog = _Watchdog()
    if rank == 0:
        value = world * B * T * K / elapsed
        # the dispatcher's own record of what it launched (pw_rollout_kernel)
        kernel = env.last_kernel() if hasattr(env, 'last_kernel') else 'stub (no device)'
        line = {
            'metric': 'env-steps/sec, simple_spread N=6 x B envs, 1/2/4/8 MI355X',
            'value': value, 'unit': 'env-steps/s', 'n_gpus': world, 'steps': K, 'warmup': W,
            'ms_per_step': elapsed * 1e3 / K, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'stub (tests only: no kernel ran)' if STUB else 'synthetic',
            'config': {'workload': '%s N=%d L=%d, B=%d envs per GPU (global %d), local obs D=%d, episode 25 with '
                                   'auto-reset, uniform int32 action indices; 1 bench step = 1 pw_rollout launch = %d '
                                   'batched env steps; %d launches timed after %d warm-up launches (before those: an '
                                   'untimed clock ramp of %d of the same launches, %.0f ms, then a reset)'
                                   % (args.scenario, N, env.num_landmarks, B, world * B, D, T, K, W, ramp['launches'], ramp['ms']),
                       'clock_ramp': ramp,
                       'env_steps_per_step': world * B * T, 'batched_env_steps_per_launch': T,
                       'batched_env_steps_timed': K * T, 'us_per_batched_env_step': elapsed * 1e6 / (K * T),
                       'global_batch': world * B, 'parallelism': 'env-shard x%d' % world,
                       'exchange': None if shard is None else dict(
                           kind='RCCL all_gather of %d freshly sampled transition rows per rank after every launch '
                                '(%d batched steps) into the root replay ring' % (shard.R, T),
                           exchanges=shard.exchanges, rows_ingested_root=shard.rows_ingested,
                           error=exchange_state['error']),
                       'outputs_finite': finite, 'timed_region_s': elapsed},
            # `bound` names what the kernel is measured to be limited by (profiles/*_summary.json SQ shares and the
            # store-ceiling probe), `frac` stays on SURVEY 8(d)'s algorithmic bytes against the 8 TB/s peak (the contract)
            'roofline': {'bound': 'latency' if B * N <= 65536 else 'valu-issue+store', 'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBPS, 'priced_against': 'hbm',
                         'traffic': traffic, 'frac_by_traffic': frac_by_traffic,
                         'frac_vs_store_ceiling': None if traffic is None else traffic / (launch_ms * 1e-3) / 1e9 / STORE_CEILING_GBPS,
                         'store_ceiling_GBps': STORE_CEILING_GBPS,
                         'traffic_source': _profile_refusals() if prof is None else
                         '%s: %.1f B/env-step (FETCH_SIZE + WRITE_SIZE passes of a %s-step launch; kernel sources %s = this '
                         'tree) x %d env-steps'
                         % (prof['_file'], prof['traffic_bytes_per_env_step'],
                            prof.get('workload', {}).get('steps_per_launch'), prof.get('kernel_source_sha16'), env_steps_per_launch),
                         'profile_launch_ms': None if prof is None else prof.get('timed_avg_ns', prof.get('avg_ns', 0.0)) * 1e-6,
                         'issue': issue,
                         'algorithmic_bytes_per_launch': bytes_per_launch, 'kernel': kernel, 'launch_ms': launch_ms,
                         'bytes_per_env_step': env.bytes_per_env_step, 'env_steps_per_launch': env_steps_per_launch,
                         'limiter': 'latency: one wave\'s dependent instruction stream per step (DESIGN.md 4), not HBM '
                                    'bandwidth -- frac_by_traffic is the share of the 8 TB/s this launch really moves'
                                    if B * N <= 65536 else 'mix of VALU issue and the HBM write path (DESIGN.md 4)',
                         'note': 'achieved = %d B (SURVEY 8(d) algorithmic bytes per env-step) x env-steps per launch / '
                                 'launch_ms; the T-step fused launch keeps state in registers, so counter traffic is BELOW '
                                 'the algorithmic bytes; frac_vs_store_ceiling = counter traffic (97 %% stores) against the '
                                 '%.1f TB/s a pure store stream reaches' % (env.bytes_per_env_step, STORE_CEILING_GBPS / 1e3)},
        }
        if world == 1:
            line['cpu_baseline'] = cpu_line
        line_holder['line'] = line
    if watchdog is not None:
        watchdog.start()
        try:
            del outs, acts, slots, last_view
            if not STUB:
                torch.cuda.empty_cache()
            policy_line = policy_in_loop(args, env if STUB else None, rank, world, dev, use_dist, sync)
        except Exception as e:  # the headline must not depend on this extra
            policy_line = dict(error=repr(e)[:300], fatal=bool(use_dist))
        if isinstance(policy_line, dict) and policy_line.get('fatal') and store is not None:
            try:
                store.set(FAIL_KEY, 'rank %d: %s' % (rank, policy_line.get('error')))
                if rank == 0:
                    store.set(LINE_KEY, '1')   # the root prints its own line below, with the error in it
            except Exception:
                pass
        if world == 1 and rank == 0 and not STUB and not use_dist:
            try:
                line_holder['line']['other_configs'] = other_configs(dev)
            except Exception as e:
                line_holder['line']['other_configs'] = dict(error=repr(e)[:300])
        watchdog.cancel()
    if rank == 0:
        line = line_holder['line']
        line['policy_in_loop'] = policy_line
        # Compact (config -> value, frac) strings where a reader of a truncated record still finds them: short scalars inside
        # `config` (records keep scalars of config / roofline and cut strings at ~120 characters) and once more as the LAST key
        # of the line (a tail of the line keeps its end); the long arrays stay under other_configs / policy_in_loop.
        summ = _compact_summary(line.get('other_configs'), policy_line)
        line['config'].update(summ)
        if use_dist:
            # what matters at N > 1, at the top level of the line: the FULL gather's per-link rate and completeness
            # (north_star's collective, measured in the policy-in-the-loop extra) and the cost of the sampled exchange
            g = policy_line.get('gather') if isinstance(policy_line, dict) else None
            line['multi_gpu'] = dict(
                full_gather=None if not g else dict(
                    GBps_per_link=g['GBps_per_link'], GBps_root_ingest=g['GBps_root_ingest'], peers=world - 1,
                    bytes_per_env_step=g['bytes_per_env_step'], transitions_ingested_root=g['transitions_ingested_root'],
                    expected_transitions=g['expected_transitions'], complete=g['transitions_ingested_root'] == g['expected_transitions'],
                    root_receive_bytes=3 * (world - 1) * g['bytes_per_chunk_per_rank'],   # three slots of one block per peer
                    env_steps_per_s=policy_line.get('value'), error=g['error']),
                sampled_exchange=dict(value_with_exchange=line['value'],
                                      value_without_exchange=world * B * T * K / elapsed_plain,
                                      overhead_frac=1.0 - elapsed_plain / elapsed, launches_timed=K,
                                      rows_per_exchange=None if shard is None else shard.R * world,
                                      error=exchange_state['error']),
                note='value = sharded rollout + sampled exchange (a full gather at this rate would need TB/s per peer: '
                     'DESIGN.md 6); the full gather is measured with the policy in the loop')
        line['other_configs_summary'] = summ
        # key order of the printed line: everything short first -- multi_gpu (the full gather's per-link rate and completeness)
        # right behind roofline / cpu_baseline -- the long arrays (policy_in_loop, other_configs) after it, the compact summary last:
        # a record that keeps only the head or only the tail of the line still carries the figures that matter
        late = ('policy_in_loop', 'other_configs', 'other_configs_summary')
        line = dict([(k, v) for k, v in line.items() if k not in late] + [(k, line[k]) for k in late if k in line])
        print(json.dumps(line, allow_nan=False), flush=True)
    if isinstance(policy_line, dict) and policy_line.get('fatal'):
        # the extra raised on THIS rank: its peers are inside collectives it will not join -- leave, loudly, once the root's
        # line is out (rank 0 printed it just above; another rank waits for the root's watchdog to have done so)
        if store is not None and rank != 0:
            try:
                store.wait([LINE_KEY], __import__('datetime').timedelta(seconds=8))
            except Exception:
                pass
        sys.stdout.flush()
        os._exit(4)
    if use_dist:
        # the line is out; a peer that died must not keep this process (and the launcher) alive in the closing barrier --
        # and a closing barrier that never completes is a failed run: non-zero
        import threading
        bye = threading.Timer(args.exit_timeout, lambda: os._exit(3))
        bye.daemon = True
        bye.start()
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception:  # a peer already left (e.g. through its watchdog or a failure of its own): the run failed
            bye.cancel()
            sys.stdout.flush()
            os._exit(3)
        bye.cancel()


def _compact_summary(other, policy):
    """-> dict of two strings of at most ~120 characters: 'other_configs_frac' (label value frac | ...) and
    'policy_in_loop_frac' (label value mfma-frac | ...)."""
    def short(name):
        for key, lab in (('simple_tag', 'C3'), ('N=3,', 'N3'), ('N=12', 'N12'), ('N=24', 'N24'), ('N=48', 'N48'), ('B=65536', 'B64k'),
                         ('simple_reference', 'ref')):
            if key in name:
                return lab
        return name[:6]
    out = {}
    if isinstance(other, list):
        out['other_configs_frac'] = '|'.join('%s %.2g %.2f' % (short(o['config']), o['value'], o['frac']) for o in other
                                             if isinstance(o, dict) and 'frac' in o)[:120]
    elif isinstance(other, dict) and other.get('error'):
        out['other_configs_frac'] = 'error: ' + other['error'][:100]
    if isinstance(policy, dict) and policy.get('value'):
        parts = []
        rf = policy.get('roofline') or {}
        lab = policy.get('label', 'C2')
        parts.append('%s %.3g mfma %.2f' % (lab, policy['value'], rf.get('frac', float('nan'))) if rf else '%s %.3g' % (lab, policy['value']))
        for o in policy.get('other_scenarios') or []:
            if 'value' in o:
                parts.append('%s %.3g %.2f' % (short(o['config']), o['value'], (o.get('roofline') or {}).get('frac', float('nan'))))
        b = policy.get('bf16x3_input_projection') or {}
        if b.get('value'):
            parts.append('bf16x3(inexact) %.3g' % b['value'])
        g = policy.get('gather') or {}
        if g.get('GBps_per_link') is not None:
            parts.append('gather %.0f GB/s/link %d B/env-step' % (g['GBps_per_link'], g.get('bytes_per_env_step', 0)))
        out['policy_in_loop_frac'] = '|'.join(parts).replace('nan', '-')[:120]
    return out


# The other BASELINE.json configurations, each at the longest launch its outputs allow: (label, scenario, B, kwargs, T)
OTHER_CONFIGS = (
    ('C3 simple_tag 4+2, B=8192', 'simple_tag', 8192, dict(num_adversaries=4, num_good=2), 1000),
    ('C5 simple_spread N=3, B=4096', 'simple_spread', 4096, dict(num_agents=3), 1000),
    ('C5 simple_spread N=12, B=4096', 'simple_spread', 4096, dict(num_agents=12), 1000),
    ('C5 simple_spread N=24, B=4096', 'simple_spread', 4096, dict(num_agents=24), 500),
    ('C5 simple_spread N=48, B=4096', 'simple_spread', 4096, dict(num_agents=48), 200),
    ('simple_spread N=6, B=65536 (store-bound regime)', 'simple_spread', 65536, dict(num_agents=6), 100),
)


def measure_rollout(env, dev, T, K, W, ring_cap_bytes=64e9, ramp_ms=60.0):
    """The headline's method for any env: the same launch untimed for ramp_ms (clock ramp), a reset, then K pw_rollout
    launches of T steps timed by ONE HIP-event pair on the launch stream after W untimed ones; every output written, a
    ring of output buffers reused round-robin (up to 4 slots)."""
    import torch
    B, N, D = env.num_envs, env.n, env.obs_dim
    slot_bytes = T * B * N * (2 * D * 4 + 4 + 4 + 1) + T * B * 5
    ring = max(1, min(4, K, int(ring_cap_bytes // max(1, slot_bytes))))
    acts = torch.randint(0, 5, (ring * T, B, N), device=dev, dtype=torch.int32)
    outs = env.alloc_outputs(ring * T, coll=False)
    launches = [env.plan_rollout(acts[i * T:(i + 1) * T], {k: v[i * T:(i + 1) * T] for k, v in outs.items()})
                for i in range(ring)]
    env.reset()
    t_r = time.perf_counter()
    while (time.perf_counter() - t_r) * 1e3 < ramp_ms:
        for i in range(4):
            launches[i % ring]()
        torch.cuda.synchronize()
    env.reset()
    for i in range(W):
        launches[i % ring]()
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(K):
        launches[(W + i) % ring]()
    ev[1].record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    launch_ms = ev[0].elapsed_time(ev[1]) / K
    finite = bool(torch.isfinite(outs['obs'][(((W + K - 1) % ring) + 1) * T - 1]).all().item())
    return dict(launch_ms=launch_ms, wall_s=wall, ring_slots=ring, outputs_finite=finite)


def other_configs(dev, K=10, W=3):
    """Not part of `value`: the other BASELINE.json configurations and the store-bound regime, measured the way the
    headline is (measure_rollout: HIP-event bracket over K launches after W, multi-slot output ring, every output
    written) at the longest launch that fits -- C3 and N <= 12: 1000 steps, N = 24: 500, N = 48: 200.  Each entry names the
    kernel the dispatcher ran and, where profiles/ holds a rocprofv3 summary of THAT kernel on that workload, its counter
    traffic per launch and the traffic-based fraction."""
    import torch
    from multiagent_rl_amd.env import BatchedParticleEnv
    out = []
    for name, scen, B, kw, T in OTHER_CONFIGS:
        env = BatchedParticleEnv(scen, B, max_episode_len=25, auto_reset=True, seed=12345678, **kw)
        m = measure_rollout(env, dev, T, K, W)
        per_launch = B * T
        rate = per_launch / (m['launch_ms'] * 1e-3)
        kernel = env.last_kernel()
        prof = _profile_lookup(scen, env.n, B, kernel)
        traffic = None if prof is None else prof['traffic_bytes_per_env_step'] * per_launch
        out.append(dict(config=name, value=rate, unit='env-steps/s', us_per_batched_env_step=m['launch_ms'] * 1e3 / T,
                        batched_env_steps_per_launch=T, launches_timed=K, warmup=W, ring_slots=m['ring_slots'],
                        kernel=kernel, launch_ms=m['launch_ms'], bytes_per_env_step=env.bytes_per_env_step,
                        frac=rate * env.bytes_per_env_step / 1e9 / HBM_PEAK_GBPS, traffic=traffic,
                        frac_by_traffic=None if traffic is None else traffic / (m['launch_ms'] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                        traffic_source=None if prof is None else prof['_file'],
                        profile_launch_ms=None if prof is None else prof.get('timed_avg_ns', prof.get('avg_ns', 0.0)) * 1e-6,
                        outputs_finite=m['outputs_finite']))
        del env
        torch.cuda.empty_cache()
    return out


def policy_in_loop(args, stub_env, rank, world, dev, use_dist, sync):
    """The same env with the reference's actor architecture in the loop: ``pw_policy_rollout`` = policy forward +
    Gumbel sampling + env step + auto-reset for a whole chunk in ONE launch.
    N = 1: transitions go straight into the device replay ring from the kernel (``pw_rollout_sink``).
    N > 1: north_star's collective -- every rank's chunk travels to the root's ring (``FullTransitionGather``)."""
    import torch
    import torch.distributed as dist
    B, N = args.envs, args.agents
    Tp, steps = max(2, args.policy_chunk), max(args.policy_chunk, args.policy_steps)
    n_chunks = steps // Tp
    if stub_env is None:
        from multiagent_rl_amd.env import BatchedParticleEnv
        from multiagent_rl_amd.policy import ActorNetwork, FusedActor
        from multiagent_rl_amd.replay_buffer import ReplayBuffer
        from multiagent_rl_amd.rollout import BatchedRollout
        torch.manual_seed(12345678)  # same weights on every rank
        penv = BatchedParticleEnv('simple_spread', B, num_agents=N, max_episode_len=25, auto_reset=True,
                                  seed=12345678, env_id_base=rank * B)
        actor = FusedActor(ActorNetwork(penv.obs_dim, 5).to(dev).eval(), seed=12345678 + rank)
    else:
        penv, actor = stub_env, None
    label = 'FusedActor (reference ActorNetwork: Linear-BiLSTM-Linear, random init) + Gumbel sampling'

    if not use_dist:
        def timed_collect(ro, chunks, ramp_ms=60.0, warm=3):
            """The headline's methodology (and tools/policy_profile_run.py's): the same chunk untimed for ramp_ms (clock ramp: a chip
            that idled through the previous leg's set-up reaches its sustained clock only after tens of milliseconds of load -- one
            warm-up chunk of 1.4 ms left the timed chunks 7 % slower than the profile run), `warm` more untimed chunks,
            then `chunks` chunks bracketed by a HIP-event pair on the launch stream (torch's current stream is the one every launch
            of BatchedRollout goes to) and by the host clock."""
            t_r = time.perf_counter()
            while (time.perf_counter() - t_r) * 1e3 < ramp_ms:
                ro.collect_one_launch(Tp, chunk=Tp)
                sync()
            ro.collect_one_launch(warm * Tp, chunk=Tp)
            sync()
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            t = time.perf_counter()
            ev[0].record()
            ro.collect_one_launch(chunks * Tp, chunk=Tp)
            ev[1].record()
            sync()
            return time.perf_counter() - t, ev[0].elapsed_time(ev[1]) * 1e-3

        def mfma_roofline(env, n_out, seconds, steps, launches):
            """The actor's dense products are the path's flops (north_star reserves MFMA for them): algorithmic flops per env-step
            x env-steps per launch / the HIP-event launch time, against the exact-f32 matrix peak."""
            fl = actor_flops_per_env_step(env.n, env.obs_dim, n_out)
            per_launch = float(fl) * env.num_envs * steps / launches
            launch_s = seconds / launches
            ach = per_launch / launch_s / 1e12
            # counter side (HBM traffic per launch, MFMA busy share) from the committed summary of THIS kernel code, if there is one
            prof = _profile_lookup(env.scenario_name, env.n, env.num_envs, env.last_kernel(), policy=True)
            return dict(bound='mfma_f32', achieved=ach, peak=F32_MFMA_PEAK_TFLOPS, unit='TFLOP/s', frac=ach / F32_MFMA_PEAK_TFLOPS,
                        flops_per_env_step=fl, algorithmic_flops_per_launch=per_launch, launch_ms=launch_s * 1e3,
                        traffic=None if prof is None else prof['traffic_bytes_per_env_step'] * env.num_envs * steps / launches,
                        mfma_busy_share=None if prof is None else prof.get('mfma_busy_share'),
                        traffic_source=_profile_refusals('policy') if prof is None else prof['_file'],
                        profile_launch_ms=None if prof is None else prof.get('timed_avg_ns', 0.0) * 1e-6,
                        kernel=env.last_kernel(), env_steps_per_launch=env.num_envs * steps // launches,
                        note='2 N (64 D + 64x256 + 2x32x128 + 64 n_out) flop per env-step (rls/model/ac_network_multi_gumbel.py:52-67), '
                             'exact f32 on v_mfma_f32_16x16x4_f32; the environment step, sampling and ring append ride in the same launch')

        ro = BatchedRollout(penv, actor, ReplayBuffer(1e6, N, penv.obs_dim))
        tp, tp_ev = timed_collect(ro, n_chunks)
        line = dict(value=B * n_chunks * Tp / tp, unit='env-steps/s', us_per_step=tp / (n_chunks * Tp) * 1e6,
                    steps=n_chunks * Tp, policy=label, label='C2' if (N, B) == (6, 4096) else 'N%d B%d' % (N, B),
                    actor_precision=penv.get_actor_precision(),
                    exact=penv.get_actor_precision() == 'f32',
                    loop='%d-step chunks, one launch each: pw_policy_rollout (actor + sampling + env step + device '
                         'replay append + episode stats); %d chunks timed after a 60 ms clock ramp + 3 warm-up chunks' % (Tp, n_chunks),
                    roofline=mfma_roofline(penv, 5, tp_ev, n_chunks * Tp, n_chunks))
        # every further figure is an extra of this extra: one that fails is recorded under its own key, the figure above stays
        # the per-step form: actor, env step, replay append + bookkeeping = 3 launches per step in a hipGraph
        try:
            ro.capture(2)
            ro.collect(50)
            sync()
            tg = time.perf_counter()
            ro.collect(500)
            sync()
            tg = time.perf_counter() - tg
            line['three_launches_per_step_hipgraph'] = dict(value=B * 500 / tg, us_per_step=tg / 500 * 1e6)
        except Exception as e:
            line['three_launches_per_step_hipgraph'] = dict(error=repr(e)[:200])
        # the other scenarios of the reference's sweep with their actors in the loop (same one-launch form, ring append included)
        others = []
        for name, mk_env, heads, cap in (
                ('simple_tag 4+2, B=8192',
                 lambda: BatchedParticleEnv('simple_tag', 8192, num_adversaries=4, num_good=2, max_episode_len=25, auto_reset=True, seed=1), 5, 1e6),
                ('simple_reference (two-head actor [5|10]), B=%d' % B,
                 lambda: BatchedParticleEnv('simple_reference', B, max_episode_len=25, auto_reset=True, seed=3), [5, 10], 8e6),
                # the reference's own scalability axis (main_scalability_1.py:30: n_agent in [6, 9, 12]) and BASELINE C5 beyond it
                ('simple_spread N=12, B=4096 (the reference\'s largest scalability setting)',
                 lambda: BatchedParticleEnv('simple_spread', 4096, num_agents=12, max_episode_len=25, auto_reset=True, seed=12345678), 5, 1e6),
                ('simple_spread N=24, B=4096 (C5 point; just-in-time form)',
                 lambda: BatchedParticleEnv('simple_spread', 4096, num_agents=24, max_episode_len=25, auto_reset=True, seed=12345678), 5, 1e6),
                ('simple_spread N=48, B=4096 (C5 point; just-in-time form, half-storage head)',
                 lambda: BatchedParticleEnv('simple_spread', 4096, num_agents=48, max_episode_len=25, auto_reset=True, seed=12345678), 5, 5e5)):
            try:
                o_env = mk_env()
                o_env.set_actor_precision('f32')     # whatever the process environment says: these rollouts serve float32 only
                two = isinstance(heads, list)
                o_ro = BatchedRollout(o_env, FusedActor(ActorNetwork(o_env.obs_dim, heads).to(dev).eval(), seed=2 if not two else 7),
                                      ReplayBuffer(int(cap), o_env.n, o_env.obs_dim, **(dict(act_heads=(5, 10)) if two else {})))
                to, to_ev = timed_collect(o_ro, 5)
                others.append(dict(config=name, value=o_env.num_envs * 5 * Tp / to, unit='env-steps/s', us_per_step=to / (5 * Tp) * 1e6,
                                   roofline=mfma_roofline(o_env, sum(heads) if two else heads, to_ev, 5 * Tp, 5)))
                del o_ro, o_env
            except Exception as e:
                others.append(dict(config=name, error=repr(e)[:200]))
        line['other_scenarios'] = others
        # labelled extra, never the figure above: the OPT-IN, NOT exact actor mode (LSTM input projection on bfloat16 matrix
        # instructions, three products per k step -- pw_set_actor_precision in include/pworld.h) on a fresh env + ring;
        # served by the third kernel form only (N <= 16 at 8+ environments per workgroup): skipped, not failed, elsewhere
        try:
            benv = BatchedParticleEnv('simple_spread', B, num_agents=N, max_episode_len=25, auto_reset=True,
                                      seed=12345678, env_id_base=rank * B)
            benv.set_actor_precision('bf16x3')
            rb = BatchedRollout(benv, actor, ReplayBuffer(1e6, N, benv.obs_dim))
            tb, _ = timed_collect(rb, n_chunks)
            line['bf16x3_input_projection'] = dict(value=B * n_chunks * Tp / tb, us_per_step=tb / (n_chunks * Tp) * 1e6, exact=False,
                                                   kernel=benv.last_kernel(),
                                                   note='opt-in (pw_set_actor_precision), within 2e-5 of the float32 logits, does '
                                                        'not reproduce the exact form\'s sampled actions; never the default, never '
                                                        'the headline')
        except Exception as e:
            line['bf16x3_input_projection'] = dict(skipped=repr(e)[:200], exact=False)
        return line

    if stub_env is None:
        from multiagent_rl_amd.dist import FullTransitionGather as Full
    else:
        from tests.dist_standins import CpuFullGather as Full
    # the learner rank keeps a STATE ring (round 5): it writes 32 N + 8 L bytes per gathered transition instead of 8 N D and rebuilds
    # the rows when a batch is sampled -- the root's ring appends are what bounds the 8-GPU figure (profiles/r5_root_ingest.txt)
    full = Full(penv, Tp, rank, world, dev, ring='state' if stub_env is None else 'rows')   # (the no-device stub env has no state wire)
    err = None
    obs0 = penv.reset()

    def one_chunk(k, obs0):
        out = full.outputs()
        if actor is not None:
            actor.rollout(penv, Tp, out)
        else:
            if os.environ.get('PW_BENCH_STUB_FAIL_RANK') == str(rank) and k >= 2:  # tests: a rank fails mid-exchange
                raise RuntimeError('injected failure on rank %d' % rank)
            if os.environ.get('PW_BENCH_STUB_HANG_RANK') == str(rank) and k >= 2:  # tests: a rank hangs mid-exchange
                time.sleep(3600)
            penv.stub_policy_chunk(out, k)
            time.sleep(2e-4)
        full(obs0)
        return out['obs'][Tp - 1]  # the next chunk's obs0: a view into the block that is now travelling (read-only)

    try:
        full.prime()
        obs0 = one_chunk(0, obs0)
        full.finish()
        if full.memory is not None and hasattr(full.memory, 'clear'):
            full.memory.clear()
        full.rows_ingested = 0
        sync()
        dist.barrier()
        sync()
        tp = time.perf_counter()
        for k in range(n_chunks):
            obs0 = one_chunk(1 + k, obs0)
        full.finish()
        sync()
        dist.barrier()
        sync()
        tp = time.perf_counter() - tp
    except Exception as e:
        # A local failure: the peers are in the barrier / irecv of the exchange, NOT in the all_reduce below -- entering it
        # would only trade one hang for another.  Report (rank 0 prints the line with this error) and leave with a non-zero
        # code; the peers end through their own error path or watchdog.
        return dict(error=repr(e)[:300], fatal=True, value=None, unit='env-steps/s', us_per_step=None, steps=n_chunks * Tp)
    t = torch.tensor([tp], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    tp = float(t.item())
    total = world * B * n_chunks * Tp
    per_link = full.lay.total_bytes * n_chunks / tp / 1e9
    return dict(value=total / tp, unit='env-steps/s', us_per_step=tp / (n_chunks * Tp) * 1e6, steps=n_chunks * Tp,
                policy=label,
                loop='%d-step chunks, one pw_policy_rollout launch each per rank, outputs written into the wire block' % Tp,
                gather=dict(kind='FULL gather: every transition of every rank to the root replay ring (%s -> grouped RCCL send/recv '
                                 'peer->root -> %s), triple-buffered, one chunk late'
                                 % (('STATE-ONLY wire blocks: pw_state_wire_begin / _finalize', 'pw_replay_add_state_wire rebuilds the rows')
                                    if getattr(full, 'state_wire', False) else ('row blocks: pw_chunk_wire_finalize', 'pw_replay_add_wire')),
                            wire='state' if getattr(full, 'state_wire', False) else 'rows', ring=getattr(full, 'ring_kind', 'rows'),
                            bytes_per_env_step=full.bytes_per_env_step, bytes_per_chunk_per_rank=full.lay.total_bytes,
                            GBps_per_link=per_link, GBps_root_ingest=None if per_link is None else per_link * (world - 1),
                            exchanges=full.exchanges, transitions_ingested_root=full.rows_ingested,
                            expected_transitions=total, error=err))


if __name__ == '__main__':
    main()
