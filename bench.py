#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched particle world on MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 it is launched by
``python -m torch.distributed.run --nproc-per-node N ...`` (one rank per GPU, RCCL).  Rank 0
prints ONE JSON line.

Workload (BASELINE.json configs[1]): simple_spread, N = 6 agents, L = 6 landmarks, B = 4096 envs per
GPU, local observation (D = 16), episode length 25 with in-kernel auto-reset, synthetic uniform
action indices pre-generated on the device, seed 12345678.  A "step" is one batched
MultiAgentEnv.step of all B envs (state update + obs + reward + done/terminal + auto-reset);
steps are issued as pw_rollout launches of ``--chunk`` (500) steps each, every step's outputs
written to their own HBM buffers.  value = n_gpus * B * K / max-over-ranks wall time.

N > 1 (weak scaling, B per GPU fixed): envs are sharded by env_id_base; the only exchange is the
RCCL gather of the replay minibatch rows sampled from each rank's local shard to rank 0 once per
chunk (DESIGN.md "Multi-GPU").

Extra objects: ``roofline`` (HIP-event timed pw_rollout launches vs the 8 TB/s HBM peak, algorithmic
bytes = 678 B per env-step) and ``cpu_baseline`` (the upstream-structured scalar NumPy oracle on one
host core, bounded sample; rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s copy ceiling)


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
                       'via the MultiAgentEnv list API, %.1f s' % (steps, steps // 25, n_agents, el))


def _cpu_worker(args):
    seconds, n_agents = args
    return cpu_baseline(seconds, n_agents)['value']


def cpu_baseline_all_cores(seconds, n_agents):
    """the same scalar oracle replicated over every host core (independent envs)."""
    import multiprocessing as mp
    # the GPU box reports every host core but grants one GPU's share (16); stay within it
    n = max(1, min(len(os.sched_getaffinity(0)), 16))
    with mp.get_context('fork').Pool(n) as pool:
        rates = pool.map(_cpu_worker, [(seconds, n_agents)] * n)
    return dict(value=float(sum(rates)), unit='env-steps/s', cores=n,
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
    """The plain-C float32 restatement (the strongest CPU form in this repo: batched, compiled, no Python in the
    loop) on every host core this process may use: what a multi-core CPU port of the same arithmetic delivers."""
    import multiprocessing as mp
    n = max(1, min(len(os.sched_getaffinity(0)), 16))
    with mp.get_context('fork').Pool(n) as pool:
        rates = pool.map(_c_worker, [(B, n_agents)] * n)
    return dict(value=float(sum(rates)), unit='env-steps/s', cores=n,
                sample='%d processes x 25 batched steps of B=%d' % (n, B))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20000,
                    help='timed batched env steps (40 launches of --chunk; ~20 ms, so that the closing barrier of an N-GPU run '
                         'is a small part of the timed region)')
    ap.add_argument('--warmup', type=int, default=500)
    ap.add_argument('--envs', type=int, default=4096, help='B per GPU')
    ap.add_argument('--agents', type=int, default=6)
    ap.add_argument('--chunk', type=int, default=500,
                    help='steps per pw_rollout launch (20 episodes; the per-launch cost, ~9 us of launch gap + tail, is '
                         '7 %% of a 100-step launch and 1.5 %% of a 500-step one)')
    ap.add_argument('--scenario', default='simple_spread')
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--batch-size', type=int, default=1024, help='replay rows gathered per exchange (N > 1)')
    ap.add_argument('--exchange-steps', type=int, default=500,
                    help='batched steps between RCCL exchanges of --batch-size sampled rows.  500 steps = ~0.6 ms: '
                         '~1700 fresh minibatches/s at the root, an order of magnitude above what one learner '
                         '(optimize() on 1024 transitions, ddpg_gumbel_fix.py:131) can consume')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))

    # CPU baseline first, on rank 0 at N = 1 only, BEFORE this process touches the GPU (it forks workers)
    cpu_line = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline and args.scenario == 'simple_spread':
        cpu_line = cpu_baseline(args.cpu_seconds, args.agents)
        cpu_line['c_oracle_f32_1core_env_steps_per_s'] = c_oracle_rate(args.envs, args.agents)
        cpu_line['all_cores'] = cpu_baseline_all_cores(min(6.0, args.cpu_seconds), args.agents)
        cpu_line['c_oracle_f32_all_cores'] = c_oracle_all_cores(args.envs, args.agents)

    import torch
    import torch.distributed as dist
    from multiagent_rl_amd.env import BatchedParticleEnv

    if args.gpus > 1 and world != args.gpus:
        raise SystemExit('--gpus %d needs torch.distributed.run with %d ranks (WORLD_SIZE=%d)' %
                         (args.gpus, args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    use_dist = world > 1 or bool(os.environ.get('PW_BENCH_FORCE_DIST'))  # the latter: 1-rank RCCL rehearsal
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29531')
        # RCCL's streams in their own (high-priority) hardware queue: the exchange then overlaps the next rollout
        # launch instead of sitting in front of it in the main stream's queue (profiles/README.md, timeline)
        os.environ.setdefault('TORCH_NCCL_HIGH_PRIORITY', '1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    B, N, K, W, T = args.envs, args.agents, args.steps, args.warmup, max(1, args.chunk)
    kw = dict(num_agents=N) if args.scenario == 'simple_spread' else dict(num_adversaries=4, num_good=2)
    env = BatchedParticleEnv(args.scenario, B, max_episode_len=25, auto_reset=True, seed=12345678,
                             env_id_base=rank * B, **kw)
    N, D = env.n, env.obs_dim
    gen = torch.Generator(device=dev)
    gen.manual_seed(12345678 + rank)

    def make_chunks(total):
        return [min(T, total - s) for s in range(0, total, T)]

    RING = 4  # output buffers of 4 launches, reused round-robin (a launch's outputs are consumed -- here: sampled by
    #           the exchange right after it -- long before 3 more launches have run)

    def alloc(total):
        rows = min(total, RING * T)
        acts = torch.randint(0, 5, (rows, B, N), generator=gen, device=dev, dtype=torch.int32)
        outs = env.alloc_outputs(rows, coll=False)
        return acts, outs

    def plan(acts, outs, chunks):
        """pw_step_io structs bound once per ring slot (as a C host would); the timed loop only launches."""
        plans, cache = [], {}
        for i, n in enumerate(chunks):
            s = (i % RING) * T
            if (s, n) not in cache:
                view = {k: v[s:s + n] for k, v in outs.items()}
                cache[(s, n)] = (env.plan_rollout(acts[s:s + n], view), view, acts[s:s + n])
            plans.append(cache[(s, n)])
        return plans

    exchange_state = {'error': None}

    def run(plans, events=None, shard=None):
        # ONE HIP-event pair brackets all launches of the timed region on the launch stream (a pair per
        # launch would put two extra packets between dependent kernels and slow what it measures)
        if events is not None:
            events[0].record()
        for launch, view, a in plans:
            launch()
            if shard is not None and exchange_state['error'] is None:
                try:
                    shard(view, a)
                except Exception as e:  # keep the sharded rollout measurable; the JSON line reports this
                    exchange_state['error'] = repr(e)[:200]
        if events is not None:
            events[1].record()
        if shard is not None and exchange_state['error'] is None:
            try:
                shard.finish()
            except Exception as e:
                exchange_state['error'] = repr(e)[:200]

    shard = None
    if use_dist:
        from multiagent_rl_amd.dist import SampledTransitionGather
        # one exchange per update_rate (100) env-steps, the learner's cadence (rls/arglist.py:18)
        shard = SampledTransitionGather(env, args.batch_size, rank, world, dev, every=max(1, args.exchange_steps // T),
                                        side_stream=not os.environ.get('PW_BENCH_NO_SIDE_STREAM'))

    env.reset()
    if W > 0:
        wa, wo = alloc(W)
        wplans = plan(wa, wo, make_chunks(W))
        run(wplans, shard=shard)
        if shard is not None:
            shard.prime(wplans[0][1], wplans[0][2])
        del wa, wo, wplans
    acts, outs = alloc(K)
    chunks = make_chunks(K)
    plans = plan(acts, outs, chunks)
    events = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))

    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(plans, events, shard)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # dominant kernel: pw_rollout_kernel<spread, local>; HIP events on the launch stream
    # average launch duration = event-bracketed time of all launches / number of launches (includes the
    # inter-launch gaps, so it is an upper bound of the kernel's own duration; rocprofv3 gives that one)
    launch_ms = events[0].elapsed_time(events[1]) / len(chunks)
    steps_per_launch = K / len(chunks)
    bytes_per_launch = env.bytes_per_env_step * B * steps_per_launch
    achieved = bytes_per_launch / (launch_ms * 1e-3) / 1e9

    # HBM traffic per launch: PMC counters cannot be collected from inside this process; use the committed
    # rocprofv3 --pmc summary of this exact workload (profiles/, collected per the MI355X guide) if present.
    traffic, traffic_src = None, None
    try:
        import glob
        for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_summary.json')), reverse=True):
            sm = json.load(open(f))
            cfgw = sm.get('bench', {}).get('config', {}).get('workload', '')
            if 'traffic_bytes_per_launch' in sm and ('N=%d ' % N) in cfgw and ('B=%d ' % B) in cfgw and \
                    ('%d steps per pw_rollout' % T) in cfgw and args.scenario in cfgw:
                traffic, traffic_src = sm['traffic_bytes_per_launch'], os.path.relpath(f, ROOT)
                break
    except Exception:
        pass

    finite = bool(torch.isfinite(outs['obs']).all().item()) and (K < 25 or bool(outs['terminal'][24].all().item()))
    steps_per_launch_f = K / len(chunks)

    # Reported next to the headline ("report policy-in-the-loop separately"): the same env with the
    # reference's actor architecture in the loop -- policy forward + Gumbel sampling + pw_step + replay append +
    # episode bookkeeping per step, captured in one hipGraph.  N = 1 only; never part of `value`.
    policy_line = None
    if world == 1 and rank == 0 and args.scenario == 'simple_spread' and not os.environ.get('PW_BENCH_NO_POLICY'):
        try:
            from multiagent_rl_amd.policy import ActorNetwork, FusedActor
            from multiagent_rl_amd.replay_buffer import ReplayBuffer
            from multiagent_rl_amd.rollout import BatchedRollout
            del outs, acts, plans
            torch.cuda.empty_cache()
            torch.manual_seed(12345678)
            penv = BatchedParticleEnv('simple_spread', B, num_agents=N, max_episode_len=25, auto_reset=True,
                                      seed=12345678)
            ro = BatchedRollout(penv, FusedActor(ActorNetwork(penv.obs_dim, 5).to(dev).eval(), seed=12345678),
                                ReplayBuffer(1e6, N, penv.obs_dim))
            # (a) ONE launch per 100 steps: pw_policy_rollout (policy + sampling + env step resident on the CU, the
            #     transitions written straight into the device replay ring, episode bookkeeping in the same kernel)
            ro.collect_one_launch(100, chunk=100)
            torch.cuda.synchronize()
            tp = time.perf_counter()
            ro.collect_one_launch(1000, chunk=100)
            torch.cuda.synchronize()
            tp = time.perf_counter() - tp
            policy_line = dict(value=B * 1000 / tp, unit='env-steps/s', us_per_step=tp / 1000 * 1e6, steps=1000,
                               policy='FusedActor (reference ActorNetwork: Linear-BiLSTM-Linear, random init) + Gumbel sampling',
                               loop='100-step chunks, one launch each: pw_policy_rollout (actor + sampling + env step + device '
                                    'replay append + episode stats)')
            # (b) the per-step form: actor, env step, replay append + bookkeeping = 3 launches per step in a hipGraph
            ro.capture(2)
            ro.collect(50)
            torch.cuda.synchronize()
            tg = time.perf_counter()
            ro.collect(500)
            torch.cuda.synchronize()
            tg = time.perf_counter() - tg
            policy_line['three_launches_per_step_hipgraph'] = dict(value=B * 500 / tg, us_per_step=tg / 500 * 1e6)
        except Exception as e:  # the headline must not depend on this extra
            policy_line = dict(error=repr(e)[:200])

    if rank == 0:
        value = world * B * K / elapsed
        line = {
            'metric': 'env-steps/sec, simple_spread N=6 x B envs, 1/2/4/8 MI355X',
            'value': value, 'unit': 'env-steps/s', 'n_gpus': world, 'steps': K, 'warmup': W,
            'ms_per_step': elapsed * 1e3 / K, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': '%s N=%d L=%d, B=%d envs per GPU (global %d), local obs D=%d, episode 25 with '
                                   'auto-reset, uniform int32 action indices, %d steps per pw_rollout launch'
                                   % (args.scenario, N, env.num_landmarks, B, world * B, D, T),
                       'global_batch': world * B, 'parallelism': 'env-shard x%d' % world,
                       'exchange': None if shard is None else dict(
                           kind='RCCL all_gather of %d sampled transition rows per rank every %d steps into the root replay ring'
                                % (shard.R, shard.every * T), exchanges=shard.exchanges,
                           rows_ingested_root=shard.rows_ingested, error=exchange_state['error']),
                       'outputs_finite': finite},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBPS, 'traffic': traffic, 'traffic_source': traffic_src,
                         'algorithmic_bytes_per_launch': bytes_per_launch,
                         'kernel': 'pw_spread_duo_kernel<6,6,true>' if (args.scenario == 'simple_spread' and N == 6) else 'pw_rollout', 'launch_ms': launch_ms,
                         'bytes_per_env_step': env.bytes_per_env_step, 'env_steps_per_launch': B * steps_per_launch},
        }
        if world == 1:
            line['cpu_baseline'] = cpu_line
            line['policy_in_loop'] = policy_line
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
