#!/usr/bin/env python3
"""Entry script of the batched engine -- the counterpart of the reference's ``main.py:29-67``.

    python examples/train_batched.py --scenario simple_spread --envs 4096 --episodes 40960
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_batched.py --envs 4096
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 examples/train_batched.py --backend gloo   (one GPU)

For every scenario and seed count ``cnt`` it does what main.py does -- build the env (``make_batched_env``: the reference's
``make_env`` contract, B worlds), seed protocol ``seed = cnt + 12345678`` (main.py:41-49), read the dims from the env
(:51-58), build ``ActorNetwork`` / ``CriticNetwork`` (:60-61), hand everything to the rollout-and-learn loop (:65-67) -- with
``multiagent_rl_amd.train.train_batched`` in the place of ``experiments.run.run``.  ``--reference`` takes the Trainer and the
critic from the reference checkout on ``sys.path`` (``rls.agent.multiagent.ddpg_gumbel_fix``); otherwise the small stock-PyTorch
learner of ``examples/madr_learner.py`` stands in (the learner is not part of this repo's scope).
With several ranks (torchrun) every rank rolls out its shard of the env batch (``env_id_base = rank * envs``) and the
transitions of all ranks reach rank 0's ring through the RCCL full gather; rank 0 learns and broadcasts the actor.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--scenario', action='append', help='simple_spread | simple_tag | simple_reference (repeatable)')
    ap.add_argument('--envs', type=int, default=4096, help='B per GPU')
    ap.add_argument('--agents', type=int, default=None, help='simple_spread: make_world(num_agents=n), scenarios.py:170')
    ap.add_argument('--episodes', type=int, default=None, help='arglist.num_episodes (finished episodes over all envs of a rank)')
    ap.add_argument('--seeds', type=int, default=1, help='cnt in range(seeds), main.py:37')
    ap.add_argument('--chunk', type=int, default=100, help='batched env steps per launch')
    ap.add_argument('--max-updates-per-chunk', type=int, default=8)
    ap.add_argument('--save-rate', type=int, default=None)
    ap.add_argument('--out-dir', default='Models')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help='process group of a multi-rank run: nccl (= RCCL, one GPU per rank) or gloo (the blocks of the full gather '
                         'travel through pinned host buffers; ranks may share a GPU -- RCCL refuses that)')
    ap.add_argument('--reference', action='store_true', help="use the reference's Trainer / CriticNetwork (rls on sys.path)")
    args = ap.parse_args(argv)

    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC for multi-process GPU work (read at the first GPU call)
    import torch
    import torch.distributed as dist
    from multiagent_rl_amd import arglist, make_batched_env
    from multiagent_rl_amd.policy import ActorNetwork
    from multiagent_rl_amd.train import dims_from_env, seed_everything, train_batched
    if args.reference:
        from rls.agent.multiagent.ddpg_gumbel_fix import Trainer
        from rls.model.ac_network_multi_gumbel import CriticNetwork
    else:
        from madr_learner import CriticNetwork, Trainer

    rank, world = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.backend == 'gloo':
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29541')
        if args.backend == 'gloo':
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    if args.episodes is not None:
        arglist.num_episodes = args.episodes
    if args.save_rate is not None:
        arglist.save_rate = args.save_rate
    results = []
    for scenario_name in (args.scenario or ['simple_spread']):
        for cnt in range(args.seeds):
            seed = seed_everything(cnt + 12345678)                             # main.py:41-49
            kw = dict(n=args.agents) if scenario_name == 'simple_spread' and args.agents else {}
            if scenario_name == 'simple_tag':
                kw = dict(num_adversaries=3, num_good=1) if args.agents is None else dict(num_adversaries=args.agents - 2, num_good=2)
            env = make_batched_env(scenario_name, args.envs, auto_reset=True, max_episode_len=arglist.max_episode_len,
                                   seed=seed, env_id_base=rank * args.envs, **kw)
            dim_obs, dim_action, action_type = dims_from_env(env)             # main.py:51-58
            actor = ActorNetwork(input_dim=dim_obs, out_dim=dim_action)       # main.py:60-61
            n_act = sum(dim_action) if isinstance(dim_action, list) else dim_action
            critic = CriticNetwork(input_dim=dim_obs + n_act, out_dim=1)
            gather = None
            if world > 1:
                from multiagent_rl_amd.dist import FullTransitionGather, broadcast_actor
                actor = actor.to(dev)
                broadcast_actor(actor, src=0)                                  # same initial weights on every rank
                # state-only blocks (simple_spread, simple_tag) land in a STATE ring on the learner rank (rows rebuilt when a batch is
                # sampled: a third of the ring writes); simple_reference travels on compact-row blocks into its two-head row ring
                state_ring = scenario_name in ('simple_spread', 'simple_tag')
                gather = FullTransitionGather(env, args.chunk, rank, world, dev, ring='state' if state_ring else 'rows')
                gather.prime()
            hist = train_batched(env, actor, critic, Trainer, scenario_name, action_type, cnt=cnt, out_dir=args.out_dir,
                                 chunk=args.chunk, max_updates_per_chunk=args.max_updates_per_chunk, gather=gather,
                                 ring='state' if (gather is None and scenario_name in ('simple_spread', 'simple_tag')) else 'rows',
                                 rank=rank, world=world, log=print if rank == 0 else (lambda *a: None))
            results.append((scenario_name, cnt, hist['stats']))
            if rank == 0:
                s = hist['stats']
                print('%s cnt=%d: %d env-steps, %d episodes, mean episode reward %.3f, %d updates, %.1f s (%.3g env-steps/s)'
                      % (scenario_name, cnt, s['env_steps'], s['episodes'], s['mean_episode_reward'], s['updates'], s['wall_s'],
                         s['env_steps'] * world / s['wall_s']))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return results


if __name__ == '__main__':
    main()
