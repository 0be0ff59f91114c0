"""simple_reference (the communication scenario of the reference's sweep, main.py:24):
oracle cross-checks on CPU, HIP parity on the GPU."""
import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import particle_oracle as po


def _py_envs(pos, vel, lm, comm, goal):
    envs = []
    for e in range(pos.shape[0]):
        env = po.make_oracle_env('simple_reference')
        po.set_world_state(env.world, pos[e], vel[e], lm[e])
        for i, ag in enumerate(env.world.agents):
            ag.state.c = comm[e, i].astype(np.float64).copy()
            ag.goal_b = env.world.landmarks[goal[e, i]]
        envs.append(env)
    return envs


def _rand(rng, B):
    return (rng.uniform(-1, 1, (B, 2, 2)).astype(np.float32), rng.uniform(-1, 1, (B, 2, 2)).astype(np.float32),
            rng.uniform(-1, 1, (B, 3, 2)).astype(np.float32), np.eye(10, dtype=np.float32)[rng.randint(0, 10, (B, 2))],
            rng.randint(0, 3, (B, 2)).astype(np.int32))


def test_oracle_surface_and_kat():
    np.random.seed(12345678)
    env = po.make_oracle_env('simple_reference')
    assert env.n == 2 and [s.shape for s in env.observation_space] == [(21,), (21,)]
    sp = env.action_space[0]
    assert hasattr(sp, 'high') and list(sp.high + 1) == [5, 10]          # main.py:52-54 -> dim_action [5, 10]
    po.set_world_state(env.world, [[0, 0], [0.5, 0.5]], np.zeros((2, 2)), [[1, 0], [0, 1], [-1, 0]])
    env.world.agents[0].goal_b, env.world.agents[1].goal_b = env.world.landmarks[0], env.world.landmarks[2]
    a0 = np.concatenate([np.eye(5)[1], np.eye(10)[3]])                    # move +x, say symbol 3
    a1 = np.concatenate([np.eye(5)[0], np.eye(10)[7]])
    obs, rew, done, info = env.step([a0, a1])
    # agent 0 moved to (0.05, 0); agent 1 stayed.  r0 = -|p1 - lm0|^2, r1 = -|p0 - lm2|^2
    np.testing.assert_allclose(rew, [-((0.5 - 1) ** 2 + 0.5 ** 2), -((0.05 + 1) ** 2)], atol=1e-15)
    want0 = np.concatenate([[0.5, 0], [0.95, 0, -0.05, 1, -1.05, 0], [0.75, 0.25, 0.25], np.eye(10)[7]])
    np.testing.assert_allclose(obs[0], want0, atol=1e-15)
    assert obs[1][8:11].tolist() == [0.25, 0.25, 0.75] and obs[1][11:].tolist() == np.eye(10)[3].tolist()
    assert done == [False, False]


def test_c_oracle_matches_python_oracle():
    rng = np.random.RandomState(1)
    B = 12
    pos, vel, lm, comm, goal = _rand(rng, B)
    cfg = co.make_config('simple_reference', 2, max_episode_len=0)
    assert co.obs_dim(cfg) == 21
    o64, o32 = co.CRefOracle(cfg, B, np.float64), co.CRefOracle(cfg, B, np.float32)
    o64.set_state(pos, vel, lm, comm, goal)
    o32.set_state(pos, vel, lm, comm, goal)
    envs = _py_envs(pos, vel, lm, comm, goal)
    for t in range(6):
        ai, ac = rng.randint(0, 5, (B, 2)), rng.randint(0, 10, (B, 2))
        r64, r32 = o64.step(act_idx=ai, act_comm=ac), o32.step(act_idx=ai, act_comm=ac)
        for e, env in enumerate(envs):
            o, rw, d, _ = env.step([np.concatenate([np.eye(5)[ai[e, i]], np.eye(10)[ac[e, i]]]) for i in range(2)])
            np.testing.assert_allclose(np.stack(o), r64['obs'][e], atol=1e-12)
            np.testing.assert_allclose(rw, r64['rew'][e], atol=1e-12)
        np.testing.assert_allclose(r32['obs'], r64['obs'], atol=1e-5)
        np.testing.assert_allclose(r32['rew'], r64['rew'], atol=1e-5)
    # soft vectors: movement part arg-maxed (force_discrete_action), communication part passed through
    soft = rng.uniform(0, 1, (B, 2, 15))
    r = o64.step(act_vec=soft)
    for e, env in enumerate(envs):
        env.force_discrete_action = True
        o, rw, _, _ = env.step([soft[e, i].copy() for i in range(2)])
        np.testing.assert_allclose(np.stack(o), r['obs'][e], atol=1e-12)
    np.testing.assert_allclose(r['obs'][:, 0, 11:], soft[:, 1, 5:], atol=0)


# ------------------------------------------------------------------------------------------------ GPU
gpu = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy()


def _same_bits(got, want, name):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    same = got.view(np.uint32) == want.view(np.uint32) if got.dtype.kind == 'f' else got == want
    assert same.all(), '%s: %d / %d differ' % (name, (~same).sum(), same.size)


@gpu
@pytest.mark.parametrize('B', [1, 33, 1000])
def test_hip_single_step_and_rollout_match_oracle_bitwise(B):
    import torch
    from multiagent_rl_amd import make_batched_env
    rng = np.random.RandomState(B)
    pos, vel, lm, comm, goal = _rand(rng, B)
    env = make_batched_env('simple_reference', B, max_episode_len=0)
    assert env.obs_dim == 21 and env.n == 2 and list(env.action_space[0].high + 1) == [5, 10]
    cfg = co.make_config('simple_reference', 2, max_episode_len=0)
    o32, o64 = co.CRefOracle(cfg, B, np.float32), co.CRefOracle(cfg, B, np.float64)
    for o in (o32, o64):
        o.set_state(pos, vel, lm, comm, goal)
    env.set_state(pos, vel, lm, comm=comm, goal=goal)
    _same_bits(_np(env.observe()), o32.observe(), 'observe')
    ai, ac = rng.randint(0, 5, (B, 2)), rng.randint(0, 10, (B, 2))
    obs, rew, done, info = env.step(torch.from_numpy(np.stack([ai, ac], -1)))
    w, w64 = o32.step(act_idx=ai, act_comm=ac), o64.step(act_idx=ai, act_comm=ac)
    _same_bits(_np(obs), w['obs'], 'obs')
    _same_bits(_np(rew), w['rew'], 'rew')
    _same_bits(_np(info['rew_shared']), (np.float32(0) + w['rew'][:, 0]) + w['rew'][:, 1], 'rew_shared')
    np.testing.assert_allclose(_np(obs), w64['obs'], atol=1e-5)
    np.testing.assert_allclose(_np(rew), w64['rew'], atol=1e-5)
    st = env.get_state()
    _same_bits(_np(st['pos']), o32.pos, 'pos')
    _same_bits(_np(st['comm']), o32.comm, 'comm')
    assert np.array_equal(_np(st['goal']), o32.goal) and not _np(done).any()
    # soft action vectors (run.py:39-41 concatenation), movement arg-maxed
    soft = rng.uniform(0, 1, (B, 2, 15)).astype(np.float32)
    obs2, rew2, _, _ = env.step(torch.from_numpy(soft))
    w2 = o32.step(act_vec=soft)
    _same_bits(_np(obs2), w2['obs'], 'obs(vec)')
    _same_bits(_np(rew2), w2['rew'], 'rew(vec)')


@gpu
def test_hip_rollout_with_auto_reset_matches_oracle_bitwise():
    import torch
    from multiagent_rl_amd import make_batched_env
    B, T = 257, 58
    env = make_batched_env('simple_reference', B, max_episode_len=25, auto_reset=True, seed=77, env_id_base=1 << 34)
    cfg = co.make_config('simple_reference', 2, max_episode_len=25, auto_reset=True, seed=77, env_id_base=1 << 34)
    o32 = co.CRefOracle(cfg, B, np.float32)
    _same_bits(_np(env.reset()), o32.reset(), 'reset')
    assert len(np.unique(o32.goal)) == 3
    rng = np.random.RandomState(4)
    acts = np.stack([rng.randint(0, 5, (T, B, 2)), rng.randint(0, 10, (T, B, 2))], -1).astype(np.int32)
    out = env.rollout(torch.from_numpy(acts))
    for t in range(T):
        w = o32.step(act_idx=acts[t, :, :, 0], act_comm=acts[t, :, :, 1])
        _same_bits(_np(out['obs'][t]), w['obs'], 'obs[%d]' % t)
        _same_bits(_np(out['rew'][t]), w['rew'], 'rew[%d]' % t)
        _same_bits(_np(out['terminal'][t]).astype(np.uint8), w['terminal'], 'terminal[%d]' % t)
        if w['terminal'].any():
            _same_bits(_np(out['final_obs'][t]), w['final_obs'], 'final_obs[%d]' % t)
    st = env.get_state()
    _same_bits(_np(st['pos']), o32.pos, 'pos')
    assert np.array_equal(_np(st['goal']), o32.goal) and np.array_equal(_np(st['ep_count']).astype(np.uint32), o32.ep_count)


@gpu
def test_multiagentenv_dropin_tracks_python_oracle():
    """make_env('simple_reference'): same NumPy seed -> same goals and initial state as the oracle env
    (np.random.choice x2 before the positions), MultiDiscrete action surface, 15-vector actions."""
    from multiagent_rl_amd import make_env
    np.random.seed(5)
    gpu_env = make_env('simple_reference')
    np.random.seed(5)
    ref = po.make_oracle_env('simple_reference')
    ref.force_discrete_action = True
    np.random.seed(6); o_gpu = gpu_env.reset()
    np.random.seed(6); o_ref = ref.reset()
    assert hasattr(gpu_env.action_space[0], 'high') and list(gpu_env.action_space[0].high + 1) == [5, 10]
    assert gpu_env.observation_space[0].shape == (21,)
    rng = np.random.RandomState(0)
    for t in range(25):
        for a, b in zip(o_gpu, o_ref):
            np.testing.assert_allclose(a, b, atol=1e-5)
        acts = [np.concatenate([np.eye(5)[rng.randint(5)], np.eye(10)[rng.randint(10)]]) for _ in range(2)]
        o_gpu, r_gpu, d_gpu, _ = gpu_env.step([a.copy() for a in acts])
        o_ref, r_ref, d_ref, _ = ref.step([a.copy() for a in acts])
        np.testing.assert_allclose(r_gpu, r_ref, atol=1e-5)
        assert d_gpu == d_ref == [False, False]


@gpu
def test_run_loop_multidiscrete_on_hip_env(tmp_path):
    """rollout.run (MultiDiscrete branch) on the HIP MultiAgentEnv: the golden reference trace's shape."""
    import json
    import os
    from multiagent_rl_amd import make_env, rollout
    from tests.trace_util import RecordingEnv, RecordingMemory, StubTrainer
    gold = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'run_multidiscrete_trace.json')))

    class Args(object):
        is_training, display = True, False
    for k, v in gold['arglist'].items():
        setattr(Args, k, v)
    np.random.seed(12345680)
    env = RecordingEnv(make_env('simple_reference'))
    np.random.seed(12345680)
    StubTrainer.trace = env.trace
    hist = rollout.run(env, None, None, StubTrainer, 'simple_reference', 'MultiDiscrete', cnt=2, arglist=Args,
                       memory=RecordingMemory(), out_dir=str(tmp_path), log=lambda *a: None)

    def skeleton(x):
        if isinstance(x, dict):
            return {k: skeleton(v) for k, v in x.items() if k not in ('sum', 'value')}
        if isinstance(x, list):
            return [skeleton(v) for v in x]
        return x
    assert skeleton(json.loads(json.dumps(env.trace))) == skeleton(gold['trace'])
    np.testing.assert_allclose(hist['reward_episodes'], gold['reward_episodes'], rtol=1e-4, atol=1e-3)


@gpu
def test_batched_two_head_policy_rollout():
    import torch
    from multiagent_rl_amd import make_batched_env
    from multiagent_rl_amd.policy import ActorNetwork, GumbelPolicy
    from multiagent_rl_amd.rollout import BatchedRollout
    torch.manual_seed(0)
    env = make_batched_env('simple_reference', 512, auto_reset=True, max_episode_len=25)
    actor = ActorNetwork(env.obs_dim, [5, 10]).cuda()
    assert sorted(k for k in actor.state_dict() if k.startswith('dense2')) == [
        'dense2_1.module.bias', 'dense2_1.module.weight', 'dense2_2.module.bias', 'dense2_2.module.weight']
    ro = BatchedRollout(env, GumbelPolicy(actor), memory=None)
    ro.collect(26)
    st = ro.stats()
    assert st['episodes'] == 512 and st['mean_episode_reward'] < 0
    a = GumbelPolicy(actor)(ro.obs)
    assert tuple(a.shape) == (512, 2, 2) and int(a[..., 0].max()) <= 4 and int(a[..., 1].max()) <= 9
    # what agent 0 hears is what agent 1 said in the last step
    obs, _, _, _ = env.step(a)
    heard = obs[:, 0, 11:].argmax(-1).cpu()
    assert torch.equal(heard, a[:, 1, 1].long().cpu()) and (obs[:, 0, 11:].sum(-1) == 1).all()


@gpu
def test_one_launch_two_head_actor_matches_pytorch_and_drives_the_rollout():
    """FusedActor on a MultiDiscrete actor ([5, 10] heads, main.py:52-54): logits of both heads within 2e-5 of plain
    PyTorch fp32, sampled (movement, symbol) pairs follow softmax of each head, and the batched simple_reference
    rollout runs with it (3 launches per step, no MIOpen RNN path)."""
    import torch
    from multiagent_rl_amd import make_batched_env
    from multiagent_rl_amd.policy import ActorNetwork, FusedActor
    from multiagent_rl_amd.rollout import BatchedRollout
    torch.manual_seed(2)
    ref = ActorNetwork(21, [5, 10])
    actor = ActorNetwork(21, [5, 10]).cuda().eval()
    actor.load_state_dict(ref.state_dict())
    fused = FusedActor(actor, seed=11)
    assert fused.heads == (5, 10) and fused.use_fused
    for B, N in [(1, 2), (257, 2), (33, 5)]:
        obs = torch.randn(B, N, 21) * 2
        with torch.no_grad():
            want = ref(obs)
        got = fused.logits(obs.cuda())
        for g, w in zip(got, want):
            np.testing.assert_allclose(g.cpu().numpy(), w.numpy(), rtol=0, atol=2e-5)
    # sampling: same two rows, 20000 independent draws per head
    obs = torch.randn(1, 2, 21).cuda().repeat(20000, 1, 1)
    p = [torch.softmax(x, -1)[0].cpu().numpy() for x in fused.logits(obs[:1])]
    a = fused(obs)
    assert tuple(a.shape) == (20000, 2, 2) and a.dtype == torch.int32
    a = a.cpu().numpy()
    for ag in range(2):
        for hd, n in ((0, 5), (1, 10)):
            freq = np.bincount(a[:, ag, hd], minlength=n) / 20000.0
            assert np.abs(freq - p[hd][ag]).max() < 0.015, (ag, hd, freq, p[hd][ag])
    # the two heads draw independent noise: the joint frequency factorises
    joint = np.zeros((5, 10))
    np.add.at(joint, (a[:, 0, 0], a[:, 0, 1]), 1.0 / 20000)
    assert np.abs(joint - np.outer(p[0][0], p[1][0])).max() < 0.01
    assert np.array_equal(FusedActor(actor, seed=11)(obs).cpu().numpy(), a)      # same seed + call index => same actions
    # in the loop
    env = make_batched_env('simple_reference', 512, auto_reset=True, max_episode_len=25)
    ro = BatchedRollout(env, FusedActor(actor, seed=3), memory=None)
    ro.collect(26)
    st = ro.stats()
    assert st['episodes'] == 512 and st['mean_episode_reward'] < 0
    obs, _, _, _ = env.step(ro.policy(ro.obs))
    assert (obs[:, 0, 11:].sum(-1) == 1).all()        # agent 0 hears exactly one symbol from agent 1
