"""simple_speaker_listener (main.py:24 lists it) with canonical MPE semantics: a fixed
speaker with a Discrete(3) action, a silent moving listener with Discrete(5), the 11-number observation the
reference patches in (experiments/scenarios.py:45-64).  Oracle cross-checks on CPU, HIP parity on the GPU."""
import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import particle_oracle as po


def _py_envs(pos, vel, lm, goal):
    envs = []
    for e in range(pos.shape[0]):
        env = po.make_oracle_env('simple_speaker_listener')
        po.set_world_state(env.world, pos[e], vel[e], lm[e])
        env.world.agents[0].goal_b = env.world.landmarks[goal[e, 0]]
        envs.append(env)
    return envs


def _rand(rng, B):
    vel = rng.uniform(-1, 1, (B, 2, 2)).astype(np.float32)
    vel[:, 0] = 0                                                # the speaker never moves
    goal = np.zeros((B, 2), np.int32)
    goal[:, 0] = rng.randint(0, 3, B)
    return (rng.uniform(-1, 1, (B, 2, 2)).astype(np.float32), vel, rng.uniform(-1, 1, (B, 3, 2)).astype(np.float32),
            np.zeros((B, 2, 3), np.float32), goal)


def test_oracle_surface_and_kat():
    np.random.seed(12345678)
    env = po.make_oracle_env('simple_speaker_listener')
    assert env.n == 2 and [s.shape for s in env.observation_space] == [(11,), (11,)]
    assert [s.n for s in env.action_space] == [3, 5] and not hasattr(env.action_space[0], 'high')   # main.py:56: Discrete
    po.set_world_state(env.world, [[0.25, -0.5], [0.5, 0.5]], np.zeros((2, 2)), [[1, 0], [0, 1], [-1, 0]])
    env.world.agents[0].goal_b = env.world.landmarks[1]
    obs, rew, done, info = env.step([np.eye(3)[2], np.eye(5)[4]])         # say symbol 2; listener moves -y
    # listener: v = (0, -5 * 0.1) = (0, -0.5), p = (0.5, 0.45); speaker untouched.  both rewards = -|p_l - lm1|^2
    np.testing.assert_allclose(rew, [-(0.5 ** 2 + 0.55 ** 2)] * 2, atol=1e-15)
    np.testing.assert_allclose(obs[0], [0, 0, 0.75, 0.5, -0.25, 1.5, -1.25, 0.5, 0.15, 0.65, 0.15], atol=1e-15)
    np.testing.assert_allclose(obs[1], [0, -0.5, 0.5, -0.45, -0.5, 0.55, -1.5, -0.45, 0, 0, 0], atol=1e-15)
    assert env.world.agents[0].state.c.tolist() == [0, 0, 1] and env.world.agents[1].state.c.tolist() == [0, 0, 0]
    assert done == [False, False]
    # upstream's own observation (local_observation=False): speaker sees only the goal colour, listener hears
    full = po.make_oracle_env('simple_speaker_listener', local_observation=False)
    assert [s.shape for s in full.observation_space] == [(3,), (11,)]


def test_reset_draw_order_on_the_global_numpy_stream():
    """reset_world: ONE np.random.choice (the speaker's goal) before the positions."""
    env = po.make_oracle_env('simple_speaker_listener')
    np.random.seed(9)
    env.reset()
    np.random.seed(9)
    g = np.random.choice(3)
    p = [np.random.uniform(-1, 1, 2) for _ in range(5)]
    assert env.world.agents[0].goal_b is env.world.landmarks[g] and env.world.agents[1].goal_b is None
    np.testing.assert_array_equal(env.world.agents[1].state.p_pos, p[1])
    np.testing.assert_array_equal(env.world.landmarks[2].state.p_pos, p[4])


def test_c_oracle_matches_python_oracle():
    rng = np.random.RandomState(1)
    B = 12
    pos, vel, lm, comm, goal = _rand(rng, B)
    cfg = co.make_config('simple_speaker_listener', max_episode_len=0)
    assert co.obs_dim(cfg) == 11
    o64, o32 = co.CRefOracle(cfg, B, np.float64), co.CRefOracle(cfg, B, np.float32)
    o64.set_state(pos, vel, lm, comm, goal)
    o32.set_state(pos, vel, lm, comm, goal)
    envs = _py_envs(pos, vel, lm, goal)
    for t in range(6):
        ai = np.stack([rng.randint(0, 3, B), rng.randint(0, 5, B)], 1)
        r64, r32 = o64.step(act_idx=ai), o32.step(act_idx=ai)
        for e, env in enumerate(envs):
            o, rw, d, _ = env.step([np.eye(3)[ai[e, 0]], np.eye(5)[ai[e, 1]]])
            np.testing.assert_allclose(np.stack(o), r64['obs'][e], atol=1e-12)
            np.testing.assert_allclose(rw, r64['rew'][e], atol=1e-12)
            np.testing.assert_array_equal(np.stack([a.state.c for a in env.world.agents]), o64.comm[e])
        np.testing.assert_allclose(r32['obs'], r64['obs'], atol=1e-5)
        np.testing.assert_allclose(r32['rew'], r64['rew'], atol=1e-5)
        np.testing.assert_array_equal(o64.pos[:, 0], pos[:, 0].astype(np.float64))      # the speaker stays put
    # soft vectors: listener movement arg-maxed (force_discrete_action), speaker vector passed through as state.c
    soft = rng.uniform(0, 1, (B, 2, 5))
    r = o64.step(act_vec=soft)
    for e, env in enumerate(envs):
        o, rw, _, _ = env.step([soft[e, 0, :3].copy(), soft[e, 1].copy()])
        np.testing.assert_allclose(np.stack(o), r['obs'][e], atol=1e-12)
    np.testing.assert_array_equal(o64.comm[:, 0], soft[:, 0, :3])
    assert not o64.comm[:, 1].any()


def test_c_oracle_philox_reset_and_auto_reset():
    cfg = co.make_config('simple_speaker_listener', max_episode_len=3, auto_reset=True, seed=5)
    o = co.CRefOracle(cfg, 64, np.float32)
    obs0 = o.reset()
    assert len(np.unique(o.goal[:, 0])) == 3 and not o.goal[:, 1].any()
    assert (np.abs(o.pos) <= 1).all() and not o.vel.any()
    col = obs0[:, 0, 8:]
    assert np.array_equal(col.argmax(-1), o.goal[:, 0]) and np.allclose(np.sort(col, -1), [0.15, 0.15, 0.65])
    assert not obs0[:, 1, 8:].any()
    p_before = o.pos.copy()
    for t in range(3):
        w = o.step(act_idx=np.tile([[1, 1]], (64, 1)))
    assert w['terminal'].all() and (o.ep_count == 2).all() and not np.array_equal(p_before, o.pos)
    assert not np.array_equal(w['final_obs'], w['obs'])


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
def test_hip_single_step_matches_oracle_bitwise(B):
    import torch
    from multiagent_rl_amd import make_batched_env
    rng = np.random.RandomState(B)
    pos, vel, lm, comm, goal = _rand(rng, B)
    env = make_batched_env('simple_speaker_listener', B, max_episode_len=0)
    assert env.obs_dim == 11 and env.n == 2 and [s.n for s in env.action_space] == [3, 5]
    cfg = co.make_config('simple_speaker_listener', max_episode_len=0)
    o32, o64 = co.CRefOracle(cfg, B, np.float32), co.CRefOracle(cfg, B, np.float64)
    for o in (o32, o64):
        o.set_state(pos, vel, lm, comm, goal)
    env.set_state(pos, vel, lm, comm=comm, goal=goal)
    _same_bits(_np(env.observe()), o32.observe(), 'observe')
    ai = np.stack([rng.randint(0, 3, B), rng.randint(0, 5, B)], 1)
    obs, rew, done, info = env.step(torch.from_numpy(ai))
    w, w64 = o32.step(act_idx=ai), o64.step(act_idx=ai)
    _same_bits(_np(obs), w['obs'], 'obs')
    _same_bits(_np(rew), w['rew'], 'rew')
    _same_bits(_np(info['rew_shared']), (np.float32(0) + w['rew'][:, 0]) + w['rew'][:, 1], 'rew_shared')
    np.testing.assert_allclose(_np(obs), w64['obs'], atol=1e-5)
    np.testing.assert_allclose(_np(rew), w64['rew'], atol=1e-5)
    st = env.get_state()
    _same_bits(_np(st['pos']), o32.pos, 'pos')
    _same_bits(_np(st['vel']), o32.vel, 'vel')
    _same_bits(_np(st['comm']), o32.comm, 'comm')
    assert tuple(st['comm'].shape) == (B, 2, 3) and np.array_equal(_np(st['goal']), o32.goal) and not _np(done).any()
    assert np.array_equal(_np(st['comm'])[:, 0].argmax(-1), ai[:, 0]) and not _np(st['comm'])[:, 1].any()
    # soft action vectors [B, 2, 5]: speaker's first three entries become state.c, listener's row is arg-maxed
    soft = rng.uniform(0, 1, (B, 2, 5)).astype(np.float32)
    obs2, rew2, _, _ = env.step(torch.from_numpy(soft))
    w2 = o32.step(act_vec=soft)
    _same_bits(_np(obs2), w2['obs'], 'obs(vec)')
    _same_bits(_np(rew2), w2['rew'], 'rew(vec)')
    _same_bits(_np(env.get_state()['comm']), o32.comm, 'comm(vec)')
    _same_bits(_np(env.reward()[0]), w2['rew'], 'pw_reward')


@gpu
def test_hip_rollout_with_auto_reset_matches_oracle_bitwise():
    import torch
    from multiagent_rl_amd import make_batched_env
    B, T = 257, 58
    env = make_batched_env('simple_speaker_listener', B, max_episode_len=25, auto_reset=True, seed=77, env_id_base=1 << 34)
    cfg = co.make_config('simple_speaker_listener', max_episode_len=25, auto_reset=True, seed=77, env_id_base=1 << 34)
    o32 = co.CRefOracle(cfg, B, np.float32)
    _same_bits(_np(env.reset()), o32.reset(), 'reset')
    assert len(np.unique(o32.goal[:, 0])) == 3
    rng = np.random.RandomState(4)
    acts = np.stack([rng.randint(0, 3, (T, B)), rng.randint(0, 5, (T, B))], -1).astype(np.int32)
    out = env.rollout(torch.from_numpy(acts))
    for t in range(T):
        w = o32.step(act_idx=acts[t])
        _same_bits(_np(out['obs'][t]), w['obs'], 'obs[%d]' % t)
        _same_bits(_np(out['rew'][t]), w['rew'], 'rew[%d]' % t)
        _same_bits(_np(out['terminal'][t]).astype(np.uint8), w['terminal'], 'terminal[%d]' % t)
        if w['terminal'].any():
            _same_bits(_np(out['final_obs'][t]), w['final_obs'], 'final_obs[%d]' % t)
    st = env.get_state()
    _same_bits(_np(st['pos']), o32.pos, 'pos')
    _same_bits(_np(st['comm']), o32.comm, 'comm')
    assert np.array_equal(_np(st['goal']), o32.goal) and np.array_equal(_np(st['ep_count']).astype(np.uint32), o32.ep_count)


@gpu
def test_multiagentenv_dropin_tracks_python_oracle():
    """make_env('simple_speaker_listener'): same NumPy seed -> same goal and initial state as the oracle env
    (one np.random.choice before the positions); per-agent Discrete(3) / Discrete(5) action surface."""
    from multiagent_rl_amd import make_env
    np.random.seed(5)
    gpu_env = make_env('simple_speaker_listener')
    np.random.seed(5)
    ref = po.make_oracle_env('simple_speaker_listener')
    np.random.seed(6); o_gpu = gpu_env.reset()
    np.random.seed(6); o_ref = ref.reset()
    assert [s.n for s in gpu_env.action_space] == [3, 5] and [s.shape for s in gpu_env.observation_space] == [(11,), (11,)]
    rng = np.random.RandomState(0)
    for t in range(25):
        for a, b in zip(o_gpu, o_ref):
            np.testing.assert_allclose(a, b, atol=1e-5)
        acts = [np.eye(3)[rng.randint(3)], np.eye(5)[rng.randint(5)]]
        o_gpu, r_gpu, d_gpu, _ = gpu_env.step([a.copy() for a in acts])
        o_ref, r_ref, d_ref, _ = ref.step([a.copy() for a in acts])
        np.testing.assert_allclose(r_gpu, r_ref, atol=1e-5)
        assert d_gpu == d_ref == [False, False]
    with pytest.raises(NotImplementedError):
        make_env('simple_speaker_listener', local_observation=False)
    # the reference's run() feeds every agent a dim_action = action_space[0].n = 3 vector (main.py:56); upstream
    # MPE then fails in _set_action on the listener (action[0][3]); the drop-in refuses the same call
    with pytest.raises(ValueError):
        gpu_env.step([np.eye(3)[0], np.eye(3)[1]])


@gpu
def test_batched_rollout_with_per_agent_action_counts():
    from multiagent_rl_amd import make_batched_env
    from multiagent_rl_amd.policy import UniformRandomPolicy
    from multiagent_rl_amd.rollout import BatchedRollout
    env = make_batched_env('simple_speaker_listener', 512, auto_reset=True, max_episode_len=25)
    ro = BatchedRollout(env, UniformRandomPolicy([3, 5]), memory=None)
    ro.collect(51)
    st = ro.stats()
    assert st['episodes'] == 2 * 512 and st['mean_episode_reward'] < 0
    a = ro.policy(ro.obs)
    assert tuple(a.shape) == (512, 2) and int(a[:, 0].max()) <= 2 and int(a[:, 1].max()) <= 4


@gpu
def test_act_comm_is_rejected_and_listener_reaches_goal_with_greedy_actions():
    """A hand-written listener policy that reads the goal from the STATE (the patched observation hides the
    spoken symbol) drives the shared reward towards 0: the dynamics are wired the right way round."""
    import torch
    from multiagent_rl_amd import make_batched_env
    B = 256
    env = make_batched_env('simple_speaker_listener', B, max_episode_len=0, seed=3)
    obs = env.reset()
    goal = env.get_state()['goal'][:, 0].long()
    first = None
    for t in range(60):
        rel = obs[:, 1, 2:8].reshape(B, 3, 2)[torch.arange(B), goal]          # goal landmark - listener position
        move = torch.where(rel[:, 0].abs() > rel[:, 1].abs(), torch.where(rel[:, 0] > 0, 1, 2), torch.where(rel[:, 1] > 0, 3, 4))
        act = torch.stack([goal.int(), move.int()], 1)
        obs, rew, _, info = env.step(act)
        first = rew.mean().item() if first is None else first
    assert rew.mean().item() > -0.05 > first and torch.equal(rew[:, 0], rew[:, 1])


# ------------------------------------------------------------------------------------------------ properties (CPU)
def test_c_oracle_invariants_over_random_states_and_actions():
    """Size-independent properties over many random worlds: the speaker never moves, the listener's state follows the
    damped-Euler recurrence of its own action only, state.c is exactly the speaker's action / zeros for the listener,
    both rewards are -|p_listener - p_goal|^2, float32 stays within 1e-5 of float64 per step."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=25, deadline=None)
    @given(st.integers(0, 2 ** 31 - 1))
    def prop(seed):
        rng = np.random.RandomState(seed)
        B = 17
        pos, vel, lm, comm, goal = _rand(rng, B)
        cfg = co.make_config('simple_speaker_listener', max_episode_len=0)
        o64, o32 = co.CRefOracle(cfg, B, np.float64), co.CRefOracle(cfg, B, np.float32)
        for o in (o64, o32):
            o.set_state(pos, vel, lm, comm, goal)
        for t in range(5):
            ai = np.stack([rng.randint(0, 3, B), rng.randint(0, 5, B)], 1)
            p_before, v_before = o64.pos.copy(), o64.vel.copy()
            w64, w32 = o64.step(act_idx=ai), o32.step(act_idx=ai)
            assert np.array_equal(o64.pos[:, 0], p_before[:, 0]) and not o64.vel[:, 0].any()
            u = np.stack([(ai[:, 1] == 1).astype(float) - (ai[:, 1] == 2), (ai[:, 1] == 3).astype(float) - (ai[:, 1] == 4)], 1) * 5.0
            v = v_before[:, 1] * 0.75 + u / 1.0 * 0.1
            np.testing.assert_allclose(o64.vel[:, 1], v, rtol=0, atol=1e-15)
            np.testing.assert_allclose(o64.pos[:, 1], p_before[:, 1] + v * 0.1, rtol=0, atol=1e-15)
            assert np.array_equal(o64.comm[:, 0], np.eye(3)[ai[:, 0]]) and not o64.comm[:, 1].any()
            d = o64.pos[:, 1] - lm.astype(np.float64)[np.arange(B), goal[:, 0]]
            np.testing.assert_allclose(w64['rew'][:, 0], -(d ** 2).sum(-1), rtol=0, atol=1e-14)
            assert np.array_equal(w64['rew'][:, 0], w64['rew'][:, 1])
            np.testing.assert_allclose(w32['obs'], w64['obs'], rtol=0, atol=1e-5)
            np.testing.assert_allclose(w32['rew'], w64['rew'], rtol=0, atol=2e-5)
    prop()
