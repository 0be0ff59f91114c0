"""Property tests of the CPU oracle (hypothesis): invariants any faithful restatement of the upstream
physics must satisfy, checked on the scalar NumPy form and on the C forms."""
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import c_oracle as co
from oracle import particle_oracle as po

coord = st.floats(-1.5, 1.5, allow_nan=False, width=32)


@settings(max_examples=200, deadline=None)
@given(coord, coord, coord, coord, st.sampled_from([0.05, 0.075, 0.15, 0.2]), st.sampled_from([0.05, 0.15]))
def test_collision_force_is_antisymmetric_and_repulsive(ax, ay, bx, by, sa, sb):
    w = po.World()
    a, b = po.Agent(), po.Agent()
    a.size, b.size = sa, sb
    a.state.p_pos, b.state.p_pos = np.array([ax, ay]), np.array([bx, by])
    if (ax, ay) == (bx, by):
        return
    fa, fb = w.get_collision_force(a, b)
    assert np.array_equal(fa, -fb)
    delta = a.state.p_pos - b.state.p_pos
    assert np.dot(fa, delta) >= 0                       # pushes a away from b
    dist = np.linalg.norm(delta)
    if dist > sa + sb + 0.75:                           # far apart: softplus underflows to exactly 0
        assert not fa.any()
    # swapping the arguments gives the mirrored pair
    ga, gb = w.get_collision_force(b, a)
    assert np.array_equal(ga, fb) and np.array_equal(gb, fa)


@settings(max_examples=50, deadline=None)
@given(st.integers(1, 7), st.integers(0, 5), st.integers(0, 2 ** 31 - 1))
def test_non_colliding_entities_get_no_force_and_obs_layout(n, l, seed):
    rng = np.random.RandomState(seed)
    env = po.make_oracle_env('simple_spread', n=n, num_landmarks=l)
    pos, vel, lm = rng.uniform(-1, 1, (n, 2)), rng.uniform(-1, 1, (n, 2)), rng.uniform(-1, 1, (l, 2))
    po.set_world_state(env.world, pos, vel, lm)
    for ag in env.world.agents:
        ag.collide = False                               # upstream: either collide False -> [None, None]
    obs, rew, done, _ = env.step([po.onehot(0)] * n)
    p2, v2, lm2 = po.get_world_state(env.world)
    np.testing.assert_allclose(v2, 0.75 * vel, atol=1e-15)           # only damping acted
    np.testing.assert_allclose(p2, pos + 0.1 * 0.75 * vel, atol=1e-15)
    assert np.array_equal(lm2, lm) and done == [False] * n
    for i in range(n):                                               # experiments/scenarios.py:6-20
        assert obs[i].shape == (4 + 2 * l,)
        np.testing.assert_array_equal(obs[i], np.concatenate([v2[i], p2[i]] + [q - p2[i] for q in lm2]))


@settings(max_examples=30, deadline=None)
@given(st.integers(1, 12), st.integers(0, 2 ** 31 - 1))
def test_c_oracle_momentum_and_determinism(n, seed):
    rng = np.random.RandomState(seed)
    cfg = co.make_config('simple_spread', n, max_episode_len=0)
    pos, lm = rng.uniform(-0.5, 0.5, (8, n, 2)), rng.uniform(-1, 1, (8, n, 2))
    a, b = co.COracle(cfg, 8, np.float64), co.COracle(cfg, 8, np.float64)
    for o in (a, b):
        o.set_state(pos, 0, lm)
        o.step(act_idx=np.zeros((8, n), np.int32))
    assert np.array_equal(a.pos, b.pos)                               # deterministic
    np.testing.assert_allclose(a.vel.sum(axis=1), 0, atol=1e-10)      # contact forces cancel pairwise
    f32 = co.COracle(cfg, 8, np.float32)
    f32.set_state(pos, 0, lm)
    f32.step(act_idx=np.zeros((8, n), np.int32))
    np.testing.assert_allclose(f32.pos, a.pos, atol=2e-5)


def test_rows_rebuilt_from_the_state_equal_the_oracles_observations():
    """The row-from-state restatements the gloo tests and the wire stand-ins use (tests/dist_standins.py rows_from_state / ref_rows: what
    pw_replay_add_state_wire, pw_replay_gather on a STATE ring and pw_replay_add_ref_wire rebuild on the GPU) against the scalar oracle's own
    observation functions on random worlds, float32 bit for bit: simple_spread (local), simple_tag (ragged rows zero-padded), simple_reference."""
    import numpy as np
    import torch
    from oracle import particle_oracle as po
    from tests.dist_standins import ref_rows, rows_from_state
    rng = np.random.RandomState(7)

    def f32_rows(env):
        return [np.asarray(env.observation_callback(a, env.world), np.float64) for a in env.world.agents]

    def set_random(world):
        for a in world.agents:
            a.state.p_pos = rng.uniform(-1, 1, 2).astype(np.float32).astype(np.float64)     # float32-representable: the rebuild is float32
            a.state.p_vel = rng.uniform(-1, 1, 2).astype(np.float32).astype(np.float64)
        for lm in world.landmarks:
            lm.state.p_pos = rng.uniform(-1, 1, 2).astype(np.float32).astype(np.float64)

    def state_of(world):
        st = torch.tensor(np.stack([np.concatenate([a.state.p_vel, a.state.p_pos]) for a in world.agents]), dtype=torch.float32)
        lm = torch.tensor(np.stack([lmk.state.p_pos for lmk in world.landmarks]), dtype=torch.float32)
        return st, lm

    for n in (3, 6):
        env = po.make_oracle_env('simple_spread', n=n)
        for _ in range(5):
            set_random(env.world)
            st, lm = state_of(env.world)
            got = rows_from_state(st, lm).numpy()
            ref32 = np.stack([np.concatenate([np.float32(a.state.p_vel), np.float32(a.state.p_pos)] +
                                            [np.float32(l.state.p_pos) - np.float32(a.state.p_pos) for l in env.world.landmarks])
                              for a in env.world.agents])
            assert np.array_equal(got, ref32)                                             # float32 arithmetic in the oracle's order
            np.testing.assert_allclose(got, np.stack(f32_rows(env)), rtol=0, atol=1e-6)   # and it is the oracle's float64 row
    for adv, good, L in ((4, 2, 2), (2, 3, 3), (1, 1, 1)):
        env = po.make_oracle_env('simple_tag', num_adversaries=adv, num_good=good, num_landmarks=L)
        N = adv + good
        D = 4 + 2 * L + 2 * (N - 1) + 2 * good
        for _ in range(5):
            set_random(env.world)
            st, lm = state_of(env.world)
            got = rows_from_state(st, lm, 'simple_tag', adv).numpy()
            assert got.shape == (N, D)
            for i, r in enumerate(f32_rows(env)):
                assert len(r) == (D if i < adv else D - 2)                               # good agents' rows are two numbers shorter: zero-padded
                np.testing.assert_allclose(got[i, :len(r)], r, rtol=0, atol=1e-6)
                assert not got[i, len(r):].any()
    env = po.make_oracle_env('simple_reference')
    for _ in range(8):
        env.reset()
        set_random(env.world)
        sym = rng.randint(0, 11, 2)                                                      # 10 = no symbol yet (zeros)
        for a, s in zip(env.world.agents, sym):
            a.state.c = np.zeros(10) if s == 10 else np.eye(10)[s]
        rows = f32_rows(env)
        head = torch.tensor(np.stack([r[:8] for r in rows]), dtype=torch.float32)
        goal = torch.tensor([env.world.landmarks.index(a.goal_b) for a in env.world.agents])
        seen = torch.tensor([255 if sym[1 - i] == 10 else int(sym[1 - i]) for i in range(2)])    # what agent i sees: the OTHER agent's symbol
        got = ref_rows(head, goal, seen).numpy()
        assert np.array_equal(got, np.stack(rows).astype(np.float32))
