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
