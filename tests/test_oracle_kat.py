"""Pins the CPU oracle: hand-derived known-answer tests and
cross-checks between its three forms (scalar NumPy f64, C f64, C f32).
The reference holds no golden vectors for this path ("parity unpinned")."""
import math

import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import particle_oracle as po


# ---------------------------------------------------------------- RNG
def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        assert po.philox4x32_10(ctr, key) == want
        assert co.philox4x32_10(ctr, key) == want


def test_kat_reset_numpy_legacy_stream():
    # main.py:41,47: seed = 12345678 + cnt; draws happen inside reset_world
    np.random.seed(12345678)
    env = po.make_oracle_env('simple_spread')  # make_world consumes draws BEFORE seeding in main.py
    np.random.seed(12345678)
    env.reset()
    pos, vel, lm = po.get_world_state(env.world)
    want_a = [(-0.5083915323501949, 0.19285721064295336), (-0.2824167431561284, -0.242179789864017),
              (-0.9510972596309135, -0.523380858782218)]
    want_l = [(-0.22413133239983196, 0.3610820860587127), (0.6786816615045046, 0.521473770985093),
              (-0.5565167970745335, -0.873116770042403)]
    assert pos.tolist() == [list(w) for w in want_a]
    assert lm.tolist() == [list(w) for w in want_l]
    assert not vel.any()


def test_philox_reset_matches_between_python_and_c():
    cfg = co.make_config('simple_tag', 6, num_adversaries=4, seed=0x1234567890abcdef, env_id_base=(1 << 33) + 5)
    o = co.COracle(cfg, 3, np.float32)
    o.reset()
    for e in range(3):
        for i in range(6):
            x, y = po.philox_entity_xy(cfg.seed, cfg.env_id_base + e, 1, i, -1.0, 1.0)
            assert (o.pos[e, i, 0], o.pos[e, i, 1]) == (x, y)
        for l in range(2):
            x, y = po.philox_entity_xy(cfg.seed, cfg.env_id_base + e, 1, 6 + l, -0.9, 0.9)
            assert (o.lm[e, l, 0], o.lm[e, l, 1]) == (x, y)
    assert np.all(np.abs(o.pos) <= 1) and np.all(np.abs(o.lm) <= np.float32(0.9))
    assert not o.vel.any() and (o.ep_count == 1).all()


# ---------------------------------------------------------------- physics KATs
def _one_agent_env():
    env = po.make_oracle_env('simple_spread', n=1)
    po.set_world_state(env.world, [[0.0, 0.0]], [[0.0, 0.0]], [[5.0, 5.0]])
    return env


def test_kat_free_flight_and_noop():
    env = _one_agent_env()
    vs, ps = [], []
    for _ in range(3):
        env.step([po.onehot(1)])
        vs.append(env.world.agents[0].state.p_vel[0])
        ps.append(env.world.agents[0].state.p_pos[0])
    assert vs == [0.5, 0.875, 1.15625]
    np.testing.assert_allclose(ps, [0.05, 0.1375, 0.253125], rtol=0, atol=1e-15)
    env.step([po.onehot(0)])
    assert env.world.agents[0].state.p_vel[0] == 0.75 * 1.15625
    # action index -> direction (U2): 1:+x 2:-x 3:+y 4:-y
    for idx, want in [(1, (0.5, 0)), (2, (-0.5, 0)), (3, (0, 0.5)), (4, (0, -0.5)), (0, (0, 0))]:
        env = _one_agent_env()
        env.step([po.onehot(idx)])
        assert tuple(env.world.agents[0].state.p_vel) == want


def test_kat_free_flight_c_oracles():
    for dt, tol in [(np.float64, 0), (np.float32, 1e-7)]:
        cfg = co.make_config('simple_spread', 1, max_episode_len=0)
        o = co.COracle(cfg, 1, dt)
        o.set_state([[[0, 0]]], [[[0, 0]]], [[[5, 5]]])
        vs = []
        for _ in range(3):
            o.step(act_idx=[[1]])
            vs.append(float(o.vel[0, 0, 0]))
        np.testing.assert_allclose(vs, [0.5, 0.875, 1.15625], rtol=0, atol=tol)


@pytest.mark.parametrize('dist,pen,force', [
    (0.2, 0.09999999999999998, 9.999999999999998),
    (0.3, 6.931471805599453e-4, 0.06931471805599454),
    (0.31, 4.5398899216864243e-08, None),
    (1e-4, 0.2999, None)])
def test_kat_contact(dist, pen, force):
    w = po.World()
    a, b = po.Agent(), po.Agent()
    for e in (a, b):
        e.size = 0.15
    a.state.p_pos, b.state.p_pos = np.array([dist, 0.0]), np.array([0.0, 0.0])
    fa, fb = w.get_collision_force(a, b)
    k = 1e-3
    assert math.isclose(np.logaddexp(0, -(dist - 0.3) / k) * k, pen, rel_tol=1e-12)
    assert np.array_equal(fa, -fb)
    if force is not None:
        assert math.isclose(fa[0], force, rel_tol=1e-12) and fa[1] == 0
    # same numbers out of the C oracles (float32: (dist - dist_min) / k amplifies
    # the rounding of the inputs by 1/k, so only ~1e-4 relative on tiny forces)
    for dt, tol in [(np.float64, 1e-12), (np.float32, 1e-4)]:
        cfg = co.make_config('simple_spread', 2, max_episode_len=0)
        o = co.COracle(cfg, 1, dt)
        o.set_state([[[dist, 0], [0, 0]]], 0, [[[9, 9], [9, 9]]])
        o.step(act_idx=[[0, 0]])
        want_dv = 100.0 * pen * 0.1
        assert math.isclose(o.vel[0, 0, 0], want_dv, rel_tol=tol, abs_tol=1e-12)
        assert o.vel[0, 1, 0] == -o.vel[0, 0, 0]


def test_kat_contact_one_step_from_rest():
    env = po.make_oracle_env('simple_spread', n=2)
    po.set_world_state(env.world, [[0.2, 0], [0, 0]], np.zeros((2, 2)), [[9, 9], [9, 9]])
    env.step([po.onehot(0), po.onehot(0)])
    pos, vel, _ = po.get_world_state(env.world)
    np.testing.assert_allclose(vel[:, 0], [1.0, -1.0], atol=1e-14)
    np.testing.assert_allclose(pos[:, 0], [0.3, -0.1], atol=1e-14)


def test_kat_reward_and_self_collision():
    env = po.make_oracle_env('simple_spread', n=1)
    po.set_world_state(env.world, [[0.25, -0.5]], [[0, 0]], [[0.25, -0.5]])
    _, rew, done, info = env.step([po.onehot(0)])
    assert rew == [-1.0] and done == [False] and info == {'n': [{}]}


def test_kat_momentum_antisymmetry():
    rng = np.random.RandomState(0)
    cfg = co.make_config('simple_spread', 6, max_episode_len=0)
    o = co.COracle(cfg, 64, np.float64)
    o.set_state(rng.uniform(-0.4, 0.4, (64, 6, 2)), 0, rng.uniform(-1, 1, (64, 6, 2)))
    o.step(act_idx=np.zeros((64, 6), np.int32))
    assert np.abs(o.vel).max() > 0.1
    np.testing.assert_allclose(o.vel.sum(axis=1), 0, atol=1e-12)


def test_kat_obs_dims_and_layout():
    for n, local, d in [(3, True, 10), (3, False, 18), (6, True, 16), (6, False, 36)]:
        env = po.make_oracle_env('simple_spread', n=n, local_observation=local)
        assert env.observation_space[0].shape == (d,) and env.action_space[0].n == 5
        assert not hasattr(env.action_space[0], 'high')
        cfg = co.make_config('simple_spread', n, obs_mode='local' if local else 'full')
        assert co.obs_dim(cfg) == d
    env = po.make_oracle_env('simple_spread', n=3)
    obs = env.reset()
    pos, vel, lm = po.get_world_state(env.world)
    np.testing.assert_array_equal(obs[1], np.concatenate([vel[1], pos[1]] + [l - pos[1] for l in lm]))
    # simple_tag canonical 3+1: 16 / 14; C3 4+2: 22 / 20
    env = po.make_oracle_env('simple_tag')
    assert [s.shape[0] for s in env.observation_space] == [16, 16, 16, 14]
    env = po.make_oracle_env('simple_tag', num_good=2, num_adversaries=4)
    assert [s.shape[0] for s in env.observation_space] == [22] * 4 + [20] * 2
    assert co.obs_dim(co.make_config('simple_tag', 6, num_adversaries=4)) == 22


def test_force_discrete_argmax_and_no_alias():
    env = po.make_oracle_env('simple_spread', n=1)
    po.set_world_state(env.world, [[0, 0]], [[0, 0]], [[5, 5]])
    a = np.array([0.1, 0.2, 0.6, 0.6, 0.0])  # first max wins -> index 2 (-x)
    env.step([a])
    assert tuple(env.world.agents[0].state.p_vel) == (-0.5, 0.0)


# ---------------------------------------------------------------- deterministic f32 math
def test_det_math_accuracy():
    L = co.lib()
    xs = np.concatenate([np.linspace(-86.9, 3.0, 20001), -np.logspace(-8, 1.9, 2000)]).astype(np.float32)
    got = np.array([L.po_exp_det_f32(float(x)) for x in xs], np.float64)
    np.testing.assert_allclose(got, np.exp(xs.astype(np.float64)), rtol=3e-7)
    assert L.po_exp_det_f32(-87.0) == 0.0 and L.po_exp_det_f32(-1e30) == 0.0
    assert L.po_exp_det_f32(0.0) == 1.0
    ts = np.concatenate([np.linspace(0, 1, 5001), np.logspace(-38, 0, 500)]).astype(np.float32)
    got = np.array([L.po_log1p_det_f32(float(t)) for t in ts], np.float64)
    np.testing.assert_allclose(got, np.log1p(ts.astype(np.float64)), rtol=3e-7, atol=0)
    xs = np.concatenate([np.linspace(-100, 400, 20001), np.linspace(-3, 3, 4001)]).astype(np.float32)
    got = np.array([L.po_softplus_det_f32(float(x)) for x in xs], np.float64)
    np.testing.assert_allclose(got, np.logaddexp(0, xs.astype(np.float64)), rtol=4e-7, atol=1e-37)
    assert L.po_softplus_det_f32(-87.0) == 0.0  # provably-zero contact force beyond dist_min + 0.087


# ---------------------------------------------------------------- cross-form agreement
def _drive_python(env_kw, pos, vel, lm, acts, steps):
    """B python worlds stepped `steps` times -> stacked per-step outputs."""
    B = pos.shape[0]
    out = []
    envs = []
    for e in range(B):
        env = po.make_oracle_env(**env_kw)
        po.set_world_state(env.world, pos[e], vel[e], lm[e])
        envs.append(env)
    for t in range(steps):
        obs_t, rew_t, pos_t, vel_t = [], [], [], []
        for e, env in enumerate(envs):
            o, r, d, _ = env.step([po.onehot(a) for a in acts[t, e]])
            assert d == [False] * env.n
            D = max(len(x) for x in o)
            obs_t.append(np.stack([np.pad(x, (0, D - len(x))) for x in o]))
            rew_t.append(np.array(r, np.float64))
            p, v, _ = po.get_world_state(env.world)
            pos_t.append(p)
            vel_t.append(v)
        out.append((np.stack(obs_t), np.stack(rew_t), np.stack(pos_t), np.stack(vel_t)))
    return out


CASES = [
    ('spread3', dict(scenario_name='simple_spread', n=3), dict(scenario='simple_spread', num_agents=3)),
    ('spread6', dict(scenario_name='simple_spread', n=6), dict(scenario='simple_spread', num_agents=6)),
    ('spread4full', dict(scenario_name='simple_spread', n=4, local_observation=False),
     dict(scenario='simple_spread', num_agents=4, obs_mode='full')),
    ('spread5_L2', dict(scenario_name='simple_spread', n=5, num_landmarks=2),
     dict(scenario='simple_spread', num_agents=5, num_landmarks=2)),
    ('tag3+1', dict(scenario_name='simple_tag'), dict(scenario='simple_tag', num_agents=4, num_adversaries=3)),
    ('tag4+2', dict(scenario_name='simple_tag', num_good=2, num_adversaries=4),
     dict(scenario='simple_tag', num_agents=6, num_adversaries=4)),
]


@pytest.mark.parametrize('name,env_kw,cfg_kw', CASES, ids=[c[0] for c in CASES])
def test_c_oracle_matches_python_oracle(name, env_kw, cfg_kw):
    rng = np.random.RandomState(abs(hash(name)) % 2 ** 31)
    cfg = co.make_config(max_episode_len=0, **cfg_kw)
    B, N, L, T = 24, cfg.num_agents, cfg.num_landmarks, 12
    # crowded boxes so contacts, speed clamps and the tag boundary penalty all occur
    pos = rng.uniform(-0.5, 0.5, (B, N, 2))
    pos[B // 2:] = rng.uniform(0.6, 1.15, (B - B // 2, N, 2)) * rng.choice([-1, 1], (B - B // 2, N, 2))
    vel = rng.uniform(-1.5, 1.5, (B, N, 2))
    lm = rng.uniform(-0.9, 0.9, (B, L, 2))
    acts = rng.randint(0, 5, (T, B, N))
    want = _drive_python(env_kw, pos, vel, lm, acts, T)
    o64 = co.COracle(cfg, B, np.float64)
    o64.set_state(pos, vel, lm)
    o32 = co.COracle(cfg, B, np.float32)
    o32.set_state(pos, vel, lm)
    saw_contact = False
    for t in range(T):
        r64 = o64.step(act_idx=acts[t])
        obs_w, rew_w, pos_w, vel_w = want[t]
        np.testing.assert_allclose(o64.pos, pos_w, rtol=0, atol=1e-11)
        np.testing.assert_allclose(o64.vel, vel_w, rtol=0, atol=1e-10)
        np.testing.assert_allclose(r64['obs'], obs_w, rtol=0, atol=1e-10)
        np.testing.assert_allclose(r64['rew'], rew_w, rtol=0, atol=1e-10)
        assert not r64['done'].any() and not r64['terminal'].any()
        saw_contact |= bool((r64['coll'] & ~(1 << np.arange(N, dtype=np.uint64))[None, :]).any())
    assert saw_contact
    # float32 form: one step from the SAME (float32-representable) state
    o32.set_state(o64.pos, o64.vel, o64.lm)
    o64.set_state(o32.pos, o32.vel, o32.lm)
    # single-step f32 vs f64 from identical states: within 1e-5 (north_star tolerance)
    a = rng.randint(0, 5, (B, N))
    r32, r64 = o32.step(act_idx=a), o64.step(act_idx=a)
    np.testing.assert_allclose(o32.pos, o64.pos, rtol=0, atol=1e-5)
    np.testing.assert_allclose(o32.vel, o64.vel, rtol=0, atol=1e-4)
    np.testing.assert_allclose(r32['obs'], r64['obs'], rtol=0, atol=1e-4)


def test_c_oracle_onehot_equals_index_path():
    rng = np.random.RandomState(3)
    cfg = co.make_config('simple_spread', 6, max_episode_len=0)
    a, b = co.COracle(cfg, 16, np.float32), co.COracle(cfg, 16, np.float32)
    pos, lm = rng.uniform(-0.5, 0.5, (16, 6, 2)), rng.uniform(-1, 1, (16, 6, 2))
    a.set_state(pos, 0, lm)
    b.set_state(pos, 0, lm)
    idx = rng.randint(0, 5, (16, 6))
    ra = a.step(act_idx=idx)
    rb = b.step(act_vec=np.eye(5)[idx])
    assert np.array_equal(a.pos, b.pos) and np.array_equal(ra['obs'], rb['obs'])
    # soft (non one-hot) input is arg-maxed when force_discrete_action (experiments/scenarios.py:191)
    soft = np.eye(5)[idx] * 0.5 + 0.1
    b.set_state(pos, 0, lm)
    b.step(act_vec=soft)
    assert np.array_equal(a.pos, b.pos)


def test_c_oracle_auto_reset_and_terminal():
    cfg = co.make_config('simple_spread', 3, max_episode_len=5, auto_reset=True, seed=7)
    o = co.COracle(cfg, 4, np.float32)
    first = o.reset()
    assert (o.ep_count == 1).all()
    for t in range(1, 11):
        r = o.step(act_idx=np.full((4, 3), 1))
        assert r['terminal'].all() == (t % 5 == 0) and r['terminal'].any() == (t % 5 == 0)
        if t % 5 == 0:
            assert (o.ep_step == 0).all() and (o.ep_count == 1 + t // 5).all()
            assert not o.vel.any()
            assert np.array_equal(r['obs'], o.observe())
            assert not np.array_equal(r['obs'], r['final_obs'])
            assert np.abs(r['final_obs'][:, :, 0]).min() > 0.5  # pre-reset velocity is in final_obs
    # same seed, fresh oracle, different batch split -> same initial states (RNG keyed by global env id)
    cfg2 = co.make_config('simple_spread', 3, max_episode_len=5, auto_reset=True, seed=7, env_id_base=2)
    o2 = co.COracle(cfg2, 2, np.float32)
    assert np.array_equal(o2.reset(), first[2:])


def test_oracle_self_regression_vectors():
    """tests/golden/oracle_vectors.npz (made by make_oracle_vectors.py from this oracle) has not drifted."""
    import os
    from tests.golden.make_oracle_vectors import CASES, trajectory
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'oracle_vectors.npz'))
    for i, (name, kw, seed) in enumerate(CASES):
        tr = trajectory(name, kw, seed)
        for k, v in tr.items():
            np.testing.assert_allclose(v, g['%d/%s' % (i, k)], rtol=0, atol=1e-12, err_msg='%s %s' % (name, k))


def test_published_math_header_equals_the_oracle_restatement_bitwise(tmp_path):
    """include/pworld_math.h is the PUBLISHED float32 contract; oracle/pworld_oracle.c restates it independently (not
    shared code).  Compiled as plain C, the header must give the oracle's bits for pw_exp / pw_log1p01 / pw_softplus over
    every exponent, the exact-zero cut, NaN and inf -- otherwise the kernels would be tested against something other than
    what a user of the header reproduces.  (The device forms are compared with the oracle in test_gpu_parity.py.)"""
    import ctypes as C
    import os
    import shutil
    import subprocess
    cc = shutil.which('gcc') or shutil.which('cc')
    if cc is None:
        pytest.skip('no C compiler')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / 'hdr.c'
    src.write_text('#include "pworld_math.h"\n'
                   'void hdr_v(int fn, const float *x, float *y, long n) {\n'
                   '  for (long i = 0; i < n; ++i) y[i] = fn == 0 ? pw_exp(x[i]) : fn == 1 ? pw_log1p01(x[i]) : pw_softplus(x[i]);\n'
                   '}\n')
    so = tmp_path / 'hdr.so'
    fma = ['-mfma'] if 'fma' in open('/proc/cpuinfo').read().split() else []
    subprocess.check_call([cc, '-O2', '-std=c11', '-ffp-contract=off', '-fno-fast-math', *fma, '-fPIC', '-shared',
                           '-I', os.path.join(root, 'include'), '-o', str(so), str(src), '-lm'])
    hdr = C.CDLL(str(so))
    L = co.lib()
    rng = np.random.RandomState(7)
    bits = rng.randint(0, 2 ** 32, 400_000, dtype=np.uint64).astype(np.uint32).view(np.float32)
    special = np.array([0.0, -0.0, 1e-45, 1.17549435e-38, 1.0, -1.0, 87.0, -87.0, -86.999, -87.001, 88.0, 300.0, -300.0,
                        np.inf, -np.inf, np.nan, 3.4e38, -3.4e38], np.float32)
    xs = np.concatenate([special, bits, rng.uniform(-100, 320, 400_000).astype(np.float32)])
    ts = np.concatenate([special[:5], np.float32(2.0) ** -rng.uniform(0, 126, 10_000).astype(np.float32),
                         rng.uniform(0, 1, 10_000).astype(np.float32)])

    def run(fn, x):
        x = np.ascontiguousarray(x, np.float32)
        y = np.empty_like(x)
        hdr.hdr_v(C.c_int(fn), x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), C.c_long(x.size))
        return y

    def same_bits(a, b):
        return ((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all()

    assert same_bits(run(0, xs), co.math_v(3, xs))                      # pw_exp
    assert same_bits(run(2, xs), co.math_v(1, xs))                      # pw_softplus
    L.po_log1p_det_f32.restype = C.c_float
    L.po_log1p_det_f32.argtypes = [C.c_float]
    want = np.array([L.po_log1p_det_f32(float(t)) for t in ts], np.float32)
    assert same_bits(run(1, ts), want)                              # pw_log1p01
