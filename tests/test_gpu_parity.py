"""GPU parity: the HIP path (through the C ABI, via multiagent_rl_amd) against the CPU oracle.

Bars (BASELINE.json north_star): integer done / collision masks BIT-EXACT, float32 state
within 1e-5 of the float64 reference semantics.  Because the kernels follow the upstream
operation order with deterministic float32 math, we assert the stronger property too:
every output equals the float32 C oracle bit for bit.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')

from oracle import c_oracle as co  # noqa: E402  (checker only)


def _mk(scenario='simple_spread', num_envs=64, num_agents=3, num_landmarks=None, num_adversaries=0,
        obs_mode='local', max_episode_len=25, auto_reset=False, seed=12345678, env_id_base=0,
        force_discrete_action=True, want_coll=True, dispatch=None):
    from multiagent_rl_amd.env import BatchedParticleEnv
    kw = dict(num_agents=num_agents, num_landmarks=num_landmarks, local_observation=obs_mode == 'local',
              max_episode_len=max_episode_len, auto_reset=auto_reset, seed=seed, env_id_base=env_id_base,
              force_discrete_action=force_discrete_action)
    if scenario == 'simple_tag':
        kw.update(num_adversaries=num_adversaries, num_good=num_agents - num_adversaries)
        kw.pop('num_agents')
    env = BatchedParticleEnv(scenario, num_envs, want_coll=want_coll, dispatch=dispatch, **kw)
    cfg = co.make_config(scenario, num_agents, num_landmarks=num_landmarks, num_adversaries=num_adversaries,
                         obs_mode=obs_mode, max_episode_len=max_episode_len, auto_reset=auto_reset, seed=seed,
                         env_id_base=env_id_base, force_discrete_action=force_discrete_action)
    return env, cfg


def _rand_state(rng, B, N, L, crowded=True):
    pos = rng.uniform(-0.5, 0.5, (B, N, 2)) if crowded else rng.uniform(-1, 1, (B, N, 2))
    h = B // 2
    pos[h:] = rng.uniform(0.6, 1.15, (B - h, N, 2)) * rng.choice([-1, 1], (B - h, N, 2))
    vel = rng.uniform(-1.5, 1.5, (B, N, 2))
    lm = rng.uniform(-0.9, 0.9, (B, L, 2))
    return pos.astype(np.float32), vel.astype(np.float32), lm.astype(np.float32)


def _np(t):
    return t.detach().cpu().numpy()


# pw_dispatch with every choice automatic (what pw_create gives in a clean environment); dict(AUTO, quad=0) etc. override
AUTO = dict(force_generic=0, no_stream=0, duo=-1, quad=-1, obs_block=-1, trio=-1, p_prio=-1, envs_per_wave=0, policy_form=0)


def _coll(t):
    return _np(t).view(np.uint64)


def _assert_same_bits(got, want, name):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    if got.dtype.kind == 'f':
        same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    else:
        same = got == want
    assert same.all(), '%s: %d / %d elements differ; max abs diff %g' % (
        name, (~same).sum(), same.size, np.nanmax(np.abs(got.astype(np.float64) - want.astype(np.float64))))


CASES = [
    dict(scenario='simple_spread', num_agents=3, num_envs=1),               # configs[0] shape
    dict(scenario='simple_spread', num_agents=3, num_envs=1000),            # 21 envs / wave, ragged tail
    dict(scenario='simple_spread', num_agents=3, num_envs=4096),            # configs[4] (C5) N = 3 point, full size
    dict(scenario='simple_spread', num_agents=6, num_envs=4096),            # configs[1] (C2) full size
    dict(scenario='simple_spread', num_agents=12, num_envs=333),
    dict(scenario='simple_spread', num_agents=9, num_envs=700),             # the reference's middle scalability setting (main_scalability_1.py:30): 'trio' = its row-wise three-wave form
    dict(scenario='simple_spread', num_agents=24, num_envs=65),
    dict(scenario='simple_spread', num_agents=48, num_envs=33),
    dict(scenario='simple_spread', num_agents=12, num_envs=4096),           # configs[4] (C5) full size
    dict(scenario='simple_spread', num_agents=24, num_envs=4096),
    dict(scenario='simple_spread', num_agents=48, num_envs=4096),
    dict(scenario='simple_spread', num_agents=64, num_envs=5),              # maximum N
    dict(scenario='simple_spread', num_agents=1, num_envs=130),
    dict(scenario='simple_spread', num_agents=4, num_envs=200, obs_mode='full'),
    dict(scenario='simple_spread', num_agents=5, num_landmarks=2, num_envs=77),
    dict(scenario='simple_spread', num_agents=3, num_landmarks=7, num_envs=77),   # L > N path
    dict(scenario='simple_spread', num_agents=3, num_landmarks=0, num_envs=9),    # empty landmark set
    dict(scenario='simple_tag', num_agents=4, num_adversaries=3, num_envs=500),   # canonical 3+1
    dict(scenario='simple_tag', num_agents=6, num_adversaries=4, num_envs=8192),  # configs[2] (C3) full size
    dict(scenario='simple_tag', num_agents=5, num_adversaries=2, num_landmarks=3, num_envs=301),  # runtime-N variant
    dict(scenario='simple_tag', num_agents=3, num_adversaries=3, num_envs=70),    # no good agents
    dict(scenario='simple_tag', num_agents=3, num_adversaries=0, num_envs=70),    # no adversaries
]


PATHS = ['duo', 'stream', 'duo+coll', 'stream+coll', 'duo+block', 'stream+block', 'fast', 'generic',
         'duo-dense', 'stream-dense', 'duo+coll-dense', 'stream+coll-dense', 'duo+block-dense',
         'stream+block-dense', 'fast-dense', 'generic-dense', 'quad', 'quad+coll', 'trio', 'trio+block', 'trio+block-dense']


class KernelPath(object):
    """One kernel form, as a pw_dispatch selection (no environment variables: the handle carries it) + whether the
    optional collision-mask output is requested."""

    def __init__(self, name):
        self.name = name
        param = name
        d = dict(quad=0)                           # every path but 'quad*' keeps the duo / stream kernels' coverage at N = 6
        if param.endswith('-dense'):
            d['envs_per_wave'] = 64
            param = param[:-6]
        self.want_coll = param not in ('stream', 'duo', 'stream+block', 'duo+block', 'quad', 'trio', 'trio+block')
        if param.startswith('quad'):
            d['quad'] = 1
            param = 'duo'
        if param.endswith('+coll'):
            param = param[:-5]
        if param.endswith('+block'):
            d['obs_block'] = 1
            param = param[:-6]
        if param == 'trio':
            d['trio'] = 1
            param = 'duo'
        if param == 'stream':
            d['duo'] = 0
        elif param == 'generic':
            d['force_generic'] = 1
        elif param == 'fast':
            d['no_stream'] = 1
        self.dispatch = d


@pytest.fixture(params=PATHS)
def kernel_path(request):
    """Four kernels serve simple_spread with homogeneous agents; all must give the same bits.
    'duo'     pw_spread_duo_kernel    (default when all standard outputs are present),
    'stream'  pw_spread_stream_kernel (pw_dispatch.duo = 0),
    'fast'    pw_spread_fast_kernel   (pw_dispatch.no_stream),
    'generic' pw_rollout_kernel       (pw_dispatch.force_generic).
    'duo' / 'stream' run WITHOUT the optional collision-mask output (the bench path's instantiations);
    'duo+coll' / 'stream+coll' are the instantiations that also store the masks;
    'quad' / 'quad+coll' (N = L = 6 only; elsewhere the default choice again) force pw_spread_quad_kernel, the four-wave
    pair-parallel form small grids of BASELINE configs[1] run, without / with the mask output (every other path here
    sets quad = 0 so that the duo kernel keeps its coverage);
    'duo+block' / 'stream+block' force the block-wise observation stores large grids use (default here: only N >= 12,
    the test batches being small), '-dense' then gives 60- and 63-row blocks;
    'trio' / 'trio+block' the three-wave variants of the duo kernels -- what BASELINE configs[2] (simple_tag, B = 8192)
    and the N = 3 / N = 12 points of configs[4] dispatch by default: simple_tag in both, simple_spread N = 3 with row-wise stores
    ('trio'), N >= 6 in its block-store form ('trio+block'; 'trio' alone is the two-wave form there).
    simple_tag has three: pw_tag_duo_kernel ('duo', 'trio'), pw_tag_stream_kernel ('stream') and the generic kernel
    ('fast'/'generic').
    Small batches are spread over ~512 workgroups (few envs per wave); '-dense' forces the packing large batches
    get (64 // N envs per wave) so that the multi-env-per-wave indexing is exercised at test sizes."""
    return KernelPath(request.param)


@pytest.mark.parametrize('case', CASES, ids=lambda c: '%s-N%d-L%s-B%d-%s' % (
    c['scenario'], c['num_agents'], c.get('num_landmarks'), c['num_envs'], c.get('obs_mode', 'local')))
def test_single_step_from_injected_states(case, kernel_path):
    env, cfg = _mk(max_episode_len=0, want_coll=kernel_path.want_coll, dispatch=kernel_path.dispatch, **case)
    B, N, L = env.num_envs, env.n, env.num_landmarks
    rng = np.random.RandomState(B * 131 + N)
    pos, vel, lm = _rand_state(rng, B, N, L)
    act = rng.randint(0, 5, (B, N)).astype(np.int32)
    env.set_state(pos, vel, lm)
    obs, rew, done, info = env.step(torch.from_numpy(act))
    st = env.get_state()
    o32 = co.COracle(cfg, B, np.float32)
    o32.set_state(pos, vel, lm)
    w = o32.step(act_idx=act)
    _assert_same_bits(_np(st['pos']), o32.pos, 'pos')
    _assert_same_bits(_np(st['vel']), o32.vel, 'vel')
    _assert_same_bits(_np(st['landmarks']), o32.lm, 'landmarks')
    _assert_same_bits(_np(obs), w['obs'], 'obs')
    _assert_same_bits(_np(rew), w['rew'], 'rew')
    if 'coll' in info:
        _assert_same_bits(_coll(info['coll']), w['coll'], 'coll')        # integer masks: bit-exact
    _assert_same_bits(_np(done).astype(np.uint8), w['done'], 'done')
    _assert_same_bits(_np(info['terminal']).astype(np.uint8), w['terminal'], 'terminal')
    assert not _np(done).any()
    # shared reward (run.py:46) = agent-order sum
    want_shared = np.zeros(B, np.float32)
    for i in range(N):
        want_shared = want_shared + w['rew'][:, i]
    _assert_same_bits(_np(info['rew_shared']), want_shared, 'rew_shared')
    # float64 reference semantics: float32 STATE within 1e-5 (north_star tolerance) -- positions, velocities and the
    # observation rows built from them, even in the crowded half of the batch (measured maxima per configuration:
    # profiles/r2_parity_error.txt, <= 4.5e-6 at N = 48 under this stress, <= 2.3e-6 along real episodes)
    o64 = co.COracle(cfg, B, np.float64)
    o64.set_state(pos, vel, lm)
    w64 = o64.step(act_idx=act)
    np.testing.assert_allclose(_np(st['pos']), o64.pos, rtol=0, atol=1e-5)
    np.testing.assert_allclose(_np(st['vel']), o64.vel, rtol=0, atol=1e-5)
    np.testing.assert_allclose(_np(obs), w64['obs'], rtol=0, atol=1e-5)
    # masks vs float64: identical except pairs whose distance is within 1e-6 of the threshold
    diff = (_coll(info['coll']) if 'coll' in info else w['coll']) ^ w64['coll']
    assert np.count_nonzero(diff) <= 2, 'collision masks differ from the float64 oracle in %d rows' % np.count_nonzero(diff)
    # rewards: 1e-5 (+ 1e-6 relative: |rew| reaches ~60 at N = 64) in every env whose integer masks agree with
    # float64 -- a flipped mask bit is a reward step of exactly 1 (or 10 in simple_tag), not a rounding error
    ok = ~(diff != 0).any(axis=1)
    np.testing.assert_allclose(_np(rew)[ok], w64['rew'][ok], rtol=1e-6, atol=1e-5)
    # the crowded half of the batch must actually exercise contacts
    assert (w['coll'] != (np.uint64(1) << np.arange(N, dtype=np.uint64))[None, :]).any() or N == 1 or B < 64


@pytest.mark.parametrize('case', [
    dict(scenario='simple_spread', num_agents=6, num_envs=257),
    dict(scenario='simple_spread', num_agents=3, num_envs=100, obs_mode='full'),
    dict(scenario='simple_tag', num_agents=6, num_adversaries=4, num_envs=123),
    dict(scenario='simple_tag', num_agents=4, num_adversaries=3, num_envs=77),
    dict(scenario='simple_tag', num_agents=7, num_adversaries=3, num_landmarks=1, num_envs=50),
    dict(scenario='simple_spread', num_agents=3, num_envs=257),      # the C5 points: the duo kernel's 3-slot ring hand-off,
    dict(scenario='simple_spread', num_agents=12, num_envs=333),     # the Philox auto-reset and the block-store path of
    dict(scenario='simple_spread', num_agents=24, num_envs=300),     # every compile-time instantiation across two resets
    dict(scenario='simple_spread', num_agents=48, num_envs=130),
], ids=['spread6', 'spread3full', 'tag4+2', 'tag3+1', 'tag3+4', 'spread3', 'spread12', 'spread24', 'spread48'])
def test_rollout_with_auto_reset_matches_oracle_bitwise(case, kernel_path):
    T, ep_len = 58, 25
    env, cfg = _mk(max_episode_len=ep_len, auto_reset=True, seed=99, env_id_base=1 << 33,
                   want_coll=kernel_path.want_coll, dispatch=kernel_path.dispatch, **case)
    B, N = env.num_envs, env.n
    rng = np.random.RandomState(5)
    acts = rng.randint(0, 5, (T, B, N)).astype(np.int32)
    obs0 = env.reset()
    o32 = co.COracle(cfg, B, np.float32)
    _assert_same_bits(_np(obs0), o32.reset(), 'reset obs')
    out = env.rollout(torch.from_numpy(acts))
    resets = 0
    for t in range(T):
        w = o32.step(act_idx=acts[t])
        _assert_same_bits(_np(out['obs'][t]), w['obs'], 'obs[%d]' % t)
        _assert_same_bits(_np(out['rew'][t]), w['rew'], 'rew[%d]' % t)
        _assert_same_bits(_np(out['terminal'][t]).astype(np.uint8), w['terminal'], 'terminal[%d]' % t)
        if 'coll' in out:
            _assert_same_bits(_coll(out['coll'][t]), w['coll'], 'coll[%d]' % t)   # pre-reset masks on reset steps
        if w['terminal'].any():
            resets += 1
            _assert_same_bits(_np(out['final_obs'][t]), w['final_obs'], 'final_obs[%d]' % t)
            assert w['terminal'].all() and (t + 1) % ep_len == 0
    assert resets == T // ep_len
    st = env.get_state()
    _assert_same_bits(_np(st['pos']), o32.pos, 'pos')
    _assert_same_bits(_np(st['vel']), o32.vel, 'vel')
    _assert_same_bits(_np(st['landmarks']), o32.lm, 'landmarks')
    assert np.array_equal(_np(st['ep_step']), o32.ep_step) and np.array_equal(_np(st['ep_count']).astype(np.uint32), o32.ep_count)


@pytest.mark.parametrize('case', [
    dict(scenario='simple_tag', num_agents=6, num_adversaries=4, num_envs=8192),
    dict(scenario='simple_tag', num_agents=2, num_adversaries=1, num_landmarks=1, num_envs=40),
    dict(scenario='simple_spread', num_agents=3, num_envs=4096),
    dict(scenario='simple_spread', num_agents=6, num_envs=4096),
    dict(scenario='simple_spread', num_agents=12, num_envs=333),
], ids=['tag4+2', 'tag1+1', 'spread3', 'spread6', 'spread12'])
def test_every_step_resets_matches_oracle_bitwise(case):
    """max_episode_len = 1 with auto-reset under the DEFAULT dispatch: every step ends an episode, so every step writes a pre-reset
    row (final_obs) and publishes a post-reset state.  The multi-wave kernel forms would need two ring slots per step here (their
    physics wave, one step ahead, overwrote the pre-reset slot the output wave was still reading: found in round 5 through simple_tag rows
    whose other-agent entries were post-reset values); the dispatcher serves this setting with the one-wave forms.  Every output of
    every step against the float32 oracle, bit for bit."""
    T = 9
    env, cfg = _mk(max_episode_len=1, auto_reset=True, seed=7, want_coll=False, **case)
    B, N = env.num_envs, env.n
    acts = np.random.RandomState(3).randint(0, 5, (T, B, N)).astype(np.int32)
    o32 = co.COracle(cfg, B, np.float32)
    _assert_same_bits(_np(env.reset()), o32.reset(), 'reset obs')
    out = env.rollout(torch.from_numpy(acts))
    assert 'duo' not in env.last_kernel() and 'quad' not in env.last_kernel(), env.last_kernel()
    for t in range(T):
        w = o32.step(act_idx=acts[t])
        assert w['terminal'].all()
        for name in ('obs', 'final_obs', 'rew', 'rew_shared'):
            _assert_same_bits(_np(out[name][t]), w[name], '%s[%d]' % (name, t))
        _assert_same_bits(_np(out['terminal'][t]).astype(np.uint8), w['terminal'], 'terminal[%d]' % t)
    st = env.get_state()
    _assert_same_bits(_np(st['pos']), o32.pos, 'pos')
    _assert_same_bits(_np(st['landmarks']), o32.lm, 'landmarks')


@pytest.mark.parametrize('ep_len', [2, 3])
@pytest.mark.parametrize('case', [
    dict(scenario='simple_tag', num_agents=6, num_adversaries=4, num_envs=8192),
    dict(scenario='simple_tag', num_agents=3, num_adversaries=2, num_landmarks=3, num_envs=77),
    dict(scenario='simple_spread', num_agents=3, num_envs=4096),
    dict(scenario='simple_spread', num_agents=6, num_envs=4096),
    dict(scenario='simple_spread', num_agents=12, num_envs=4096),
], ids=['tag4+2', 'tag2+1', 'spread3', 'spread6', 'spread12'])
def test_very_short_episodes_on_the_multi_wave_forms_match_oracle_bitwise(case, ep_len):
    """Episodes of 2 and 3 steps under the DEFAULT dispatch (the multi-wave forms: quad / duo / trio): a reset every second or third step
    is the tightest schedule their LDS ring serves (two slots in a resetting step, the physics wave one step ahead) -- every output of
    every step, the pre-reset rows included, against the float32 oracle."""
    T = 13
    env, cfg = _mk(max_episode_len=ep_len, auto_reset=True, seed=5, want_coll=False, **case)
    B, N = env.num_envs, env.n
    acts = np.random.RandomState(4).randint(0, 5, (T, B, N)).astype(np.int32)
    o32 = co.COracle(cfg, B, np.float32)
    _assert_same_bits(_np(env.reset()), o32.reset(), 'reset obs')
    out = env.rollout(torch.from_numpy(acts))
    assert any(k in env.last_kernel() for k in ('duo', 'quad')), env.last_kernel()
    for t in range(T):
        w = o32.step(act_idx=acts[t])
        for name in ('obs', 'rew', 'rew_shared'):
            _assert_same_bits(_np(out[name][t]), w[name], '%s[%d]' % (name, t))
        _assert_same_bits(_np(out['terminal'][t]).astype(np.uint8), w['terminal'], 'terminal[%d]' % t)
        if w['terminal'].any():
            _assert_same_bits(_np(out['final_obs'][t]), w['final_obs'], 'final_obs[%d]' % t)
    assert (T // ep_len) >= 4
    st = env.get_state()
    _assert_same_bits(_np(st['pos']), o32.pos, 'pos')
    _assert_same_bits(_np(st['landmarks']), o32.lm, 'landmarks')


def test_rollout_equals_repeated_steps_and_onehot_equals_index():
    T = 30
    rng = np.random.RandomState(11)
    acts = rng.randint(0, 5, (T, 300, 6)).astype(np.int32)
    a, _ = _mk(num_agents=6, num_envs=300, max_episode_len=25, auto_reset=True, seed=3, want_coll=False)
    b, _ = _mk(num_agents=6, num_envs=300, max_episode_len=25, auto_reset=True, seed=3)
    c, _ = _mk(num_agents=6, num_envs=300, max_episode_len=25, auto_reset=True, seed=3)
    for e in (a, b, c):
        e.reset()
    out = a.rollout(torch.from_numpy(acts))
    eye = np.eye(5, dtype=np.float32)
    for t in range(T):
        obs, rew, done, info = b.step(torch.from_numpy(acts[t]))
        soft = eye[acts[t]] * 0.6 + 0.05  # arg-maxed because force_discrete_action (scenarios.py:191)
        obs_c, rew_c, _, info_c = c.step(torch.from_numpy(soft))
        _assert_same_bits(_np(obs), _np(out['obs'][t]), 'obs')
        _assert_same_bits(_np(rew), _np(out['rew'][t]), 'rew')
        _assert_same_bits(_np(obs_c), _np(out['obs'][t]), 'obs(one-hot)')
        _assert_same_bits(_np(info['terminal']), _np(out['terminal'][t]), 'terminal')
        if _np(info['terminal']).any():
            _assert_same_bits(_np(info['final_obs']), _np(out['final_obs'][t]), 'final_obs')
            _assert_same_bits(_np(info_c['final_obs']), _np(out['final_obs'][t]), 'final_obs(one-hot)')


def test_soft_actions_without_force_discrete(kernel_path):
    env, cfg = _mk(num_agents=3, num_envs=50, max_episode_len=0, force_discrete_action=False,
                   want_coll=kernel_path.want_coll, dispatch=kernel_path.dispatch)
    rng = np.random.RandomState(2)
    pos, vel, lm = _rand_state(rng, 50, 3, 3)
    soft = rng.uniform(0, 1, (50, 3, 5)).astype(np.float32)
    env.set_state(pos, vel, lm)
    env.step(torch.from_numpy(soft))
    o32 = co.COracle(cfg, 50, np.float32)
    o32.set_state(pos, vel, lm)
    o32.step(act_vec=soft)
    _assert_same_bits(_np(env.get_state()['pos']), o32.pos, 'pos')


def test_masked_reset_and_shard_invariance():
    full, cfg = _mk(num_agents=6, num_envs=96, seed=2024)
    o = co.COracle(cfg, 96, np.float32)
    obs = full.reset()
    _assert_same_bits(_np(obs), o.reset(), 'reset')
    # the same 96 envs as two shards (env_id_base) draw the same states
    lo, _ = _mk(num_agents=6, num_envs=40, seed=2024, env_id_base=0)
    hi, _ = _mk(num_agents=6, num_envs=56, seed=2024, env_id_base=40)
    _assert_same_bits(np.concatenate([_np(lo.reset()), _np(hi.reset())]), _np(obs), 'sharded reset')
    # masked reset: only masked-in envs change; their episode counter advances
    mask = (np.arange(96) % 3 == 0).astype(np.uint8)
    before = full.get_state()
    obs2 = full.reset(torch.from_numpy(mask))
    _assert_same_bits(_np(obs2), o.reset(mask), 'masked reset')
    after = full.get_state()
    keep = mask == 0
    assert np.array_equal(_np(after['pos'])[keep], _np(before['pos'])[keep])
    assert (_np(after['pos'])[~keep] != _np(before['pos'])[~keep]).any()
    assert np.array_equal(_np(after['ep_count']), 1 + mask.astype(np.int32))


def test_coincident_agents_propagate_nan_like_upstream(kernel_path):
    env, cfg = _mk(num_agents=3, num_envs=2, max_episode_len=0, want_coll=kernel_path.want_coll, dispatch=kernel_path.dispatch)
    pos = np.array([[[0.1, 0.1], [0.1, 0.1], [0.7, 0.7]], [[0, 0], [0.5, 0.5], [-0.5, 0.5]]], np.float32)
    lm = np.zeros((2, 3, 2), np.float32)
    env.set_state(pos, None, lm)
    env.step(torch.zeros(2, 3, dtype=torch.int32))
    o32 = co.COracle(cfg, 2, np.float32)
    o32.set_state(pos, 0, lm)
    o32.step(act_idx=np.zeros((2, 3), np.int32))
    got = _np(env.get_state()['pos'])
    assert np.isnan(got[0, :2]).all() and np.isnan(o32.pos[0, :2]).all()   # 0/0 in delta/dist, as NumPy
    _assert_same_bits(got, o32.pos, 'pos')


def test_full_size_properties_c2():
    """BASELINE configs[1]: size-independent invariants at B=4096, N=6 over one episode."""
    env, cfg = _mk(num_agents=6, num_envs=4096, max_episode_len=25, auto_reset=True, want_coll=True)
    obs0 = env.reset()
    T = 25
    acts = torch.zeros(T, 4096, 6, dtype=torch.int32, device='cuda')      # no-op: only contact forces act
    st0 = env.get_state()
    out = env.rollout(acts)
    obs, coll = _np(out['obs']), _coll(out['coll'])
    # obs layout (experiments/scenarios.py:6-20): [vel, pos, landmark - pos]
    lm = _np(st0['landmarks'])
    for t in (0, 10, 23):
        pos_t = obs[t, :, :, 2:4]
        want = (lm[:, None, :, :] - pos_t[:, :, None, :]).reshape(4096, 6, 12)
        np.testing.assert_array_equal(obs[t, :, :, 4:], want)
    # momentum: contact forces are pairwise antisymmetric, v' = 0.75 v + F dt  =>  sum_i v_i decays by 0.75
    mom = obs[:24, :, :, 0:2].astype(np.float64).sum(axis=2)
    np.testing.assert_allclose(mom[1:], 0.75 * mom[:-1], rtol=0, atol=2e-5)
    # collision masks: symmetric, self bit always set, and consistent with the reward's -1 terms
    bits = (coll[..., None] >> np.arange(6, dtype=np.uint64)) & np.uint64(1)
    assert (bits == bits.swapaxes(-1, -2)).all() and (np.diagonal(bits, axis1=-2, axis2=-1) == 1).all()
    rew = _np(out['rew']).astype(np.float64)
    shared_term = rew + bits.sum(-1)
    np.testing.assert_allclose(shared_term, shared_term[:, :, :1].repeat(6, 2), atol=2e-5)
    # the episode boundary: step 25 resets every env; post-reset velocities are zero
    assert _np(out['terminal'])[24].all() and not _np(out['terminal'])[:24].any()
    assert not obs[24, :, :, 0:2].any() and obs[23, :, :, 0:2].any()
    assert np.array_equal(_np(env.get_state()['ep_count']), np.full(4096, 2, np.int32))
    assert not _np(out['done']).any()


def test_long_rollout_quad_and_duo_equal_stream_bitwise():
    """Soak: 1500 steps (60 episodes with auto-reset) at C2 size through the four-wave quad kernel (the default there),
    the two-wave duo kernel and the single-wave stream kernel give identical outputs and final state -- the LDS ring /
    barrier hand-offs never drop or reorder a step.  The common result is anchored on the oracle over the first 30 steps."""
    T, B, N = 1500, 4096, 6
    acts = torch.randint(0, 5, (T, B, N), dtype=torch.int32, generator=torch.Generator().manual_seed(9)).cuda()
    outs = {}
    want_kernel = dict(quad='pw_spread_quad_kernel', duo='pw_spread_duo_kernel', stream='pw_spread_stream_kernel')
    for path, disp in (('quad', dict(AUTO)), ('duo', dict(AUTO, quad=0)), ('stream', dict(AUTO, duo=0))):
        env, cfg = _mk(num_agents=N, num_envs=B, max_episode_len=25, auto_reset=True, seed=5, want_coll=False, dispatch=disp)
        env.reset()
        chunks = []
        for s0 in range(0, T, 300):                       # 5 launches of 300 steps
            o = env.rollout(acts[s0:s0 + 300])
            chunks.append((o['obs'].sum(dim=(2, 3)).double().sum(1), o['rew'].double().sum(dim=(1, 2)),
                           o['rew_shared'].double().sum(1), o['terminal'].sum(1), o['obs'][-1].clone(),
                           o['final_obs'][24].clone(), o['obs'][:30].clone() if s0 == 0 else None))
        assert env.last_kernel().startswith(want_kernel[path]), env.last_kernel()
        outs[path] = (chunks, env.get_state())
    (ca, sa) = outs['stream']
    for other in ('quad', 'duo'):
        cb, sb = outs[other]
        for x, y in zip(ca, cb):
            for u, v in zip(x[:6], y[:6]):
                assert torch.equal(u, v), other
        for k in ('pos', 'vel', 'landmarks', 'ep_step', 'ep_count'):
            assert torch.equal(sa[k], sb[k]), (other, k)
    assert int(sa['ep_count'][0]) == 1 + T // 25
    # anchor the common result on the oracle: first 30 steps
    o32 = co.COracle(cfg, B, np.float32)
    o32.reset()
    for t in range(30):
        w = o32.step(act_idx=_np(acts[t]))
        _assert_same_bits(_np(ca[0][6][t]), w['obs'], 'obs[%d]' % t)


def test_device_math_primitives_match_cpu_contract_bitwise():
    """The kernels' float32 building blocks, element by element over ~6M inputs each: correctly rounded
    sqrt (both device forms), IEEE division, pw_exp and both softplus forms against the CPU restatement of
    include/pworld_math.h (oracle/pworld_oracle.c) -- including the exact-zero cut, NaN, inf, subnormals."""
    import ctypes as C
    from multiagent_rl_amd import _lib
    lib = _lib.load()
    rng = np.random.RandomState(0)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def dev(fn, x, aux=1.0):
        xd = torch.from_numpy(x).cuda()
        yd = torch.empty_like(xd)
        assert lib.pw_debug_math(fn, C.c_void_p(xd.data_ptr()), C.c_float(aux), C.c_void_p(yd.data_ptr()), xd.numel(), stream) == 0
        return yd.cpu().numpy()

    special = np.array([0.0, -0.0, 1e-45, 1e-38, 1.17549435e-38, 8.077935669463161e-28, 8.0779e-28, 1.0, 0.09, 0.1509,
                        1.2379400392853803e+27, 1.3e27, 3.4e38, np.inf, np.nan, -1.0, 87.0, -87.0, -86.999, -87.001, 88.0,
                        -300.0, 300.0, 17.0, -17.0], np.float32)
    bits = rng.randint(0, 2 ** 32, 3_000_000, dtype=np.uint64).astype(np.uint32).view(np.float32)     # every exponent
    d2 = np.concatenate([special, np.abs(bits), rng.uniform(0, 0.2, 3_000_000).astype(np.float32)])
    for fn in (0, 4):
        _assert_same_bits(dev(fn, d2), co.math_v(fn, d2), 'sqrt fn=%d' % fn)
    xs = np.concatenate([special, bits, rng.uniform(-100, 320, 3_000_000).astype(np.float32),
                         rng.uniform(-1, 1, 1_000_000).astype(np.float32)])
    for fn in (1, 2):
        _assert_same_bits(dev(fn, xs), co.math_v(fn, xs), 'softplus fn=%d' % fn)
    _assert_same_bits(dev(3, xs), co.math_v(3, xs), 'pw_exp')
    for k in (1e-3, 0.3, 1.0):
        _assert_same_bits(dev(5, xs, k), co.math_v(5, xs, k), 'x / %g' % k)


def test_scaling_free_division_chain_is_ieee_division_bitwise():
    """The hot loops divide through a bare FMA chain (pw_common.hpp div_chain) inside a guarded operand range.
    Inside that range it must BE IEEE division: compared with the device's own a / b and with the CPU's,
    over tens of millions of operand pairs, dense around powers of two and at the range edges; the chain's
    softplus against the branch-free one over every exponent, NaN, inf, the exact-zero cut."""
    import ctypes as C
    from multiagent_rl_amd import _lib
    lib = _lib.load()
    rng = np.random.RandomState(5)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def dev(fn, x, aux=1.0):
        xd = torch.from_numpy(np.ascontiguousarray(x)).cuda()
        yd = torch.empty_like(xd)
        assert lib.pw_debug_math(fn, C.c_void_p(xd.data_ptr()), C.c_float(aux), C.c_void_p(yd.data_ptr()), xd.numel(), stream) == 0
        return yd.cpu().numpy()

    def mags(lo_exp, hi_exp, n):  # random sign, exponent uniform in [lo_exp, hi_exp), random mantissa
        e = rng.randint(lo_exp + 127, hi_exp + 127, n).astype(np.uint32) << 23
        m = rng.randint(0, 1 << 23, n).astype(np.uint32)
        sg = rng.randint(0, 2, n).astype(np.uint32) << 31
        return (sg | e | m).view(np.float32)

    def near_pow2(lo_exp, hi_exp, n):  # mantissas within a few ulps of 1.0 and of 2.0 (rounding-boundary stress)
        e = rng.randint(lo_exp + 127, hi_exp + 127, n).astype(np.uint32) << 23
        m = np.where(rng.randint(0, 2, n) == 0, rng.randint(0, 8, n), (1 << 23) - 1 - rng.randint(0, 8, n)).astype(np.uint32)
        return (e | m).view(np.float32)

    # numerators of the three kinds the kernels produce: -(dist - dist_min) (incl. exact zero), contact_force * delta
    num = np.concatenate([mags(-69, 52, 4_000_000), near_pow2(-69, 52, 500_000), -near_pow2(-69, 52, 500_000),
                          rng.uniform(-0.5, 0.5, 2_000_000).astype(np.float32), np.array([0.0, -0.0], np.float32)])
    for k in (1e-3, 1e-2, 0.3, 1.0, 2.5, 3.0, 7.62939453125e-06, 9.094947017729282e-13, 1099511627776.0,
              0.99999994, 1.9999999, 1.0000001, 35184372088832.0 / 2, 2.842170943040401e-14 * 2):
        k = float(np.float32(k))
        got, want = dev(6, num, k), dev(5, num, k)
        nz = num != 0                       # a zero numerator gives a zero of either sign (callers do not care)
        _assert_same_bits(got[nz], want[nz], 'chain x / %g vs device IEEE' % k)
        assert (got[~nz] == 0).all()
        _assert_same_bits(want, co.math_v(5, num, k), 'device IEEE x / %g vs CPU' % k)
    # divisors: dist in [2^-45, 2^45], and log1p's 2 + t in (2, 3]
    den = np.concatenate([np.abs(mags(-45, 45, 4_000_000)), near_pow2(-45, 45, 500_000),
                          rng.uniform(0.05, 0.6, 2_000_000).astype(np.float32), rng.uniform(2.0, 3.0, 1_000_000).astype(np.float32)])
    for a in (100.0, -100.0, 1.0, 3.0517578125e-05, 8.673617379884035e-19, 4503599627370495.0, 0.33333334, -17.123457,
              1.0000001, 1.9999999):
        a = float(np.float32(a))
        lo, hi = abs(a) / 2.0 ** 100, min(abs(a) * 2.0 ** 100, 3.0e38)     # keep the quotient well inside the normal range
        d = den[(den > lo) & (den < hi)]
        _assert_same_bits(dev(7, d, a), dev(9, d, a), 'chain %g / x vs device IEEE' % a)
        _assert_same_bits(dev(9, d, a), (np.float32(a) / d).astype(np.float32), 'device IEEE %g / x vs CPU' % a)
    # softplus through the chain == the branch-free softplus == the CPU contract, everywhere
    special = np.array([0.0, -0.0, 1e-45, 1e-38, 1.0, -1.0, 87.0, -87.0, -86.999, -87.001, 88.0, -300.0, 300.0, 17.0, -17.0,
                        15.9, 16.0, 16.1, -15.9, -16.0, -16.1, np.inf, -np.inf, np.nan], np.float32)
    bits = rng.randint(0, 2 ** 32, 3_000_000, dtype=np.uint64).astype(np.uint32).view(np.float32)
    xs = np.concatenate([special, bits, rng.uniform(-100, 320, 3_000_000).astype(np.float32),
                         rng.uniform(-20, 20, 3_000_000).astype(np.float32)])
    _assert_same_bits(dev(8, xs), dev(1, xs), 'softplus_fastdiv vs softplus_branchless')
    _assert_same_bits(dev(8, xs), co.math_v(1, xs), 'softplus_fastdiv vs CPU contract')


def test_one_correction_sqrt_is_the_correctly_rounded_sqrt_exhaustively():
    """The hot loops' sqrt (pw_common.hpp sqrt_rn_core: hardware estimate + ONE fused correction) against the device's sqrtf and
    against the two-test form it replaced, over EVERY float32 of its range [2^-90, 2^90): 1.5 x 10^9 arguments; one binade also
    against the CPU's correctly rounded sqrt."""
    import ctypes as C
    from multiagent_rl_amd import _lib
    lib = _lib.load()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    sig = torch.arange(1 << 23, dtype=torch.int32, device='cuda')
    o4, o11, o12 = [torch.empty(1 << 23, device='cuda') for _ in range(3)]
    for e in range(-90, 90):
        x = (sig | ((e + 127) << 23)).view(torch.float32)
        for fn, o in ((4, o4), (11, o11), (12, o12)):
            assert lib.pw_debug_math(fn, C.c_void_p(x.data_ptr()), C.c_float(1.0), C.c_void_p(o.data_ptr()), x.numel(), stream) == 0
        assert torch.equal(o4.view(torch.int32), o11.view(torch.int32)), 'one-correction sqrt, exponent %d' % e
        assert torch.equal(o4.view(torch.int32), o12.view(torch.int32)), 'two-test sqrt, exponent %d' % e
        if e in (-90, -3, 0, 1, 89):
            want = np.sqrt(x.cpu().numpy().astype(np.float64)).astype(np.float32)   # float64 sqrt of a float32, rounded: correctly rounded
            _assert_same_bits(o11.cpu().numpy(), want, 'sqrt vs CPU, exponent %d' % e)


def test_one_correction_division_by_the_contact_margin_is_ieee_division_exhaustively():
    """BASELINE configs[1]'s kernel divides by the contact margin with ONE Newton correction (pw_common.hpp div_chain1) when the
    host decides the margin qualifies (pw_margin_one_correction: its refined reciprocal is the correctly rounded one).  For
    every margin the decision accepts, the chain must BE IEEE division: compared with the device's own x / k over EVERY float32
    significand (2^23), both signs, every exponent of the chain's operand range -- 2 x 10^9 operands per margin.  And the
    decision must refuse what the theorem excludes (a significand of all ones) or what is out of range."""
    import ctypes as C
    from multiagent_rl_amd import _lib
    lib = _lib.load()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.pw_margin_one_correction(C.c_float(1e-3)) == 1           # the canonical margin: what the bench runs
    assert lib.pw_margin_one_correction(C.c_float(0.99999994)) == 0      # significand all ones
    assert lib.pw_margin_one_correction(C.c_float(1e-13)) == 0 and lib.pw_margin_one_correction(C.c_float(2e12)) == 0
    sig = torch.arange(1 << 23, dtype=torch.int32, device='cuda')
    out5, out10 = torch.empty(1 << 23, device='cuda'), torch.empty(1 << 23, device='cuda')
    tried = accepted = 0
    rng = np.random.RandomState(7)
    margins = [1e-3, 1e-2, 0.3, 2.5, 3.0, 7.62939453125e-06, 1.0000001, 1.9999998, 0.1, 5e-4, 2e-3, 1e-4, 0.75] + \
        list(np.exp(rng.uniform(np.log(1e-6), np.log(10.0), 12)))
    for k in margins:
        k = float(np.float32(k))
        tried += 1
        if not lib.pw_margin_one_correction(C.c_float(k)):
            continue
        accepted += 1
        exps = range(-69, 52) if k == float(np.float32(1e-3)) else (-69, -30, -1, 0, 1, 17, 51)
        for e in exps:
            for sign in (0, 1):
                x = (sig | ((e + 127) << 23) | (-(1 << 31) if sign else 0)).view(torch.float32)
                for fn, o in ((5, out5), (10, out10)):
                    assert lib.pw_debug_math(fn, C.c_void_p(x.data_ptr()), C.c_float(k), C.c_void_p(o.data_ptr()), x.numel(), stream) == 0
                assert torch.equal(out5.view(torch.int32), out10.view(torch.int32)), 'x / %r: exponent %d sign %d' % (k, e, sign)
    assert accepted >= 8 and tried - accepted >= 0, (tried, accepted)


@pytest.mark.parametrize('margin, k1', [(1e-3, True), (2e-3, None), (0.0009765624417923391, False), (0.01, None)])
def test_quad_kernel_with_other_contact_margins_matches_oracle_bitwise(margin, k1):
    """The quad kernel's two instantiations of the margin division (one correction where pw_margin_one_correction says so,
    two otherwise -- 0.00097656244 = 0x3A7FFFFF has a significand of all ones and must take the general chain): 30 steps across a
    reset against the float32 oracle built with the same margin."""
    import ctypes as C
    from multiagent_rl_amd import _lib
    from multiagent_rl_amd.env import BatchedParticleEnv
    margin = float(np.float32(margin))
    got_k1 = bool(_lib.load().pw_margin_one_correction(C.c_float(margin)))
    if k1 is not None:
        assert got_k1 == k1
    B, N, T = 777, 6, 30
    env = BatchedParticleEnv('simple_spread', B, num_agents=N, max_episode_len=25, auto_reset=True, seed=3, contact_margin=margin,
                             dispatch=dict(AUTO, quad=1))
    cfg = co.make_config('simple_spread', N, max_episode_len=25, auto_reset=True, seed=3, contact_margin=margin)
    o32 = co.COracle(cfg, B, np.float32)
    _assert_same_bits(_np(env.reset()), o32.reset(), 'reset')
    # crowd the agents so that contacts are frequent
    st = env.get_state()
    env.set_state(st['pos'] * 0.3, st['vel'], st['landmarks'], ep_step=st['ep_step'], ep_count=st['ep_count'])
    o32.set_state(_np(st['pos'] * 0.3), _np(st['vel']), _np(st['landmarks']))
    acts = np.random.RandomState(1).randint(0, 5, (T, B, N)).astype(np.int32)
    out = env.rollout(torch.from_numpy(acts))
    assert env.last_kernel() == ('pw_spread_quad_kernel<true,false,true>' if got_k1 else 'pw_spread_quad_kernel<true>')
    for t in range(T):
        w = o32.step(act_idx=acts[t])
        _assert_same_bits(_np(out['obs'][t]), w['obs'], 'obs[%d]' % t)
        _assert_same_bits(_np(out['rew'][t]), w['rew'], 'rew[%d]' % t)
    _assert_same_bits(_np(env.get_state()['vel']), o32.vel, 'vel')


@pytest.mark.parametrize('case, disp, kernel, kernel_coll', [
    (dict(scenario='simple_spread', num_agents=6, num_envs=4096), {},                  # C2 as the bench runs it
     'pw_spread_quad_kernel<true,false,true>', 'pw_spread_quad_kernel<true,true,true>'),
    (dict(scenario='simple_spread', num_agents=6, num_envs=4096), dict(quad=0),          # C2, two-wave form
     'pw_spread_duo_kernel<6,6,true>', 'pw_spread_duo_kernel<6,6,true,true>'),
    (dict(scenario='simple_tag', num_agents=6, num_adversaries=4, num_envs=8192), {},  # C3 as the bench runs it: three waves
     'pw_tag_duo_kernel<6,4,2,true,false,true>', 'pw_tag_duo_kernel<6,4,2,true,true,true>'),
    (dict(scenario='simple_tag', num_agents=6, num_adversaries=4, num_envs=8192), dict(trio=0),
     'pw_tag_duo_kernel<6,4,2,true,false>', 'pw_tag_duo_kernel<6,4,2,true,true>'),
], ids=['C2', 'C2-duo', 'C3', 'C3-duo'])
def test_bench_path_full_size_every_output_bitwise(case, disp, kernel, kernel_coll):
    """The headline kernels under the DEFAULT dispatch at BASELINE's full sizes, over 27 steps across an auto-reset,
    without and with the optional collision-mask output: EVERY output of both instantiations (observations, rewards,
    shared rewards, terminal / done flags, pre-reset observations, collision masks, final world state) equals the
    float32 oracle bit for bit at every step -- and the dispatcher reports the kernels the bench names."""
    T = 27
    plain, cfg = _mk(max_episode_len=25, auto_reset=True, seed=77, want_coll=False, dispatch=dict(AUTO, **disp), **case)
    env, _ = _mk(max_episode_len=25, auto_reset=True, seed=77, want_coll=True, dispatch=dict(AUTO, **disp), **case)
    B, N = env.num_envs, env.n
    acts = np.random.RandomState(3).randint(0, 5, (T, B, N)).astype(np.int32)
    o32 = co.COracle(cfg, B, np.float32)
    want0 = o32.reset()
    _assert_same_bits(_np(env.reset()), want0, 'reset obs (coll form)')
    _assert_same_bits(_np(plain.reset()), want0, 'reset obs')
    out = env.rollout(torch.from_numpy(acts))
    ref = plain.rollout(torch.from_numpy(acts))
    assert plain.last_kernel() == kernel and env.last_kernel() == kernel_coll, (plain.last_kernel(), env.last_kernel())
    assert 'coll' in out and 'coll' not in ref
    hits = 0
    for t in range(T):
        w = o32.step(act_idx=acts[t])
        for name, o in (('plain', ref), ('coll', out)):
            _assert_same_bits(_np(o['obs'][t]), w['obs'], '%s obs[%d]' % (name, t))
            _assert_same_bits(_np(o['rew'][t]), w['rew'], '%s rew[%d]' % (name, t))
            _assert_same_bits(_np(o['rew_shared'][t]), w['rew_shared'], '%s rew_shared[%d]' % (name, t))
            _assert_same_bits(_np(o['terminal'][t]).astype(np.uint8), w['terminal'], '%s terminal[%d]' % (name, t))
            _assert_same_bits(_np(o['done'][t]).astype(np.uint8), w['done'], '%s done[%d]' % (name, t))
            if w['terminal'].any():
                _assert_same_bits(_np(o['final_obs'][t]), w['final_obs'], '%s final_obs[%d]' % (name, t))
        _assert_same_bits(_coll(out['coll'][t]), w['coll'], 'coll[%d]' % t)
        hits += int((w['coll'] != (np.uint64(1) << np.arange(N, dtype=np.uint64))[None, :]).sum())
    assert hits > 100        # real collisions were seen, not only the self bits
    for e in (env, plain):
        st = e.get_state()
        _assert_same_bits(_np(st['pos']), o32.pos, 'pos')
        _assert_same_bits(_np(st['vel']), o32.vel, 'vel')
        _assert_same_bits(_np(st['landmarks']), o32.lm, 'landmarks')
        assert np.array_equal(_np(st['ep_step']), o32.ep_step)


@pytest.mark.parametrize('case, kernel', [
    (dict(scenario='simple_spread', num_agents=3, num_envs=4096), 'pw_spread_duo_kernel<3,3,true,false,false,true>'),   # three waves, row-wise stores
    (dict(scenario='simple_spread', num_agents=12, num_envs=4096), 'pw_spread_duo_kernel<12,12,true,false,true,true>'),  # three waves
    (dict(scenario='simple_spread', num_agents=24, num_envs=4096), 'pw_spread_duo_kernel<24,24,true,false,true>'),
    (dict(scenario='simple_spread', num_agents=48, num_envs=4096), 'pw_spread_duo_kernel<48,48,true,false,true>'),
    (dict(scenario='simple_spread', num_agents=6, num_envs=16384), 'pw_spread_duo_kernel<6,6,true,false,true>'),
    # the reference's scalability settings off the C5 grid (main_scalability_1.py:30: n_agent in [6, 9, 12]): N = 9 never stores blocks
    # below 700 workgroups and takes the row-wise three-wave form, as N = 12 does on small batches
    (dict(scenario='simple_spread', num_agents=9, num_envs=4096), 'pw_spread_duo_kernel<9,9,true,false,false,true>'),
    (dict(scenario='simple_spread', num_agents=12, num_envs=1024), 'pw_spread_duo_kernel<12,12,true,false,false,true>'),
], ids=['N3', 'N12', 'N24', 'N48', 'B16384', 'N9', 'N12-B1024'])
def test_c5_points_default_dispatch_rollout_across_two_resets_bitwise(case, kernel):
    """BASELINE configs[4] at full size under the DEFAULT dispatch (what bench.py's sweep times): 52 steps across two
    auto-resets, every output and the final state bit-identical to the float32 oracle."""
    T = 52
    env, cfg = _mk(max_episode_len=25, auto_reset=True, seed=31, want_coll=False, dispatch=dict(AUTO), **case)
    B, N = env.num_envs, env.n
    acts = np.random.RandomState(N).randint(0, 5, (T, B, N)).astype(np.int32)
    o32 = co.COracle(cfg, B, np.float32)
    _assert_same_bits(_np(env.reset()), o32.reset(), 'reset obs')
    out = env.rollout(torch.from_numpy(acts))
    assert env.last_kernel() == kernel, env.last_kernel()
    resets = 0
    for t in range(T):
        w = o32.step(act_idx=acts[t])
        _assert_same_bits(_np(out['obs'][t]), w['obs'], 'obs[%d]' % t)
        _assert_same_bits(_np(out['rew'][t]), w['rew'], 'rew[%d]' % t)
        _assert_same_bits(_np(out['rew_shared'][t]), w['rew_shared'], 'rew_shared[%d]' % t)
        _assert_same_bits(_np(out['terminal'][t]).astype(np.uint8), w['terminal'], 'terminal[%d]' % t)
        if w['terminal'].any():
            resets += 1
            _assert_same_bits(_np(out['final_obs'][t]), w['final_obs'], 'final_obs[%d]' % t)
    assert resets == 2
    st = env.get_state()
    _assert_same_bits(_np(st['pos']), o32.pos, 'pos')
    _assert_same_bits(_np(st['vel']), o32.vel, 'vel')
    _assert_same_bits(_np(st['landmarks']), o32.lm, 'landmarks')


def test_c4_partition_full_size_equals_the_unsharded_batch():
    """BASELINE configs[3] (C4): simple_spread N = 6, B = 32768 split over 8 GPUs as contiguous shards of 4096 envs.
    One GPU plays the eight ranks in turn (env_id_base = rank * 4096, its own action slice) and the concatenation of
    their outputs -- reset observations, 30 rollout steps across an auto-reset, final world state -- must equal the
    unsharded B = 32768 run bit for bit: the partition has no data-path coupling and the Philox reset is keyed by the
    GLOBAL env id.  (The exchange that follows the rollout is covered by the gloo world-size tests.)"""
    world, Bs, N, T = 8, 4096, 6, 30
    B = world * Bs
    whole, _ = _mk(num_agents=N, num_envs=B, max_episode_len=25, auto_reset=True, seed=12345678, want_coll=False)
    acts = torch.randint(0, 5, (T, B, N), dtype=torch.int32, generator=torch.Generator().manual_seed(4)).cuda()
    obs0 = whole.reset()
    out = whole.rollout(acts)
    st = whole.get_state()
    for rank in range(world):
        lo, hi = rank * Bs, (rank + 1) * Bs
        shard, _ = _mk(num_agents=N, num_envs=Bs, max_episode_len=25, auto_reset=True, seed=12345678, env_id_base=lo,
                       want_coll=False)
        assert torch.equal(shard.reset(), obs0[lo:hi]), 'reset of shard %d' % rank
        o = shard.rollout(acts[:, lo:hi].contiguous())
        for k in ('obs', 'rew', 'rew_shared', 'terminal', 'done'):
            assert torch.equal(o[k], out[k][:, lo:hi]), (k, rank)
        assert torch.equal(o['final_obs'][24], out['final_obs'][24, lo:hi])
        ss = shard.get_state()
        for k in ('pos', 'vel', 'landmarks', 'ep_step', 'ep_count'):
            assert torch.equal(ss[k], st[k][lo:hi]), (k, rank)
    assert out['terminal'][24].all() and int(st['ep_count'][0]) == 2


@pytest.mark.parametrize('B,ep_len', [(4096, 25), (77, 3), (9, 1), (1000, 2)])
def test_quad_kernel_equals_duo_with_desynchronised_episode_clocks(B, ep_len):
    """pw_spread_quad_kernel keeps ONE ring-slot sequence per workgroup: a step in which any of its 8 envs resets makes
    every env publish two slots.  Envs whose episode clocks are out of step (masked resets, restored checkpoints)
    therefore reset in different -- also consecutive -- steps of one workgroup; episodes of 1, 2 and 3 steps stress the
    4-slot ring.  Every output and the final state must equal the duo kernel's (which is anchored on the oracle)."""
    T, N = 31, 6
    acts = torch.randint(0, 5, (T, B, N), dtype=torch.int32, generator=torch.Generator().manual_seed(B)).cuda()
    res = {}
    for form in ('duo', 'quad'):
        env, cfg = _mk(num_agents=N, num_envs=B, max_episode_len=ep_len, auto_reset=True, seed=41, want_coll=False,
                       dispatch=dict(AUTO, quad=0 if form == 'duo' else 1))
        env.reset()
        st = env.get_state()
        env.set_state(st['pos'], st['vel'], st['landmarks'],
                      ep_step=((torch.arange(B, device='cuda') * 7) % ep_len).int(), ep_count=st['ep_count'])
        out = env.rollout(acts)
        res[form] = (out, env.get_state())
    (oa, sa), (ob, sb) = res['duo'], res['quad']
    for k in ('obs', 'rew', 'rew_shared', 'terminal', 'done'):
        assert torch.equal(oa[k], ob[k]), k
    term = oa['terminal']
    assert term.any() and (ep_len == 1 or not term.all())
    assert torch.equal(oa['final_obs'][term], ob['final_obs'][term])
    for k in ('pos', 'vel', 'landmarks', 'ep_step', 'ep_count'):
        assert torch.equal(sa[k], sb[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize('case, want', [
    (dict(scenario='simple_spread', num_agents=6, num_envs=4096), 'pw_spread_quad_kernel<true,false,true>'),       # C2: the bench path
    (dict(scenario='simple_spread', num_agents=6, num_envs=16384), 'pw_spread_duo_kernel<6,6,true,false,true>'),
    (dict(scenario='simple_spread', num_agents=3, num_envs=4096), 'pw_spread_duo_kernel<3,3,true,false,false,true>'),
    (dict(scenario='simple_spread', num_agents=48, num_envs=4096), 'pw_spread_duo_kernel<48,48,true,false,true>'),
    (dict(scenario='simple_tag', num_agents=6, num_adversaries=4, num_envs=8192), 'pw_tag_duo_kernel<6,4,2,true,false,true>'),   # three-wave form
], ids=['C2', 'B16384', 'N3', 'N48', 'C3'])
def test_dispatcher_reports_the_kernel_it_launched(case, want):
    """pw_rollout_kernel(): bench.py and the profile tools name the dominant kernel from the dispatcher's own record, so the
    record has to be right -- the BASELINE configurations land on the kernels DESIGN.md says they do."""
    env, _ = _mk(max_episode_len=25, auto_reset=True, seed=5, want_coll=False, dispatch=dict(AUTO), **case)
    assert env.last_kernel() == ''            # nothing launched yet on this handle
    env.reset()
    acts = torch.randint(0, 5, (3, env.num_envs, env.n), dtype=torch.int32)
    env.rollout(acts)
    assert env.last_kernel().startswith(want), env.last_kernel()


def test_dispatch_is_frozen_in_the_handle_not_read_from_the_environment_at_launch(monkeypatch):
    """pw_create reads the PWORLD_* overrides ONCE; a later change of the process environment does not reach a live handle
    (pw_step / pw_rollout never call getenv), pw_set_dispatch does, and a bad selection is refused."""
    from multiagent_rl_amd import _lib
    for k in ('PWORLD_EPW', 'PWORLD_FORCE_GENERIC', 'PWORLD_NO_STREAM', 'PWORLD_NO_DUO', 'PWORLD_NO_QUAD', 'PWORLD_FORCE_QUAD',
              'PWORLD_OBS_BLOCK', 'PWORLD_FORCE_DUO', 'PWORLD_SPREAD_TRIO', 'PWORLD_TAG_TRIO', 'PWORLD_P_PRIO'):
        monkeypatch.delenv(k, raising=False)
    clean, _ = _mk(num_agents=6, num_envs=512, max_episode_len=25, auto_reset=True, want_coll=False)
    assert clean.get_dispatch() == AUTO
    monkeypatch.setenv('PWORLD_NO_DUO', '1')
    monkeypatch.setenv('PWORLD_EPW', '64')
    frozen, _ = _mk(num_agents=6, num_envs=512, max_episode_len=25, auto_reset=True, want_coll=False)
    assert frozen.get_dispatch() == dict(AUTO, duo=0, envs_per_wave=64)
    monkeypatch.delenv('PWORLD_NO_DUO')
    monkeypatch.delenv('PWORLD_EPW')
    acts = torch.randint(0, 5, (3, 512, 6), dtype=torch.int32)
    for e in (clean, frozen):
        e.reset()
    a, b = clean.rollout(acts), frozen.rollout(acts)
    assert clean.last_kernel().startswith('pw_spread_quad_kernel') and frozen.last_kernel().startswith('pw_spread_stream_kernel')
    assert all(torch.equal(a[k], b[k]) for k in ('obs', 'rew', 'rew_shared', 'terminal'))
    frozen.set_dispatch(duo=-1, envs_per_wave=0)
    frozen.rollout(acts)
    assert frozen.last_kernel().startswith('pw_spread_quad_kernel')
    with pytest.raises(_lib.PworldError):
        frozen.set_dispatch(duo=7)
    with pytest.raises(TypeError):
        frozen.set_dispatch(no_such_field=1)


def test_output_planes_beyond_four_gibi_elements():
    """Sized for the card (288 GB): B = 262144 envs x 200 steps of simple_spread N = 6 write observation planes of 5.03e9 floats (20 GB each) --
    element offsets past 2^32.  Envs from both ends of the batch must equal single-env shards (env_id_base = e, the same actions: the partition
    property of C4) at every step, for the synthetic-action rollout AND for the policy-in-the-loop rollout replayed on its own sampled actions."""
    from multiagent_rl_amd.policy import ActorNetwork, FusedActor
    B, N, T = 262144, 6, 200
    free = torch.cuda.mem_get_info()[0]
    if free < 120e9:
        pytest.skip('needs ~100 GB of device memory')
    picks = (0, 1, 131077, B - 2, B - 1)
    env, _ = _mk(num_agents=N, num_envs=B, max_episode_len=25, auto_reset=True, seed=12345678, want_coll=False)
    assert T * B * N * env.obs_dim > 2 ** 32
    acts = torch.randint(0, 5, (T, B, N), dtype=torch.int32, device='cuda', generator=torch.Generator(device='cuda').manual_seed(9))
    obs0 = env.reset()
    out = env.rollout(acts)
    torch.cuda.synchronize()
    assert out['obs'].numel() > 2 ** 32 and out['terminal'][24].all() and out['terminal'][199].all()

    def check(big, big_obs0, actions, what):
        for e in picks:
            shard, _ = _mk(num_agents=N, num_envs=1, max_episode_len=25, auto_reset=True, seed=12345678, env_id_base=e, want_coll=False)
            assert torch.equal(shard.reset(), big_obs0[e:e + 1]), (what, 'reset', e)
            o = shard.rollout(actions[:, e:e + 1].contiguous())
            for k in ('obs', 'rew', 'rew_shared', 'terminal', 'final_obs'):
                got, want = big[k][:, e:e + 1], o[k]
                if k == 'final_obs':   # defined where an episode ended
                    m = o['terminal'][:, 0]
                    got, want = got[m], want[m]
                assert torch.equal(got, want), (what, k, e)

    check(out, obs0, acts, 'rollout')
    del out, acts
    torch.cuda.empty_cache()
    # the policy in the loop at the same size (one launch per 100 steps), replayed through single-env shards on its own actions
    torch.manual_seed(0)
    penv, _ = _mk(num_agents=N, num_envs=B, max_episode_len=25, auto_reset=True, seed=12345678, want_coll=False)
    actor = FusedActor(ActorNetwork(penv.obs_dim, 5).cuda().eval(), seed=5)
    pobs0 = penv.reset().clone()
    pout = {k: torch.empty((T,) + tuple(s), dtype=d, device='cuda') for k, s, d in (
        ('obs', (B, N, penv.obs_dim), torch.float32), ('final_obs', (B, N, penv.obs_dim), torch.float32), ('rew', (B, N), torch.float32),
        ('rew_shared', (B,), torch.float32), ('terminal', (B,), torch.bool), ('done', (B, N), torch.bool), ('act', (B, N), torch.int32))}
    for c in range(T // 100):
        actor.rollout(penv, 100, {k: v[c * 100:(c + 1) * 100] for k, v in pout.items()})
    torch.cuda.synchronize()
    assert int(pout['act'].min()) >= 0 and int(pout['act'].max()) <= 4
    check(pout, pobs0, pout['act'], 'policy rollout')
