"""GPU: the pieces either side of the step kernels -- device replay ring, transition packing,
the B = 1 MultiAgentEnv drop-in, the reference-shaped run() loop on the HIP env, BatchedRollout."""
import json
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')

from oracle import particle_oracle as po  # noqa: E402  (checker only)
from tests.trace_util import RecordingEnv, RecordingMemory, StubTrainer  # noqa: E402

GOLD_DIR = os.path.join(os.path.dirname(__file__), 'golden')


def test_replay_buffer_matches_reference_semantics():
    """tests/golden/replay_buffer.json was produced by rls.replay_buffer.ReplayBuffer itself."""
    from multiagent_rl_amd.replay_buffer import ReplayBuffer
    g = json.load(open(os.path.join(GOLD_DIR, 'replay_buffer.json')))
    rb = ReplayBuffer(g['size'])
    for tr in g['transitions']:
        rb.add([np.array(o) for o in tr['obs']], [np.array(a) for a in tr['act']], tr['rew'],
               [np.array(o) for o in tr['next_obs']], tr['done'])
    assert len(rb) == g['len'] and rb._next_idx == g['next_idx']
    np.testing.assert_allclose(rb.rew.cpu().numpy(), g['storage_rewards'], rtol=1e-6)   # ring overwrite order
    random.seed(0)
    idx = rb.make_index(4)
    assert idx == g['make_index_seed0']
    enc = rb.sample_index(idx)
    assert [list(e.shape) for e in enc] == g['encode_shapes']
    for got, want in zip(enc, g['encode']):
        np.testing.assert_allclose(got.cpu().numpy(), np.asarray(want), rtol=1e-6, atol=1e-6)
    big = ReplayBuffer(1e6, 3, 10)
    big._len = 8
    random.seed(0)
    assert big.make_index(4) == g['make_index_seed0_len8_batch4'] == [6, 6, 0, 4]
    allb = rb.collect()
    assert allb[0].shape[0] == 5


def test_device_side_index_draw_is_uniform_over_the_filled_ring_and_feeds_sample_index():
    """ReplayBuffer(device_index=True): make_index is one torch.randint on the device (the reference draws batch_size times from
    Python's never-seeded random: rls/replay_buffer.py:51-52 -- uniform with replacement either way); the tensor goes into
    sample_index as it is.  The default keeps the list of the golden test above."""
    import pickle
    from multiagent_rl_amd.replay_buffer import ReplayBuffer
    rb = ReplayBuffer(5000, 3, 10, device_index=True)
    obs = torch.arange(3000, dtype=torch.float32, device='cuda')[:, None, None].expand(3000, 3, 10).contiguous()
    rb.add_batch(obs, torch.zeros(3000, 3, dtype=torch.int32, device='cuda'), torch.zeros(3000, device='cuda'), obs + 0.5)
    assert len(rb) == 3000
    idx = rb.make_index(200000)
    assert torch.is_tensor(idx) and idx.is_cuda and idx.dtype == torch.int64 and int(idx.min()) == 0 and int(idx.max()) == 2999
    counts = torch.bincount(idx, minlength=3000).float()
    assert abs(float(counts.mean()) - 200000 / 3000) < 1e-3 and float(counts.std()) < 2.0 * (200000 / 3000) ** 0.5   # Poisson-like spread
    o, a, r, n, d = rb.sample_index(idx[:1024])
    assert torch.equal(o[:, 0, 0], idx[:1024].float()) and torch.equal(n[:, 0, 0], idx[:1024].float() + 0.5)
    rb2 = pickle.loads(pickle.dumps(rb))
    assert rb2.device_index and len(rb2) == 3000
    plain = ReplayBuffer(10, 3, 10)
    plain._len = 5
    assert isinstance(plain.make_index(4), list)


def test_add_batch_ring_and_pre_reset_next_obs():
    from multiagent_rl_amd.replay_buffer import ReplayBuffer
    rb = ReplayBuffer(10, 2, 6)
    rng = np.random.RandomState(0)
    log = []
    for k in range(4):
        B = 4
        obs, nxt, fin = (torch.from_numpy(rng.randn(B, 2, 6).astype(np.float32)) for _ in range(3))
        act = torch.from_numpy(rng.randint(0, 5, (B, 2)).astype(np.int32))
        rew = torch.from_numpy(rng.randn(B).astype(np.float32))
        term = torch.tensor([False, True, False, True])
        rb.add_batch(obs.cuda(), act.cuda(), rew.cuda(), nxt.cuda(), fin.cuda(), term.cuda())
        for b in range(B):
            log.append((obs[b], act[b], rew[b], fin[b] if term[b] else nxt[b]))
    assert len(rb) == 10 and rb._next_idx == 16 % 10
    o, a, r, n, d = rb.sample_index(list(range(10)))
    for slot in range(10):
        src = log[slot + 10] if slot < 6 else log[slot]     # slots 0..5 were overwritten by transitions 10..15
        assert torch.equal(o[slot].cpu(), src[0]) and torch.equal(n[slot].cpu(), src[3])
        assert torch.equal(a[slot].cpu().argmax(-1).int(), src[1]) and r[slot].item() == src[2].item()
        assert (a[slot].sum(-1) == 1).all() and d[slot].item() == 0.0


def test_pack_transitions_and_packed_ingest():
    from tests.dist_standins import pack_reference
    from multiagent_rl_amd.dist import SampledTransitionGather
    from multiagent_rl_amd.env import BatchedParticleEnv
    env = BatchedParticleEnv('simple_spread', 64, num_agents=3, max_episode_len=5, auto_reset=True)
    env.reset()
    T = 12
    acts = torch.randint(0, 5, (T, 64, 3), device='cuda', dtype=torch.int32)
    out = env.rollout(acts)
    gat = SampledTransitionGather.__new__(SampledTransitionGather)   # single process: no process group needed
    gat.rank, gat.world, gat.device, gat.B, gat.N, gat.D = 0, 1, torch.device('cuda', 0), 64, 3, env.obs_dim
    gat.R, gat._gen, gat._seed = 40, None, 1
    sel_t, sel_e = gat._selection(T)
    rows = torch.zeros(40, 2 * 3 * env.obs_dim + 3 + 2, device='cuda')
    gat._pack(out, acts, sel_t, sel_e, rows)
    cpu = {k: v.cpu() for k, v in out.items()}
    want = pack_reference(cpu, acts.cpu(), sel_t.cpu(), sel_e.cpu())
    assert torch.equal(rows.cpu(), want)
    assert cpu['terminal'][sel_t.cpu().long(), sel_e.cpu().long()].any()   # some rows used final_obs
    gat.memory = gat._make_memory()
    gat._ingest(rows)
    gat._ingest(rows[:7])
    m = gat.memory
    assert len(m) == 47 and m._next_idx == 47
    o, a, r, n, d = m.sample_index(list(range(40)))
    nd = 3 * env.obs_dim
    assert torch.equal(o.reshape(40, -1).cpu(), want[:, :nd]) and torch.equal(n.reshape(40, -1).cpu(), want[:, nd:2 * nd])
    assert torch.equal(a.argmax(-1).float().cpu(), want[:, 2 * nd:2 * nd + 3]) and torch.equal(r.cpu(), want[:, -2])


def test_fused_exchange_launch_equals_pack_then_ingest():
    """pw_exchange (one launch: append previous rows + pack new rows) == pw_replay_add_packed + pw_pack_transitions."""
    from tests.dist_standins import pack_reference
    from multiagent_rl_amd.dist import SampledTransitionGather
    from multiagent_rl_amd.env import BatchedParticleEnv
    env = BatchedParticleEnv('simple_spread', 128, num_agents=6, max_episode_len=7, auto_reset=True)
    env.reset()
    acts = torch.randint(0, 5, (20, 128, 6), device='cuda', dtype=torch.int32)
    out = env.rollout(acts)
    g = SampledTransitionGather.__new__(SampledTransitionGather)
    g.rank, g.world, g.device, g.B, g.N, g.D = 0, 1, torch.device('cuda', 0), 128, 6, env.obs_dim
    g.R, g._gen, g._seed = 96, None, 5
    g.memory = g._make_memory()
    sel_t, sel_e = g._selection(20)
    W = 2 * 6 * env.obs_dim + 6 + 2
    prev = torch.randn(200, W, device='cuda')
    prev[:, 2 * 6 * env.obs_dim:2 * 6 * env.obs_dim + 6] = torch.randint(0, 5, (200, 6), device='cuda').float()
    rows = torch.zeros(96, W, device='cuda')
    g.memory._next_idx, g.memory._len = 999_990, 999_990          # wraps around the ring end
    g._ingest_and_pack(prev, out, acts, sel_t, sel_e, rows)
    want = pack_reference({k: v.cpu() for k, v in out.items()}, acts.cpu(), sel_t.cpu(), sel_e.cpu())
    assert torch.equal(rows.cpu(), want)
    m = g.memory
    assert m._next_idx == 190 and len(m) == 1_000_000
    idx = [(999_990 + i) % 1_000_000 for i in range(200)]
    o, a, r, n, d = m.sample_index(idx)
    nd = 6 * env.obs_dim
    assert torch.equal(o.reshape(200, -1), prev[:, :nd]) and torch.equal(n.reshape(200, -1), prev[:, nd:2 * nd])
    assert torch.equal(a.argmax(-1).float(), prev[:, 2 * nd:2 * nd + 6]) and torch.equal(r, prev[:, -2]) and torch.equal(d, prev[:, -1])


def _drive_pair(scenario, steps, seed, **kw):
    """The HIP MultiAgentEnv and the float64 scalar oracle env under the same NumPy seed and actions."""
    from multiagent_rl_amd import make_env
    np.random.seed(seed)
    gpu = make_env(scenario, **kw)
    np.random.seed(seed)
    ref = po.make_oracle_env(scenario, **kw)
    np.random.seed(seed)
    o_gpu = gpu.reset()
    np.random.seed(seed)
    o_ref = ref.reset()
    rng = np.random.RandomState(1)
    worst = 0.0
    for t in range(steps):
        for a, b in zip(o_gpu, o_ref):
            assert a.shape == b.shape and a.dtype == np.float64
            worst = max(worst, float(np.abs(a - b).max()))
        idx = rng.randint(0, 5, gpu.n)
        acts = [np.eye(5)[i] for i in idx]
        o_gpu, r_gpu, d_gpu, i_gpu = gpu.step([a.copy() for a in acts])
        o_ref, r_ref, d_ref, i_ref = ref.step([a.copy() for a in acts])
        assert d_gpu == d_ref == [False] * gpu.n and i_gpu == i_ref == {'n': [{}] * gpu.n}
        assert all(isinstance(r, float) for r in r_gpu)
        np.testing.assert_allclose(r_gpu, r_ref, atol=2e-4)
    return gpu, ref, worst


def test_multiagentenv_dropin_tracks_float64_oracle():
    """configs[0]: simple_spread, 3 agents, 1 env, seed protocol of main.py:41-49."""
    gpu, ref, worst = _drive_pair('simple_spread', 25, 12345678)
    assert gpu.n == 3 and gpu.observation_space[0].shape == (10,) and gpu.action_space[0].n == 5
    assert not hasattr(gpu.action_space[0], 'high')
    assert worst < 1e-4   # float32 trajectory vs float64 over an episode (per-step bound is 1e-5, see parity tests)
    gpu6, _, worst6 = _drive_pair('simple_spread', 25, 12345679, n=6)
    assert gpu6.observation_space[0].shape == (16,) and worst6 < 1e-4
    tag, _, worst_t = _drive_pair('simple_tag', 25, 5)
    assert [s.shape[0] for s in tag.observation_space] == [16, 16, 16, 14] and worst_t < 1e-4


def test_reset_matches_numpy_legacy_stream_kat():
    from multiagent_rl_amd import make_env
    env = make_env('simple_spread')
    np.random.seed(12345678)
    obs = env.reset()
    # SURVEY.md 8(c) KAT-reset, rounded to float32 by the upload
    want_pos0 = np.float32([-0.5083915323501949, 0.19285721064295336])
    want_lm0 = np.float32([-0.22413133239983196, 0.3610820860587127])
    np.testing.assert_array_equal(obs[0][2:4].astype(np.float32), want_pos0)
    np.testing.assert_array_equal(obs[0][4:6].astype(np.float32), want_lm0 - want_pos0)
    assert not obs[0][:2].any()


def test_benchmark_info_and_full_observation():
    from multiagent_rl_amd import make_env
    np.random.seed(3)
    env = make_env('simple_spread', benchmark=True, local_observation=False)
    np.random.seed(3)
    ref = po.make_oracle_env('simple_spread', benchmark=True, local_observation=False)
    np.random.seed(3); env.reset()
    np.random.seed(3); ref.reset()
    assert env.observation_space[0].shape == (18,)
    a = [np.eye(5)[1] for _ in range(3)]
    _, _, _, info = env.step([x.copy() for x in a])
    _, _, _, info_ref = ref.step([x.copy() for x in a])
    for got, want in zip(info['n'], info_ref['n']):
        assert got[1] == want[1] and got[3] == want[3]
        np.testing.assert_allclose([got[0], got[2]], [want[0], want[2]], atol=1e-4)


def test_run_loop_on_hip_env_has_reference_call_shape(tmp_path):
    """rollout.run (mirror of experiments/run.py) on the HIP MultiAgentEnv: same event kinds, shapes and
    dtypes as the golden reference trace; values agree with the float64 oracle run to float32 accuracy."""
    from multiagent_rl_amd import make_env, rollout
    gold = json.load(open(os.path.join(GOLD_DIR, 'run_trace.json')))

    class Args(object):
        is_training, display = True, False
    for k, v in gold['arglist'].items():
        setattr(Args, k, v)
    np.random.seed(12345678)
    env = RecordingEnv(make_env('simple_spread'))
    np.random.seed(12345678)
    StubTrainer.trace = env.trace
    hist = rollout.run(env, None, None, StubTrainer, 'simple_spread', 'Discrete', cnt=0, arglist=Args,
                       memory=RecordingMemory(), out_dir=str(tmp_path), log=lambda *a: None)

    def skeleton(x):
        if isinstance(x, dict):
            return {k: skeleton(v) for k, v in x.items() if k not in ('sum', 'value')}
        if isinstance(x, list):
            return [skeleton(v) for v in x]
        return x
    assert skeleton(json.loads(json.dumps(env.trace))) == skeleton(gold['trace'])
    np.testing.assert_allclose(hist['reward_episodes'], gold['reward_episodes'], rtol=1e-4, atol=1e-3)


def test_batched_rollout_feeds_replay():
    from multiagent_rl_amd import make_batched_env
    from multiagent_rl_amd.policy import ActorNetwork, GumbelPolicy
    from multiagent_rl_amd.replay_buffer import ReplayBuffer
    from multiagent_rl_amd.rollout import BatchedRollout
    torch.manual_seed(0)
    env = make_batched_env('simple_spread', 256, n=6, auto_reset=True, max_episode_len=25)
    actor = ActorNetwork(env.obs_dim, 5).cuda()
    mem = ReplayBuffer(256 * 30, 6, env.obs_dim)
    ro = BatchedRollout(env, GumbelPolicy(actor), mem)
    first_obs = ro.obs.clone()
    ro.collect(27)
    st = ro.stats()
    assert st['env_steps'] == 27 * 256 and st['episodes'] == 256 and len(mem) == 27 * 256
    assert -2000 < st['mean_episode_reward'] < -100      # ~ 6 agents x 25 steps x -(sum of 6 min-dists + 1)
    o, a, r, n, d = mem.sample_index(list(range(256)))          # the first batched step's transitions
    assert torch.equal(o, first_obs) and (a.sum(-1) == 1).all() and not d.any()
    # transition 24 (the terminal step) stores the PRE-reset next_obs: velocities are non-zero there
    o24, _, _, n24, _ = mem.sample_index(list(range(24 * 256, 25 * 256)))
    assert n24[:, :, :2].abs().sum() > 0
    o25, _, _, _, _ = mem.sample_index(list(range(25 * 256, 26 * 256)))
    assert o25[:, :, :2].abs().sum() == 0                        # first obs of the next episode: at rest


def test_fused_actor_matches_pytorch_float32_reference():
    """HIP BiLSTM + head vs plain PyTorch fp32 (CPU) on the reference's own weights (golden state_dict)."""
    from multiagent_rl_amd.policy import ActorNetwork, FusedActor
    g = np.load(os.path.join(GOLD_DIR, 'actor_forward.npz'))
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('sd/')}
    ref = ActorNetwork(16, 5)
    ref.load_state_dict(sd)
    fused = FusedActor(ActorNetwork(16, 5).cuda().eval())
    fused.actor.load_state_dict(sd)
    fused.refresh()
    np.testing.assert_allclose(fused.logits(torch.from_numpy(g['obs']).cuda()).cpu().numpy(), g['logits'],
                               rtol=0, atol=2e-5)                                     # the reference's own logits
    torch.manual_seed(1)
    for B, N in [(1, 3), (257, 6), (1000, 12), (33, 1)]:
        obs = torch.randn(B, N, 16) * 2
        with torch.no_grad():
            want = ref(obs)
            hid = torch.relu(ref.bilstm(torch.relu(ref.dense1(obs)))[0])
        np.testing.assert_allclose(fused.hidden(obs.cuda()).cpu().numpy(), hid.numpy(), rtol=0, atol=2e-5)
        np.testing.assert_allclose(fused.logits(obs.cuda()).cpu().numpy(), want.numpy(), rtol=0, atol=2e-5)


def test_bf16x3_input_projection_mode_within_the_float32_tolerance():
    """The OPT-IN, not exact actor mode (pw_actor_set_bf16x3 / pw_set_actor_precision: LSTM input projection on bfloat16 matrix
    instructions, operands split in high and low halves, three products per k step) against the reference's own logits
    (tests/golden/actor_forward.npz) and plain PyTorch fp32 at the 2e-5 bound the exact form is held to; it must differ from
    the exact form somewhere (else the switch did nothing); and the exact form is back, bit for bit, once it is off."""
    from multiagent_rl_amd import _lib
    from multiagent_rl_amd.policy import ActorNetwork, FusedActor
    lib = _lib.load()
    g = np.load(os.path.join(GOLD_DIR, 'actor_forward.npz'))
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('sd/')}
    ref = ActorNetwork(16, 5)
    ref.load_state_dict(sd)
    fused = FusedActor(ActorNetwork(16, 5).cuda().eval())
    fused.actor.load_state_dict(sd)
    fused.refresh()
    gobs = torch.from_numpy(g['obs']).cuda()
    exact = fused.logits(gobs).clone()
    assert lib.pw_actor_set_bf16x3(1) == 0
    try:
        got = fused.logits(gobs)
        np.testing.assert_allclose(got.cpu().numpy(), g['logits'], rtol=0, atol=2e-5)
        assert not torch.equal(got, exact)
        torch.manual_seed(1)
        worst = 0.0
        for B, N in [(1, 3), (257, 6), (1000, 12), (33, 1), (4096, 6)]:
            obs = torch.randn(B, N, 16) * 2
            with torch.no_grad():
                want = ref(obs)
            got = fused.logits(obs.cuda()).cpu()
            np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=0, atol=2e-5)
            worst = max(worst, float((got - want).abs().max()))
        assert worst > 0.0
    finally:
        assert lib.pw_actor_set_bf16x3(0) == 1
    assert torch.equal(fused.logits(gobs), exact)


def test_bf16x3_mode_in_the_one_launch_rollout():
    """pw_set_actor_precision on a handle: the one-launch policy rollout runs in the bf16x3 mode (outputs finite, episode
    statistics those of the exact form to sampling noise) and returns to the exact bits when the mode is switched back."""
    from multiagent_rl_amd import make_batched_env
    from multiagent_rl_amd.policy import ActorNetwork, FusedActor
    torch.manual_seed(4)
    mk = lambda: make_batched_env('simple_spread', 2048, n=6, auto_reset=True, max_episode_len=25, seed=21)  # noqa: E731
    actor = ActorNetwork(16, 5).cuda().eval()
    outs = {}
    for mode in ('f32', 'bf16x3', 'f32-again'):
        env = mk()
        env.reset()
        assert env.get_actor_precision() == 'f32'
        if mode == 'bf16x3':
            env.set_actor_precision('bf16x3')
            assert env.get_actor_precision() == 'bf16x3'
        out = FusedActor(actor, seed=9).rollout(env, 100)
        outs[mode] = {k: v.clone() for k, v in out.items() if torch.is_tensor(v)}
    for k in outs['f32']:
        if k != 'final_obs':            # written at terminal steps only
            assert torch.equal(outs['f32'][k], outs['f32-again'][k]), k
    tag = make_batched_env('simple_tag', 64, num_adversaries=4, num_good=2, auto_reset=True, max_episode_len=25, seed=1)
    tag.reset()
    tag.set_actor_precision('bf16x3')
    with pytest.raises(Exception, match='simple_spread'):   # refused, not silently run in float32
        FusedActor(ActorNetwork(tag.obs_dim, 5).cuda().eval(), seed=2).rollout(tag, 10)
    a, b = outs['f32'], outs['bf16x3']
    assert torch.isfinite(b['obs']).all() and torch.isfinite(b['rew_shared']).all()
    same = (a['act'] == b['act']).float().mean().item()
    assert 0.5 < same <= 1.0            # the first steps agree; trajectories then part wherever one arg-max flips
    assert (a['act'][0] == b['act'][0]).float().mean().item() > 0.999
    assert abs(a['rew_shared'].mean().item() - b['rew_shared'].mean().item()) < 0.05 * abs(a['rew_shared'].mean().item())


def test_fused_gumbel_sampling_distribution():
    """act = argmax(logits + Gumbel): frequencies follow softmax(logits); streams differ per call and row."""
    from multiagent_rl_amd.policy import ActorNetwork, FusedActor
    torch.manual_seed(0)
    fused = FusedActor(ActorNetwork(16, 5).cuda().eval(), seed=123)
    obs = torch.randn(1, 2, 16).cuda().repeat(20000, 1, 1)        # same two rows, 20000 independent draws
    p = torch.softmax(fused.logits(obs[:1]), -1)[0].cpu().numpy()
    a = fused(obs).cpu().numpy()
    b = fused(obs).cpu().numpy()
    assert a.dtype == np.int32 and a.min() >= 0 and a.max() <= 4 and (a != b).any()
    for ag in range(2):
        freq = np.bincount(a[:, ag], minlength=5) / 20000.0
        assert np.abs(freq - p[ag]).max() < 0.015, (freq, p[ag])
    fused2 = FusedActor(fused.actor, seed=123)
    assert np.array_equal(fused2(obs).cpu().numpy(), a)            # same seed + call index => same actions


def test_hipgraph_rollout_equals_eager_rollout():
    """BatchedRollout.capture(): policy + fused env step + replay append replayed as ONE hipGraph
    (device-side ring cursor and Philox step) fills the ring with exactly what the eager loop stores."""
    from multiagent_rl_amd import make_batched_env
    from multiagent_rl_amd.policy import ActorNetwork, FusedActor
    from multiagent_rl_amd.replay_buffer import ReplayBuffer
    from multiagent_rl_amd.rollout import BatchedRollout
    torch.manual_seed(0)
    actor = ActorNetwork(16, 5).cuda().eval()
    rings = []
    for graph in (False, True):
        env = make_batched_env('simple_spread', 192, n=6, auto_reset=True, max_episode_len=5, seed=11)
        mem = ReplayBuffer(192 * 8, 6, env.obs_dim)          # wraps: 12 steps into 8 slots-of-B
        ro = BatchedRollout(env, FusedActor(actor, seed=7), mem)
        if graph:
            ro.capture(steps_per_replay=2)                    # 2 warm-up steps ran eagerly inside capture()
            ro.collect(10)
        else:
            ro.collect(12)
        st = ro.stats()
        assert st['env_steps'] == 12 * 192 and st['episodes'] == 2 * 192 and len(mem) == 192 * 8
        assert mem._next_idx == (12 * 192) % (192 * 8)
        rings.append((mem.obs.clone(), mem.next_obs.clone(), mem.act.clone(), mem.rew.clone(), st['mean_episode_reward'],
                      ro.obs.clone()))
    for a, b in zip(rings[0][:4], rings[1][:4]):
        assert torch.equal(a, b)
    assert rings[0][4] == rings[1][4] and torch.equal(rings[0][5], rings[1][5])


def test_pw_dense_matches_torch():
    """pw_dense (weight-stationary skinny dense layer) vs torch fp32 for templated and runtime K, 64..256 outputs."""
    import ctypes as C
    from multiagent_rl_amd import _lib
    lib = _lib.load()
    torch.manual_seed(3)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for rows, K, out, relu in [(1, 16, 64, 1), (1000, 16, 64, 1), (777, 10, 64, 1), (513, 13, 128, 0),
                               (300, 64, 256, 0), (4099, 22, 64, 1)]:
        x, w, b = torch.randn(rows, K).cuda(), torch.randn(out, K).cuda() * 0.3, torch.randn(out).cuda()
        y = torch.full((rows, out), float('nan'), device='cuda')
        assert lib.pw_dense(p(x), p(w), p(b), rows, K, out, relu, p(y), stream) == 0
        want = x.double() @ w.double().t() + b.double()
        want = torch.relu(want) if relu else want
        np.testing.assert_allclose(y.cpu().numpy(), want.float().cpu().numpy(), rtol=0, atol=2e-5)
    assert lib.pw_dense(p(x), p(w), p(b), 10, 65, 64, 1, p(y), stream) == -1      # in_dim > 64
    assert lib.pw_dense(p(x), p(w), p(b), 10, 16, 96, 1, p(y), stream) == -1      # out_dim not a multiple of 64


def test_pw_actor_front_mfma_matches_torch():
    """G = relu(X W1^T + b1) Wih^T + bih on the matrix cores vs float64 torch, several in_dims and ragged row counts."""
    import ctypes as C
    from multiagent_rl_amd import _lib
    lib = _lib.load()
    torch.manual_seed(5)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for rows, D in [(1, 16), (31, 16), (128, 16), (24576, 16), (1000, 10), (333, 21), (129, 22), (64, 1), (200, 64)]:
        x = (torch.randn(rows, D) * 2).cuda()
        w1, b1 = (torch.randn(64, D) * 0.4).cuda(), torch.randn(64).cuda()
        wih, bih = (torch.randn(256, 64) * 0.2).cuda(), torch.randn(256).cuda()
        g = torch.full((rows, 256), float('nan'), device='cuda')
        frag = torch.empty(lib.pw_actor_front_pack_floats(D), device='cuda')
        assert lib.pw_actor_front_pack(p(w1), p(wih), D, p(frag), stream) == 0
        assert lib.pw_actor_front(p(x), p(frag), p(b1), p(bih), rows, D, p(g), stream) == 0
        want = torch.relu(x.double() @ w1.double().t() + b1.double()) @ wih.double().t() + bih.double()
        np.testing.assert_allclose(g.cpu().numpy(), want.float().cpu().numpy(), rtol=0, atol=5e-5,
                                   err_msg='rows=%d D=%d' % (rows, D))
    # asymmetric-weight identity check (catches row/col swaps): W1 = [I | 0], Wih picks hidden unit (u % 64)
    D = 16
    x = torch.rand(70, D).cuda() + 0.5
    w1 = torch.zeros(64, D).cuda(); w1[:D] = torch.eye(D).cuda()
    wih = torch.zeros(256, 64).cuda(); wih[torch.arange(256), torch.arange(256) % 64] = torch.arange(1, 257).float().cuda()
    g = torch.empty(70, 256, device='cuda')
    z = torch.zeros(256).cuda()
    frag = torch.empty(lib.pw_actor_front_pack_floats(D), device='cuda')
    assert lib.pw_actor_front_pack(p(w1), p(wih), D, p(frag), stream) == 0
    assert lib.pw_actor_front(p(x), p(frag), p(z[:64].contiguous()), p(z), 70, D, p(g), stream) == 0
    want = torch.zeros(70, 256).cuda()
    for u in range(256):
        if u % 64 < D:
            want[:, u] = x[:, u % 64] * (u + 1)
    assert torch.equal(g, want)


def test_one_launch_actor_equals_three_launch_chain(monkeypatch):
    """pw_actor_fused (everything in LDS) vs pw_actor_front + pw_bilstm_forward + pw_actor_head chained: the
    hidden state, the logits and the sampled actions must be IDENTICAL (same arithmetic, same Philox keying),
    for ragged batches (last workgroup partly filled), N that does not divide 96, and several in_dims."""
    from multiagent_rl_amd.policy import ActorNetwork, FusedActor
    torch.manual_seed(9)
    for D in (16, 10, 21, 40):
        actor = ActorNetwork(D, 5).cuda().eval()
        one = FusedActor(actor, seed=77)
        monkeypatch.setenv('PW_ACTOR_NO_FUSE', '1')
        three = FusedActor(actor, seed=77)
        monkeypatch.delenv('PW_ACTOR_NO_FUSE')
        assert one.use_fused and not three.use_fused
        for B, N in [(1, 1), (1, 6), (16, 6), (17, 6), (4096, 6), (100, 3), (33, 7), (5, 48), (9, 96), (70, 2), (3, 50)]:
            obs = (torch.randn(B, N, D) * 2).cuda()
            assert torch.equal(one.hidden(obs), three.hidden(obs)), (D, B, N)
            assert torch.equal(one.logits(obs), three.logits(obs)), (D, B, N)
            a1, a3 = one(obs), three(obs)
            assert a1.dtype == torch.int32 and torch.equal(a1, a3), (D, B, N)
    # N > 96 falls back to the chain
    obs = torch.randn(2, 100, 40).cuda()
    assert torch.equal(one(obs), three(obs))


@pytest.mark.parametrize('form', ['default', 'v3', 'v3j'])
@pytest.mark.parametrize('B,N,T', [(16, 6, 3), (4096, 6, 30), (512, 6, 300), (8192, 6, 130), (100, 3, 60), (37, 7, 27), (5, 10, 4), (70, 2, 26),
                                   (9, 12, 26), (33, 16, 26), (7, 24, 5), (4, 30, 3), (40, 24, 27), (21, 30, 4), (1, 1, 27)])
def test_one_launch_policy_rollout_equals_the_step_loop(B, N, T, form):
    """pw_policy_rollout (T x (actor + sampling + env step) in ONE launch, everything resident on the CU) vs the
    loop of FusedActor() + env.step(): sampled actions, observations, rewards, terminals, pre-reset observations
    and the final world state must be IDENTICAL, across auto-resets, ragged batches and N that does not divide 96.
    Both kernel forms ('v3': the whole BiLSTM on v_mfma_f32_16x16x4_f32, one timestep per barrier -- N = 30 exceeds its LDS:
    skipped there; 'v3j': the same with dense1 just in time and no observation rows in LDS, which serves every N and is what
    long agent axes get by default) and the default choice between them."""
    from multiagent_rl_amd import make_batched_env
    from multiagent_rl_amd.policy import ActorNetwork, FusedActor
    if form == 'v3' and N >= 30:
        pytest.skip('the plain third form does not hold 30 agents\' dense1 output and rows in LDS (the launch returns PW_EINVAL)')
    torch.manual_seed(4)
    mk = lambda: make_batched_env('simple_spread', B, n=N, auto_reset=True, max_episode_len=25, seed=21)  # noqa: E731
    env_a, env_b = mk(), mk()
    env_b.set_dispatch(policy_form=dict(default=0, v3=3, v3j=4)[form])   # pw_dispatch: the handle carries the selection
    actor = ActorNetwork(env_a.obs_dim, 5).cuda().eval()
    loop, one = FusedActor(actor, seed=9), FusedActor(actor, seed=9)
    obs = env_a.reset()
    env_b.reset()
    want = dict(obs=[], act=[], rew=[], rew_shared=[], terminal=[], final_obs=[])
    for t in range(T):
        act = loop(obs)
        obs, rew, done, info = env_a.step(act)
        for k, v in (('obs', obs), ('act', act), ('rew', rew), ('rew_shared', info['rew_shared']),
                     ('terminal', info['terminal']), ('final_obs', info['final_obs'])):
            want[k].append(v.clone())
    got = one.rollout(env_b, T)
    assert one.calls == loop.calls == T
    for k in ('act', 'obs', 'rew', 'rew_shared', 'terminal'):
        assert torch.equal(got[k], torch.stack(want[k])), k
    term = got['terminal']
    if term.any():
        assert torch.equal(got['final_obs'][term], torch.stack(want['final_obs'])[term])
    sa, sb = env_a.get_state(), env_b.get_state()
    for k in ('pos', 'vel', 'landmarks', 'ep_step', 'ep_count'):
        assert torch.equal(sa[k], sb[k]), k
    assert not got['done'].any()
    # continue both for a second chunk: the one-launch form resumes from the stored state and Philox step
    act = loop(obs)
    obs2, _, _, _ = env_a.step(act)
    got2 = one.rollout(env_b, 1)
    assert torch.equal(got2['act'][0], act) and torch.equal(got2['obs'][0], obs2)


def test_retired_policy_forms_are_refused():
    """policy_form 1 / 2 (the phase-by-phase and the role-specialised-waves rollout kernels) were retired in 0.1.5: PW_EINVAL, not a
    silent other form; and a forced form that cannot hold the rows says so."""
    from multiagent_rl_amd import make_batched_env
    from multiagent_rl_amd.policy import ActorNetwork, FusedActor
    env = make_batched_env('simple_spread', 32, n=6, auto_reset=True, seed=1)
    env.reset()
    actor = FusedActor(ActorNetwork(env.obs_dim, 5).cuda().eval(), seed=1)
    for form in (1, 2):
        env.set_dispatch(policy_form=form)
        with pytest.raises(Exception, match='retired'):
            actor.rollout(env, 3)
    big = make_batched_env('simple_spread', 8, n=30, auto_reset=True, seed=1)
    big.reset()
    big.set_dispatch(policy_form=3)
    with pytest.raises(Exception, match='does not fit'):
        FusedActor(ActorNetwork(big.obs_dim, 5).cuda().eval(), seed=1).rollout(big, 2)


def test_collect_one_launch_fills_the_ring_like_the_step_loop():
    """BatchedRollout.collect_one_launch (2 launches per chunk: pw_policy_rollout + pw_replay_add_rollout) vs
    BatchedRollout.collect (3 launches per step): identical ring contents (wrapping), cursor, final observation
    and world state; episode statistics equal up to float64 summation order."""
    from multiagent_rl_amd import make_batched_env
    from multiagent_rl_amd.policy import ActorNetwork, FusedActor
    from multiagent_rl_amd.replay_buffer import ReplayBuffer
    from multiagent_rl_amd.rollout import BatchedRollout
    torch.manual_seed(0)
    actor = ActorNetwork(16, 5).cuda().eval()
    res = []
    for one in (False, True):
        env = make_batched_env('simple_spread', 200, n=6, auto_reset=True, max_episode_len=25, seed=11)
        mem = ReplayBuffer(200 * 40, 6, env.obs_dim)              # 57 steps into 40 slots-of-B: wraps
        ro = BatchedRollout(env, FusedActor(actor, seed=7), mem)
        if one:
            ro.collect_one_launch(57, chunk=20)                   # chunks of 20, 20, 17
        else:
            ro.collect(57)
        st = ro.stats()
        assert st['env_steps'] == 57 * 200 and st['episodes'] == 2 * 200 and len(mem) == 200 * 40
        assert mem._next_idx == (57 * 200) % (200 * 40)
        res.append((mem.obs.clone(), mem.next_obs.clone(), mem.act.clone(), mem.rew.clone(), mem.done.clone(),
                    ro.obs.clone(), env.get_state()['pos'].clone(), ro.episode_return.clone(), st['mean_episode_reward']))
    for a, b in zip(res[0][:8], res[1][:8]):
        assert torch.equal(a, b)
    assert abs(res[0][8] - res[1][8]) < 1e-9 * abs(res[0][8])


def test_accelerate_trainer_patches_get_exploration_action():
    """accelerate_trainer(): the reference Trainer's get_exploration_action surface (list of N obs arrays ->
    one-hot ndarray [1,N,5], ddpg_gumbel_fix.py:86-107) served by the one-launch actor; the snapshot follows
    optimize()."""
    from multiagent_rl_amd.policy import ActorNetwork, accelerate_trainer
    torch.manual_seed(0)

    class Trainer(object):                      # the attributes / methods of the reference's Trainer that matter here
        action_type = 'Discrete'

        def __init__(self):
            self.actor = ActorNetwork(16, 5).cuda()
            self.optimized = 0

        def optimize(self):
            with torch.no_grad():
                self.actor.dense2.module.bias[:] = torch.tensor([0.0, 0.0, 50.0, 0.0, 0.0])   # "learning": always act 2
            self.optimized += 1
            return 0.0, 0.0

    tr = Trainer()
    accelerate_trainer(tr, seed=5)
    obs_n = [np.random.RandomState(i).randn(16) for i in range(6)]
    a = tr.get_exploration_action(obs_n)
    assert isinstance(a, np.ndarray) and a.shape == (1, 6, 5) and a.dtype == np.float32
    assert (a.sum(-1) == 1).all() and set(np.unique(a)) == {0.0, 1.0}
    draws = np.stack([tr.get_exploration_action(obs_n)[0] for _ in range(300)])
    with torch.no_grad():
        p = torch.softmax(tr.actor(torch.from_numpy(np.stack(obs_n)[None].astype(np.float32)).cuda()), -1)[0].cpu().numpy()
    assert np.abs(draws.mean(0) - p).max() < 0.12
    assert tr.optimize() == (0.0, 0.0) and tr.optimized == 1
    assert (tr.get_exploration_action(obs_n)[0].argmax(-1) == 2).all()       # the refreshed snapshot is in use
    # MultiDiscrete: two one-hots, as the reference returns them
    class Trainer2(object):
        action_type = 'MultiDiscrete'

        def __init__(self):
            self.actor = ActorNetwork(21, [5, 10]).cuda()
    t2 = Trainer2()
    accelerate_trainer(t2)
    a2 = t2.get_exploration_action([np.zeros(21), np.ones(21)])
    assert isinstance(a2, list) and a2[0].shape == (1, 2, 5) and a2[1].shape == (1, 2, 10)
    assert (a2[0].sum(-1) == 1).all() and (a2[1].sum(-1) == 1).all()


def test_add_rollout_equals_a_loop_of_add_batch():
    """ReplayBuffer.add_rollout (a whole chunk of step outputs in one launch, incl. the episode bookkeeping) vs T x
    add_batch + pw_episode_stats on the same outputs."""
    from multiagent_rl_amd import make_batched_env
    from multiagent_rl_amd.policy import UniformRandomPolicy
    from multiagent_rl_amd.replay_buffer import ReplayBuffer
    from multiagent_rl_amd.rollout import BatchedRollout
    torch.manual_seed(1)
    B, T = 300, 31
    env = make_batched_env('simple_spread', B, n=3, auto_reset=True, max_episode_len=25, seed=2)
    obs0 = env.reset()
    acts = torch.randint(0, 5, (T, B, 3), device='cuda', dtype=torch.int32)
    out = env.rollout(acts)
    out['act'] = acts
    a, b = ReplayBuffer(B * 40, 3, env.obs_dim), ReplayBuffer(B * 40, 3, env.obs_dim)
    ro = BatchedRollout(make_batched_env('simple_spread', B, n=3, auto_reset=True), UniformRandomPolicy(), None)
    prev = obs0
    for t in range(T):
        a.add_batch(prev, acts[t], out['rew_shared'][t], out['obs'][t], out['final_obs'][t], out['terminal'][t])
        ro._bookkeeping(out['rew_shared'][t], out['terminal'][t])
        prev = out['obs'][t]
    ret = torch.zeros(B, device='cuda')
    fs, fc = torch.zeros((), dtype=torch.float64, device='cuda'), torch.zeros((), dtype=torch.int64, device='cuda')
    b.add_rollout(obs0, out, ret, fs, fc)
    b.add_rollout(out['obs'][T - 1], {k: v[:1] for k, v in out.items()})        # a second chunk, no bookkeeping
    a.add_batch(out['obs'][T - 1], acts[0], out['rew_shared'][0], out['obs'][0], out['final_obs'][0], out['terminal'][0])
    assert a._next_idx == b._next_idx == (T + 1) * B and len(a) == len(b)
    n = (T + 1) * B
    for x, y in ((a.obs, b.obs), (a.next_obs, b.next_obs), (a.act, b.act), (a.rew, b.rew), (a.done, b.done)):
        assert torch.equal(x[:n], y[:n])
    assert torch.equal(ret, ro.episode_return) and int(fc.item()) == int(ro.finished_episodes.item()) == B
    assert abs(fs.item() - ro.finished_return_sum.item()) < 1e-9 * abs(fs.item())


@pytest.mark.parametrize('B,N,T,ep', [(300, 3, 31, 25), (4096, 6, 100, 25), (77, 6, 26, 7), (50, 5, 12, 0), (64, 2, 9, 1)])
def test_chunk_wire_finalize_and_add_wire_equal_add_rollout(B, N, T, ep):
    """The full-gather path on one GPU: the rollout writes obs / rew_shared INTO the wire block, pw_chunk_wire_finalize
    condenses the rest, pw_replay_add_wire appends the block -- the ring must equal pw_replay_add_rollout on the
    sender's dense outputs bit for bit (incl. a wrap of the ring end), and the block must equal the torch
    restatement the gloo test uses (tests/dist_standins.py).  ep = 0: episodes never end (F = 0)."""
    from tests.dist_standins import wire_finalize_reference, wire_transitions_reference
    from multiagent_rl_amd import make_batched_env
    from multiagent_rl_amd.dist import FullTransitionGather
    from multiagent_rl_amd.replay_buffer import ReplayBuffer
    torch.manual_seed(3)
    env = make_batched_env('simple_spread', B, n=N, auto_reset=True, max_episode_len=ep, seed=5)
    D = env.obs_dim
    cap = T * B * 3 + 17
    mem = ReplayBuffer(cap, N, D)
    mem._next_idx = mem._len = cap - 5                        # the first chunk wraps around the ring end
    full = FullTransitionGather(env, T, 0, 1, torch.device('cuda', 0), memory=mem, wire='rows')
    assert not full.state_wire and full.lay.F == (0 if ep == 0 else -(-T // ep))
    assert abs(full.bytes_per_env_step - full.lay.total_bytes / (T * B)) < 1e-9
    want = ReplayBuffer(cap, N, D)
    want._next_idx = want._len = cap - 5
    for rb in (mem, want):                                    # 17 slots stay unwritten: make them comparable
        for plane in (rb.obs, rb.next_obs, rb.act, rb.rew, rb.done):
            plane.zero_()
    obs0 = env.reset()
    # desynchronise the episode clocks so that episode ends are not in lockstep
    if ep > 1:
        st = env.get_state()
        env.set_state(st['pos'], st['vel'], st['landmarks'],
                      ep_step=(torch.arange(B, device='cuda') % ep).int(), ep_count=st['ep_count'])
    for k in range(3):
        acts = torch.randint(0, 5, (T, B, N), device='cuda', dtype=torch.int32)
        out = full.outputs()
        out['act'].copy_(acts)
        env.rollout(acts, out={n_: v for n_, v in out.items() if n_ != 'act'})
        block = full.wire[full.exchanges % full.SLOTS]
        dense = {n_: v.clone() for n_, v in out.items()}
        full(obs0)
        # the block equals the CPU restatement of finalize
        cpu = FullTransitionGather.__new__(FullTransitionGather)
        cpu.__dict__.update(T=T, B=B, N=N, D=D, lay=full.lay, state_wire=False,
                            side={n_: (None if v is None else v.cpu()) for n_, v in full.side.items()})
        ref_block = block.cpu().clone()
        for name in ('obs0', 'final_rows', 'act', 'fin_slot'):
            cpu.views(ref_block)[name].zero_()
        wire_finalize_reference(cpu, ref_block, obs0.cpu())
        got_v, ref_v = full.views(block), cpu.views(ref_block)
        for name in ('obs0', 'act', 'fin_slot'):
            assert torch.equal(got_v[name].cpu(), ref_v[name]), name
        fs = ref_v['fin_slot'].long()
        tt, ee = torch.nonzero(fs != 255, as_tuple=True)
        if tt.numel():
            assert torch.equal(got_v['final_rows'].cpu()[fs[tt, ee], ee], ref_v['final_rows'][fs[tt, ee], ee])
        assert (fs != 255).equal(dense['terminal'].cpu() & (ep > 0))
        want.add_rollout(obs0, dense)
        tr = wire_transitions_reference(cpu, block.cpu())
        assert torch.equal(tr['next_obs'].reshape(T, B, N, D),
                           torch.where(dense['terminal'][:, :, None, None], dense['final_obs'], dense['obs']).cpu()
                           if ep > 0 else dense['obs'].cpu())
        obs0 = dense['obs'][T - 1]
    full.finish()
    assert full.rows_ingested == 3 * T * B and mem._next_idx == want._next_idx and len(mem) == len(want) == cap
    for x, y in ((mem.obs, want.obs), (mem.next_obs, want.next_obs), (mem.act, want.act), (mem.rew, want.rew)):
        assert torch.equal(x, y)
    lo, n = (cap - 5) % cap, 3 * T * B
    idx = [(lo + i) % cap for i in range(0, n, max(1, n // 999))]
    assert not mem.sample_index(idx)[4].any()


def _reset_landmarks_numpy(seed, env_ids, episodes, N, L):
    """pw_reset_xy (include/pworld_math.h) for the landmark entities, restated with the oracle's Philox: counter = (entity,
    episode, env id lo, hi), key = (seed lo, hi); u = (r >> 8) * 2^-24; coordinate = 2 u - 1 in float32."""
    from oracle import c_oracle as co
    out = np.zeros((len(env_ids), L, 2), np.float32)
    for i, (e, ep) in enumerate(zip(env_ids, episodes)):
        for l in range(L):
            r = co.philox4x32_10((N + l, int(ep) & 0xFFFFFFFF, int(e) & 0xFFFFFFFF, int(e) >> 32), (seed & 0xFFFFFFFF, seed >> 32))
            u = (np.array(r[:2], np.uint32) >> 8).astype(np.float32) * np.float32(5.9604644775390625e-8)
            out[i, l] = np.float32(2.0) * u + np.float32(-1.0)
    return out


@pytest.mark.parametrize('B,N,L,T,ep', [(300, 3, None, 31, 25), (4096, 6, None, 100, 25), (77, 6, None, 26, 7), (50, 5, None, 12, 0),
                                        (64, 2, None, 9, 1), (90, 6, 4, 30, 11), (33, 12, None, 27, 25)])
def test_state_wire_ring_equals_add_rollout_bitwise(B, N, L, T, ep):
    """The full gather on STATE-ONLY wire blocks, on one GPU: pw_state_wire_begin snapshots the chunk's start, the rollout writes
    its rows to the sender's side buffer and rew_shared into the block, pw_state_wire_finalize condenses rows to states (+ the
    pre-reset states, the landmarks each reset drew, byte actions, the episode map), pw_replay_add_state_wire REBUILDS the rows:
    the ring must equal pw_replay_add_rollout on the sender's dense outputs BIT FOR BIT (obs, the pre-reset next_obs, actions,
    rewards; incl. a wrap of the ring end), and the block must equal the torch restatement the gloo tests use.
    Odd L (float2 path), L != N, ep = 0 (episodes never end) and ep = 1 (every step ends one) included."""
    from tests.dist_standins import (state_wire_begin_reference, state_wire_finalize_reference,
                                     state_wire_transitions_reference)
    from multiagent_rl_amd import make_batched_env
    from multiagent_rl_amd.dist import FullTransitionGather
    from multiagent_rl_amd.replay_buffer import ReplayBuffer
    torch.manual_seed(3)
    seed = 5
    kw = {} if L is None else dict(num_landmarks=L)
    env = make_batched_env('simple_spread', B, n=N, auto_reset=True, max_episode_len=ep, seed=seed, **kw)
    D, L = env.obs_dim, env.num_landmarks
    cap = T * B * 3 + 17
    mem = ReplayBuffer(cap, N, D)
    mem._next_idx = mem._len = cap - 5                        # the first chunk wraps around the ring end
    full = FullTransitionGather(env, T, 0, 1, torch.device('cuda', 0), memory=mem)
    assert full.state_wire and full.lay.F == (0 if ep == 0 else -(-T // ep))
    rows = FullTransitionGather(env, T, 0, 1, torch.device('cuda', 0), wire='rows')
    assert not rows.state_wire and full.bytes_per_env_step < (0.5 if ep != 1 else 0.7) * rows.bytes_per_env_step
    want = ReplayBuffer(cap, N, D)
    want._next_idx = want._len = cap - 5
    for rb in (mem, want):                                    # 17 slots stay unwritten: make them comparable
        for plane in (rb.obs, rb.next_obs, rb.act, rb.rew, rb.done):
            plane.zero_()
    obs0 = env.reset()
    if ep > 1:                                                # desynchronise the episode clocks
        st = env.get_state()
        env.set_state(st['pos'], st['vel'], st['landmarks'],
                      ep_step=(torch.arange(B, device='cuda') % ep).int(), ep_count=st['ep_count'])
    lm_fn = lambda e, epn: torch.from_numpy(_reset_landmarks_numpy(seed, e.tolist(), epn.tolist(), N, L))  # noqa: E731
    for k in range(3):
        acts = torch.randint(0, 5, (T, B, N), device='cuda', dtype=torch.int32)
        st0 = env.get_state()
        out = full.outputs()                                   # also: pw_state_wire_begin on this chunk's block
        out['act'].copy_(acts)
        env.rollout(acts, out={n_: v for n_, v in out.items() if n_ != 'act'})
        block = full.wire[full.exchanges % full.SLOTS]
        dense = {n_: v.clone() for n_, v in out.items()}
        full(obs0)
        # the block equals the CPU restatement of begin + finalize
        cpu = FullTransitionGather.__new__(FullTransitionGather)
        cpu.__dict__.update(T=T, B=B, N=N, D=D, L=L, lay=full.lay, state_wire=True,
                            side={n_: (None if v is None else v.cpu()) for n_, v in full.side.items()})
        ref_block = torch.zeros_like(block, device='cpu')
        cpu.views(ref_block)['rew_shared'].copy_(dense['rew_shared'].cpu())
        state0 = torch.cat([st0['vel'], st0['pos']], -1).cpu()
        state_wire_begin_reference(cpu, ref_block, state0, st0['landmarks'].cpu(), st0['ep_count'].cpu())
        state_wire_finalize_reference(cpu, ref_block, lm_fn)
        got_v, ref_v = full.views(block), cpu.views(ref_block)
        for name in ('state0', 'state', 'ep0', 'rew_shared', 'act', 'epi'):
            assert torch.equal(got_v[name].cpu(), ref_v[name]), name
        assert torch.equal(got_v['state0'].cpu(), obs0[..., :4].cpu())
        epi = ref_v['epi'].long()
        ended, kk = (epi & 128) != 0, epi & 127
        assert ended.equal(dense['terminal'].cpu() & (ep > 0))
        tt, ee = torch.nonzero(ended, as_tuple=True)
        if tt.numel():
            assert torch.equal(got_v['final_state'].cpu()[kk[tt, ee], ee], ref_v['final_state'][kk[tt, ee], ee])
            assert torch.equal(got_v['lm'].cpu()[kk[tt, ee] + 1, ee], ref_v['lm'][kk[tt, ee] + 1, ee])
        assert torch.equal(got_v['lm'][0].cpu(), ref_v['lm'][0])
        # the landmarks the block carries for the LAST episode are the ones the env itself holds now
        last_k = (kk[-1] + ended[-1].long())
        assert torch.equal(got_v['lm'].cpu()[last_k, torch.arange(B)], env.get_state()['landmarks'].cpu())
        want.add_rollout(obs0, dense)
        tr = state_wire_transitions_reference(cpu, block.cpu())
        assert torch.equal(tr['next_obs'].reshape(T, B, N, D),
                           torch.where(dense['terminal'][:, :, None, None], dense['final_obs'], dense['obs']).cpu()
                           if ep > 0 else dense['obs'].cpu())
        assert torch.equal(tr['obs'].reshape(T, B, N, D), torch.cat([obs0[None], dense['obs'][:-1]], 0).cpu())
        obs0 = dense['obs'][T - 1]
    full.finish()
    assert full.rows_ingested == 3 * T * B and mem._next_idx == want._next_idx and len(mem) == len(want) == cap
    for name, x, y in (('obs', mem.obs, want.obs), ('next_obs', mem.next_obs, want.next_obs), ('act', mem.act, want.act),
                       ('rew', mem.rew, want.rew), ('done', mem.done, want.done)):
        assert torch.equal(x, y), name


def _gather_three_chunks(env, T, ep, rings, policy=None):
    """Three chunks of `env` through one FullTransitionGather per entry of `rings` (ring kind -> gather), all fed from the SAME
    rollout outputs (the first gather's wire block is the one the rollout writes into; the others ingest copies of it), plus the
    reference ring filled by pw_replay_add_rollout from the dense outputs.  -> (gathers, want)."""
    from multiagent_rl_amd.dist import FullTransitionGather
    from multiagent_rl_amd.replay_buffer import ReplayBuffer
    B, N, D = env.num_envs, env.n, env.obs_dim
    dev = torch.device('cuda', 0)
    cap = T * B * 3 + 17
    gs = {}
    for kind in rings:
        g = FullTransitionGather(env, T, 0, 1, dev, capacity=cap, ring=kind, overlap_ingest=False)
        assert g.state_wire
        g.memory._next_idx = g.memory._len = cap - 5          # the first chunk wraps around the ring end
        gs[kind] = g
    want = ReplayBuffer(cap, N, D)
    want._next_idx = want._len = cap - 5
    lead = gs[rings[0]]
    obs0 = env.reset()
    if ep > 1:                                                # desynchronise the episode clocks
        st = env.get_state()
        env.set_state(st['pos'], st['vel'], st['landmarks'], ep_step=(torch.arange(B, device='cuda') % ep).int(), ep_count=st['ep_count'])
    for k in range(3):
        acts = torch.randint(0, 5, (T, B, N), device='cuda', dtype=torch.int32)
        out = lead.outputs()
        out['act'].copy_(acts)
        env.rollout(acts, out={n_: v for n_, v in out.items() if n_ != 'act'})
        dense = {n_: v.clone() for n_, v in out.items()}
        slot = lead.exchanges % lead.SLOTS
        lead(obs0)
        for kind in rings[1:]:                                # the same finished block, appended by the other ring kind
            g = gs[kind]
            g.wire[slot].copy_(lead.wire[slot])
            g._pending = ([], slot)
            g._complete()
        want.add_rollout(obs0, dense)
        obs0 = dense['obs'][T - 1]
    lead.finish()
    torch.cuda.synchronize()
    return gs, want


@pytest.mark.parametrize('N,L,B,T,ep', [(3, None, 300, 31, 25), (6, None, 512, 60, 25), (12, None, 33, 27, 25), (6, 5, 90, 30, 11),
                                        (3, 1, 64, 9, 1), (5, None, 50, 12, 0), (6, 9, 40, 26, 7)])
def test_state_ring_sample_index_equals_the_row_ring_bitwise(N, L, B, T, ep):
    """The learner rank's STATE ring (pw_replay_store.state_rows: {vel, pos} of every agent before / after + the episode's landmarks per
    transition, 32 N + 8 L bytes instead of 8 N D): filled from the same state-only wire blocks as the row ring, over three chunks with
    desynchronised episode clocks (>= two resets per env) and a wrap of the ring end; sample_index on it -- rows REBUILT by
    pw_replay_gather -- equals the row ring's batch bit for bit (obs, one-hot actions, reward, the pre-reset next_obs, done), and the
    row ring equals pw_replay_add_rollout on the sender's dense outputs.  Odd L, L != N, L > N, ep = 0 / 1 included."""
    from multiagent_rl_amd import make_batched_env
    kw = {} if L is None else dict(num_landmarks=L)
    env = make_batched_env('simple_spread', B, n=N, auto_reset=True, max_episode_len=ep, seed=5, **kw)
    gs, want = _gather_three_chunks(env, T, ep, ['state', 'rows'])
    st, rows = gs['state'].memory, gs['rows'].memory
    assert st.state_ring and tuple(st.obs.shape[1:]) == (N, 4) and tuple(st.lm.shape[1:]) == (max(env.num_landmarks, 1), 2)
    assert len(st) == len(rows) == len(want) and st._next_idx == rows._next_idx == want._next_idx
    cap = want._maxsize
    written = [(cap - 5 + i) % cap for i in range(3 * T * B)]
    for lo in range(0, len(written), 4096):
        idx = written[lo:lo + 4096]
        a, b, c = st.sample_index(idx), rows.sample_index(idx), want.sample_index(idx)
        for name, x, y, z in zip(('obs', 'act', 'rew', 'next_obs', 'done'), a, b, c):
            assert torch.equal(x, y), name
            assert torch.equal(y, z), name
    # bytes the root writes per transition
    per_state = sum(t.element_size() * t[0].numel() for t in (st.obs, st.next_obs, st.lm, st.act, st.rew, st.done))
    per_rows = sum(t.element_size() * t[0].numel() for t in (rows.obs, rows.next_obs, rows.act, rows.rew, rows.done))
    assert per_state == 32 * N + 8 * max(env.num_landmarks, 1) + N + 8 and per_rows == 8 * N * env.obs_dim + N + 8
    # a state ring refuses rows
    with pytest.raises(Exception, match='STATE ring'):
        st.add_batch(torch.zeros(1, N, env.obs_dim, device='cuda'), torch.zeros(1, N, dtype=torch.int32, device='cuda'),
                     torch.zeros(1, device='cuda'), torch.zeros(1, N, env.obs_dim, device='cuda'))


@pytest.mark.parametrize('A,G,L,B,T,ep', [(4, 2, 2, 8192, 50, 25), (3, 1, 2, 77, 26, 7), (2, 3, 3, 60, 30, 11), (1, 1, 1, 40, 9, 1),
                                          (4, 2, 2, 50, 12, 0)])
def test_tag_state_wire_ring_equals_add_rollout_bitwise(A, G, L, B, T, ep):
    """simple_tag on STATE-ONLY wire blocks (0.1.6): 16 B per agent and step, the landmarks (drawn from U(-0.9, 0.9)) once per episode,
    the pre-reset state at episode ends; the root rebuilds the 4 + 2L + 2(N - 1) + 2G-number rows (other agents' relative positions, the
    good agents' velocities, zero padding for the good agents' own rows) -- the row ring equals pw_replay_add_rollout on the sender's
    dense outputs BIT FOR BIT, and so does the batch a STATE ring rebuilds.  C3's 4 + 2 roster at full size: <= 150 B per env-step."""
    from multiagent_rl_amd import make_batched_env
    from multiagent_rl_amd.dist import FullTransitionGather
    env = make_batched_env('simple_tag', B, num_adversaries=A, num_good=G, num_landmarks=L, auto_reset=True, max_episode_len=ep, seed=9)
    N = A + G
    assert env.obs_dim == 4 + 2 * L + 2 * (N - 1) + 2 * G
    gs, want = _gather_three_chunks(env, T, ep, ['rows', 'state'])
    rows, st = gs['rows'].memory, gs['state'].memory
    for name in ('obs', 'next_obs', 'act', 'rew', 'done'):
        cap = want._maxsize
        sl = torch.tensor([(cap - 5 + i) % cap for i in range(3 * T * B)], device='cuda')
        assert torch.equal(getattr(rows, name)[sl], getattr(want, name)[sl]), name
    written = [(want._maxsize - 5 + i) % want._maxsize for i in range(3 * T * B)]
    for lo in range(0, len(written), 8192):
        idx = written[lo:lo + 8192]
        for name, x, y in zip(('obs', 'act', 'rew', 'next_obs', 'done'), st.sample_index(idx), want.sample_index(idx)):
            assert torch.equal(x, y), name
    per_step = gs['rows'].bytes_per_env_step
    row_block = FullTransitionGather(env, T, 0, 1, torch.device('cuda', 0), wire='rows').bytes_per_env_step
    assert per_step < (0.45 if ep != 1 else 0.7) * row_block   # (ep = 1: every step ships a pre-reset state and fresh landmarks too)
    if (A, G, B) == (4, 2, 8192):
        assert per_step <= 150.0, per_step                   # <= 150 B per env-step for tag 4 + 2 (from ~565)


@pytest.mark.parametrize('B,T,ep', [(4096, 100, 25), (77, 26, 7), (40, 9, 1), (50, 12, 0)])
def test_reference_compact_wire_ring_equals_add_rollout_bitwise(B, T, ep):
    """simple_reference (MultiDiscrete, main.py:24,52-54) through the full gather on COMPACT-ROW wire blocks (pw_ref_wire, 0.1.6): the
    first eight numbers of every 21-number row, one goal byte per agent and episode, both action heads as bytes; the root rebuilds
    goal colours and the other agent's one-hot symbol (zeros after a reset) -- the two-head ring equals pw_replay_add_rollout on the
    sender's dense outputs BIT FOR BIT over three chunks with desynchronised episode clocks and a ring wrap.  <= 80 B per env-step."""
    from multiagent_rl_amd import make_batched_env
    from multiagent_rl_amd.dist import FullTransitionGather
    from multiagent_rl_amd.replay_buffer import ReplayBuffer
    torch.manual_seed(4)
    env = make_batched_env('simple_reference', B, auto_reset=True, max_episode_len=ep, seed=21)
    N, D = env.n, env.obs_dim
    assert (N, D) == (2, 21)
    cap = 3 * T * B + 17
    full = FullTransitionGather(env, T, 0, 1, torch.device('cuda', 0), capacity=cap)
    assert full.ref_wire and not full.state_wire and full.memory.act_heads == (5, 10)
    want = ReplayBuffer(cap, N, D, act_heads=(5, 10))
    full.memory._next_idx = full.memory._len = want._next_idx = want._len = cap - 5
    obs0 = env.reset()
    if ep > 1:
        st = env.get_state()
        env.set_state(st['pos'], st['vel'], st['landmarks'], ep_step=(torch.arange(B, device='cuda') % ep).int(), ep_count=st['ep_count'])
    for k in range(3):
        acts = torch.stack([torch.randint(0, 5, (T, B, N), device='cuda'), torch.randint(0, 10, (T, B, N), device='cuda')], -1).int()
        out = full.outputs()
        out['act'].copy_(acts)
        env.rollout(acts, out={n_: v for n_, v in out.items() if n_ != 'act'})
        dense = {n_: v.clone() for n_, v in out.items()}
        block = full.wire[full.exchanges % full.SLOTS]
        full(obs0)
        # the block equals the torch restatement the gloo tests run (tests/dist_standins.py), and so do the rebuilt transitions
        from tests.dist_standins import ref_wire_finalize_reference, ref_wire_transitions_reference
        cpu = FullTransitionGather.__new__(FullTransitionGather)
        cpu.__dict__.update(T=T, B=B, N=N, D=D, lay=full.lay, state_wire=False, ref_wire=True,
                            side={n_: (None if v is None else v.cpu()) for n_, v in full.side.items()})
        ref_block = torch.zeros_like(block, device='cpu')
        cpu.views(ref_block)['rew_shared'].copy_(dense['rew_shared'].cpu())
        ref_wire_finalize_reference(cpu, ref_block, obs0.cpu())
        got_v, ref_v = full.views(block), cpu.views(ref_block)
        for name in ('head0', 'head', 'comm0', 'rew_shared', 'act', 'epi'):
            assert torch.equal(got_v[name].cpu(), ref_v[name]), name
        epi = ref_v['epi'].long()
        ended, kk = (epi & 128) != 0, epi & 127
        assert torch.equal(got_v['goal'][0].cpu(), ref_v['goal'][0])
        tt, ee = torch.nonzero(ended, as_tuple=True)
        if tt.numel():
            assert torch.equal(got_v['final_head'].cpu()[kk[tt, ee], ee], ref_v['final_head'][kk[tt, ee], ee])
            assert torch.equal(got_v['goal'].cpu()[kk[tt, ee] + 1, ee], ref_v['goal'][kk[tt, ee] + 1, ee])
        tr = ref_wire_transitions_reference(cpu, block.cpu())
        assert torch.equal(tr['obs'].reshape(T, B, N, D), torch.cat([obs0[None], dense['obs'][:-1]], 0).cpu())
        assert torch.equal(tr['next_obs'].reshape(T, B, N, D),
                           (torch.where(dense['terminal'][:, :, None, None], dense['final_obs'], dense['obs']) if ep > 0 else dense['obs']).cpu())
        want.add_rollout(obs0, dense)
        obs0 = dense['obs'][T - 1].clone()
    full.finish()
    torch.cuda.synchronize()
    assert full.rows_ingested == 3 * T * B and full.memory._next_idx == want._next_idx
    sl = torch.tensor([(cap - 5 + i) % cap for i in range(3 * T * B)], device='cuda')
    for name in ('obs', 'next_obs', 'act', 'rew', 'done'):
        assert torch.equal(getattr(full.memory, name)[sl], getattr(want, name)[sl]), name
    if T == 100:
        assert full.bytes_per_env_step <= 80.0, full.bytes_per_env_step     # <= 80 B per env-step (the row block: 181)
    with pytest.raises(ValueError, match='two-head'):
        FullTransitionGather(env, T, 0, 1, torch.device('cuda', 0), wire='rows')


def test_reference_compact_wire_carries_the_two_head_policy_rollout():
    """The two-head actor in the loop (pw_policy_rollout on simple_reference, act [T,B,N,2]) writing into the gather's outputs: the root's
    two-head ring equals the ring the same rollout's own sink fills, bit for bit (B = 4096, 100-step chunks)."""
    from multiagent_rl_amd import make_batched_env
    from multiagent_rl_amd.dist import FullTransitionGather
    from multiagent_rl_amd.policy import ActorNetwork, FusedActor
    from multiagent_rl_amd.replay_buffer import ReplayBuffer
    torch.manual_seed(0)
    B, T = 4096, 100
    mk = lambda: make_batched_env('simple_reference', B, auto_reset=True, max_episode_len=25, seed=3)  # noqa: E731
    env_a, env_b = mk(), mk()
    actor = ActorNetwork(env_a.obs_dim, [5, 10]).cuda().eval()
    wire_actor, sink_actor = FusedActor(actor, seed=7), FusedActor(actor, seed=7)
    full = FullTransitionGather(env_a, T, 0, 1, torch.device('cuda', 0), capacity=2 * T * B)
    want = ReplayBuffer(2 * T * B, 2, env_b.obs_dim, act_heads=(5, 10))
    env_a.reset()
    env_b.reset()
    obs0 = env_a.observe()
    for k in range(2):
        out = full.outputs()
        assert tuple(out['act'].shape) == (T, B, 2, 2)
        wire_actor.rollout(env_a, T, out)
        full(obs0)
        obs0 = out['obs'][T - 1].clone()
        sink_actor.rollout(env_b, T, False, memory=want)
    full.finish()
    torch.cuda.synchronize()
    assert full.rows_ingested == 2 * T * B == len(full.memory) == len(want)
    for name in ('obs', 'next_obs', 'act', 'rew', 'done'):
        assert torch.equal(getattr(full.memory, name), getattr(want, name)), name
    # a single-head action buffer under the two-head actor is refused, not overrun
    bad = dict(out, act=torch.zeros(T, B, 2, dtype=torch.int32, device='cuda'))
    with pytest.raises(ValueError, match='two-head'):
        wire_actor.rollout(env_a, T, bad)


@pytest.mark.parametrize('scenario,B,kw,T', [('simple_spread', 4096, dict(n=6), 60), ('simple_spread', 70, dict(n=24), 30),
                                             ('simple_spread', 40, dict(n=33), 27), ('simple_spread', 100, dict(n=3), 55),
                                             ('simple_tag', 8192, dict(num_adversaries=4, num_good=2), 53),
                                             ('simple_tag', 37, dict(num_adversaries=2, num_good=3), 30)],
                         ids=['C2-form3', 'N24-3j', 'N33-3j-half', 'N3', 'C3-tag', 'tag2+3'])
def test_policy_rollout_sink_into_a_state_ring_samples_the_row_rings_batch(scenario, B, kw, T):
    """The one-launch policy rollouts' own ring sink into a STATE ring (0.1.6): the launch leaves {vel, pos} before / after + the
    episode's landmarks per transition instead of the two observation rows; sample_index on it (rows rebuilt by pw_replay_gather)
    equals the batch of the ROW ring the same rollout fills from the same seeds, bit for bit -- every kernel form, across two resets,
    chunks of 20 steps, a ring wrap."""
    from multiagent_rl_amd import make_batched_env
    from multiagent_rl_amd.policy import ActorNetwork, FusedActor
    from multiagent_rl_amd.replay_buffer import ReplayBuffer
    torch.manual_seed(2)
    mk = lambda: make_batched_env(scenario, B, auto_reset=True, max_episode_len=25, seed=17, **kw)  # noqa: E731
    env_a, env_b = mk(), mk()
    N, D, L = env_a.n, env_a.obs_dim, env_a.num_landmarks
    net = ActorNetwork(D, 5).cuda().eval()
    cap = T * B - 3 * B // 2                                  # the last chunk wraps around the ring end
    rows = ReplayBuffer(cap, N, D)
    state = ReplayBuffer(cap, N, D, state_ring=dict(scenario=scenario, num_landmarks=L, num_adversaries=kw.get('num_adversaries', 0)))
    for env, mem in ((env_a, rows), (env_b, state)):
        env.reset()
        actor = FusedActor(net, seed=5)
        done = 0
        while done < T:
            n = min(20, T - done)
            actor.rollout(env, n, False, memory=mem)
            done += n
    torch.cuda.synchronize()
    assert len(rows) == len(state) == cap and rows._next_idx == state._next_idx
    assert tuple(state.obs.shape[1:]) == (N, 4)
    for lo in range(0, cap, 8192):
        idx = list(range(lo, min(cap, lo + 8192)))
        for name, x, y in zip(('obs', 'act', 'rew', 'next_obs', 'done'), state.sample_index(idx), rows.sample_index(idx)):
            assert torch.equal(x, y), name
    # a STATE ring of another scenario / shape is refused
    other = ReplayBuffer(cap, N, D, state_ring=dict(scenario=scenario, num_landmarks=L + 1, num_adversaries=kw.get('num_adversaries', 0))) \
        if scenario == 'simple_tag' else None
    if other is not None:
        with pytest.raises(Exception):
            FusedActor(net, seed=5).rollout(env_a, 5, False, memory=other)


def test_state_wire_carries_the_policy_rollout_and_stays_under_130_bytes_per_env_step():
    """C2 (B = 4096, N = 6, 100-step chunks) with the policy in the loop, as bench.py --gpus N drives it: FusedActor.rollout
    writes into FullTransitionGather.outputs(); the root ring equals the ring the same launch's own sink fills
    (pw_rollout_sink) bit for bit; the block is <= 130 B per env-step (the row block: 414)."""
    from multiagent_rl_amd import make_batched_env
    from multiagent_rl_amd.dist import FullTransitionGather
    from multiagent_rl_amd.policy import ActorNetwork, FusedActor
    from multiagent_rl_amd.replay_buffer import ReplayBuffer
    torch.manual_seed(0)
    B, N, T = 4096, 6, 100
    mk = lambda: make_batched_env('simple_spread', B, n=N, auto_reset=True, max_episode_len=25, seed=12345678)  # noqa: E731
    env_a, env_b = mk(), mk()
    actor = ActorNetwork(env_a.obs_dim, 5).cuda().eval()
    wire_actor, sink_actor = FusedActor(actor, seed=3), FusedActor(actor, seed=3)
    full = FullTransitionGather(env_a, T, 0, 1, torch.device('cuda', 0), capacity=2 * T * B)
    assert full.state_wire and full.bytes_per_env_step <= 130.0, full.bytes_per_env_step
    assert 113.0 < full.bytes_per_env_step < 116.0            # 17 N + 5 per step + (1 + F) state / landmark batches per chunk
    want = ReplayBuffer(2 * T * B, N, env_b.obs_dim)
    env_a.reset()
    env_b.reset()
    obs0 = env_a.observe()
    for k in range(2):
        out = full.outputs()
        wire_actor.rollout(env_a, T, out)
        full(obs0)
        obs0 = out['obs'][T - 1]
        sink_actor.rollout(env_b, T, False, memory=want)
    full.finish()
    assert full.rows_ingested == 2 * T * B == len(full.memory) == len(want)
    for name in ('obs', 'next_obs', 'act', 'rew', 'done'):
        assert torch.equal(getattr(full.memory, name), getattr(want, name)), name


def test_wire_functions_reject_foreign_layouts():
    import ctypes as C
    from multiagent_rl_amd import _lib
    lib = _lib.load()
    lay = _lib.PwChunkWire()
    assert lib.pw_chunk_wire_layout(10, 8, 3, 10, 25, C.byref(lay)) == 0 and lay.F == 1
    assert lib.pw_chunk_wire_layout(1000, 8, 3, 10, 2, C.byref(lay)) < 0      # 500 episode ends per env: too many
    assert lib.pw_chunk_wire_layout(10, 8, 3, 10, 25, C.byref(lay)) == 0
    block = torch.zeros(lay.total_bytes, dtype=torch.uint8, device='cuda')
    bad = _lib.PwChunkWire.from_buffer_copy(lay)
    bad.obs = lay.obs + 256
    z = torch.zeros(8 * 3 * 10, device='cuda')
    term, act = torch.zeros(10, 8, dtype=torch.uint8, device='cuda'), torch.zeros(10, 8, 3, dtype=torch.int32, device='cuda')
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    assert lib.pw_chunk_wire_finalize(C.byref(bad), p(block), p(z), None, p(term), p(act), None) < 0
    assert b'pw_chunk_wire_layout' in lib.pw_last_error()
    assert lib.pw_chunk_wire_finalize(C.byref(lay), C.c_void_p(block.data_ptr() + 4), p(z), None, p(term), p(act), None) < 0
    assert lib.pw_chunk_wire_finalize(C.byref(lay), p(block), p(z), None, p(term), p(act), None) == 0
    torch.cuda.synchronize()


def test_state_wire_functions_reject_foreign_layouts_and_scenarios():
    """pw_state_wire_*: a layout not produced by pw_state_wire_layout, a misaligned block, a handle of another scenario / another
    shape / the full observation are refused with PW_EINVAL (and dist.FullTransitionGather falls back to row blocks there)."""
    import ctypes as C
    from multiagent_rl_amd import _lib, make_batched_env
    from multiagent_rl_amd.dist import FullTransitionGather
    lib = _lib.load()
    env = make_batched_env('simple_spread', 8, n=3, auto_reset=True, max_episode_len=25, seed=1)
    env.reset()
    lay = _lib.PwStateWire()
    assert lib.pw_state_wire_layout(10, 8, 3, 3, 25, C.byref(lay)) == 0 and lay.F == 1 and lay.D == 10
    assert lib.pw_state_wire_layout(1000, 8, 3, 3, 2, C.byref(lay)) < 0 and b'126' in lib.pw_last_error()
    assert lib.pw_state_wire_layout(10, 8, 3, 3, 25, C.byref(lay)) == 0
    block = torch.zeros(lay.total_bytes, dtype=torch.uint8, device='cuda')
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    obs = torch.zeros(10, 8, 3, 10, device='cuda')
    term, act = torch.zeros(10, 8, dtype=torch.uint8, device='cuda'), torch.zeros(10, 8, 3, dtype=torch.int32, device='cuda')
    bad = _lib.PwStateWire.from_buffer_copy(lay)
    bad.state = lay.state + 256
    assert lib.pw_state_wire_begin(env._h, C.byref(bad), p(block), None) < 0 and b'pw_state_wire_layout' in lib.pw_last_error()
    assert lib.pw_state_wire_begin(env._h, C.byref(lay), C.c_void_p(block.data_ptr() + 4), None) < 0
    assert lib.pw_state_wire_begin(env._h, C.byref(lay), p(block), None) == 0
    assert lib.pw_state_wire_finalize(env._h, C.byref(lay), p(block), p(obs), p(obs), p(term), p(act), None) == 0
    other = make_batched_env('simple_spread', 8, n=6, auto_reset=True, seed=1)          # another shape
    other.reset()
    assert lib.pw_state_wire_begin(other._h, C.byref(lay), p(block), None) < 0 and b'shape mismatch' in lib.pw_last_error()
    tag = make_batched_env('simple_tag', 8, num_adversaries=3, num_good=1, auto_reset=True, seed=1)
    tag.reset()
    assert lib.pw_state_wire_begin(tag._h, C.byref(lay), p(block), None) < 0 and b'shape mismatch' in lib.pw_last_error()   # a simple_spread block
    tlay = _lib.PwStateWire()
    assert lib.pw_state_wire_layout_scn(_lib.PW_SIMPLE_TAG, 10, 8, 4, 2, 3, 25, C.byref(tlay)) == 0 and tlay.D == 4 + 4 + 6 + 2
    tblock = torch.zeros(tlay.total_bytes, dtype=torch.uint8, device='cuda')
    assert lib.pw_state_wire_begin(tag._h, C.byref(tlay), p(tblock), None) == 0                 # simple_tag is served since 0.1.6
    assert lib.pw_state_wire_begin(env._h, C.byref(tlay), p(tblock), None) < 0
    assert lib.pw_state_wire_layout_scn(_lib.PW_SIMPLE_REFERENCE, 10, 8, 2, 3, 0, 25, C.byref(tlay)) < 0
    full_obs = make_batched_env('simple_spread', 8, n=3, local_observation=False, auto_reset=True, seed=1)
    full_obs.reset()
    assert lib.pw_state_wire_begin(full_obs._h, C.byref(lay), p(block), None) < 0
    ring = _lib.PwReplayStore()
    torch.cuda.synchronize()
    # the host side picks the block kind by the same rule
    dev = torch.device('cuda', 0)
    assert FullTransitionGather(env, 10, 0, 1, dev).state_wire
    assert FullTransitionGather(tag, 10, 0, 1, dev).state_wire and not FullTransitionGather(full_obs, 10, 0, 1, dev).state_wire
    with pytest.raises(ValueError, match='state-only'):
        FullTransitionGather(full_obs, 10, 0, 1, dev, wire='state')
    with pytest.raises(ValueError, match="ring='state'"):
        FullTransitionGather(full_obs, 10, 0, 1, dev, ring='state')
    ref = make_batched_env('simple_reference', 8, auto_reset=True, seed=1)
    assert FullTransitionGather(ref, 10, 0, 1, dev).ref_wire
    with pytest.raises(ValueError, match='two-head'):
        FullTransitionGather(ref, 10, 0, 1, dev, wire='rows')
    with pytest.raises(ValueError, match='auto-resetting'):
        FullTransitionGather(make_batched_env('simple_spread', 8, n=3, auto_reset=False, seed=1), 10, 0, 1, dev)


def test_run_test_on_hip_env_with_device_ring_pickles_history(tmp_path):
    """rollout.run_test (experiments/run.py:106-200) on the HIP MultiAgentEnv with the DEFAULT memory -- the device
    ReplayBuffer -- reproduces the reference's evaluation call trace (event kinds, shapes, dtypes; values to float32
    accuracy) and pickles the history WITH the memory inside (run.py:186-191); the pickle loads back and serves the
    same samples."""
    import pickle
    from multiagent_rl_amd import make_env, rollout
    from multiagent_rl_amd.replay_buffer import ReplayBuffer
    gold = json.load(open(os.path.join(GOLD_DIR, 'run_test_trace.json')))

    class Args(object):
        is_training, display = True, False
    for k, v in gold['arglist'].items():
        setattr(Args, k, v)
    np.random.seed(12345679)
    env = RecordingEnv(make_env('simple_spread', n=4))
    np.random.seed(12345679)
    StubTrainer.trace = env.trace
    hist = rollout.run_test(env, None, None, StubTrainer, 'simple_spread', 'Discrete', cnt=1, arglist=Args,
                            out_dir=str(tmp_path), log=lambda *a: None)

    def skeleton(x):
        if isinstance(x, dict):
            return {k: skeleton(v) for k, v in x.items() if k not in ('sum', 'value')}
        if isinstance(x, list):
            return [skeleton(v) for v in x]
        return x
    assert skeleton(json.loads(json.dumps(env.trace))) == skeleton(gold['trace'])
    assert env.trace[1] == ['load_models', 'pfx/simple_spread_fin_1'] and not any(e[0] == 'save_models' for e in env.trace)
    np.testing.assert_allclose(hist['reward_episodes'], gold['reward_episodes'], rtol=1e-4, atol=1e-3)
    mem = hist['memory']
    assert isinstance(mem, ReplayBuffer) and len(mem) == gold['memory_len'] and sorted(os.listdir(tmp_path)) == gold['files']
    with open(os.path.join(str(tmp_path), gold['files'][0]), 'rb') as fp:
        back = pickle.load(fp)
    assert sorted(back.keys()) == gold['history_keys']
    m2 = back['memory']
    assert len(m2) == len(mem) and m2._next_idx == mem._next_idx and m2._store is None   # no GPU touched by loading
    planes = m2.host_planes()
    assert planes['obs'].shape == (len(mem), 4, mem.obs_dim) and planes['act'].shape == (len(mem), 4)
    idx = list(range(0, len(mem), 3))
    for a, b in zip(mem.sample_index(idx), m2.sample_index(idx)):
        assert torch.equal(a, b)
    m2.add(*[[np.zeros(mem.obs_dim)] * 4, [np.eye(5)[1]] * 4, 0.5, [np.ones(mem.obs_dim)] * 4, 0.0])   # still a ring
    assert len(m2) == len(mem) + 1


def test_device_ring_multidiscrete_and_bicnet_variants(tmp_path):
    """The transitions run() stores for MultiDiscrete scenarios (15-wide concatenated one-hots, run.py:39-41,52) and
    for BiCNet (per-agent rew_n / done_n, run_BIC.py:46,50) go through the DEFAULT device memory and come back from
    sample_index exactly as the reference's tuple ring would return them."""
    import pickle
    from multiagent_rl_amd import make_env, rollout
    from multiagent_rl_amd.replay_buffer import ReplayBuffer
    gold = json.load(open(os.path.join(GOLD_DIR, 'run_multidiscrete_trace.json')))

    class Args(object):
        is_training, display = True, False
    for k, v in gold['arglist'].items():
        setattr(Args, k, v)

    class Spy(StubTrainer):                      # keeps what add() received, next to the device ring
        def __init__(self, actor, critic, memory, action_type='Discrete'):
            super().__init__(actor, critic, memory, action_type)
            self.log, inner = [], memory.add

            def add(*a):
                self.log.append(a)
                inner(*a)
            memory.add = add
    np.random.seed(12345680)
    env = RecordingEnv(make_env('simple_reference'))
    Spy.trace = env.trace
    rollout.run(env, None, None, Spy, 'simple_reference', 'MultiDiscrete', cnt=2, arglist=Args,
                out_dir=str(tmp_path), log=lambda *a: None)
    tr, mem = Spy.last.log, Spy.last.memory
    assert isinstance(mem, ReplayBuffer) and mem.act_heads == (5, 10) and not mem.per_agent and len(mem) == len(tr) > 20
    o, a, r, n, d = mem.sample_index(list(range(len(tr))))
    assert a.shape == (len(tr), 2, 15) and r.shape == (len(tr),)
    np.testing.assert_array_equal(a.cpu().numpy(), np.array([np.stack(t[1]) for t in tr], dtype=np.float32))
    np.testing.assert_allclose(o.cpu().numpy(), np.array([np.stack(t[0]) for t in tr]), rtol=0, atol=1e-7)
    np.testing.assert_allclose(n.cpu().numpy(), np.array([np.stack(t[3]) for t in tr]), rtol=0, atol=1e-7)
    np.testing.assert_allclose(r.cpu().numpy(), np.array([t[2] for t in tr]), rtol=1e-6)
    m2 = pickle.loads(pickle.dumps(mem))
    assert m2.act_heads == (5, 10) and torch.equal(m2.sample_index([0, 5, 7])[1], a[[0, 5, 7]])

    # BiCNet tuple
    gold = json.load(open(os.path.join(GOLD_DIR, 'run_trace.json')))
    for k, v in gold['arglist'].items():
        setattr(Args, k, v)
    np.random.seed(12345678)
    env = RecordingEnv(make_env('simple_spread'))
    Spy.trace = env.trace
    rollout.run(env, None, None, Spy, 'simple_spread', 'Discrete', cnt=0, arglist=Args, out_dir=None,
                log=lambda *a: None, per_agent_transition=True)
    tr, mem = Spy.last.log, Spy.last.memory
    assert mem.per_agent and mem.act_heads == (5,) and len(mem) == 75
    o, a, r, n, d = mem.sample_index(list(range(75)))
    assert r.shape == d.shape == (75, 3) and a.shape == (75, 3, 5)
    np.testing.assert_allclose(r.cpu().numpy(), np.array([t[2] for t in tr]), rtol=1e-6)
    np.testing.assert_array_equal(d.cpu().numpy(), np.array([t[4] for t in tr], dtype=np.float32))
    np.testing.assert_array_equal(a.cpu().numpy(), np.array([np.stack(t[1]) for t in tr], dtype=np.float32))

    # what the index ring cannot hold is refused loudly, never squeezed through an argmax
    rb = ReplayBuffer(8)
    two = [np.concatenate([np.eye(5)[1], np.eye(10)[3]])] * 2
    with pytest.raises(ValueError, match='act_heads'):
        rb.add([np.zeros(21)] * 2, two, 0.0, [np.zeros(21)] * 2, 0.0)                 # 15-wide rows, heads not given
    with pytest.raises(ValueError, match='different lengths'):
        ReplayBuffer(8).add([np.zeros(11)] * 2, [np.eye(3)[0], np.eye(5)[0]], 0.0, [np.zeros(11)] * 2, 0.0)
    with pytest.raises(ValueError, match='one-hot'):
        ReplayBuffer(8).add([np.zeros(10)] * 3, [np.full(5, 0.2)] * 3, 0.0, [np.zeros(10)] * 3, 0.0)   # soft action
    ok = ReplayBuffer(8, act_heads=(5, 10))
    ok.add([np.zeros(21)] * 2, two, 0.0, [np.zeros(21)] * 2, 0.0)
    with pytest.raises(ValueError, match='shared scalar'):
        ok.add([np.zeros(21)] * 2, two, [0.0, 1.0], [np.zeros(21)] * 2, [0.0, 0.0])


def test_batched_rollout_multidiscrete_feeds_a_two_head_ring():
    """simple_reference, B envs: the two-head FusedActor's [B,N,2] actions go into a ring built with act_heads=(5,10)."""
    from multiagent_rl_amd import make_batched_env
    from multiagent_rl_amd.policy import ActorNetwork, FusedActor
    from multiagent_rl_amd.replay_buffer import ReplayBuffer
    from multiagent_rl_amd.rollout import BatchedRollout
    torch.manual_seed(0)
    B = 128
    env = make_batched_env('simple_reference', B, auto_reset=True, max_episode_len=25, seed=3)
    actor = ActorNetwork(env.obs_dim, [5, 10]).cuda().eval()
    mem = ReplayBuffer(B * 40, 2, env.obs_dim, act_heads=(5, 10))
    ro = BatchedRollout(env, FusedActor(actor, seed=4), mem)
    first = ro.obs.clone()
    ro.collect(30)
    assert len(mem) == 30 * B and ro.stats()['episodes'] == B
    o, a, r, n, d = mem.sample_index(list(range(B)))
    assert torch.equal(o, first) and a.shape == (B, 2, 15)
    assert (a[..., :5].sum(-1) == 1).all() and (a[..., 5:].sum(-1) == 1).all() and not d.any()
    assert mem.act.shape == (B * 40, 2, 2) and int(mem.act[:30 * B, :, 1].max()) > 4      # symbols use the 10-wide head


@pytest.mark.parametrize('B,adv,good,T', [(8192, 4, 2, 27), (100, 3, 1, 55), (37, 2, 3, 30)])
def test_one_launch_policy_rollout_simple_tag_equals_the_step_loop(B, adv, good, T):
    """BASELINE configs[2] with the policy in the loop: pw_policy_rollout on simple_tag (4 adversaries + 2 good agents,
    B = 8192 at full size; the canonical 3+1; a 2+3 roster) vs the loop of FusedActor() + env.step(): identical
    sampled actions, observations (ragged rows zero-padded), rewards, terminals, pre-reset observations, final state."""
    from multiagent_rl_amd import make_batched_env
    from multiagent_rl_amd.policy import ActorNetwork, FusedActor
    from multiagent_rl_amd.replay_buffer import ReplayBuffer
    from multiagent_rl_amd.rollout import BatchedRollout
    torch.manual_seed(5)
    mk = lambda: make_batched_env('simple_tag', B, num_adversaries=adv, num_good=good, auto_reset=True,  # noqa: E731
                                  max_episode_len=25, seed=31)
    env_a, env_b = mk(), mk()
    N = adv + good
    actor = ActorNetwork(env_a.obs_dim, 5).cuda().eval()
    loop, one = FusedActor(actor, seed=9), FusedActor(actor, seed=9)
    obs = env_a.reset()
    env_b.reset()
    want = dict(obs=[], act=[], rew=[], rew_shared=[], terminal=[], final_obs=[])
    for t in range(T):
        act = loop(obs)
        obs, rew, done, info = env_a.step(act)
        for k, v in (('obs', obs), ('act', act), ('rew', rew), ('rew_shared', info['rew_shared']),
                     ('terminal', info['terminal']), ('final_obs', info['final_obs'])):
            want[k].append(v.clone())
    got = one.rollout(env_b, T)
    for k in ('act', 'obs', 'rew', 'rew_shared', 'terminal'):
        assert torch.equal(got[k], torch.stack(want[k])), k
    term = got['terminal']
    assert term.any() and torch.equal(got['final_obs'][term], torch.stack(want['final_obs'])[term])
    sa, sb = env_a.get_state(), env_b.get_state()
    for k in ('pos', 'vel', 'landmarks', 'ep_step', 'ep_count'):
        assert torch.equal(sa[k], sb[k]), k
    assert got['act'].shape == (T, B, N) and int(got['act'].max()) == 4 and got['rew'].abs().sum() > 0
    # the ring sink: collect_one_launch == collect on the same scenario
    if B <= 100:
        res = []
        for one_launch in (False, True):
            env = mk()
            mem = ReplayBuffer(B * 60, N, env.obs_dim)
            ro = BatchedRollout(env, FusedActor(actor, seed=7), mem)
            ro.collect_one_launch(40, chunk=17) if one_launch else ro.collect(40)
            st = ro.stats()
            res.append((mem.obs[:40 * B].clone(), mem.next_obs[:40 * B].clone(), mem.act[:40 * B].clone(),
                        mem.rew[:40 * B].clone(), ro.episode_return.clone(), st['episodes'], st['mean_episode_reward']))
        for x, y in zip(res[0][:5], res[1][:5]):
            assert torch.equal(x, y)
        assert res[0][5] == res[1][5] == B and abs(res[0][6] - res[1][6]) < 1e-9 * max(1.0, abs(res[0][6]))


def test_exchanges_on_real_rccl_single_rank():
    """Both exchanges through torch.distributed's "nccl" backend (= RCCL) with ONE rank, in this process (no worker is
    spawned): the sampled gather's all_gather_into_tensor is a real RCCL launch, the full gather runs its finalize /
    ingest launches behind the same process-group plumbing bench.py uses.  The multi-rank choreography is covered by
    the gloo tests; what this adds on a GPU box is that the RCCL path itself initialises and runs."""
    import socket
    import torch.distributed as dist
    from multiagent_rl_amd import make_batched_env
    from multiagent_rl_amd.dist import FullTransitionGather, SampledTransitionGather
    from multiagent_rl_amd.policy import ActorNetwork, FusedActor
    if dist.is_initialized():
        pytest.skip('a process group already exists in this process')
    sk = socket.socket()
    sk.bind(('127.0.0.1', 0))
    port = sk.getsockname()[1]
    sk.close()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dev = torch.device('cuda', 0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    try:
        torch.manual_seed(0)
        B, N, T = 512, 6, 50
        env = make_batched_env('simple_spread', B, n=N, auto_reset=True, max_episode_len=25, seed=3)
        env.reset()
        gat = SampledTransitionGather(env, 256, 0, 1, dev, every=1, side_stream=True)
        for k in range(3):
            acts = torch.randint(0, 5, (T, B, N), device='cuda', dtype=torch.int32)
            out = env.rollout(acts)
            gat(out, acts)
        gat.finish()
        torch.cuda.synchronize()
        assert gat.exchanges == 3 and gat.rows_ingested == 3 * 256 and len(gat.memory) == 3 * 256
        o, a, r, n, d = gat.memory.sample_index(list(range(3 * 256)))
        assert torch.isfinite(o).all() and (a.sum(-1) == 1).all() and not d.any()
        # full gather: the policy-in-the-loop rollout writes into the wire block; every transition reaches the ring
        actor = FusedActor(ActorNetwork(env.obs_dim, 5).cuda().eval(), seed=5)
        full = FullTransitionGather(env, T, 0, 1, dev)
        full.prime()
        obs0 = env.observe()
        first = obs0.clone()
        for k in range(2):
            out = full.outputs()
            actor.rollout(env, T, out)
            full(obs0)
            obs0 = out['obs'][T - 1]
        full.finish()
        torch.cuda.synchronize()
        assert full.rows_ingested == 2 * T * B and len(full.memory) == 2 * T * B
        o, a, r, n, d = full.memory.sample_index(list(range(B)))
        assert torch.equal(o, first) and (a.sum(-1) == 1).all()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('B,T', [(4096, 27), (100, 55), (37, 30), (16, 3), (17, 26)])
def test_one_launch_policy_rollout_simple_reference_equals_the_step_loop(B, T):
    """The MultiDiscrete scenario of the reference's sweep (main.py:24,52-54; run.py:39-41) with the two-head actor in the loop
    as ONE launch: pw_policy_rollout on simple_reference vs the loop of FusedActor() + env.step((movement, symbol)) --
    identical sampled pairs, observations (incl. the partner's spoken symbol), rewards, terminals, pre-reset observations,
    final world / communication state, across auto-resets and ragged batches; a second chunk continues both."""
    from multiagent_rl_amd import make_batched_env
    from multiagent_rl_amd.policy import ActorNetwork, FusedActor
    torch.manual_seed(6)
    mk = lambda: make_batched_env('simple_reference', B, auto_reset=True, max_episode_len=25, seed=17)  # noqa: E731
    env_a, env_b = mk(), mk()
    actor = ActorNetwork(env_a.obs_dim, [5, 10]).cuda().eval()
    loop, one = FusedActor(actor, seed=9), FusedActor(actor, seed=9)
    obs = env_a.reset()
    env_b.reset()

    def run_loop(steps):
        nonlocal obs
        want = dict(obs=[], act=[], rew=[], rew_shared=[], terminal=[], final_obs=[])
        for t in range(steps):
            act = loop(obs)
            assert act.shape == (B, 2, 2)
            obs, rew, done, info = env_a.step(act)
            for k, v in (('obs', obs), ('act', act), ('rew', rew), ('rew_shared', info['rew_shared']),
                         ('terminal', info['terminal']), ('final_obs', info['final_obs'])):
                want[k].append(v.clone())
        return {k: torch.stack(v) for k, v in want.items()}

    for chunk in (T, 7):
        want = run_loop(chunk)
        got = one.rollout(env_b, chunk)
        assert got['act'].shape == (chunk, B, 2, 2) and int(got['act'][..., 1].max()) > 4      # symbols use the 10-wide head
        for k in ('act', 'obs', 'rew', 'rew_shared', 'terminal'):
            assert torch.equal(got[k], want[k]), k
        term = got['terminal']
        if term.any():
            assert torch.equal(got['final_obs'][term], want['final_obs'][term])
        assert not got['done'].any()
        sa, sb = env_a.get_state(), env_b.get_state()
        for k in ('pos', 'vel', 'landmarks', 'ep_step', 'ep_count', 'comm', 'goal'):
            assert torch.equal(sa[k], sb[k]), k
    assert one.calls == loop.calls == T + 7


def test_collect_one_launch_multidiscrete_fills_the_two_head_ring_like_the_step_loop():
    """BatchedRollout.collect_one_launch on simple_reference (pw_policy_rollout + pw_replay_add_rollout into a ring built with
    act_heads = (5, 10): 2 launches per chunk) vs BatchedRollout.collect (3 launches per step): identical ring contents
    (wrapping), cursor, final observation and state; statistics equal up to float64 summation order."""
    from multiagent_rl_amd import make_batched_env
    from multiagent_rl_amd.policy import ActorNetwork, FusedActor
    from multiagent_rl_amd.replay_buffer import ReplayBuffer
    from multiagent_rl_amd.rollout import BatchedRollout
    torch.manual_seed(0)
    B = 150
    res = []
    actor = None
    for one in (False, True):
        env = make_batched_env('simple_reference', B, auto_reset=True, max_episode_len=25, seed=3)
        if actor is None:
            actor = ActorNetwork(env.obs_dim, [5, 10]).cuda().eval()
        mem = ReplayBuffer(B * 40, 2, env.obs_dim, act_heads=(5, 10))              # 57 steps into 40 slots-of-B: wraps
        ro = BatchedRollout(env, FusedActor(actor, seed=7), mem)
        if one:
            ro.collect_one_launch(57, chunk=20)                                     # chunks of 20, 20, 17
        else:
            ro.collect(57)
        st = ro.stats()
        assert st['env_steps'] == 57 * B and st['episodes'] == 2 * B and len(mem) == B * 40
        assert mem._next_idx == (57 * B) % (B * 40)
        res.append((mem.obs.clone(), mem.next_obs.clone(), mem.act.clone(), mem.rew.clone(), mem.done.clone(),
                    ro.obs.clone(), env.get_state()['pos'].clone(), ro.episode_return.clone(), st['mean_episode_reward']))
    for a, b in zip(res[0][:8], res[1][:8]):
        assert torch.equal(a, b)
    assert abs(res[0][8] - res[1][8]) < 1e-9 * abs(res[0][8])


@pytest.mark.gpu
@pytest.mark.parametrize('state_ring', [False, True], ids=['row-ring', 'state-ring'])
def test_reference_sized_ring_at_n48_indexes_past_four_gibi_elements(state_ring):
    """The reference's ring size (1e6 transitions, run.py:20) at BASELINE's largest agent count (N = L = 48, D = 100): one observation plane
    is 4.8e9 floats (19.2 GB) -- element offsets past 2^32 in the ring's upper half.  A chunk appended across the ring's END (add_rollout:
    row ring; the policy rollout's own sink: both rings) and sampled back must equal the same chunk in a small ring, bit for bit."""
    from multiagent_rl_amd import make_batched_env
    from multiagent_rl_amd.policy import ActorNetwork, FusedActor
    from multiagent_rl_amd.replay_buffer import ReplayBuffer
    if torch.cuda.mem_get_info()[0] < 60e9:
        pytest.skip('needs ~45 GB of device memory')
    cap, N, B, T = int(1e6), 48, 64, 20
    kw = dict(state_ring=dict(scenario='simple_spread', num_landmarks=N, num_adversaries=0)) if state_ring else {}
    mk = lambda: make_batched_env('simple_spread', B, n=N, auto_reset=True, max_episode_len=7, seed=5)  # noqa: E731
    torch.manual_seed(1)
    net = ActorNetwork(100, 5).cuda().eval()
    start = cap - T * B // 2                                   # the chunk wraps around the ring's end
    res = []
    for size in (cap, 4 * T * B):
        env = mk()
        D = env.obs_dim
        assert D == 100 and (state_ring or cap * N * D > 2 ** 32)
        ring = ReplayBuffer(size, N, D, **kw)
        env.reset()
        ring._next_idx, ring._len = (start if size == cap else size - T * B // 2), size
        first = ring._next_idx
        actor = FusedActor(net, seed=3)
        actor.rollout(env, T, False, memory=ring)              # the launch's own sink
        idx = [(first + i) % size for i in range(T * B)]
        got = [x.clone() for x in ring.sample_index(idx)]
        if not state_ring:                                      # the same transitions again through pw_replay_add_rollout, one chunk further
            env2 = mk()
            o0 = env2.reset().clone()
            out = {k: torch.empty((T,) + tuple(s), dtype=d, device='cuda') for k, s, d in (
                ('obs', (B, N, D), torch.float32), ('final_obs', (B, N, D), torch.float32), ('rew', (B, N), torch.float32),
                ('rew_shared', (B,), torch.float32), ('terminal', (B,), torch.bool), ('done', (B, N), torch.bool), ('act', (B, N), torch.int32))}
            FusedActor(net, seed=3).rollout(env2, T, out)
            ring._next_idx = first
            ring.add_rollout(o0, out)
            again = ring.sample_index(idx)
            for nm, x, y in zip(('obs', 'act', 'rew', 'next_obs', 'done'), got, again):
                assert torch.equal(x, y), ('add_rollout vs sink', size, nm)
        res.append(got)
        del ring
        torch.cuda.empty_cache()
    for nm, x, y in zip(('obs', 'act', 'rew', 'next_obs', 'done'), *res):
        assert torch.equal(x, y), nm
    assert torch.isfinite(res[0][0]).all() and (res[0][1].sum(-1) == 1).all()
