"""GPU: the one-launch policy rollouts (pw_policy_rollout: actor + Gumbel sampling + env step for a whole chunk in
ONE kernel) anchored DIRECTLY on the CPU oracle.

Every rollout kernel (forms 3 / 3j of simple_spread, the simple_tag and the simple_reference kernels) carries its
own copy of the environment step.  tests/test_gpu_engine.py compares them with the FusedActor() + env.step() loop --
HIP against HIP.  Here the actions a launch sampled (``out['act']``, an output) are replayed through the float32 C
oracle from the same Philox reset, and every environment output of the launch -- observations, per-agent rewards,
shared reward (run.py:46), terminal flags (run.py:50), pre-reset observations at terminal steps (what run.py:52 stores
as new_obs_n), done flags, the final world state -- must equal the oracle's BIT FOR BIT; likewise the rows the launch
leaves in the replay ring (rls/replay_buffer.py:30-37 tuples).  Reference loop: experiments/run.py:37-60.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')

from oracle import c_oracle as co  # noqa: E402  (checker only)
from tests.test_gpu_parity import _assert_same_bits, _np  # noqa: E402

FORMS = dict(default=0, v3=3, v3j=4)
KERNEL_OF_FORM = {3: 'pw_policy_rollout3_kernel',
                  4: 'pw_policy_rollout3j_kernel'}


def _replay_through_oracle(o32, got, T, two_head=False):
    """Steps ``o32`` with the launch's own sampled actions and bit-compares every environment output of ``got``."""
    resets = 0
    for t in range(T):
        a = _np(got['act'][t])
        w = o32.step(act_idx=a[..., 0], act_comm=a[..., 1]) if two_head else o32.step(act_idx=a)
        if two_head:
            shared = np.zeros(o32.B, np.float32)
            for i in range(o32.N):
                shared = shared + w['rew'][:, i]
            w['rew_shared'] = shared
        _assert_same_bits(_np(got['obs'][t]), w['obs'], 'obs[%d]' % t)
        _assert_same_bits(_np(got['rew'][t]), w['rew'], 'rew[%d]' % t)
        _assert_same_bits(_np(got['rew_shared'][t]), w['rew_shared'], 'rew_shared[%d]' % t)
        _assert_same_bits(_np(got['terminal'][t]).astype(np.uint8), w['terminal'], 'terminal[%d]' % t)
        _assert_same_bits(_np(got['done'][t]).astype(np.uint8), w['done'], 'done[%d]' % t)
        if w['terminal'].any():
            resets += 1
            m = w['terminal'].astype(bool)
            _assert_same_bits(_np(got['final_obs'][t])[m], w['final_obs'][m], 'final_obs[%d]' % t)
    return resets


def _assert_final_state(env, o32, extra=()):
    st = env.get_state()
    _assert_same_bits(_np(st['pos']), o32.pos, 'pos')
    _assert_same_bits(_np(st['vel']), o32.vel, 'vel')
    _assert_same_bits(_np(st['landmarks']), o32.lm, 'landmarks')
    assert np.array_equal(_np(st['ep_step']), o32.ep_step)
    assert np.array_equal(_np(st['ep_count']).astype(np.uint32), o32.ep_count)
    for k in extra:
        _assert_same_bits(_np(st[k]), getattr(o32, k), k)


@pytest.mark.parametrize('form', ['default', 'v3', 'v3j'])
@pytest.mark.parametrize('B,N,T', [(4096, 6, 53), (100, 3, 60), (37, 7, 27), (9, 12, 26), (33, 16, 26), (7, 24, 5), (520, 24, 53),
                                   (19, 30, 27)], ids=['C2', 'N3', 'N7', 'N12', 'N16', 'N24', 'N24-B520', 'N30'])
def test_spread_policy_rollout_outputs_equal_the_oracle_on_its_own_actions(B, N, T, form):
    """simple_spread, every kernel form (C2 at full size: B = 4096, N = 6, 53 steps across two auto-resets)."""
    from multiagent_rl_amd import make_batched_env
    from multiagent_rl_amd.policy import ActorNetwork, FusedActor
    if form == 'v3' and N >= 30:
        pytest.skip('the plain third form does not hold 30 agents\' dense1 output and rows in LDS')
    torch.manual_seed(4)
    env = make_batched_env('simple_spread', B, n=N, auto_reset=True, max_episode_len=25, seed=21)
    env.set_dispatch(policy_form=FORMS[form])
    cfg = co.make_config('simple_spread', N, max_episode_len=25, auto_reset=True, seed=21)
    o32 = co.COracle(cfg, B, np.float32)
    actor = FusedActor(ActorNetwork(env.obs_dim, 5).cuda().eval(), seed=9)
    _assert_same_bits(_np(env.reset()), o32.reset(), 'reset obs')
    got = actor.rollout(env, T)
    if form == 'default':                 # automatic choice: form 3 wherever 16 environments per workgroup fit its LDS,
        assert env.last_kernel() == KERNEL_OF_FORM[3 if N <= 12 else 4], env.last_kernel()   # its just-in-time variant beyond
    else:
        assert env.last_kernel() == KERNEL_OF_FORM[FORMS[form]], env.last_kernel()
    a = _np(got['act'])
    assert a.shape == (T, B, N) and a.min() >= 0 and a.max() <= 4
    assert len(np.unique(a)) == 5                       # a real policy sample, not a constant
    resets = _replay_through_oracle(o32, got, T)
    assert resets == T // 25
    _assert_final_state(env, o32)
    # a second chunk continues from the stored state (and the oracle from its own)
    got2 = actor.rollout(env, 3)
    _replay_through_oracle(o32, got2, 3)
    _assert_final_state(env, o32)


@pytest.mark.parametrize('B,N,T', [(24, 48, 27), (11, 33, 26), (70, 40, 4), (17, 31, 27), (40, 50, 26), (33, 32, 5)],
                         ids=['N48', 'N33', 'N40', 'N31', 'N50', 'N32'])
def test_spread_policy_rollout_with_rows_longer_than_64_numbers(B, N, T):
    """BASELINE configs[4]'s largest point with the policy in the loop: N = L = 48 (D = 100).  Only the just-in-time form holds
    such rows (round 5: 16 environments per workgroup -- the half-storage head, environment slots; N = 31 is the smallest such N, 50 the largest,
    B = 17 / 24 / 40 / 70 leave ragged last workgroups); its outputs equal the oracle's on its own actions bit for bit, and -- with a
    sharpened head, so that the Gumbel noise cannot decide -- its actions are the arg-max of PyTorch's float32 logits on the
    oracle's observation rows (the per-step FusedActor does not serve D > 64, so this is the actor's own check here)."""
    from multiagent_rl_amd import make_batched_env
    from multiagent_rl_amd.policy import ActorNetwork, FusedActor
    torch.manual_seed(8)
    env = make_batched_env('simple_spread', B, n=N, auto_reset=True, max_episode_len=25, seed=33)
    cfg = co.make_config('simple_spread', N, max_episode_len=25, auto_reset=True, seed=33)
    o32 = co.COracle(cfg, B, np.float32)
    net = ActorNetwork(env.obs_dim, 5).cuda().eval()
    with torch.no_grad():
        net.dense2.module.weight *= 2e5
        net.dense2.module.bias *= 2e5
    actor = FusedActor(net, seed=9)
    obs0 = o32.reset()
    _assert_same_bits(_np(env.reset()), obs0, 'reset obs')
    got = actor.rollout(env, T)
    assert env.last_kernel() == 'pw_policy_rollout3j_kernel', env.last_kernel()
    assert _replay_through_oracle(o32, got, T) == T // 25
    _assert_final_state(env, o32)
    # the actor: logits of the stock PyTorch network on the rows the oracle produced
    rows = torch.from_numpy(np.concatenate([obs0[None], _np(got['obs'][:-1])], 0)).cuda()        # what the policy saw at step t
    with torch.no_grad():
        lg = net(rows.reshape(T * B, N, env.obs_dim)).reshape(T, B, N, 5)
    top2 = lg.topk(2, dim=-1).values
    clear = (top2[..., 0] - top2[..., 1]) > 200.0                  # Gumbel noise lives in [-17, 3]; float32 logits of ~1e5 carry ~1e-1 of error
    want = lg.argmax(-1).to(torch.int32)
    assert clear.float().mean().item() > 0.8
    assert torch.equal(got['act'][clear], want[clear])
    env.set_dispatch(policy_form=3)                                # the plain third form refuses such rows instead of running something else
    with pytest.raises(Exception, match='longer than 64'):
        actor.rollout(env, 2)


@pytest.mark.parametrize('B,adv,good,T', [(8192, 4, 2, 53), (100, 3, 1, 55), (37, 2, 3, 30)], ids=['C3', '3+1', '2+3'])
def test_tag_policy_rollout_outputs_equal_the_oracle_on_its_own_actions(B, adv, good, T):
    """simple_tag (BASELINE configs[2]: 4 adversaries + 2 good agents, B = 8192; ragged rows zero-padded)."""
    from multiagent_rl_amd import make_batched_env
    from multiagent_rl_amd.policy import ActorNetwork, FusedActor
    torch.manual_seed(5)
    N = adv + good
    env = make_batched_env('simple_tag', B, num_adversaries=adv, num_good=good, auto_reset=True, max_episode_len=25, seed=31)
    cfg = co.make_config('simple_tag', N, num_adversaries=adv, max_episode_len=25, auto_reset=True, seed=31)
    o32 = co.COracle(cfg, B, np.float32)
    actor = FusedActor(ActorNetwork(env.obs_dim, 5).cuda().eval(), seed=9)
    _assert_same_bits(_np(env.reset()), o32.reset(), 'reset obs')
    got = actor.rollout(env, T)
    assert 'policy_rollout_tag' in env.last_kernel(), env.last_kernel()
    assert len(np.unique(_np(got['act']))) == 5
    assert _replay_through_oracle(o32, got, T) == T // 25
    assert float(got['rew'].abs().sum()) > 0
    _assert_final_state(env, o32)


@pytest.mark.parametrize('B,T', [(4096, 53), (100, 55), (17, 26)])
def test_reference_policy_rollout_outputs_equal_the_oracle_on_its_own_actions(B, T):
    """simple_reference (main.py:24; MultiDiscrete [5, 10], run.py:39-41) with the two-head actor in the launch."""
    from multiagent_rl_amd import make_batched_env
    from multiagent_rl_amd.policy import ActorNetwork, FusedActor
    torch.manual_seed(6)
    env = make_batched_env('simple_reference', B, auto_reset=True, max_episode_len=25, seed=17)
    cfg = co.make_config('simple_reference', max_episode_len=25, auto_reset=True, seed=17)
    o32 = co.CRefOracle(cfg, B, np.float32)
    actor = FusedActor(ActorNetwork(env.obs_dim, [5, 10]).cuda().eval(), seed=9)
    _assert_same_bits(_np(env.reset()), o32.reset(), 'reset obs')
    got = actor.rollout(env, T)
    assert 'policy_rollout_ref' in env.last_kernel(), env.last_kernel()
    a = _np(got['act'])
    assert a.shape == (T, B, 2, 2) and a[..., 0].max() <= 4 and 4 < a[..., 1].max() <= 9
    assert _replay_through_oracle(o32, got, T, two_head=True) == T // 25
    _assert_final_state(env, o32, extra=('comm', 'goal'))


def test_bf16x3_policy_rollout_environment_half_is_still_exact():
    """The opt-in bf16x3 input projection changes which actions are sampled, never the environment arithmetic: the
    launch's outputs still equal the oracle bit for bit on the actions it sampled."""
    from multiagent_rl_amd import make_batched_env
    from multiagent_rl_amd.policy import ActorNetwork, FusedActor
    torch.manual_seed(4)
    B, N, T = 2048, 6, 53
    env = make_batched_env('simple_spread', B, n=N, auto_reset=True, max_episode_len=25, seed=21)
    env.set_actor_precision('bf16x3')
    cfg = co.make_config('simple_spread', N, max_episode_len=25, auto_reset=True, seed=21)
    o32 = co.COracle(cfg, B, np.float32)
    _assert_same_bits(_np(env.reset()), o32.reset(), 'reset obs')
    got = FusedActor(ActorNetwork(env.obs_dim, 5).cuda().eval(), seed=9).rollout(env, T)
    assert _replay_through_oracle(o32, got, T) == 2
    _assert_final_state(env, o32)


def _oracle_transitions(o32, obs0, acts, two_head=False):
    """The tuples experiments/run.py:52 hands to memory.add, built from the oracle: (obs_t, action_t, rew_shared,
    new_obs BEFORE the reset, done = 0.0), in (step, env) order."""
    T = acts.shape[0]
    obs, nxt, rew = [], [], []
    cur = obs0
    for t in range(T):
        w = o32.step(act_idx=acts[t][..., 0], act_comm=acts[t][..., 1]) if two_head else o32.step(act_idx=acts[t])
        shared = np.zeros(o32.B, np.float32)
        for i in range(o32.N):
            shared = shared + w['rew'][:, i]
        m = w['terminal'].astype(bool)
        n = w['obs'].copy()
        n[m] = w['final_obs'][m]
        obs.append(cur)
        nxt.append(n)
        rew.append(shared)
        cur = w['obs']
    return np.concatenate(obs), np.concatenate(nxt), np.concatenate(rew)


@pytest.mark.parametrize('scenario,B,kw', [('simple_spread', 256, dict(n=6)), ('simple_spread', 64, dict(n=3)),
                                          ('simple_tag', 100, dict(num_adversaries=4, num_good=2)),
                                          ('simple_reference', 150, {}), ('simple_spread', 40, dict(n=33)), ('simple_spread', 21, dict(n=24))],
                         ids=['spread6', 'spread3', 'tag4+2', 'reference', 'spread33-half-head', 'spread24'])
def test_collect_one_launch_ring_rows_equal_oracle_transitions(scenario, B, kw):
    """BatchedRollout.collect_one_launch: the rows the launches leave in the device ring -- obs, action, shared reward,
    pre-reset next_obs, done -- equal the transitions built from the oracle stepped with the ring's own actions."""
    from multiagent_rl_amd import make_batched_env
    from multiagent_rl_amd.policy import ActorNetwork, FusedActor
    from multiagent_rl_amd.replay_buffer import ReplayBuffer
    from multiagent_rl_amd.rollout import BatchedRollout
    torch.manual_seed(0)
    two = scenario == 'simple_reference'
    env = make_batched_env(scenario, B, auto_reset=True, max_episode_len=25, seed=11, **kw)
    N = env.n
    if scenario == 'simple_tag':
        cfg = co.make_config(scenario, N, num_adversaries=4, max_episode_len=25, auto_reset=True, seed=11)
    else:
        cfg = co.make_config(scenario, N, max_episode_len=25, auto_reset=True, seed=11)
    o32 = (co.CRefOracle if two else co.COracle)(cfg, B, np.float32)
    T = 57
    mem = ReplayBuffer(B * 64, N, env.obs_dim, **(dict(act_heads=(5, 10)) if two else {}))
    ro = BatchedRollout(env, FusedActor(ActorNetwork(env.obs_dim, [5, 10] if two else 5).cuda().eval(), seed=7), mem)
    obs0 = o32.reset()
    _assert_same_bits(_np(ro.obs), obs0, 'reset obs')
    ro.collect_one_launch(T, chunk=20)                                          # chunks of 20, 20, 17
    assert len(mem) == T * B
    acts = _np(mem.act[:T * B]).astype(np.int32).reshape((T, B, N, 2) if two else (T, B, N))
    want_obs, want_next, want_rew = _oracle_transitions(o32, obs0, acts, two)
    _assert_same_bits(_np(mem.obs[:T * B]).reshape(want_obs.shape), want_obs, 'ring obs')
    _assert_same_bits(_np(mem.next_obs[:T * B]).reshape(want_next.shape), want_next, 'ring next_obs')
    _assert_same_bits(_np(mem.rew[:T * B]).reshape(-1), want_rew, 'ring rew')
    assert not _np(mem.done[:T * B]).any()
    _assert_same_bits(_np(ro.obs), o32.observe(), 'observation after the last chunk')
    st = ro.stats()
    assert st['episodes'] == 2 * B
