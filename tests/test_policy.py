"""CPU: the actor mirror loads the reference's state_dict and reproduces its logits; the batched
Gumbel sampler picks the same actions as ddpg_gumbel_fix.Trainer.gumbel_softmax(hard=True)
(tests/golden/actor_forward.npz, generated from the reference by tests/golden/make_golden.py)."""
import os

import numpy as np
import torch

from multiagent_rl_amd.policy import ActorNetwork, GumbelPolicy, UniformRandomPolicy

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'actor_forward.npz'))


def _actor():
    sd = {k[3:]: torch.from_numpy(G[k]) for k in G.files if k.startswith('sd/')}
    actor = ActorNetwork(input_dim=G['obs'].shape[-1], out_dim=5)
    assert sorted(actor.state_dict().keys()) == sorted(sd.keys())     # same parameter names
    actor.load_state_dict(sd)
    return actor.eval()


def test_actor_matches_reference_logits():
    with torch.no_grad():
        logits = _actor()(torch.from_numpy(G['obs']))
    np.testing.assert_allclose(logits.numpy(), G['logits'], rtol=0, atol=1e-6)


def test_gumbel_policy_matches_reference_one_hot():
    pol = GumbelPolicy(_actor())
    torch.manual_seed(int(G['gumbel_seed']))
    idx = pol(torch.from_numpy(G['obs']))
    assert idx.dtype == torch.int32 and tuple(idx.shape) == G['onehot'].shape[:2]
    assert np.array_equal(np.eye(5, dtype=np.float32)[idx.numpy()], G['onehot'])
    assert (G['onehot'].sum(-1) == 1).all()


def test_uniform_policy_shape():
    a = UniformRandomPolicy()(torch.zeros(4, 6, 16))
    assert tuple(a.shape) == (4, 6) and a.dtype == torch.int32 and int(a.min()) >= 0 and int(a.max()) <= 4
