"""The action producer of the rollout: the reference's actor architecture and its
Gumbel-softmax exploration, batched over [B envs x N agents] and kept on the device.

``ActorNetwork`` has the layer structure AND parameter names of
``rls/model/ac_network_multi_gumbel.py:24-67`` (``dense1.module``, ``bilstm``,
``dense2.module``), so state_dicts saved by the reference's Trainer
(``ddpg_gumbel_fix.py:221-229``) load unchanged.  It stays stock PyTorch-ROCm
(rocBLAS / MIOpen on MFMA) as BASELINE.json's north_star prescribes; the
hand-written HIP work is the environment, not the policy GEMMs.

``GumbelPolicy`` is the batched form of ``Trainer.get_exploration_action`` /
``gumbel_softmax(hard=True)`` (``ddpg_gumbel_fix.py:86-116``): the hard one-hot of
``F.gumbel_softmax`` is ``argmax(logits + g)`` with ``g = -log(Exponential(1))``, drawn here with
the same RNG call, but the result stays an int32 index tensor [B,N] in HBM -- no
``.cpu().numpy()`` round trip per env-step.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class TimeDistributed(nn.Module):
    """Applies ``module`` over the last axis of [batch, agents, features]."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, x):
        if x.dim() <= 2:
            return self.module(x)
        lead = x.shape[:2]
        return self.module(x.reshape(lead[0] * lead[1], x.shape[2])).reshape(lead[0], lead[1], -1)


class ActorNetwork(nn.Module):
    """Linear(D, 64) -> ReLU -> BiLSTM(64 -> 2x32) over the AGENT axis -> ReLU -> Linear(64, 5)."""

    def __init__(self, input_dim, out_dim):
        super().__init__()
        self.out_dim = out_dim
        self.dense1 = TimeDistributed(nn.Linear(input_dim, 64))
        self.bilstm = nn.LSTM(64, 32, num_layers=1, batch_first=True, bidirectional=True)
        self.dense2 = TimeDistributed(nn.Linear(64, out_dim))

    def forward(self, obs):
        hid = F.relu(self.dense1(obs))
        hid, _ = self.bilstm(hid, None)
        return self.dense2(F.relu(hid))


class GumbelPolicy(object):
    """obs [B,N,D] (device) -> action index [B,N] int32 (device)."""

    def __init__(self, actor, generator=None):
        self.actor = actor
        self.generator = generator

    @torch.no_grad()
    def logits(self, obs):
        return self.actor(obs)

    @torch.no_grad()
    def __call__(self, obs):
        logits = self.actor(obs)
        gumbels = -torch.empty_like(logits).exponential_(generator=self.generator).log()
        return (logits + gumbels).argmax(dim=-1).to(torch.int32)


class UniformRandomPolicy(object):
    """i.i.d. uniform action indices (the synthetic-action workload of bench.py)."""

    def __init__(self, num_actions=5, generator=None):
        self.num_actions, self.generator = num_actions, generator

    def __call__(self, obs):
        return torch.randint(0, self.num_actions, obs.shape[:2], device=obs.device, dtype=torch.int32,
                             generator=self.generator)
