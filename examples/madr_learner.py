"""A minimal stock-PyTorch learner with the Trainer surface ``experiments/run.py:21,37,52,81,102`` consumes, so that
``examples/train_batched.py`` runs where the reference checkout (and its ``rls`` package) is not importable.

NOT part of the product and not the reference's algorithm in detail: the learner is out of scope of this repo (north_star: it
stays stock PyTorch-ROCm); inside a checkout of the reference, pass its own
``rls.agent.multiagent.ddpg_gumbel_fix.Trainer`` / ``rls.model.ac_network_multi_gumbel.CriticNetwork`` instead (``--reference``).
What it keeps: DDPG with hard Gumbel-softmax categorical actions, one shared reward per transition, target networks with
soft updates, Adam, batches drawn through ``memory.make_index`` / ``memory.sample_index``.
"""
import copy
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

GAMMA, TAU = 0.95, 1e-2   # rls/agent/multiagent/ddpg_gumbel_fix.py:10, rls/arglist.py:12


class CriticNetwork(nn.Module):
    """Q(obs [b,N,D], action one-hots [b,N,A]) -> [b,1]: per-agent embedding, mean over the agent axis, two dense layers."""

    def __init__(self, input_dim, out_dim=1):
        super().__init__()
        self.embed = nn.Linear(input_dim, 64)
        self.mix = nn.Linear(64, 64)
        self.out = nn.Linear(64, out_dim)

    def forward(self, obs, action):
        h = F.relu(self.embed(torch.cat([obs, action], dim=-1))).mean(dim=1)
        return self.out(F.relu(self.mix(h)))


class Trainer(object):
    def __init__(self, actor, critic, memory, action_type='Discrete', batch_size=1024, lr=1e-2, device=None):
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.actor, self.critic = actor.to(self.device), critic.to(self.device)
        self.target_actor, self.target_critic = copy.deepcopy(self.actor).eval(), copy.deepcopy(self.critic).eval()
        self.actor_optimizer = torch.optim.Adam(self.actor.parameters(), lr)
        self.critic_optimizer = torch.optim.Adam(self.critic.parameters(), lr)
        self.memory, self.action_type, self.batch_size = memory, action_type, batch_size
        self.iter = 0

    @staticmethod
    def _sample(logits):
        if isinstance(logits, (list, tuple)):       # MultiDiscrete: one one-hot per head, concatenated (run.py:39-41)
            return torch.cat([F.gumbel_softmax(x, hard=True, dim=-1) for x in logits], dim=-1)
        return F.gumbel_softmax(logits, hard=True, dim=-1)

    @torch.no_grad()
    def get_exploration_action(self, state):
        import numpy as np
        obs = torch.from_numpy(np.array([np.stack(state)], dtype='float32')).to(self.device)
        logits = self.actor(obs)
        if isinstance(logits, (list, tuple)):
            return [F.gumbel_softmax(x, hard=True, dim=-1).cpu().numpy() for x in logits]
        return F.gumbel_softmax(logits, hard=True, dim=-1).cpu().numpy()

    def optimize(self):
        s0, a0, r, s1, d = (torch.as_tensor(x, dtype=torch.float32, device=self.device)
                            for x in self.memory.sample_index(self.memory.make_index(self.batch_size)))
        with torch.no_grad():
            y = r[:, None] + GAMMA * (1.0 - d[:, None]) * self.target_critic(s1, self._sample(self.target_actor(s1)))
        loss_critic = F.smooth_l1_loss(self.critic(s0, a0), y)
        self.critic_optimizer.zero_grad()
        loss_critic.backward()
        nn.utils.clip_grad_norm_(self.critic.parameters(), 0.5)
        self.critic_optimizer.step()
        loss_actor = -self.critic(s0, self._sample(self.actor(s0))).mean()
        self.actor_optimizer.zero_grad()
        loss_actor.backward()
        nn.utils.clip_grad_norm_(self.actor.parameters(), 0.5)
        self.actor_optimizer.step()
        with torch.no_grad():
            for tgt, src in ((self.target_actor, self.actor), (self.target_critic, self.critic)):
                for pt, ps in zip(tgt.parameters(), src.parameters()):
                    pt.mul_(1.0 - TAU).add_(ps, alpha=TAU)
        self.iter += 1
        return float(loss_actor.detach()), float(loss_critic.detach())

    def save_models(self, name, out_dir='Models'):
        """Target nets' state_dicts as ``<name>_actor.pt`` / ``<name>_critic.pt`` (ddpg_gumbel_fix.py:221-229)."""
        os.makedirs(out_dir, exist_ok=True)
        torch.save(self.target_actor.state_dict(), os.path.join(out_dir, name + '_actor.pt'))
        torch.save(self.target_critic.state_dict(), os.path.join(out_dir, name + '_critic.pt'))

    def load_models(self, name, out_dir='Models'):
        self.actor.load_state_dict(torch.load(os.path.join(out_dir, name + '_actor.pt')))
        self.critic.load_state_dict(torch.load(os.path.join(out_dir, name + '_critic.pt')))
        self.target_actor.load_state_dict(self.actor.state_dict())
        self.target_critic.load_state_dict(self.critic.state_dict())
