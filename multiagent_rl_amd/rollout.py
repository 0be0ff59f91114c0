"""The rollout loop: host mirror of ``experiments/run.py`` plus its batched, device-resident form.

``run(env, actor, critic, Trainer, scenario_name, action_type, cnt)`` keeps the reference's
signature and call sequence (``experiments/run.py:11-103``): one ``get_exploration_action`` and one
``env.step`` per iteration, ``rew_shared = np.sum(rew_n)`` (:46), ``terminal`` at
``arglist.max_episode_len`` (:50), ``memory.add(obs_n, action_n_env, rew_shared, new_obs_n,
float(done))`` (:52), ``env.reset()`` on done/terminal (:59-60), ``optimize()`` every
``update_rate`` steps after ``warmup_steps`` (:78-81), history pickle + ``save_models`` at the
end (:96-103).  ``env`` and ``Trainer`` are duck-typed exactly as there, so the reference's own
Trainer classes plug in unchanged.

``BatchedRollout`` is what the MI355X is for: B envs stepped by one fused kernel per step, the
policy evaluated on [B,N,D] in HBM, transitions appended to the device replay ring -- no host
synchronisation inside the loop.
"""
import os
import pickle
import time

import numpy as np
import torch

from . import arglist as _default_arglist


def run(env, actor, critic, Trainer, scenario_name=None, action_type='Discrete', cnt=0,
        arglist=None, memory=None, out_dir='Models', log=print, per_agent_transition=False):
    """Training rollout, one env (B = 1), host loop: experiments/run.py:11-103.
    ``per_agent_transition=True`` stores the BiCNet tuple instead (per-agent ``rew_n`` and float
    ``done_n``; experiments/run_BIC.py:46,50).  Returns the history dict it also pickles."""
    return _loop(env, actor, critic, Trainer, scenario_name, action_type, cnt, arglist, memory, out_dir, log,
                 evaluate=False, per_agent_transition=per_agent_transition)


def run_test(env, actor, critic, Trainer, scenario_name=None, action_type='Discrete', cnt=0,
             arglist=None, memory=None, out_dir='Models', log=print):
    """Evaluation rollout: experiments/run.py:106-200 -- ``load_models(appx + scenario + '_fin_' + cnt)``
    first, a progress line every 10 episodes, history pickled to ``test_history_<scenario>_<cnt>.pkl``
    with the replay memory inside, no ``save_models``.  (It still calls ``optimize()``: the reference
    leaves ``arglist.is_training`` True)"""
    return _loop(env, actor, critic, Trainer, scenario_name, action_type, cnt, arglist, memory, out_dir, log,
                 evaluate=True, per_agent_transition=False)


def _loop(env, actor, critic, Trainer, scenario_name, action_type, cnt, arglist, memory, out_dir, log,
          evaluate, per_agent_transition):
    cfg = _default_arglist if arglist is None else arglist
    if action_type != 'Discrete':
        raise NotImplementedError('MultiDiscrete scenarios (communication actions) are outside the hot path')
    log('observation shape: ', env.observation_space)
    log('action shape: ', env.action_space)
    if memory is None:
        from .replay_buffer import ReplayBuffer
        memory = ReplayBuffer(size=1e+6)
    learner = Trainer(actor, critic, memory, action_type=action_type)
    if evaluate:
        learner.load_models(getattr(cfg, 'appx', '') + scenario_name + '_fin_' + str(cnt))
    report_every = 10 if evaluate else cfg.save_rate

    episode_rewards = [0.0]
    agent_rewards = [[0.0] for _ in range(env.n)]
    final_ep_rewards, final_ep_ag_rewards = [], []
    obs_n = env.reset()
    episode_step = train_step = 0
    t_start = time.time()
    log('Starting iterations...')
    while True:
        action_n = learner.get_exploration_action(obs_n)[0]
        action_n_env = [np.array(row) for row in action_n.tolist()]
        new_obs_n, rew_n, done_n, info_n = env.step(action_n_env)
        episode_step += 1
        done = all(done_n)
        terminal = episode_step >= cfg.max_episode_len
        if per_agent_transition:
            learner.memory.add(obs_n, action_n_env, rew_n, new_obs_n, [float(d) for d in done_n])
        else:
            learner.memory.add(obs_n, action_n_env, np.sum(rew_n), new_obs_n, float(done))
        obs_n = new_obs_n
        for i, rew in enumerate(rew_n):
            episode_rewards[-1] += rew
            agent_rewards[i][-1] += rew
        if done or terminal:
            obs_n = env.reset()
            episode_step = 0
            episode_rewards.append(0)
            for track in agent_rewards:
                track.append(0)
        train_step += 1
        if cfg.display:
            time.sleep(0.1)
            env.render()
            continue
        if train_step > cfg.warmup_steps and train_step % cfg.update_rate == 0 and cfg.is_training:
            learner.optimize()
        if terminal and len(episode_rewards) % report_every == 0:
            log('steps: {}, episodes: {}, mean episode reward: {}, time: {}'.format(
                train_step, len(episode_rewards), np.mean(episode_rewards[-report_every:]),
                round(time.time() - t_start, 3)))
            t_start = time.time()
            final_ep_rewards.append(np.mean(episode_rewards[-report_every:]))
            for track in agent_rewards:
                final_ep_ag_rewards.append(np.mean(track[-report_every:]))
        if len(episode_rewards) > cfg.num_episodes:
            hist = {'reward_episodes': episode_rewards, 'reward_episodes_by_agents': agent_rewards}
            if evaluate:
                hist['memory'] = memory
            if out_dir is not None:
                os.makedirs(out_dir, exist_ok=True)
                name = ('test_history_' if evaluate else 'history_') + scenario_name + '_' + str(cnt) + '.pkl'
                with open(os.path.join(out_dir, name), 'wb') as fp:
                    pickle.dump(hist, fp)
            log('...Finished total of {} episodes.'.format(len(episode_rewards)))
            if not evaluate:
                learner.save_models(scenario_name + '_fin_' + str(cnt))
            return hist


class BatchedRollout(object):
    """Device-resident rollout of a ``BatchedParticleEnv`` (auto_reset=True).

    Per step: ``actions = policy(obs)`` -> ``env.step(actions)`` (one fused launch) ->
    ``memory.add_batch(...)`` with the reference's transition tuple (obs, action, shared reward,
    next obs BEFORE reset, done) -- the same bookkeeping as run.py:44-65, vectorised over B.
    Episode returns are accumulated on the device; nothing is read back inside ``collect``.
    """

    def __init__(self, env, policy, memory=None):
        assert env.cfg.auto_reset, 'BatchedRollout needs an auto-resetting env'
        self.env, self.policy, self.memory = env, policy, memory
        self.obs = env.reset()
        B = env.num_envs
        dev = self.obs.device
        self.env_steps = 0
        self.episode_return = torch.zeros(B, device=dev)
        self.finished_return_sum = torch.zeros((), device=dev)
        self.finished_episodes = torch.zeros((), device=dev)

    def step(self):
        obs = self.obs
        actions = self.policy(obs)
        nxt, rew, done, info = self.env.step(actions)
        if self.memory is not None:
            self.memory.add_batch(obs, actions, info['rew_shared'], nxt, info.get('final_obs'), info['terminal'])
        self.episode_return += info['rew_shared']
        term = info['terminal']
        self.finished_return_sum += (self.episode_return * term).sum()
        self.finished_episodes += term.sum()
        self.episode_return *= ~term
        self.obs = nxt
        self.env_steps += self.env.num_envs
        return nxt, rew, done, info

    def collect(self, num_steps):
        for _ in range(num_steps):
            self.step()
        return self.env_steps

    def stats(self):
        """Host read-back (synchronises): mean return of the episodes finished so far."""
        n = float(self.finished_episodes.item())
        return dict(env_steps=self.env_steps, episodes=int(n),
                    mean_episode_reward=float(self.finished_return_sum.item()) / n if n else float('nan'))
