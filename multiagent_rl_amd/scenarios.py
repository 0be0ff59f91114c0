"""Mirror of ``experiments/scenarios.py`` (the reference's environment factory).

``make_env`` keeps the reference's name, arguments and behaviour
(experiments/scenarios.py:124-192) for the scenarios on the hot path
(BASELINE.json: simple_spread, simple_tag) and returns an object with the
``MultiAgentEnv`` surface ``experiments/run.py`` consumes, backed by the HIP
kernels.  ``make_batched_env`` is the [B x N] tensor form of the same env.
"""
from .env import BatchedParticleEnv, MultiAgentEnv

SUPPORTED = ('simple_spread', 'simple_tag', 'simple_reference', 'simple_speaker_listener')


def make_env(scenario_name, n=None, local_observation=True, benchmark=False, discrete_action=True, **kw):
    """Same contract as experiments/scenarios.py:124: per-agent rewards
    (``world.collaborative = False``, :171), ``force_discrete_action = True`` (:191),
    local observation patched in for simple_spread (:151-153), ``n`` ->
    ``make_world(num_agents=n)`` (:167-170)."""
    if scenario_name not in SUPPORTED:
        # the reference prints 'error: unsupported scenario!' and then fails inside MPE
        raise ValueError('error: unsupported scenario! %r (supported: %s)' % (scenario_name, ', '.join(SUPPORTED)))
    env = MultiAgentEnv(scenario_name, n=n, local_observation=local_observation, benchmark=benchmark,
                        discrete_action=discrete_action, **kw)
    env.force_discrete_action = True
    return env


def make_batched_env(scenario_name, num_envs, n=None, local_observation=True, **kw):
    """[B x N] env on the current GPU; same scenario constants as ``make_env``."""
    if scenario_name not in SUPPORTED:
        raise ValueError('error: unsupported scenario! %r (supported: %s)' % (scenario_name, ', '.join(SUPPORTED)))
    if scenario_name == 'simple_tag':
        if n is not None:
            kw.setdefault('num_agents', n)
    elif scenario_name == 'simple_spread':
        kw['num_agents'] = n
    kw.setdefault('force_discrete_action', True)
    return BatchedParticleEnv(scenario_name, num_envs, local_observation=local_observation, **kw)
