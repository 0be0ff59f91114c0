"""multiagent_rl_amd -- MI355X-native batched particle world + rollout engine.

Drop-in for the ``MultiAgentEnv.step/reset`` surface and the ``experiments/run.py``
rollout loop of yjpark1/multiagent_rl; the arithmetic runs in hand-written HIP
kernels (csrc/pworld.hip) behind the C ABI of include/pworld.h.
"""
__version__ = '0.1.0'

from . import _lib  # noqa: F401
from .scenarios import make_env, make_batched_env  # noqa: F401
from .env import BatchedParticleEnv, MultiAgentEnv  # noqa: F401
