"""Measurement helper: latency of the B = 1 drop-in MultiAgentEnv.step (lists of ndarrays in and out)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiagent_rl_amd import make_env
env = make_env('simple_spread', n=6)
np.random.seed(0)
obs = env.reset()
acts = [np.eye(5)[i % 5] for i in range(6)]
for _ in range(20): env.step(acts)
t0 = time.perf_counter()
for _ in range(500): o, r, d, i = env.step(acts)
dt = (time.perf_counter() - t0) / 500
print('MultiAgentEnv.step (B=1 compat, N=6): %.1f us/step -> %.0f env-steps/s' % (dt * 1e6, 1 / dt))
