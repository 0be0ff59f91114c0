"""Recording wrappers shared by tests/golden/make_golden.py (reference loop) and
tests/test_rollout_host.py (this repo's loop): same env, same stub Trainer, comparable traces."""
import numpy as np


def fingerprint(x):
    """Type/shape/dtype skeleton + a rounded checksum of nested containers of arrays."""
    if isinstance(x, (list, tuple)):
        return {'type': type(x).__name__, 'items': [fingerprint(i) for i in x]}
    if isinstance(x, np.ndarray):
        return {'type': 'ndarray', 'shape': list(x.shape), 'dtype': str(x.dtype),
                'sum': round(float(np.sum(x)), 9)}
    if isinstance(x, (bool, np.bool_)):
        return {'type': 'bool', 'value': bool(x)}
    if isinstance(x, (float, np.floating)):
        return {'type': type(x).__name__, 'value': round(float(x), 9)}
    if isinstance(x, (int, np.integer)):
        return {'type': 'int', 'value': int(x)}
    if isinstance(x, dict):
        return {'type': 'dict', 'keys': sorted(x.keys())}
    return {'type': type(x).__name__}


class RecordingEnv(object):
    def __init__(self, env):
        self.env = env
        self.n = env.n
        self.observation_space = env.observation_space
        self.action_space = env.action_space
        self.trace = []

    def seed(self, s=None):
        return self.env.seed(s)

    def reset(self):
        obs = self.env.reset()
        self.trace.append(['reset', fingerprint(obs)])
        return obs

    def step(self, action_n):
        self.trace.append(['step_in', fingerprint(action_n)])
        out = self.env.step(action_n)
        self.trace.append(['step_out', fingerprint(list(out[:3])), fingerprint(out[3])])
        return out

    def render(self):
        self.trace.append(['render'])


class RecordingMemory(object):
    """Host stand-in with the ReplayBuffer.add signature (rls/replay_buffer.py:30)."""

    def __init__(self):
        self._storage = []

    def add(self, obs_t, action, reward, obs_tp1, done):
        self._storage.append((obs_t, action, reward, obs_tp1, done))

    def __len__(self):
        return len(self._storage)


class StubTrainer(object):
    """Trainer surface of experiments/run.py:21,37,52,81,102 with deterministic random one-hot actions."""
    trace = None
    last = None

    def __init__(self, actor, critic, memory, action_type='Discrete'):
        self.memory = memory
        self.action_type = action_type
        self.rng = np.random.RandomState(7)
        StubTrainer.last = self
        self.trace.append(['trainer_init', action_type])

    def get_exploration_action(self, obs_n):
        self.trace.append(['act', fingerprint(obs_n)])
        n = len(obs_n)
        if self.action_type == 'MultiDiscrete':  # one array per head, as ddpg_gumbel_fix.py:101-105 returns
            return [np.eye(5, dtype=np.float32)[self.rng.randint(0, 5, n)][None],
                    np.eye(10, dtype=np.float32)[self.rng.randint(0, 10, n)][None]]
        return np.eye(5, dtype=np.float32)[self.rng.randint(0, 5, n)][None]

    def optimize(self):
        self.trace.append(['optimize', len(self.memory)])
        return 0.0, 0.0

    def save_models(self, name):
        self.trace.append(['save_models', name])

    def load_models(self, name):
        self.trace.append(['load_models', name])
