#!/usr/bin/env python3
"""Self-regression vectors of THIS repo's CPU oracle (not of the reference -- its physics is absent, see
oracle/particle_oracle.py): fixed seeds and actions -> trajectories from the scalar NumPy float64 oracle.
They pin the oracle across rounds: an accidental change of the restatement shows up as a diff here.
Run from the repo root:  python tests/golden/make_oracle_vectors.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import particle_oracle as po  # noqa: E402

CASES = [('simple_spread', dict(), 12345678), ('simple_spread', dict(n=6), 12345679),
         ('simple_tag', dict(), 12345680), ('simple_tag', dict(num_good=2, num_adversaries=4), 12345681),
         ('simple_reference', dict(), 12345682), ('simple_speaker_listener', dict(), 12345683)]


def trajectory(name, kw, seed, steps=30):
    np.random.seed(seed)                       # main.py:47
    env = po.make_oracle_env(name, **kw)
    np.random.seed(seed)
    obs = env.reset()
    rng = np.random.RandomState(seed % 1000)
    multi = hasattr(env.action_space[0], 'high')
    out = dict(obs0=np.stack([np.pad(o, (0, 32 - len(o))) for o in obs]), actions=[], obs=[], rew=[])
    for t in range(steps):
        idx = rng.randint(0, 5, env.n)
        cidx = rng.randint(0, 10, env.n)
        if name == 'simple_speaker_listener':  # per-agent heads: Discrete(3) speaker, Discrete(5) listener
            idx[0] %= 3
            acts = [np.eye(sp.n)[i] for sp, i in zip(env.action_space, idx)]
        else:
            acts = [np.concatenate([np.eye(5)[i], np.eye(10)[c]]) if multi else np.eye(5)[i] for i, c in zip(idx, cidx)]
        o, r, d, _ = env.step(acts)
        out['actions'].append(np.stack([idx, cidx], -1))
        out['obs'].append(np.stack([np.pad(x, (0, 32 - len(x))) for x in o]))
        out['rew'].append(np.array(r, np.float64))
        if (t + 1) % 25 == 0:                  # run.py:59-60
            env.reset()
    return {k: np.asarray(v) for k, v in out.items()}


if __name__ == '__main__':
    arrays = {}
    for i, (name, kw, seed) in enumerate(CASES):
        for k, v in trajectory(name, kw, seed).items():
            arrays['%d/%s' % (i, k)] = v
    np.savez_compressed(os.path.join(HERE, 'oracle_vectors.npz'), **arrays)
    print('oracle_vectors.npz:', len(arrays), 'arrays')
