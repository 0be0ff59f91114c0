#!/usr/bin/env python3
"""Measured float32 (HIP) vs float64 (C oracle) error per field and configuration -> profiles/r2_parity_error.txt.

Two kinds of states, one step each from IDENTICAL float32 states (north_star: "float32 state within 1e-5"):
  stress      the injected states of tests/test_gpu_parity.py (half the batch crowded into [-0.5, 0.5]^2 with
              |v| <= 1.5: dozens of simultaneous deep overlaps per agent at N >= 24 -- far denser than any
              reachable state);
  trajectory  states along real episodes: Philox reset, uniform random actions, 25-step episodes, 4 episodes.
Reported: max |delta| of pos, vel, obs, rew, the largest |force| seen (contact stiffness is 100 / unit of
penetration, so a float32 rounding of the distance, ~6e-8, is a force error of ~6e-6 per contact and a velocity
error of ~6e-7 per contact and step), and how many collision-mask rows differ from float64.
Run on the GPU box: python tools/parity_error.py > profiles/r2_parity_error.txt"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import c_oracle as co  # noqa: E402  (checker)
from tests.test_gpu_parity import _mk, _np, _coll, _rand_state  # noqa: E402

CONFIGS = [
    ('C1 spread N=3 B=1000', dict(scenario='simple_spread', num_agents=3, num_envs=1000)),
    ('C2 spread N=6 B=4096', dict(scenario='simple_spread', num_agents=6, num_envs=4096)),
    ('C5 spread N=12 B=4096', dict(scenario='simple_spread', num_agents=12, num_envs=4096)),
    ('C5 spread N=24 B=4096', dict(scenario='simple_spread', num_agents=24, num_envs=4096)),
    ('C5 spread N=48 B=4096', dict(scenario='simple_spread', num_agents=48, num_envs=4096)),
    ('C3 tag 4+2 B=8192', dict(scenario='simple_tag', num_agents=6, num_adversaries=4, num_envs=8192)),
]


def one_step(env, cfg, pos, vel, lm, act):
    B = pos.shape[0]
    env.set_state(pos, vel, lm)
    obs, rew, done, info = env.step(torch.from_numpy(act))
    st = env.get_state()
    o64 = co.COracle(cfg, B, np.float64)
    o64.set_state(pos, vel, lm)
    w = o64.step(act_idx=act)
    d = dict(pos=np.abs(_np(st['pos']) - o64.pos).max(), vel=np.abs(_np(st['vel']) - o64.vel).max(),
             obs=np.abs(_np(obs) - w['obs']).max(), rew=np.abs(_np(rew) - w['rew']).max(),
             rew_rel=(np.abs(_np(rew) - w['rew']) / np.maximum(1.0, np.abs(w['rew']))).max(),
             mask_rows=int(np.count_nonzero(_coll(info['coll']) ^ w['coll'])))
    # force actually applied: dv = (F / m) dt after damping  =>  |F| = |v' - 0.75 v| / dt
    f = np.abs((o64.vel - 0.75 * vel.astype(np.float64)) / 0.1)
    d['fmax'] = f.max()
    return d, _np(st['pos']), _np(st['vel'])


def main():
    print(__doc__.split('Run on')[0])
    print('%-24s %-11s %10s %10s %10s %10s %10s %9s %s' % ('config', 'states', 'max|dpos|', 'max|dvel|', 'max|dobs|',
                                                            'max|drew|', 'drew(rel)', 'max|F|', 'mask rows != f64'))
    for name, case in CONFIGS:
        env, cfg = _mk(max_episode_len=0, want_coll=True, **case)
        B, N, L = env.num_envs, env.n, env.num_landmarks
        rng = np.random.RandomState(B * 131 + N)
        pos, vel, lm = _rand_state(rng, B, N, L)
        act = rng.randint(0, 5, (B, N)).astype(np.int32)
        d, _, _ = one_step(env, cfg, pos, vel, lm, act)
        print('%-24s %-11s %10.2e %10.2e %10.2e %10.2e %10.2e %9.1f %d / %d' % (
            name, 'stress', d['pos'], d['vel'], d['obs'], d['rew'], d['rew_rel'], d['fmax'], d['mask_rows'], B * N))
        agg = dict(pos=0.0, vel=0.0, obs=0.0, rew=0.0, rew_rel=0.0, fmax=0.0, mask_rows=0)
        steps = 0
        for ep in range(4):
            env.reset()
            st = env.get_state()
            pos, vel, lm = _np(st['pos']), _np(st['vel']), _np(st['landmarks'])
            for t in range(25):
                act = rng.randint(0, 5, (B, N)).astype(np.int32)
                d, pos, vel = one_step(env, cfg, pos, vel, lm, act)
                for k in agg:
                    agg[k] = max(agg[k], d[k]) if k != 'mask_rows' else agg[k] + d[k]
                steps += 1
        print('%-24s %-11s %10.2e %10.2e %10.2e %10.2e %10.2e %9.1f %d / %d' % (
            name, 'trajectory', agg['pos'], agg['vel'], agg['obs'], agg['rew'], agg['rew_rel'], agg['fmax'],
            agg['mask_rows'], B * N * steps))


if __name__ == '__main__':
    main()
