#!/usr/bin/env python3
"""Checks the output of examples/c_host (a plain-C caller of libpworld.so) against the CPU oracle, bit for bit.
Uses the oracle only (no GPU, no torch):  python tools/check_c_host.py gpurun_out/c_host.txt"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import c_oracle as co  # noqa: E402  (checker only)

lines = open(sys.argv[1]).read().split('\n')
hdr = lines[0].split()
B, N, D, T, seed = int(hdr[3]), int(hdr[5]), int(hdr[7]), int(hdr[9]), int(hdr[11])
got = {k: np.array([int(l[2:], 16) for l in lines if l.startswith(k + ' ')], np.uint32) for k in 'ors'}
cfg = co.make_config('simple_spread', N, max_episode_len=25, auto_reset=True, seed=seed)
o = co.COracle(cfg, B, np.float32)
o.reset()
s, acts = 12345, np.zeros(T * B * N, np.int32)
for i in range(acts.size):
    s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
    acts[i] = (s >> 16) % 5
acts = acts.reshape(T, B, N)
terms = 0
for t in range(T):
    w = o.step(act_idx=acts[t])
    terms += int(w['terminal'].sum())
assert int(lines[1].split()[1]) == terms == B, (lines[1], terms)
ring = lines[2].split()
assert ring[0] == 'state_ring' and ring[3:] == ['next_obs_equal', '1', 'obs_equal', '1', 'rew_equal', '1'], lines[2]   # the C side's own bitwise check
assert np.array_equal(got['o'], w['obs'].reshape(-1).view(np.uint32)), 'observations differ'
assert np.array_equal(got['r'], w['rew'].reshape(-1).view(np.uint32)), 'rewards differ'
shared = ((np.float32(0) + w['rew'][:, 0]) + w['rew'][:, 1]) + w['rew'][:, 2]
assert np.array_equal(got['s'], shared.view(np.uint32)), 'shared rewards differ'
print('c_host ok: %d observations, %d rewards, %d shared rewards identical to the float32 oracle; state-only wire block (%s B per env-step) -> STATE ring -> '
      'pw_replay_gather rebuilt the rows bit for bit; %s' % (got['o'].size, got['r'].size, got['s'].size, ring[2], lines[0]))
