#!/usr/bin/env python3
"""What the LEARNER rank of an 8-GPU policy-in-the-loop run pays per chunk, measured on ONE GPU: its own rollout chunk
(pw_policy_rollout, C2: B = 4096, N = 6, T = 100), the sender-side wire launches (pw_state_wire_begin / _finalize) and the
ingest of `--blocks` wire blocks per chunk (its own + 7 peers'; here the same block appended 8 times: the root's ingest work
does not depend on whose block it is).  Modes: no ingest / ingest on the rollout's stream (what FullTransitionGather did) /
ingest on a side stream overlapping the NEXT chunk's rollout.
`--ring state` (round 5): the learner rank keeps a STATE ring (pw_replay_store.state_rows: 32 N + 8 L bytes written per transition
instead of 8 N D; rows rebuilt by pw_replay_gather when a batch is sampled); also times one append alone and one sample_index(1024).
  python3 tools/root_ingest_budget.py [--blocks 8] [--wire state|rows] [--ring rows|state]"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multiagent_rl_amd import _lib, make_batched_env
from multiagent_rl_amd.dist import FullTransitionGather
from multiagent_rl_amd.policy import ActorNetwork, FusedActor

ap = argparse.ArgumentParser()
ap.add_argument('--blocks', type=int, default=8)
ap.add_argument('--wire', default='state')
ap.add_argument('--ring', default='rows')
ap.add_argument('--chunks', type=int, default=12)
a = ap.parse_args()
B, N, T = 4096, 6, 100
dev = torch.device('cuda', 0)
lib = _lib.load()


def run(mode):
    torch.manual_seed(0)
    env = make_batched_env('simple_spread', B, n=N, auto_reset=True, max_episode_len=25, seed=1)
    actor = FusedActor(ActorNetwork(env.obs_dim, 5).to(dev).eval(), seed=2)
    full = FullTransitionGather(env, T, 0, 1, dev, wire=a.wire, capacity=int(4e6), overlap_ingest=False, ring=a.ring)
    side = torch.cuda.Stream(dev, priority=-1)
    add = lib.pw_replay_add_state_wire if full.state_wire else lib.pw_replay_add_wire
    m = full.memory
    obs0 = env.reset()
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    done = [None, None, None]
    blocks = [torch.zeros_like(full.wire[0]) for _ in range(3)]       # a third slot so that a side-stream ingest may lag a chunk
    full.wire = blocks

    def ingest(block, stream):
        for _ in range(a.blocks):
            _lib.check(add(C.byref(m._store), m._next_idx, C.byref(full.lay), C.c_void_p(block.data_ptr()),
                           C.c_void_p(stream.cuda_stream)))
            m._next_idx = (m._next_idx + T * B) % m._maxsize

    for k in range(a.chunks + 2):
        if k == 2:
            torch.cuda.synchronize()
            ev[0].record()
        slot = k % 3
        if done[slot] is not None:
            torch.cuda.current_stream().wait_event(done[slot])          # the block is free again
        full.exchanges = slot                                           # outputs() / __call__ pick wire[exchanges & 1]: steer it
        blk = blocks[slot]
        full.wire = [blk, blk, blk]
        out = full.outputs()
        actor.rollout(env, T, out)
        full._finalize(blk, obs0)
        obs0 = out['obs'][T - 1]
        if mode == 'main':
            ingest(blk, torch.cuda.current_stream())
        elif mode == 'side':
            ready = torch.cuda.Event()
            ready.record()
            side.wait_event(ready)
            ingest(blk, side)
            done[slot] = torch.cuda.Event()
            done[slot].record(side)
    torch.cuda.current_stream().wait_stream(side)
    ev[1].record()
    torch.cuda.synchronize()
    per_step = ev[0].elapsed_time(ev[1]) * 1e3 / (a.chunks * T)
    if mode == 'none':       # once: one append alone (20 back to back), and what the learner pays to sample a batch
        m._len = m._maxsize
        e2 = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        a_blocks, a.blocks = a.blocks, 20
        ingest(blocks[0], torch.cuda.current_stream())
        torch.cuda.synchronize()
        e2[0].record()
        ingest(blocks[0], torch.cuda.current_stream())
        e2[1].record()
        torch.cuda.synchronize()
        a.blocks = a_blocks
        idx = torch.randint(0, m._maxsize, (1024,), device=dev)
        m.sample_index(idx)
        torch.cuda.synchronize()
        e3 = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        e3[0].record()
        for _ in range(50):
            m.sample_index(idx)
        e3[1].record()
        torch.cuda.synchronize()
        print('    one append of %d transitions alone: %.1f us; sample_index(1024): %.1f us'
              % (T * B, e2[0].elapsed_time(e2[1]) * 1e3 / 20, e3[0].elapsed_time(e3[1]) * 1e3 / 50), flush=True)
    return per_step


for mode in ('none', 'main', 'side', 'none', 'main', 'side'):
    print('%s wire -> %s ring, %d blocks per chunk, ingest %-4s: %.2f us per batched step'
          % (a.wire, a.ring, a.blocks, mode, run(mode)), flush=True)
