"""-m gpu: TWO (and three) processes sharing GPU 0 -- the multi-rank path with the product's HIP launches in every rank.

The gloo tests (tests/test_dist_gloo.py, tests/test_train_dist_gloo.py) run torch stand-ins for the HIP launches on the CPU; the
single-rank RCCL test (test_gpu_engine.py::test_exchanges_on_real_rccl_single_rank) runs the real launches with nobody to talk to.
RCCL refuses two ranks on one device, so here the blocks of the full gather travel through pinned host buffers over a gloo group
(FullTransitionGather(transport='host')); the rollout, finalize, append and gather launches are the real ones in each process."""
import json
import os
import pickle
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    sk = socket.socket()
    sk.bind(('127.0.0.1', 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def _launch(world, argv, cwd=ROOT, timeout=420):
    """`world` ranks of `argv` (RANK / WORLD_SIZE / MASTER_* in the env), all on GPU 0; returns rank 0's stdout."""
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   PYTHONPATH=ROOT + os.pathsep + os.environ.get('PYTHONPATH', ''))
        procs.append(subprocess.Popen([sys.executable] + argv, cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, 'rank %d exited %s\n%s\n%s' % (r, p.returncode, so[-2000:], se[-4000:])
    return outs[0][0]


@pytest.mark.timeout(600)
def test_full_gather_of_two_ranks_on_one_gpu_fills_the_root_ring_rank_by_rank():
    """Every wire / ring pairing (state blocks -> STATE ring, state blocks -> row ring, row blocks, simple_tag both rings, the N = 24
    rollout form, simple_reference's compact rows -> two-head ring), 4-5 chunks each (the three block slots wrap): the root's ring
    holds rank r's chunk x at [(x*2 + r)*T*B, +T*B), bit-identical to the ring rank r's own rollout sink fills."""
    out = _launch(2, [os.path.join(ROOT, 'tests', 'two_rank_gpu_worker.py')])
    res = [json.loads(l) for l in out.splitlines() if l.startswith('{')]
    assert [r['case'] for r in res] == ['spread_state_ring', 'spread_n3_rows_ring', 'spread_row_blocks', 'spread_n24_state_ring',
                                        'tag_state_ring', 'tag_rows_ring', 'reference_two_head']
    assert all(r['ok'] and r['world'] == 2 for r in res)
    by = {r['case']: r for r in res}
    assert by['spread_state_ring']['wire'] == 'state' and by['spread_row_blocks']['wire'] == 'rows' and by['reference_two_head']['wire'] == 'ref'
    assert by['spread_state_ring']['transitions'] == 2 * 5 * 40 * 512


@pytest.mark.timeout(600)
def test_full_gather_of_three_ranks_on_one_gpu():
    out = _launch(3, [os.path.join(ROOT, 'tests', 'two_rank_gpu_worker.py'), 'spread_state_ring', 'tag_state_ring', 'reference_two_head'])
    res = [json.loads(l) for l in out.splitlines() if l.startswith('{')]
    assert len(res) == 3 and all(r['ok'] and r['world'] == 3 for r in res)


@pytest.mark.timeout(600)
def test_entry_script_on_two_ranks_sharing_the_gpu(tmp_path):
    """examples/train_batched.py --backend gloo as two ranks (256 envs each, 50-step chunks): both ranks roll out, rank 0 owns the
    STATE ring and the learner, the actor goes back after every batch of updates; the pickled history carries both ranks' episodes."""
    out = _launch(2, [os.path.join(ROOT, 'examples', 'train_batched.py'), '--backend', 'gloo', '--scenario', 'simple_spread', '--envs', '256',
                      '--agents', '3', '--episodes', '1024', '--chunk', '50', '--save-rate', '512', '--max-updates-per-chunk', '3',
                      '--out-dir', str(tmp_path / 'Models')], cwd=str(tmp_path))
    assert 'simple_spread cnt=0: 25600 env-steps, 1024 episodes' in out, out[-2000:]
    hist = pickle.load(open(tmp_path / 'Models' / 'history_simple_spread_0.pkl', 'rb'))
    assert hist['episodes_per_rank'] == [1024 + 256, 1024 + 256] and hist['open_episodes'] == [256, 256]
    assert len(hist['reward_episodes']) == 2 * (1024 + 256) and len(hist['reward_episodes_by_agents']) == 3
    st = hist['stats']
    # 2 x 256 envs x 50 steps = 25 600 env-steps per chunk -> ~250 gate openings per chunk, capped at 3; the first chunk's are skipped
    # (the ring is filled one chunk late), the second chunk's run on the 25 600 transitions of chunk one
    assert st['world'] == 2 and st['updates_run'] == 3 and 400 < st['updates_owed'] <= 512 and np.isfinite(hist['reward_episodes']).all()
    assert os.path.exists(tmp_path / 'Models' / 'simple_spread_fin_0_actor.pt')
    # the two ranks' shards are different worlds (env_id_base): their first episodes differ
    assert hist['reward_episodes'][:256] != hist['reward_episodes'][1280:1280 + 256]
