"""CPU: ``python bench.py --gpus 2 --steps K --warmup W`` from a plain shell starts its own rank processes
(torch.distributed.run children of a parent that never imports torch) and rank 0 prints ONE JSON line -- the
driver's command line, exercised with ``PW_BENCH_STUB=1`` (gloo, a rollout "env" that launches nothing and torch
stand-ins for the HIP launches of the exchanges; the line says ``data: stub``).  What this covers is the launcher,
the rank plumbing, both exchanges' choreography inside bench.py and the shape of the line; the kernels are
covered by the -m gpu tests."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None, timeout=240):
    env = dict(os.environ, PW_BENCH_STUB='1', PYTHONPATH=ROOT)
    env.pop('RANK', None)
    env.pop('WORLD_SIZE', None)
    env.update(extra_env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, text=True)
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    return p, lines


@pytest.mark.timeout(300)
@pytest.mark.skipif(torch.cuda.device_count() > 0, reason='CPU-container test (stub kernel, gloo)')
def test_bench_gpus2_self_launch_prints_one_line():
    p, lines = _run(['--gpus', '2', '--steps', '20', '--warmup', '5', '--envs', '8', '--agents', '3', '--chunk', '50',
                     '--batch-size', '8', '--policy-steps', '100', '--policy-chunk', '50'])
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['steps'] == 20 and line['warmup'] == 5 and line['scaling'] == 'weak'
    assert line['data'].startswith('stub')
    cfg = line['config']
    assert cfg['batched_env_steps_per_launch'] == 50 and cfg['batched_env_steps_timed'] == 1000
    assert cfg['env_steps_per_step'] == 2 * 8 * 50 and cfg['global_batch'] == 16
    assert '50 batched env steps' in cfg['workload'] and '20 launches timed' in cfg['workload']
    ex = cfg['exchange']
    assert ex['error'] is None and ex['exchanges'] == 20 and ex['rows_ingested_root'] == 20 * 8
    assert abs(line['value'] - 2 * 8 * 50 * 20 / cfg['timed_region_s']) < 1e-6 * line['value']
    assert abs(line['ms_per_step'] - cfg['timed_region_s'] * 1e3 / 20) < 1e-9
    g = line['policy_in_loop']['gather']
    assert g['error'] is None and g['transitions_ingested_root'] == g['expected_transitions'] == 2 * 8 * 100
    assert abs(g['bytes_per_env_step'] - g['bytes_per_chunk_per_rank'] / (50 * 8)) < 1e-9
    assert 'cpu_baseline' not in line  # N = 1 only


@pytest.mark.timeout(400)
@pytest.mark.skipif(torch.cuda.device_count() > 0, reason='CPU-container test (stub kernel, gloo)')
def test_bench_gpus8_c4_shape_self_launch():
    """BASELINE configs[3]'s rank count (8 x a small B): the sampled exchange of the headline and the 7-receive full gather
    of the policy extra at the real fan-in, rank-ordered ingest, and the N > 1 line's top-level exchange summary."""
    p, lines = _run(['--gpus', '8', '--steps', '6', '--warmup', '2', '--envs', '4', '--agents', '3', '--chunk', '50',
                     '--batch-size', '16', '--policy-steps', '100', '--policy-chunk', '50'], timeout=380)
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line['n_gpus'] == 8 and line['config']['global_batch'] == 32 and line['config']['parallelism'] == 'env-shard x8'
    ex = line['config']['exchange']
    assert ex['error'] is None and ex['exchanges'] == 6 and ex['rows_ingested_root'] == 6 * 16
    g = line['policy_in_loop']['gather']
    assert g['error'] is None and g['transitions_ingested_root'] == g['expected_transitions'] == 8 * 4 * 100
    mg = line['multi_gpu']
    assert mg['full_gather']['transitions_ingested_root'] == mg['full_gather']['expected_transitions'] == 8 * 4 * 100
    assert mg['full_gather']['GBps_per_link'] == g['GBps_per_link'] and mg['full_gather']['peers'] == 7
    assert mg['full_gather']['root_receive_bytes'] == 3 * 7 * g['bytes_per_chunk_per_rank']      # three slots of one block per peer
    assert mg['sampled_exchange']['value_without_exchange'] > 0 and mg['sampled_exchange']['launches_timed'] == 6
    assert abs(mg['sampled_exchange']['value_with_exchange'] - line['value']) < 1e-6 * line['value']


@pytest.mark.timeout(300)
@pytest.mark.skipif(torch.cuda.device_count() > 0, reason='CPU-container test (stub kernel, gloo)')
def test_bench_under_torchrun_env_does_not_relaunch():
    """Started by the driver's own torch.distributed.run (RANK set): no second launcher, one rank = one process."""
    p, lines = _run(['--gpus', '1', '--steps', '3', '--warmup', '1', '--envs', '4', '--agents', '3', '--chunk', '25',
                     '--policy-steps', '50', '--policy-chunk', '25'],
                    extra_env=dict(RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', PW_BENCH_FORCE_DIST='1',
                                   MASTER_ADDR='127.0.0.1', MASTER_PORT='29577'))
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(lines[0])
    assert line['n_gpus'] == 1 and line['config']['exchange']['exchanges'] == 3
    assert line['policy_in_loop']['gather']['transitions_ingested_root'] == 4 * 50


@pytest.mark.timeout(300)
@pytest.mark.skipif(torch.cuda.device_count() > 0, reason='CPU-container test (stub kernel, gloo)')
def test_a_dying_peer_in_the_policy_extra_does_not_cost_the_line():
    """Rank 1 fails inside the full-gather extra; the root would wait for its block for ever.  The failing rank flags it in
    the process group's store, the root's watchdog prints the (already measured) headline line with that error within a
    second -- not after the policy timeout -- and the run exits NON-ZERO: a run that did not finish is not a success."""
    p, lines = _run(['--gpus', '2', '--steps', '4', '--warmup', '1', '--envs', '8', '--agents', '3', '--chunk', '50',
                     '--batch-size', '8', '--policy-steps', '400', '--policy-chunk', '50', '--policy-timeout', '60',
                     '--exit-timeout', '4'], extra_env=dict(PW_BENCH_STUB_FAIL_RANK='1'), timeout=200)
    assert p.returncode != 0, 'a run with a dead peer must not report success'
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])           # strict JSON: no NaN
    assert line['n_gpus'] == 2 and line['value'] > 0 and line['config']['exchange']['error'] is None
    err = line['policy_in_loop']['error']
    assert 'did not finish' in err and 'rank 1' in err and 'injected failure' in err


@pytest.mark.timeout(300)
@pytest.mark.skipif(torch.cuda.device_count() > 0, reason='CPU-container test (stub kernel, gloo)')
def test_a_hung_policy_extra_times_out_with_the_line_and_a_nonzero_code():
    """The same, but the store flag never arrives (the peer hangs instead of raising): the watchdog's deadline prints the
    line and the exit code is non-zero."""
    p, lines = _run(['--gpus', '2', '--steps', '4', '--warmup', '1', '--envs', '8', '--agents', '3', '--chunk', '50',
                     '--batch-size', '8', '--policy-steps', '400', '--policy-chunk', '50', '--policy-timeout', '5',
                     '--exit-timeout', '4'], extra_env=dict(PW_BENCH_STUB_HANG_RANK='1'), timeout=200)
    assert p.returncode != 0
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line['value'] > 0 and 'did not finish within 5 s' in line['policy_in_loop']['error']


def test_profiler_preload_refuses_the_rank_launcher():
    """bench.py --gpus N from a process a profiler has already GPU-initialised must not exec the launcher."""
    p, lines = _run(['--gpus', '2', '--steps', '2'], extra_env=dict(LD_PRELOAD='librocprofiler-sdk-tool.so'))
    assert p.returncode != 0 and not lines
    assert 'profiler preload' in p.stderr and 'PW_BENCH_FORCE_DIST=1' in p.stderr


def test_self_launch_command_line():
    """The child command is torch.distributed.run on 127.0.0.1 with the same arguments; built without torch."""
    sys.path.insert(0, ROOT)
    import bench
    seen = {}
    real = subprocess.call
    try:
        subprocess.call = lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 7
        rc = bench.self_launch(4, ['--gpus', '4', '--steps', '20', '--warmup', '5'])
    finally:
        subprocess.call = real
    cmd = seen['cmd']
    assert rc == 7 and cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nproc-per-node' in cmd
    assert cmd[cmd.index('--nproc-per-node') + 1] == '4' and cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert cmd[-6:] == ['--gpus', '4', '--steps', '20', '--warmup', '5'] and cmd[-7].endswith('bench.py')
    assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
