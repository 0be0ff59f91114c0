#!/usr/bin/env python3
"""Turn rocprofv3 CSV output (kernel trace + separate --pmc FETCH_SIZE / WRITE_SIZE passes) into the
summaries committed under profiles/.  Usage:
  python tools/summarize_prof.py --tag r1 --stats gpurun_out/prof_r1 --fetch gpurun_out/pmc_fetch_r1 \
      --write gpurun_out/pmc_write_r1 --kernel pw_spread_stream_kernel --bench gpurun_out/bench_r1.json
PMC handling follows /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 PMC slots): FETCH_SIZE and
WRITE_SIZE in separate passes (TCC slots), unit KiB; WRITE_SIZE is exact for 16-B-per-lane streaming stores
(this kernel's observation rows); FETCH_SIZE under-reports WIDE (16 B/lane) coalesced reads by 2x -- this
kernel's reads are 4 B/lane action indices, a width the guide leaves uncalibrated, so the read side is
reported as measured (it is 5 % of the traffic)."""
import argparse
import collections
import csv
import glob
import json
import os
import shutil


def pmc_mean(d, counter, kernel):
    f = glob.glob(os.path.join(d, '**', '*_counter_collection.csv'), recursive=True)[0]
    vals = [float(r['Counter_Value']) for r in csv.DictReader(open(f))
            if r['Counter_Name'] == counter and kernel in r['Kernel_Name']]
    return sum(vals) / len(vals), len(vals)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--tag', required=True)
    ap.add_argument('--stats', required=True)
    ap.add_argument('--fetch')
    ap.add_argument('--write')
    ap.add_argument('--kernel', default='pw_spread_stream_kernel')
    ap.add_argument('--bench')
    ap.add_argument('--sq', nargs='*', default=[], help='directories of the SQ counter passes (SQ_INSTS_VALU, SQ_WAVES, '
                                                       'SQ_WAVE_CYCLES, SQ_INSTS_LDS, SQ_INSTS_SALU, SQ_BUSY_CYCLES)')
    ap.add_argument('--out', default='profiles')
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    ks = glob.glob(os.path.join(a.stats, '**', '*_kernel_stats.csv'), recursive=True)[0]
    shutil.copy(ks, os.path.join(a.out, '%s_kernel_stats.csv' % a.tag))
    rows = list(csv.DictReader(open(ks)))
    k = [r for r in rows if a.kernel in r['Name']][0]
    summ = collections.OrderedDict(tag=a.tag, kernel=k['Name'], calls=int(k['Calls']),
                                   avg_ns=float(k['AverageNs']), min_ns=float(k['MinNs']),
                                   max_ns=float(k['MaxNs']), pct_of_gpu_time=float(k['Percentage']))
    # per-dispatch trace: the timed launches are the LAST ceil(steps / chunk) dispatches of the kernel
    kt = glob.glob(os.path.join(a.stats, '**', '*_kernel_trace.csv'), recursive=True)
    if kt and a.bench:
        line = [l for l in open(a.bench).read().splitlines() if l.startswith('{')][-1]
        b = json.loads(line)
        # round 2 on: one bench step IS one launch; round-1 lines counted batched env steps
        n_timed = b['steps'] if 'batched_env_steps_per_launch' in b['config'] else \
            max(1, round(b['steps'] * b['config']['global_batch'] / b['n_gpus'] / b['roofline']['env_steps_per_launch']))
        durs = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']) - int(r['Start_Timestamp']))
                      for r in csv.DictReader(open(kt[0])) if a.kernel in r['Kernel_Name'])
        timed = [d for _, d in durs][-n_timed:]
        summ['timed_launches'] = len(timed)
        summ['timed_avg_ns'] = sum(timed) / len(timed)
    if a.fetch and a.write:
        fk, nf = pmc_mean(a.fetch, 'FETCH_SIZE', a.kernel)
        wk, nw = pmc_mean(a.write, 'WRITE_SIZE', a.kernel)
        summ.update(fetch_size_kib_per_launch=fk, write_size_kib_per_launch=wk, pmc_launches=[nf, nw],
                    traffic_bytes_per_launch=(fk + wk) * 1024.0,
                    pmc_note='separate --pmc passes; WRITE_SIZE exact for 16-B/lane stores; FETCH_SIZE as '
                             'measured (4-B/lane loads: width uncalibrated on gfx950, no 2x correction applied)')
    if a.bench:
        line = [l for l in open(a.bench).read().splitlines() if l.startswith('{')][-1]
        b = json.loads(line)
        summ['bench'] = b
        summ['hip_event_launch_ms'] = b['roofline']['launch_ms']
        summ['rocprof_vs_hip_event'] = summ.get('timed_avg_ns', summ['avg_ns']) * 1e-6 / b['roofline']['launch_ms']
    if a.bench:
        # what bench.py looks up: the workload this profile belongs to, per-ENV-STEP traffic, the issue-side figures
        cfg = b['config']
        per_launch = b['roofline']['env_steps_per_launch']
        summ['workload'] = dict(scenario='simple_spread' if 'simple_spread' in cfg['workload'] else 'simple_tag',
                                N=int(cfg['workload'].split(' N=')[1].split(' ')[0]),
                                B=int(cfg['global_batch'] // b['n_gpus']), steps_per_launch=cfg['batched_env_steps_per_launch'])
        if 'traffic_bytes_per_launch' in summ:
            # PMC passes ran the same launch size (bench --steps 4 --warmup 1: same --chunk)
            summ['traffic_bytes_per_env_step'] = summ['traffic_bytes_per_launch'] / per_launch
    if a.sq:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from pmc_table import table
        t = {c: sum(v) / len(v) for (k, c), v in table(a.sq, a.kernel).items()}
        summ['sq_counters_per_launch'] = t
        if a.bench and all(k in t for k in ('SQ_INSTS_VALU', 'SQ_WAVES', 'SQ_WAVE_CYCLES')):
            wave_steps = t['SQ_WAVES'] * summ['workload']['steps_per_launch']
            summ['valu_insts_per_wave_step'] = t['SQ_INSTS_VALU'] / wave_steps
            summ['clocks_per_wave_step'] = t['SQ_WAVE_CYCLES'] * 4.0 / wave_steps      # SQ cycle counters tick once per 4 clocks
            # a wave64 VALU instruction occupies its SIMD's issue for 4 clocks
            summ['valu_issue_share'] = 4.0 * summ['valu_insts_per_wave_step'] / summ['clocks_per_wave_step']
            summ['sq_source'] = 'profiles/%s_summary.json (rocprofv3 --pmc SQ passes, mean over the launches of this kernel)' % a.tag
    # which kernel code this was measured on: bench.py quotes the counters only while the tree still holds exactly these sources
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from multiagent_rl_amd import build_native
    summ['kernel_family'] = build_native.kernel_family(summ['kernel'])
    summ['kernel_source_sha16'] = build_native.kernel_source_hash(summ['kernel_family'])
    summ['kernel_source_files'] = build_native.KERNEL_FAMILIES[summ['kernel_family']]
    json.dump(summ, open(os.path.join(a.out, '%s_summary.json' % a.tag), 'w'), indent=1)
    print(json.dumps(summ, indent=1)[:1500])


if __name__ == '__main__':
    main()
