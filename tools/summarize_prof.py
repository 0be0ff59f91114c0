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
        n_timed = max(1, round(b['steps'] * b['config']['global_batch'] / b['n_gpus'] / b['roofline']['env_steps_per_launch']))
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
    json.dump(summ, open(os.path.join(a.out, '%s_summary.json' % a.tag), 'w'), indent=1)
    print(json.dumps(summ, indent=1)[:1500])


if __name__ == '__main__':
    main()
