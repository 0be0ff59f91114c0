#!/usr/bin/env python3
"""rocprofv3 CSV output of tools/collect_profiles_r4.sh -> profiles/r4_policy_<tag>_summary.json: kernel-trace average of the
timed dispatches of the policy rollout kernel, its HIP-event figure from the same command, the f32-MFMA roofline (algorithmic
flops of the actor / launch time / 157.3 TF), and the counters per launch: MFMA ops and busy cycles, VALU / LDS instruction and
wait shares, LDS bank conflicts, HBM traffic (FETCH_SIZE + WRITE_SIZE, separate passes, KiB; MI355X guide)."""
import argparse
import collections
import csv
import glob
import json
import os
import shutil


def counters(d, kernel):
    """{counter: mean over the dispatches of `kernel`} of every *_counter_collection.csv under d."""
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, '**', '*_counter_collection.csv'), recursive=True):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if kernel in r['Kernel_Name']:
                per[(r['Dispatch_Id'], r['Counter_Name'])] += float(r['Counter_Value'])
        for (_, c), v in per.items():
            acc[c].append(v)
    return {c: sum(v) / len(v) for c, v in acc.items()}, {c: len(v) for c, v in acc.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--tag', required=True)
    ap.add_argument('--dir', required=True)
    ap.add_argument('--kernel', required=True)
    ap.add_argument('--out', default='profiles')
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    line = json.loads([l for l in open(os.path.join(a.dir, 'line.json')).read().splitlines() if l.startswith('{')][-1])
    ks = glob.glob(os.path.join(a.dir, 'trace', '**', '*_kernel_stats.csv'), recursive=True)[0]
    shutil.copy(ks, os.path.join(a.out, '%s_kernel_stats.csv' % a.tag))
    k = [r for r in csv.DictReader(open(ks)) if a.kernel in r['Name']][0]
    summ = collections.OrderedDict(tag=a.tag, kernel=k['Name'], calls=int(k['Calls']), avg_ns=float(k['AverageNs']),
                                   min_ns=float(k['MinNs']), max_ns=float(k['MaxNs']), pct_of_gpu_time=float(k['Percentage']))
    kt = glob.glob(os.path.join(a.dir, 'trace', '**', '*_kernel_trace.csv'), recursive=True)
    if kt:
        durs = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']) - int(r['Start_Timestamp']))
                      for r in csv.DictReader(open(kt[0])) if a.kernel in r['Kernel_Name'])
        timed = [d for _, d in durs][-line['steps']:]
        summ['timed_launches'], summ['timed_avg_ns'] = len(timed), sum(timed) / len(timed)
    summ['bench'] = line
    summ['hip_event_launch_ms'] = line['launch_ms']
    kern_ms = summ.get('timed_avg_ns', summ['avg_ns']) * 1e-6
    summ['rocprof_vs_hip_event'] = kern_ms / line['launch_ms']
    per_launch = line['B'] * line['chunk']
    fl = line['flops_per_env_step'] * per_launch
    summ['workload'] = dict(scenario=line['scenario'], N=line['N'], B=line['B'], steps_per_launch=line['chunk'], policy=True)
    summ['roofline'] = dict(bound='mfma_f32', flops_per_launch=fl, achieved_TFLOPs=fl / (kern_ms * 1e-3) / 1e12, peak_TFLOPs=157.3,
                            frac=fl / (kern_ms * 1e-3) / 1e12 / 157.3, frac_hip_event=line['roofline']['frac'])
    c, n = counters(a.dir, a.kernel)
    summ['counters_per_launch'], summ['counter_launches'] = c, n
    if 'SQ_INSTS_VALU_MFMA_MOPS_F32' in c:
        # MOPS counts 512-flop units?  Keep the raw figure and the ratio to the algorithmic flops: it says how many MFMA
        # operations the kernel ISSUED per algorithmic flop (padding of tiles, the head on 16-row tiles, ...)
        summ['mfma_mops_f32_per_launch'] = c['SQ_INSTS_VALU_MFMA_MOPS_F32']
        summ['algorithmic_flops_per_mfma_mop'] = fl / c['SQ_INSTS_VALU_MFMA_MOPS_F32']
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'SQ_BUSY_CYCLES' in c:
        # SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs; SQ_BUSY_CYCLES counts per SE (guide: unit table).  The
        # robust share is against the kernel's own duration: busy cycles / (SIMDs x duration x clock)
        clock_hz = 2.4e9
        summ['mfma_busy_share'] = c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * kern_ms * 1e-3 * clock_hz)
        summ['mfma_busy_note'] = 'SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel duration x 2.4 GHz nominal clock)'
    if all(x in c for x in ('SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_VALU')):
        w = c['SQ_WAVE_CYCLES']
        summ['wave_time_shares'] = dict(wait_any=c['SQ_WAIT_ANY'] / w, wait_inst_any=c['SQ_WAIT_INST_ANY'] / w,
                                        active_valu=c['SQ_ACTIVE_INST_VALU'] / w,
                                        active_lds=c.get('SQ_ACTIVE_INST_LDS', 0.0) / w)
    if 'SQ_LDS_BANK_CONFLICT' in c and 'SQ_LDS_IDX_ACTIVE' in c and c['SQ_LDS_IDX_ACTIVE']:
        summ['lds_bank_conflict_share'] = c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']
    if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
        summ['traffic_bytes_per_launch'] = (c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024.0
        summ['traffic_bytes_per_env_step'] = summ['traffic_bytes_per_launch'] / per_launch
        summ['hbm_GBps'] = summ['traffic_bytes_per_launch'] / (kern_ms * 1e-3) / 1e9
    # which kernel code this was measured on: bench.py quotes the counters only while the tree still holds exactly these sources
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from multiagent_rl_amd import build_native
    summ['kernel_family'] = build_native.kernel_family(summ['kernel'])
    summ['kernel_source_sha16'] = build_native.kernel_source_hash(summ['kernel_family'])
    summ['kernel_source_files'] = build_native.KERNEL_FAMILIES[summ['kernel_family']]
    json.dump(summ, open(os.path.join(a.out, '%s_summary.json' % a.tag), 'w'), indent=1)
    print(json.dumps(summ, indent=1)[:2500])


if __name__ == '__main__':
    main()
