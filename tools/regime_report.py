#!/usr/bin/env python3
"""gpurun_out/prof_r2 (tools/collect_profiles.sh) -> profiles/r2_regime_counters.txt: the counter tables of the two
large regimes with the derived per-env-step traffic and chip-wide VALU issue share.
Usage: python tools/regime_report.py gpurun_out/prof_r2 > profiles/r2_regime_counters.txt"""
import csv
import glob
import os
import re
import sys

ALG = {6: 678, 48: 21552}


def main():
    O = sys.argv[1]
    print("# Round 2: counters of the two regimes VERDICT r1 asked to back with numbers (MI355X, rocprofv3, separate --pmc passes:\n"
          "# FETCH_SIZE | WRITE_SIZE | SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES | SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU;\n"
          "# kernel trace in its own run).  Workload: tools/sweep.py, 100-step pw_rollout launches, all outputs written; kernel =\n"
          "# the two-wave duo form with block-wise observation stores and the revision-3 math contract (final state of this round).\n"
          "# tools/collect_profiles.sh + tools/regime_report.py reproduce it.  Stamps per kernel segment and before/after\n"
          "# throughput of the block-wise stores: r2_regimes_before.txt / r2_regimes_after.txt (revision-2 math).\n")
    for tag, B, N in (('B65536_N6', 65536, 6), ('B4096_N48', 4096, 48)):
        txt = open(os.path.join(O, tag + '_counters.txt')).read().splitlines()
        sweep = [l for l in txt if l.startswith('B=')][0]
        rows = [l for l in txt if l.startswith('pw_spread')]
        c = {}
        for l in rows:
            m = re.match(r'(\S.*?)\s{2,}(\S+)\s+mean (\S+) over (\d+)', l)
            c[m.group(2)] = float(m.group(3))
        ks = glob.glob(os.path.join(O, tag + '_trace', '**', '*_kernel_stats.csv'), recursive=True)[0]
        k = [r for r in csv.DictReader(open(ks)) if 'pw_spread' in r['Name']][0]
        avg_us, calls = float(k['AverageNs']) / 1e3, int(k['Calls'])
        T = 100
        per = B * T
        f, w = c['FETCH_SIZE'] * 1024 / per, c['WRITE_SIZE'] * 1024 / per
        ws = c['SQ_WAVES'] * T
        print('== simple_spread N=%d, B=%d  (sweep under the profiler: %s)' % (N, B, sweep))
        print('\n'.join(rows))
        print('kernel trace: avg %.1f us per 100-step launch (%d launches)' % (avg_us, calls))
        print('derived:')
        print('  HBM traffic / env-step   FETCH %.1f B + WRITE %.1f B = %.1f B  (algorithmic %d B)' % (f, w, f + w, ALG[N]))
        print('  achieved                 %.0f GB/s by algorithmic bytes = %.1f %% of 8 TB/s; %.0f GB/s by counter traffic = %.1f %%'
              % (ALG[N] * per / avg_us / 1e3, ALG[N] * per / avg_us / 1e3 / 80, (f + w) * per / avg_us / 1e3, (f + w) * per / avg_us / 1e3 / 80))
        print('  per wave and step        %.0f VALU, %.1f LDS, %.0f scalar instructions in %.0f wave-resident clocks'
              % (c['SQ_INSTS_VALU'] / ws, c['SQ_INSTS_LDS'] / ws, c['SQ_INSTS_SALU'] / ws, c['SQ_WAVE_CYCLES'] * 4 / ws))
        clk = c['SQ_INSTS_VALU'] * 4 / 1024
        print('  VALU issue, whole chip   %.3g wave-instructions x 4 clocks / 1024 SIMDs = %.3g clocks per SIMD = %.0f %% of the kernel\'s %.0f us at 2.4 GHz'
              % (c['SQ_INSTS_VALU'], clk, 100 * clk / (avg_us * 2400), avg_us))
        print()
    print("=> both regimes sit at roughly two thirds VALU issue (at 2.4 GHz; higher at the clock a loaded chip holds) AND\n"
          "   ~3.7-4 TB/s of real (almost all write) HBM traffic: no single resource is saturated; the write path no longer\n"
          "   dominates (before the block-wise stores they were 70-77 % of a wave's step, r2_regimes_before.txt).  What remains on\n"
          "   the VALU side is the divergent near-pair loop (a wave runs max-over-lanes iterations of the contact chain) and, at\n"
          "   N = 48, the two O(N) partner passes.")


if __name__ == '__main__':
    main()
