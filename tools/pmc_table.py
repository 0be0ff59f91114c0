#!/usr/bin/env python3
"""rocprofv3 --pmc CSV output -> one line per (kernel, counter): mean value per dispatch and the dispatch count.
Usage: python tools/pmc_table.py DIR [DIR ...] [--kernel SUBSTR]"""
import collections
import csv
import glob
import os
import sys


def table(dirs, kernel=None):
    acc = collections.OrderedDict()
    for d in dirs:
        for f in glob.glob(os.path.join(d, '**', '*_counter_collection.csv'), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r['Kernel_Name']
                if kernel and kernel not in k:
                    continue
                short = k.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
                acc.setdefault((short, r['Counter_Name']), []).append(float(r['Counter_Value']))
    return acc


if __name__ == '__main__':
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    kern = None
    if '--kernel' in sys.argv:
        kern = sys.argv[sys.argv.index('--kernel') + 1]
        args = [a for a in args if a != kern]
    for (k, c), v in table(args, kern).items():
        print('%-60s %-22s mean %.6g over %d dispatches' % (k[:60], c, sum(v) / len(v), len(v)))
