#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 rocpd database (the default output format of ROCm 7.2's rocprofv3):
python tools/rocpd_stats.py <results.db> [top_n]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
q = ("select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), sum(d.end-d.start) "
     "from %s d join %s s on d.kernel_id=s.id group by s.kernel_name order by 6 desc limit %d"
     % (kd, ks, int(sys.argv[2]) if len(sys.argv) > 2 else 15))
print('%-80s %7s %10s %10s %10s %12s' % ('kernel', 'calls', 'avg_us', 'min_us', 'max_us', 'total_us'))
for r in c.execute(q):
    print('%-80s %7d %10.2f %10.2f %10.2f %12.1f' % (r[0][:80], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3))
