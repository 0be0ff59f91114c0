#!/usr/bin/env python3
"""Timeline of a rocprofv3 rocpd database: every kernel dispatch in start order with its duration and the idle
gap before it (per process).  python tools/rocpd_timeline.py <results.db> [first] [count]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = list(c.execute("select d.start, d.end, s.kernel_name, d.queue_id, d.stream_id from %s d join %s s on d.kernel_id=s.id "
                      "order by d.start" % (kd, ks)))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 60
if first < 0:
    first = max(0, len(rows) + first)
prev_end = None
for i, (st, en, name, q, sid) in enumerate(rows):
    if i >= first and i < first + count:
        gap = 0 if prev_end is None else (st - prev_end) / 1e3
        print('%5d  +%9.2f us  dur %9.2f us  q%-3s s%-3s %s' % (i, gap, (en - st) / 1e3, q, sid, name[:70]))
    prev_end = en if prev_end is None else max(prev_end, en)
