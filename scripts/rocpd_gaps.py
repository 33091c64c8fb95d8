#!/usr/bin/env python
"""Idle gaps between consecutive kernels of a rocprofv3 (rocpd sqlite) kernel trace, grouped by the kernel that
follows the gap.  usage: rocpd_gaps.py results.db [skip_first_n]"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute("select %s, start, end from kernels order by start" % name_col).fetchall()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 4
rows = rows[skip:]
gaps, durs = defaultdict(list), defaultdict(list)
for (n0, s0, e0), (n1, s1, e1) in zip(rows[:-1], rows[1:]):
    gaps[n1.split("(")[0][:40]].append(s1 - e0)
    durs[n1.split("(")[0][:40]].append(e1 - s1)
span = rows[-1][2] - rows[0][1]
busy = sum(e - s for _, s, e in rows)
print("kernels %d  span %.1f us  busy %.1f us (%.1f%%)" % (len(rows), span / 1e3, busy / 1e3, 100.0 * busy / span))
print("%-42s %6s %9s %9s %9s" % ("kernel (gap BEFORE it)", "n", "gap_avg", "gap_p50", "dur_avg"))
for k in sorted(gaps, key=lambda k: -sum(gaps[k])):
    g = sorted(gaps[k])
    print("%-42s %6d %9.2f %9.2f %9.2f" % (k, len(g), sum(g) / len(g) / 1e3, g[len(g) // 2] / 1e3, sum(durs[k]) / len(durs[k]) / 1e3))
