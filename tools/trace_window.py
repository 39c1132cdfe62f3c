#!/usr/bin/env python
"""Timeline around the largest main-queue gap of the last step: every dispatch (both queues) from `before` us in front of
the gap to `after` us behind it.  usage: trace_window.py <kernel_trace.csv> [before_us after_us [rank]]  (rank 0 = largest gap)"""
import csv, re, sys
from collections import defaultdict

def short(n):
    n = re.sub(r"^void\s+", "", n); n = re.sub(r"\(anonymous namespace\)::", "", n)
    return n.split("(")[0][:70]

path = sys.argv[1]
before, after = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (400.0, 150.0)
rank = int(sys.argv[4]) if len(sys.argv) > 4 else 0
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), short(r["Kernel_Name"]), int(r.get("Workgroup_Size_X", 0) or 0),
               int(r.get("Grid_Size_X", 0) or 0)) for r in csv.DictReader(open(path)))
adam = [i for i, r in enumerate(rows) if r[3].startswith("adam")]
a, b = adam[-2], adam[-1]
ks = rows[a + 1:b + 1]
qc = defaultdict(int)
for r in ks: qc[r[2]] += 1
mainq = max(qc, key=qc.get)
gaps, prev = [], rows[a][1]
for r in ks:
    if r[2] != mainq: continue
    gaps.append((r[0] - prev, prev, r[0])); prev = max(prev, r[1])
gaps.sort(reverse=True)
g, g0, g1 = gaps[rank]
print(f"gap {g / 1e3:.1f} us; t = 0 at its start; main queue = {mainq}")
for s, e, q, n, wg, grid in ks:
    if e < g0 - before * 1e3 or s > g1 + after * 1e3: continue
    print(f"{'main' if q == mainq else 'side'}  {(s - g0) / 1e3:9.1f} .. {(e - g0) / 1e3:9.1f}  ({(e - s) / 1e3:7.1f} us)  wgs {grid // max(wg, 1):6d}  {n}")
