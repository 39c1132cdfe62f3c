#!/usr/bin/env python
"""Largest main-queue gaps of the last traced train step, each with what the host (HIP runtime API calls), the copy engines
(memory copies) and the other queues (kernels) were doing inside the gap.  Input: a rocprofv3 output directory holding
*kernel_trace.csv, *memory_copy_trace.csv, *hip_api_trace.csv (tools/trace_bubbles.sh).
usage: trace_bubbles.py <dir> [n_gaps]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r"^void\s+", "", n)
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return n.split("<")[0].split("(")[0][:40]


def find(d, pat):
    fs = [f for f in glob.glob(os.path.join(d, "**", pat), recursive=True)]
    return max(fs, key=os.path.getsize) if fs else None


def main():
    d = sys.argv[1]
    ngaps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    kt = find(d, "*kernel_trace.csv")
    rows = []
    for r in csv.DictReader(open(kt)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), short(r["Kernel_Name"])))
    rows.sort()
    adam = [i for i, r in enumerate(rows) if r[3].startswith("adam")]
    a, b = adam[-2], adam[-1]
    ks = rows[a + 1:b + 1]
    qcount = defaultdict(int)
    for s, e, q, n in ks: qcount[q] += 1
    mainq = max(qcount, key=qcount.get)
    api = []
    f = find(d, "*hip_api_trace.csv")
    if f:
        for r in csv.DictReader(open(f)):
            api.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Function", r.get("Name", "?")), r.get("Thread_Id", "")))
    cp = []
    f = find(d, "*memory_copy_trace.csv")
    if f:
        for r in csv.DictReader(open(f)):
            cp.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", r.get("Name", "?")), r.get("Size", "")))
    gaps, prev = [], (rows[a][1], "adam(prev step)")
    for s, e, q, n in ks:
        if q != mainq: continue
        if s > prev[0]: gaps.append((s - prev[0], prev[0], s, prev[1], n))
        if e > prev[0]: prev = (e, n)
    gaps.sort(reverse=True)
    t_step0 = rows[a][1]
    print(f"step wall {(rows[b][1] - t_step0) / 1e6:.3f} ms, {len(ks)} kernels, queues {dict(qcount)}; sum of main-queue gaps {sum(g[0] for g in gaps) / 1e6:.3f} ms")
    for gp, g0, g1, pn, nn in gaps[:ngaps]:
        print(f"\ngap {gp / 1e3:8.1f} us at +{(g0 - t_step0) / 1e6:7.3f} ms   {pn} -> {nn}")
        for s, e, q, n in ks:
            if q != mainq and e > g0 and s < g1:
                print(f"    other queue {q}: {n}  [{(s - g0) / 1e3:+.1f} .. {(e - g0) / 1e3:+.1f} us]")
        for s, e, n, sz in cp:
            if e > g0 - 50000 and s < g1:
                print(f"    copy {n} {sz} B  [{(s - g0) / 1e3:+.1f} .. {(e - g0) / 1e3:+.1f} us]")
        calls = [(s, e, n) for s, e, n, _ in api if e > g0 and s < g1]
        agg = defaultdict(lambda: [0, 0])
        for s, e, n in calls:
            agg[n][0] += 1; agg[n][1] += e - s
        if calls:
            print("    host API inside the gap: " + ", ".join(f"{n} x{c} ({t / 1e3:.0f} us)" for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]))
            longest = sorted(calls, key=lambda c: c[0] - c[1])[:3]
            for s, e, n in longest:
                print(f"      longest: {n}  [{(s - g0) / 1e3:+.1f} .. {(e - g0) / 1e3:+.1f} us]")
        else:
            print("    no host API call overlaps the gap (the host is elsewhere: ahead or blocked)")


if __name__ == "__main__":
    main()
