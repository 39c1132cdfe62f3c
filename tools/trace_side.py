#!/usr/bin/env python
"""The side queue (weight gradients) of one traced train step against the main queue: reads the rocprofv3 --kernel-trace CSV
tools/trace_step.sh writes.  Prints (a) for the span in which the side queue has work: its busy share, its kernels by name with
their durations alone-in-trace, the idle time in front of each kind; (b) every main-queue gap over 5 us inside that span with
the side-queue kernel that was running when the gap ended -- i.e. whether the data-gradient chain was waiting for the side queue
(c3d_stage_bwd's lag ring) and for which kernel.
usage: trace_side.py <kernel_trace.csv>"""
import csv
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r"^void\s+", "", n)
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    m = re.match(r"([A-Za-z0-9_:]+)(<[^>]*>)?", n)
    return (m.group(1) + (m.group(2) or "")).replace("unsigned short", "bf16")[:48]


def main():
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), short(r["Kernel_Name"]),
                     int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))))
    rows.sort()
    adam = [i for i, r in enumerate(rows) if r[3].startswith("adam")]
    a, b = adam[-2], adam[-1]
    ks = rows[a + 1:b + 1]
    qc = defaultdict(int)
    for k in ks: qc[k[2]] += 1
    mainq = max(qc, key=qc.get)
    side = [k for k in ks if k[2] != mainq]
    main_ = [k for k in ks if k[2] == mainq]
    if not side:
        print("no side-queue kernels"); return
    t0, t1 = side[0][0], max(k[1] for k in side)
    busy = sum(k[1] - k[0] for k in side)
    print(f"side queue: {len(side)} kernels over a span of {(t1 - t0) / 1e3:.0f} us, busy {busy / 1e3:.0f} us = {100 * busy / (t1 - t0):.0f} %")
    by = defaultdict(lambda: [0, 0.0, 0.0])
    prev_end = t0
    for s, e, q, n, g in side:
        key = f"{n} wgs={g}"
        by[key][0] += 1; by[key][1] += e - s; by[key][2] += max(0, s - prev_end)
        prev_end = max(prev_end, e)
    print(f"{'side-queue kernel':64s} {'n':>4s} {'avg us':>8s} {'idle before, avg us':>20s}")
    for key, (n, d, gp) in sorted(by.items(), key=lambda kv: -kv[1][1]):
        print(f"{key:64s} {n:4d} {d / n / 1e3:8.1f} {gp / n / 1e3:20.1f}")
    # main-queue gaps inside the span
    print("main-queue gaps > 5 us inside that span: gap us | after -> before | side kernel running at the end of the gap (its remaining us)")
    prev = None
    tot = 0.0
    shown = 0
    agg = defaultdict(lambda: [0, 0.0])
    for s, e, q, n, g in main_:
        if prev is not None and s - prev[1] > 5000 and t0 <= s <= t1:
            running = [k for k in side if k[0] < s and k[1] > prev[1]]
            tag = running[-1][3] if running else "-"
            agg[(prev[3], n, tag)][0] += 1; agg[(prev[3], n, tag)][1] += s - prev[1]
            tot += s - prev[1]
        if prev is None or e > prev[1]: prev = (s, e, q, n)
    for (pn, nn, tag), (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"  {c:3d} x {d / c / 1e3:6.1f} us   {pn} -> {nn}   [{tag}]")
    print(f"  total {tot / 1e3:.0f} us")
    # main-queue kernel durations inside / outside the span (contention)
    ins, outs = defaultdict(lambda: [0, 0.0]), defaultdict(lambda: [0, 0.0])
    for s, e, q, n, g in main_:
        d = ins if (s >= t0 and e <= t1) else outs
        d[f"{n} wgs={g}"][0] += 1; d[f"{n} wgs={g}"][1] += e - s
    print("main-queue kernels inside the span (beside the side queue): n, avg us")
    for key, (c, d) in sorted(ins.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"  {key:64s} {c:4d} {d / c / 1e3:8.1f}")


if __name__ == "__main__":
    main()
