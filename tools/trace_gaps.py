#!/usr/bin/env python
"""Where is the GPU idle inside a train step?  Reads a rocprofv3 --kernel-trace CSV (Start/End timestamps per dispatch),
takes the steady-state steps (between consecutive c3d Adam launches) and reports, per step: wall time, time with at least
one kernel running, time with kernels of both queues running, and -- per kernel name on the main queue -- the average
launch gap in front of it (previous main-queue kernel end -> its start) and its duration.
usage: trace_gaps.py <kernel_trace.csv> [n_last_steps]"""
import csv
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r"^void\s+", "", n)
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return n.split("<")[0].split("(")[0][:34]


def main():
    path = sys.argv[1]
    nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), short(r["Kernel_Name"])))
    rows.sort()
    adam = [i for i, r in enumerate(rows) if r[3].startswith("adam")]
    if len(adam) < 3:
        print("not enough steps"); return
    steps = list(zip(adam[:-1], adam[1:]))[-nlast:]
    tot = defaultdict(float)
    gap_by, dur_by, cnt_by = defaultdict(float), defaultdict(float), defaultdict(int)
    for a, b in steps:
        ks = rows[a + 1:b + 1]
        t0, t1 = rows[a][1], rows[b][1]
        tot["wall"] += t1 - t0
        # union / overlap via sweep
        ev = []
        for s, e, q, n in ks:
            ev.append((s, 1)); ev.append((e, -1))
        ev.sort()
        depth, last, busy, both = 0, t0, 0, 0
        for t, d in ev:
            if depth >= 1: busy += t - last
            if depth >= 2: both += t - last
            depth += d; last = t
        tot["busy"] += busy; tot["both"] += both
        qcount = defaultdict(int)
        for s, e, q, n in ks: qcount[q] += 1
        mainq = max(qcount, key=qcount.get)
        prev_end = t0
        for s, e, q, n in ks:
            if q != mainq:
                tot["side_kernel_time"] += e - s
                continue
            tot["main_kernel_time"] += e - s
            gap_by[n] += max(0, s - prev_end); dur_by[n] += e - s; cnt_by[n] += 1
            prev_end = max(prev_end, e)
    # the largest individual main-queue gaps of the last step, with the kernels on either side and whether the other queue
    # was busy meanwhile
    a, b = steps[-1]
    ks = rows[a + 1:b + 1]
    qcount = defaultdict(int)
    for s_, e_, q_, n_ in ks: qcount[q_] += 1
    mainq = max(qcount, key=qcount.get)
    side = [(s_, e_) for s_, e_, q_, n_ in ks if q_ != mainq]
    gaps, prev = [], (rows[a][1], "adam(prev step)")
    for s_, e_, q_, n_ in ks:
        if q_ != mainq: continue
        if s_ > prev[0]:
            ov = sum(max(0, min(e2, s_) - max(s2, prev[0])) for s2, e2 in side)
            gaps.append((s_ - prev[0], prev[1], n_, ov))
        if e_ > prev[0]: prev = (e_, n_)
    gaps.sort(reverse=True)
    print("largest main-queue gaps of the last step: gap us | side queue busy us | after -> before")
    for gp, pn, nn, ov in gaps[:14]:
        print(f"  {gp/1e3:8.1f} {ov/1e3:8.1f}   {pn} -> {nn}")
    print(f"  (sum of all {len(gaps)} gaps {sum(g_[0] for g_ in gaps)/1e6:.3f} ms, of which side queue busy {sum(g_[3] for g_ in gaps)/1e6:.3f} ms)")
    n = len(steps)
    print(f"{n} steps: wall {tot['wall']/n/1e6:.3f} ms  busy(any kernel) {tot['busy']/n/1e6:.3f}  idle {(tot['wall']-tot['busy'])/n/1e6:.3f}  "
          f"two queues at once {tot['both']/n/1e6:.3f}  main-queue kernel time {tot['main_kernel_time']/n/1e6:.3f}  "
          f"side-queue kernel time {tot['side_kernel_time']/n/1e6:.3f}")
    print(f"{'main-queue kernel':36s} {'n/step':>7s} {'avg us':>8s} {'gap us':>8s} {'dur ms/step':>12s} {'gap ms/step':>12s}")
    for k in sorted(dur_by, key=lambda k: -(dur_by[k] + gap_by[k])):
        c = cnt_by[k]
        print(f"{k:36s} {c/n:7.1f} {dur_by[k]/c/1e3:8.2f} {gap_by[k]/c/1e3:8.2f} {dur_by[k]/n/1e6:12.3f} {gap_by[k]/n/1e6:12.3f}")


if __name__ == "__main__":
    main()
