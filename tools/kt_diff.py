#!/usr/bin/env python3
"""Per-shape comparison of two bench kernel tables: tools/kt_diff.py gpurun_out/a_kernels.json gpurun_out/b_kernels.json [min_ms]"""
import json, sys
a = {r['kernel']: r for r in json.load(open(sys.argv[1]))['pointwise_by_shape']}
b = {r['kernel']: r for r in json.load(open(sys.argv[2]))['pointwise_by_shape']}
lim = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1
rows = []
for k in a:
    if k in b and max(a[k]['ms_total'], b[k]['ms_total']) >= lim:
        rows.append((b[k]['ms_total'] - a[k]['ms_total'], k, a[k]['ms_total'], b[k]['ms_total'], a[k]['launches']))
for d, k, x, y, n in sorted(rows):
    print(f"{k:62s} x{n:3d} {x:7.3f} -> {y:7.3f} ms ({d:+.3f})")
print("sum", round(sum(r['ms_total'] for r in a.values()), 3), "->", round(sum(r['ms_total'] for r in b.values()), 3))
