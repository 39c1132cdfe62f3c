#!/usr/bin/env python
"""Dependent load -> s_waitcnt vmcnt(0) chains per kernel of a disassembled object (tools/isa_extract.sh): the number of
vmcnt(0) waits that directly follow (within 12 instructions) a global / buffer load.  A kernel that shows many of them in its
prologue loads its parameters one memory round trip after the other (c3d_block_out_bwd: eight, before round 5); one that shows
them inside its tile loop has lost its prefetch (c3d_pw_gemm, before round 5).
usage: isa_chains.py <dir>/k.s [min_count]"""
import re
import subprocess
import sys


def main():
    lines = open(sys.argv[1]).read().split("\n")
    thr = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    cur, res, hist = None, {}, []
    for l in lines:
        m = re.match(r"^[0-9a-f]+ <(.*)>:", l)
        if m:
            cur = m.group(1); res[cur] = [0, 0]; hist = []
            continue
        if cur is None: continue
        t = l.strip().split("//")[0].strip()
        if not t: continue
        if re.match(r"(global_load|buffer_load)", t):
            res[cur][1] += 1; hist.append("L")
        elif "vmcnt(0)" in t:
            if "L" in hist[-12:]: res[cur][0] += 1
            hist.append("W")
        else:
            hist.append(".")
    names = subprocess.run(["c++filt"], input="\n".join(res.keys()), capture_output=True, text=True).stdout.split("\n")
    for n, (k, v) in sorted(zip(names, res.items()), key=lambda kv: -kv[1][1][0]):
        if v[0] >= thr:
            print("%4d chained waits / %4d loads  %s" % (v[0], v[1], n.replace("(anonymous namespace)::", "")[:120]))


if __name__ == "__main__":
    main()
