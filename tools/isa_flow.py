#!/usr/bin/env python
"""Memory-operation skeleton of one kernel's ISA: every global / buffer / scratch load and store, every s_waitcnt that
names vmcnt, barriers, branches (with the line they go to) and the MFMA batches between them, in file order.  This is
how round 5 found that the pointwise GEMM's tile loop waited vmcnt(0) for every prefetch slot it had just re-issued:
the counted waits are in the object code, not in any profile.
usage: tools/isa_extract.sh <obj> <dir>;  isa_flow.py <dir>/k.s "<demangled kernel name substring>" [first_line last_line]"""
import re
import subprocess
import sys


def kernel_lines(path, want):
    lines = open(path).read().split("\n")
    heads = [(i, l) for i, l in enumerate(lines) if re.match(r"^[0-9a-f]+ <.*>:", l)]
    names = subprocess.run(["c++filt"], input="\n".join(l for _, l in heads), capture_output=True, text=True).stdout.split("\n")
    for k, ((i, _), n) in enumerate(zip(heads, names)):
        if want in n.replace("(anonymous namespace)::", ""):
            end = heads[k + 1][0] if k + 1 < len(heads) else len(lines)
            return lines[i:end]
    sys.exit(f"no kernel matching {want!r}")


def main():
    lines = kernel_lines(sys.argv[1], sys.argv[2])
    lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    hi = int(sys.argv[4]) if len(sys.argv) > 4 else 10 ** 9
    addr2line = {}
    for i, l in enumerate(lines):
        m = re.search(r"//\s*([0-9A-Fa-f]{8,12}):", l)
        if m: addr2line[int(m.group(1), 16)] = i + 1
    mf = 0
    for i, l in enumerate(lines):
        n = i + 1
        if n < lo or n > hi: continue
        m = re.search(r"//\s*([0-9A-Fa-f]{8,12}):", l)
        a = int(m.group(1), 16) if m else 0
        t = l.strip().split("//")[0].strip()
        if "v_mfma" in t:
            mf += 1
            continue
        key = None
        if re.match(r"(buffer_load|global_load|scratch_load)", t): key = "LOAD  " + " ".join(t.split()[:2])
        elif re.match(r"(buffer_store|global_store|scratch_store|global_atomic|buffer_atomic)", t): key = "STORE " + t.split()[0]
        elif "vmcnt" in t: key = "WAIT  " + t
        elif t.startswith("s_cbranch") or t.startswith("s_branch"):
            off = int(t.split()[1])
            off = off - 65536 if off > 32767 else off
            key = f"{t.split()[0]} -> L{addr2line.get(a + 4 + off * 4, '?')}"
        elif t.startswith("s_barrier"): key = "BARRIER"
        if key:
            if mf:
                print(f"        [{mf} mfma]")
                mf = 0
            print(f"L{n}: {key}")


if __name__ == "__main__":
    main()
