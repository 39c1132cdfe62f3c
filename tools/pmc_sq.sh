#!/bin/bash
# SQ wave-state counters per kernel (one eager step, side stream off): tools/pmc_sq.sh <tag>
#   -> gpurun_out/<tag>_sq.txt : per kernel name, fractions of wave cycles (active / parked / issue-stalled), LDS and VALU shares
set -u
cd "${GRAFT_REPO_ROOT:-.}"
root=$(pwd)
export TMPDIR=/tmp
tag=$1; shift
mkdir -p gpurun_out
rm -rf /tmp/pmc_$tag
(cd /tmp && C3D_WGRAD_SIDE=0 timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT \
   --kernel-trace --output-format csv -d /tmp/pmc_$tag -- python $root/bench.py --no-cpu-baseline --no-also --no-graph --no-kernel-profile --steps 1 --warmup 1 "$@" > /tmp/pmc_$tag.log 2>&1)
f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
python - "$f" > gpurun_out/${tag}_sq.txt <<'PY'
import csv, sys, collections, re
rows = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:70]
    rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"])
    if key not in seen: seen.add(key); cnt[k] += 1
out = []
for k, c in rows.items():
    wc = c.get("SQ_WAVE_CYCLES", 0)
    if wc <= 0: continue
    out.append((wc, k, cnt[k], c))
print("%-70s %5s %10s %6s %6s %6s %6s %6s %6s %6s" % ("kernel", "n", "wave_cyc", "act", "park", "istall", "valu", "lds", "w_lds", "bankc"))
for wc, k, n, c in sorted(out, reverse=True)[:40]:
    g = lambda x: c.get(x, 0) / wc
    print("%-70s %5d %10.3g %6.3f %6.3f %6.3f %6.3f %6.3f %6.3f %6.3f" % (k, n, wc, g("SQ_ACTIVE_INST_ANY"), g("SQ_WAIT_ANY"), g("SQ_WAIT_INST_ANY"), g("SQ_ACTIVE_INST_VALU"), g("SQ_ACTIVE_INST_LDS"), g("SQ_WAIT_INST_LDS"), g("SQ_LDS_BANK_CONFLICT")))
PY
head -45 gpurun_out/${tag}_sq.txt
