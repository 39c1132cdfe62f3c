#!/bin/bash
# serial kernel trace at the current library: top kernels per step
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/kt_now
C3D_WGRAD_SIDE=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_now -- \
  python bench.py --no-cpu-baseline --no-also --no-kernel-profile --steps 20 --warmup 3 "$@" > gpurun_out/kt_now.log 2>&1
f=$(ls -t gpurun_out/kt_now/*/*kernel_stats.csv | head -1)
python - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
n1 = [int(r["Calls"]) for r in rows if "adam" in r["Name"]][0]
print(f"total kernel time {tot/1e6/n1:.3f} ms/step ({n1} steps)")
for r in rows[:42]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    print(f'{float(r["TotalDurationNs"])/1e6/n1:8.3f} ms/step  calls/step {int(r["Calls"])/n1:6.1f}  avg {float(r["AverageNs"])/1e3:7.1f} us  {n[:120]}')
PY
