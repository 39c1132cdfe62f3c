#!/bin/bash
# per-kernel times with the cooperative conv_a data + weight gradient off / on: serial kernel trace
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_pw_wg_gpu.py -x -q -m gpu 2>&1 | tail -3
for v in 0 1; do
  rm -rf gpurun_out/kt_cdg$v
  C3D_WGRAD_SIDE=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_cdg$v -- \
    python bench.py --no-cpu-baseline --no-also --no-kernel-profile --steps 20 --warmup 3 --option PW_CDG=$v > gpurun_out/kt_cdg$v.log 2>&1
  f=$(ls -t gpurun_out/kt_cdg$v/*/*kernel_stats.csv | head -1)
  echo "== PW_CDG=$v"; python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
n1 = [int(r["Calls"]) for r in rows if "adam" in r["Name"]][0]
print(f"total kernel time {tot/1e6/n1:.3f} ms/step ({n1} steps)")
for r in rows:
    n = r["Name"]
    if "pw_cdg" in n or "pw_wgrad" in n or ("pw_gemm_kernel" in n and ", 2, 3, 8" in n):
        print(f'{float(r["TotalDurationNs"])/1e6/n1:8.3f} ms/step  calls/step {int(r["Calls"])/n1:6.1f}  avg {float(r["AverageNs"])/1e3:7.1f} us  {n[:140]}')
PY
done
