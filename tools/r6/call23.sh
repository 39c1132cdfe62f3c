#!/bin/bash
# per-kernel times with conv_a forward on the first kernel (PW_CFWD=1) and on the cooperative kernel (3): serial kernel trace
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in 1 3; do
  rm -rf gpurun_out/kt_cfwd$v
  C3D_WGRAD_SIDE=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_cfwd$v -- \
    python bench.py --no-cpu-baseline --no-also --no-kernel-profile --steps 20 --warmup 3 --option PW_CFWD=$v > gpurun_out/kt_cfwd$v.log 2>&1
  f=$(ls -t gpurun_out/kt_cfwd$v/*/*kernel_stats.csv | head -1)
  echo "== PW_CFWD=$v"; python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if "pw_cfwd" in n or ("pw_gemm_kernel" in n or "pw_kernel" in n) :
        print(f'{float(r["TotalDurationNs"])/1e6/23:8.3f} ms/step  calls/step {int(r["Calls"])/23:6.1f}  avg {float(r["AverageNs"])/1e3:7.1f} us  {n[:150]}')
PY
done
