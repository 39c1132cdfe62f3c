cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; rm -rf gpurun_out/fx3
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fx3 -- python tools/pw_fixed_cost.py > gpurun_out/fx3.log 2>&1
python - <<'PY'
import csv, glob, collections
f = sorted(glob.glob('gpurun_out/fx3/**/*kernel_trace.csv', recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
agg = collections.OrderedDict()
for r in rows:
    if 'pw_gemm' not in r['Kernel_Name']: continue
    key = (r['Kernel_Name'][40:100], r['Grid_Size_X'], r['Workgroup_Size_X'])
    agg.setdefault(key, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in agg.items():
    v = sorted(v)
    print(k, 'n', len(v), 'median us', round(v[len(v)//2], 2), 'min', round(v[0], 2))
PY
