#!/bin/bash
# round 5, call 35: full GPU suite, smoke and the default bench line at the final library
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/c35_pytest.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/c35_smoke.txt
python bench.py 2>gpurun_out/final_bench.err | tail -1 > gpurun_out/final_bench.json
python -c "
import json
d=json.loads(open('gpurun_out/final_bench.json').read())
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['step_roofline']['frac'], d['cpu_baseline']['value'], {k:v['value'] for k,v in d['also'].items()})"
