cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for b in 32 16 2; do python bench.py --batch $b --no-cpu-baseline --no-kernel-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B=$b', d['value'], 'img/s', d['ms_per_step'], 'ms/step  host', d['config']['host_enqueue_ms_per_step'])"; done
uptime
