cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 600 python -m cProfile -o gpurun_out/host.prof bench.py --no-cpu-baseline --no-kernel-profile --steps 40 --warmup 5 > /dev/null 2> gpurun_out/host_prof.err
python - <<'PY' > gpurun_out/host_prof.txt 2>&1
import pstats
p = pstats.Stats('gpurun_out/host.prof')
p.sort_stats('tottime').print_stats(40)
p.sort_stats('cumulative').print_stats('change3d_amd|bench.py', 60)
PY
