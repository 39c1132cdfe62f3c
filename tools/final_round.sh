set -u
cd "${GRAFT_REPO_ROOT:-.}"
bash tools/profile_round.sh > gpurun_out/profile_round.log 2>&1
python bench.py --task scd --no-cpu-baseline > gpurun_out/bench_scd.json 2> gpurun_out/bench_scd.err
python tools/infer_bench.py > gpurun_out/infer.json 2> gpurun_out/infer.err
python bench.py --no-cpu-baseline --kernel-table gpurun_out/kernels_final.json > gpurun_out/bench_kt.json 2> gpurun_out/bench_kt.err
python tools/pw_phase_clock.py > gpurun_out/pw_phase_clock_gemm.txt 2>&1
python tools/pw_phase_clock.py --wgrad > gpurun_out/pw_phase_clock_wgrad.txt 2>&1
tail -5 gpurun_out/profile_round.log; tail -c 600 gpurun_out/bench_scd.json; tail -c 400 gpurun_out/infer.json
