cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py -x -q -k "dw333 or bn_b_finalize or folded_bn" 2>&1 | tail -2
python tools/pw_phase_clock.py --fb 2>&1 | head -32
for i in 1 2; do python bench.py --no-cpu-baseline --no-kernel-profile --no-also --steps 30 --warmup 10 | cut -c1-160; done
python bench.py --no-cpu-baseline --no-also --steps 5 --warmup 2 --kernel-table gpurun_out/ktab.json > /dev/null 2>gpurun_out/ktab.err
