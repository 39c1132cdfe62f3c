cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py -x -q -k "pw or wide" 2>&1 | tail -2
python -m pytest tests/test_cc_gpu.py -x -q 2>&1 | tail -2
python bench.py --task cc --no-cpu-baseline --no-also --steps 20 --warmup 10 --kernel-table gpurun_out/ktab_cc.json 2>/dev/null | cut -c1-150
