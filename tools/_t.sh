cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py -x -q -k "wgrad" 2>&1 | tail -3
python tools/pw_phase_clock.py --wgrad 2>&1 | grep -E "us/launch|mfma|convert|barrier" 
for i in 1 2; do python bench.py --no-cpu-baseline --no-kernel-profile --no-also --steps 30 --warmup 10 | cut -c1-160; done
