cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
python tools/time_choose.py 2>&1 | grep -v "amdgpu\|WARNING" > gpurun_out/time_choose_last.txt; cat gpurun_out/time_choose_last.txt
timeout 900 python -m pytest tests/test_gpu_round5.py -x -q -m gpu -k "initial_report" 2>&1 | tail -2
