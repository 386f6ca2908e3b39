cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round2.py tests/test_resume_cli.py -x -q -m gpu > gpurun_out/quick_tests.log 2>&1; echo "rc=$?" >> gpurun_out/quick_tests.log
grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" gpurun_out/quick_tests.log | tail -6
python tools/time_e2e.py 2>&1 | grep -v "amdgpu\|WARNING" > gpurun_out/time_e2e_last.txt; cat gpurun_out/time_e2e_last.txt
