cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round5.py tests/test_resume_cli.py tests/test_sparse_plus.py -x -q -m gpu > gpurun_out/quick_tests.log 2>&1; echo "rc=$?" >> gpurun_out/quick_tests.log
tail -5 gpurun_out/quick_tests.log
