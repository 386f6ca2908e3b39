cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
TSEM_TRACE=1 timeout 300 python tools/time_setup.py 2>&1 | grep -v "^{\|amdgpu" > gpurun_out/r5_setup_trace6.txt
python tools/time_setup.py 2>&1 | grep -v "^{\|amdgpu" >> gpurun_out/r5_setup_trace6.txt
cat gpurun_out/r5_setup_trace6.txt
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round5.py tests/test_gpu_round4.py -x -q -m gpu > gpurun_out/r5_t8.log 2>&1; echo "rc=$?" >> gpurun_out/r5_t8.log
tail -6 gpurun_out/r5_t8.log
