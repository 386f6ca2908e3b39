cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
TSEM_TRACE=1 timeout 300 python tools/time_setup.py 2>&1 | grep -v "^{" > gpurun_out/r5_setup_trace4.txt
python tools/time_setup.py 2>&1 | grep -v "^{" >> gpurun_out/r5_setup_trace4.txt
python tools/time_setup.py 2>&1 | grep -v "^{" >> gpurun_out/r5_setup_trace4.txt
cat gpurun_out/r5_setup_trace4.txt | grep -v amdgpu
timeout 1800 python -m pytest tests/test_gpu_round5.py -x -q -m gpu > gpurun_out/r5_t3.log 2>&1; echo "rc=$?" >> gpurun_out/r5_t3.log
tail -12 gpurun_out/r5_t3.log
