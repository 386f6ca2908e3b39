cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
TSEM_TRACE=1 timeout 300 python tools/time_setup.py 2>&1 | grep -v "^{" > gpurun_out/r5_setup_trace3.txt
python tools/time_setup.py 2>&1 | grep -v "^{" >> gpurun_out/r5_setup_trace3.txt
cat gpurun_out/r5_setup_trace3.txt
timeout 2400 python -m pytest tests/ -x -q -m gpu > gpurun_out/r5_gpu_full.log 2>&1; echo "rc=$?" >> gpurun_out/r5_gpu_full.log
tail -8 gpurun_out/r5_gpu_full.log
