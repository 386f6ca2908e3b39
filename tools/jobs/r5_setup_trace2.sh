cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== default" > gpurun_out/r5_setup_trace2.txt
TSEM_TRACE=1 timeout 300 python tools/time_setup.py 2>&1 | grep -v "^{" >> gpurun_out/r5_setup_trace2.txt
echo "== HIP_ENABLE_DEFERRED_LOADING=0" >> gpurun_out/r5_setup_trace2.txt
HIP_ENABLE_DEFERRED_LOADING=0 TSEM_TRACE=1 timeout 300 python tools/time_setup.py 2>&1 | grep -v "^{" >> gpurun_out/r5_setup_trace2.txt
echo "== two engines in one process" >> gpurun_out/r5_setup_trace2.txt
TSEM_TRACE=1 timeout 300 python tools/time_setup_twice.py 2>&1 | grep -v "^{" >> gpurun_out/r5_setup_trace2.txt
cat gpurun_out/r5_setup_trace2.txt
