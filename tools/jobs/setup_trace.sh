cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
TSEM_TRACE=1 timeout 300 python tools/time_setup.py 2>&1 | grep -v "^{\|amdgpu" > gpurun_out/setup_trace_last.txt
python tools/time_setup.py 2>&1 | grep -v "^{\|amdgpu" >> gpurun_out/setup_trace_last.txt
python tools/time_setup.py 2>&1 | grep -v "^{\|amdgpu" >> gpurun_out/setup_trace_last.txt
cat gpurun_out/setup_trace_last.txt
