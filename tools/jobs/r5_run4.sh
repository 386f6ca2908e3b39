cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
TSEM_TRACE=1 timeout 300 python tools/time_setup.py 2>&1 | grep -v "^{" > gpurun_out/r5_setup_trace5.txt
python tools/time_setup.py 2>&1 | grep -v "^{" >> gpurun_out/r5_setup_trace5.txt
cat gpurun_out/r5_setup_trace5.txt | grep -v amdgpu
