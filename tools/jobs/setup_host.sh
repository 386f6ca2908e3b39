cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
TSEM_TRACE=1 timeout 600 python tools/time_setup_host.py 2>&1 | grep -v amdgpu > gpurun_out/setup_host_last.txt
echo "---- device-generated, same size" >> gpurun_out/setup_host_last.txt
TSEM_TRACE=1 timeout 600 python tools/time_setup.py 2000000 2>&1 | grep -v "amdgpu\|^{" >> gpurun_out/setup_host_last.txt
cat gpurun_out/setup_host_last.txt
