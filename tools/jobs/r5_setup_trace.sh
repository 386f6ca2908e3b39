cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
TSEM_TRACE=1 timeout 300 python tools/time_setup.py > gpurun_out/r5_setup_trace.txt 2>&1
TSEM_TRACE=1 timeout 300 python tools/time_setup.py value_format=1 >> gpurun_out/r5_setup_trace.txt 2>&1
rm -rf gpurun_out/prof_setup
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_setup -- python $GRAFT_REPO_ROOT/tools/time_setup.py > /dev/null 2>&1 )
python tools/kernel_table.py gpurun_out/prof_setup > gpurun_out/r5_setup_kernels.txt 2>&1
rm -rf gpurun_out/prof_setup
cat gpurun_out/r5_setup_trace.txt; cat gpurun_out/r5_setup_kernels.txt | head -40
