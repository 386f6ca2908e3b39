cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/prof_rep
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_rep -- python $GRAFT_REPO_ROOT/tools/time_report.py > /dev/null 2>&1 )
python tools/kernel_table.py gpurun_out/prof_rep k_report > gpurun_out/r5_report_kernels.txt 2>&1
rm -rf gpurun_out/prof_rep
cat gpurun_out/r5_report_kernels.txt
