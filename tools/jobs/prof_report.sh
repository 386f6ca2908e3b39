cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/prof_rep
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_rep -- python $GRAFT_REPO_ROOT/tools/time_report.py > $GRAFT_REPO_ROOT/gpurun_out/time_report_last.txt 2>&1 )
python tools/kernel_table.py gpurun_out/prof_rep k_report > gpurun_out/report_kernels_last.txt 2>&1
rm -rf gpurun_out/prof_rep
cat gpurun_out/report_kernels_last.txt; grep "report_e8\|codes only" gpurun_out/time_report_last.txt
