cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for nz in 18 10; do for dbg in 0 8192; do echo "== codes $nz/row fused_dbg=$dbg"; timeout 300 python tools/time_lnl.py nnz_row=$nz value_format=2 fused_dbg=$dbg 2>&1 | tail -3; done; done > gpurun_out/r5_time_lnl_short.txt 2>&1
cat gpurun_out/r5_time_lnl_short.txt
