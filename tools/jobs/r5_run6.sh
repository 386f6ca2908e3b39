cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_round5.py -x -q -m gpu > gpurun_out/r5_t6.log 2>&1; echo "rc=$?" >> gpurun_out/r5_t6.log
tail -15 gpurun_out/r5_t6.log
for dbg in 0 16384 8192; do echo "== codes 40/row fused_dbg=$dbg"; timeout 300 python tools/time_lnl.py value_format=2 fused_dbg=$dbg 2>&1 | tail -3; done > gpurun_out/r5_time_lnl_lin.txt 2>&1
cat gpurun_out/r5_time_lnl_lin.txt
