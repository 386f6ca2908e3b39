set -x
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_round5.py -x -q -m gpu > gpurun_out/r5_t2.log 2>&1; echo "rc=$?" >> gpurun_out/r5_t2.log
tail -5 gpurun_out/r5_t2.log
timeout 600 python tools/time_lnl.py > gpurun_out/r5_time_lnl.txt 2>&1
cat gpurun_out/r5_time_lnl.txt | tail -20
for fmt in 1 2; do for dbg in 0 8192; do echo "== value_format=$fmt fused_dbg=$dbg"; timeout 300 python tools/time_lnl.py value_format=$fmt fused_dbg=$dbg 2>&1 | tail -3; done; done > gpurun_out/r5_time_lnl_ab.txt 2>&1
cat gpurun_out/r5_time_lnl_ab.txt
