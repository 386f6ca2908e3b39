cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
python tools/time_lnl.py 50000000 value_format=1 2>&1 | grep -v amdgpu | tail -3
timeout 1200 python tests/fuzz_reports.py 0 150 sharded > gpurun_out/fuzz_sharded.log 2>&1; grep -c "^[0-9]* ok" gpurun_out/fuzz_sharded.log; grep "FAILED\|failures\|Error\|Traceback" gpurun_out/fuzz_sharded.log | head -5
timeout 900 python tests/fuzz_reports.py 0 400 > gpurun_out/fuzz_reports.log 2>&1; grep -c "^[0-9]* ok" gpurun_out/fuzz_reports.log; grep "FAILED\|failures\|Error\|Traceback" gpurun_out/fuzz_reports.log | head -5
bash tools/jobs/gpu_suite_and_bench.sh 2>&1 | grep "passed\|failed\|smoke\|rc="
