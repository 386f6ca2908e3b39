#!/bin/bash
# Round 6, near-ties: the new parity tests, the soak legs with ONE attempt per case (seed 502 included), the end-to-end leg with the
# oracle's own parameters (a count, not an assertion), the report pass's time before / after.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r6_neartie; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round6.py -x -q > $O/tests_round6.log 2>&1; echo "rc=$?" >> $O/tests_round6.log; tail -5 $O/tests_round6.log
timeout 1500 python tests/fuzz_reports.py 400 200 > $O/fuzz_one_400_200.log 2>&1; tail -1 $O/fuzz_one_400_200.log
timeout 900 python tests/fuzz_reports.py 0 200 > $O/fuzz_one_0_200.log 2>&1; tail -1 $O/fuzz_one_0_200.log
timeout 900 python tests/fuzz_reports.py 0 300 own > $O/fuzz_own_0_300.log 2>&1; tail -2 $O/fuzz_own_0_300.log
timeout 600 python tests/fuzz_reports.py 0 120 lookups > $O/fuzz_lookups.log 2>&1; tail -1 $O/fuzz_lookups.log
timeout 600 python tests/fuzz_reports.py 0 120 groups > $O/fuzz_groups.log 2>&1; tail -1 $O/fuzz_groups.log
timeout 600 python tests/fuzz_reports.py 0 80 sharded > $O/fuzz_sharded.log 2>&1; tail -1 $O/fuzz_sharded.log
timeout 600 python tools/time_report.py 2>&1 | grep -v amdgpu > $O/time_report.txt; cat $O/time_report.txt
timeout 600 python tools/time_e2e.py 2>&1 | grep -v "amdgpu\|WARNING" > $O/time_e2e.txt; tail -15 $O/time_e2e.txt
