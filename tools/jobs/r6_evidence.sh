#!/bin/bash
# Round 6, first evidence of the session: the GPU suite, smoke, the bench line, then the near-tie soak legs (one attempt per case)
# and the report-pass timings.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r6_evidence; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/gpu_suite.log 2>&1; echo "rc=$?" >> $O/gpu_suite.log; tail -4 $O/gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py --steps 20 --warmup 5 > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log > $O/bench.json; tail -c 400 $O/bench.json
timeout 900 python tests/fuzz_reports.py 400 200 > $O/fuzz_one_400_200.log 2>&1; tail -1 $O/fuzz_one_400_200.log
timeout 900 python tests/fuzz_reports.py 0 200 > $O/fuzz_one_0_200.log 2>&1; tail -1 $O/fuzz_one_0_200.log
timeout 900 python tests/fuzz_reports.py 0 300 own > $O/fuzz_own_0_300.log 2>&1; tail -2 $O/fuzz_own_0_300.log
timeout 600 python tests/fuzz_reports.py 0 80 sharded > $O/fuzz_sharded.log 2>&1; tail -1 $O/fuzz_sharded.log
timeout 600 python tools/time_report.py 2>&1 | grep -v amdgpu > $O/time_report.txt; cat $O/time_report.txt
timeout 600 python tools/time_e2e.py 2>&1 | grep -v "amdgpu\|WARNING" > $O/time_e2e.txt; tail -15 $O/time_e2e.txt
