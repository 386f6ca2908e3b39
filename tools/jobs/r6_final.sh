#!/bin/bash
# Round 6 closing session on one box: the round's profile of the bench (kernel trace + stats, FETCH / WRITE), the report profile,
# the GPU suite, smoke, the soak legs, the bench line.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r6_final; rm -rf $O; mkdir -p $O
STAGES="trace pmc" timeout 1200 bash tools/profile.sh r06 > $O/profile_sh.log 2>&1
# (tools/profile_summary.py runs afterwards in the dev container, on the CSVs merged back: it writes into profiles/)
bash tools/jobs/r6_report_profile.sh > $O/report_profile.log 2>&1; tail -3 $O/report_profile.log
timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/gpu_suite.log 2>&1; echo "rc=$?" >> $O/gpu_suite.log; grep -E "passed|failed|rc=" $O/gpu_suite.log | tail -3
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python tests/fuzz_reports.py 0 300 > $O/fuzz_one_0_300.log 2>&1; tail -1 $O/fuzz_one_0_300.log
timeout 900 python tests/fuzz_reports.py 400 300 > $O/fuzz_one_400_300.log 2>&1; tail -1 $O/fuzz_one_400_300.log
timeout 900 python tests/fuzz_reports.py 0 300 own > $O/fuzz_own_0_300.log 2>&1; tail -2 $O/fuzz_own_0_300.log
timeout 600 python tests/fuzz_reports.py 0 80 sharded > $O/fuzz_sharded.log 2>&1; tail -1 $O/fuzz_sharded.log
timeout 600 python tests/fuzz_reports.py 0 80 groups > $O/fuzz_groups.log 2>&1; tail -1 $O/fuzz_groups.log
timeout 600 python tests/fuzz_reports.py 0 80 lookups > $O/fuzz_lookups.log 2>&1; tail -1 $O/fuzz_lookups.log
python bench.py --steps 20 --warmup 5 > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log > $O/bench.json; cp $O/bench.json gpurun_out/prof/r06/bench.json; tail -c 300 $O/bench.json
timeout 600 python tools/time_e2e.py 2>&1 | grep -v "amdgpu\|WARNING" > $O/time_e2e.txt
find gpurun_out/prof/r06 -name "*.db" -delete
