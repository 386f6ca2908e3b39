#!/bin/bash
# the GPU suite, smoke, the soak legs (one attempt per case), the report timings, the bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r6_suite; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/gpu_suite.log 2>&1; echo "rc=$?" >> $O/gpu_suite.log; grep -E "passed|failed|rc=" $O/gpu_suite.log | tail -3
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python tests/fuzz_reports.py 0 ${FUZZ_N:-200} > $O/fuzz_one_0.log 2>&1; tail -1 $O/fuzz_one_0.log
timeout 900 python tests/fuzz_reports.py 400 ${FUZZ_N:-200} > $O/fuzz_one_400.log 2>&1; tail -1 $O/fuzz_one_400.log
timeout 900 python tests/fuzz_reports.py 0 ${FUZZ_N:-200} own > $O/fuzz_own_0.log 2>&1; tail -2 $O/fuzz_own_0.log
timeout 600 python tests/fuzz_reports.py 0 60 sharded > $O/fuzz_sharded.log 2>&1; tail -1 $O/fuzz_sharded.log
timeout 600 python tests/fuzz_reports.py 0 60 groups > $O/fuzz_groups.log 2>&1; tail -1 $O/fuzz_groups.log
timeout 600 python tools/time_e2e.py 2>&1 | grep -v "amdgpu\|WARNING" > $O/time_e2e.txt; tail -12 $O/time_e2e.txt
if [ -z "$NO_BENCH" ]; then python bench.py --steps 20 --warmup 5 > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log > $O/bench.json; tail -c 300 $O/bench.json; fi
