#!/bin/bash
# after the fix of the idle lanes' entry loads: the rest of the 0 .. 2300 soak (the first 1128 cases ran before the fault), the round-6 tests,
# the profile stages of the bench (kernel trace + stats, FETCH / WRITE) for the final sources' stamp, the bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r6_soak_rest; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round6.py -x -q > $O/tests_round6.log 2>&1; tail -2 $O/tests_round6.log
timeout 3000 python tests/fuzz_reports.py 1100 1200 > $O/fuzz_one_1100_1200.log 2>&1; echo "rc=$?" >> $O/fuzz_one_1100_1200.log; tail -2 $O/fuzz_one_1100_1200.log
rm -rf gpurun_out/prof/r06; STAGES="trace pmc" bash tools/profile.sh r06 > /dev/null 2>&1
grep -a "^{\"metric\"" gpurun_out/prof/r06/trace.log | tail -1 | cut -c1-160
python bench.py --steps 20 --warmup 5 > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log > $O/bench.json; tail -c 200 $O/bench.json
find gpurun_out/prof/r06 -name "*.db" -delete
