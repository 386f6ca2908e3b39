#!/bin/bash
# VERDICT r5 #1 "Done": python tests/fuzz_reports.py 0 2300 (seed 502 included) with ONE attempt per case -> 0 failures with the same parameters;
# the end-to-end leg (the oracle's own parameters) over 1000 seeds -> the flip count.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r6_soak_full; rm -rf $O; mkdir -p $O
timeout 3300 python tests/fuzz_reports.py 0 2300 > $O/fuzz_one_0_2300.log 2>&1; echo "rc=$?" >> $O/fuzz_one_0_2300.log; tail -2 $O/fuzz_one_0_2300.log
grep -a "^502 " $O/fuzz_one_0_2300.log | head -2
timeout 2400 python tests/fuzz_reports.py 0 1000 own > $O/fuzz_own_0_1000.log 2>&1; tail -2 $O/fuzz_own_0_1000.log
