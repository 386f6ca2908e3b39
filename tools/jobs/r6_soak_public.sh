#!/bin/bash
# the soak legs the closing session did not repeat, on the final tree: the public methods (estep / mstep / calculate_lnl with caller-supplied
# parameters) and em() to convergence (same stopping iteration as the oracle), at seeds 0 .. 270 as in round 5 and at fresh ones
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r6_soak_public; rm -rf $O; mkdir -p $O
timeout 500 python tests/fuzz_reports.py 0 271 public > $O/public_0_271.log 2>&1; tail -1 $O/public_0_271.log
timeout 500 python tests/fuzz_reports.py 0 271 converge > $O/converge_0_271.log 2>&1; tail -1 $O/converge_0_271.log
timeout 300 python tests/fuzz_reports.py 1000 150 public > $O/public_1000_150.log 2>&1; tail -1 $O/public_1000_150.log
timeout 300 python tests/fuzz_reports.py 1000 150 converge > $O/converge_1000_150.log 2>&1; tail -1 $O/converge_1000_150.log
