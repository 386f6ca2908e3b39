#!/bin/bash
# the soak of the report passes against the oracle at seeds no earlier run of the round has seen (2300 .. 3399), ONE attempt per case, on the
# final tree (padding behind indices / raw / rid16 zeroed), + the bench line against the restamped profiles/pmc_traffic.json
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r6_soak_fresh; rm -rf $O; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log > $O/bench.json; tail -c 200 $O/bench.json; echo
timeout 1500 python tests/fuzz_reports.py 2300 1100 > $O/fuzz_one_2300_1100.log 2>&1; echo "rc=$?" >> $O/fuzz_one_2300_1100.log; tail -2 $O/fuzz_one_2300_1100.log
timeout 300 python tests/fuzz_reports.py 100 60 sharded > $O/fuzz_sharded_100.log 2>&1; tail -1 $O/fuzz_sharded_100.log
timeout 300 python tests/fuzz_reports.py 100 60 groups > $O/fuzz_groups_100.log 2>&1; tail -1 $O/fuzz_groups_100.log
timeout 300 python tests/fuzz_reports.py 100 60 lookups > $O/fuzz_lookups_100.log 2>&1; tail -1 $O/fuzz_lookups_100.log
