#!/bin/bash
# Round profile of the bench workload on the GPU box: kernel trace + the two HBM PMC passes + timeline.
# Writes under gpurun_out/prof/<tag>/ ; summarise with tools/profile_summary.py.
#   tools/profile.sh <tag>
set -u
TAG=${1:-r}
OUT=gpurun_out/prof/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-precision-sweep --kernel-timing 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- $B > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- $B > $OUT/write.log 2>&1
python tools/fused_prof.py 20000000 value_format=2 > $OUT/timeline_code16.txt 2>&1
python tools/fused_prof.py 20000000 value_format=1 > $OUT/timeline_f64.txt 2>&1
tools/ubench/lds > $OUT/lds_ubench.log 2>&1
tools/ubench/stream > $OUT/stream_ubench.log 2>&1
tools/profile_lds.sh ${TAG}_lds > $OUT/lds_counters.log 2>&1
python bench.py --steps 20 --warmup 3 > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log > $OUT/bench.json
find $OUT -name "*.csv" | head -40
