#!/bin/bash
# Round profile of the bench workload on the GPU box -> gpurun_out/prof/<tag>/ ; summarise with
# tools/profile_summary.py <dir> <round-prefix>.   tools/profile.sh <tag>
# Every --pmc pass is its own rocprofv3 run with --kernel-trace only (no other trace domain).
# STAGES="trace pmc lds rest" (default all) lets a stage be rerun alone; every rocprofv3 run is under `timeout`.
set -u
TAG=${1:-r}
OUT=gpurun_out/prof/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
# (round 6: the driver's own step counts — with 2 warm-up steps the 6 timed launches were still on the ramp after the generator: 4.5 .. 4.14 ms)
B="python bench.py --steps ${PROF_STEPS:-20} --warmup ${PROF_WARMUP:-5} --no-cpu-baseline --no-precision-sweep --no-reproducible-leg --kernel-timing 1"
STAGES=${STAGES:-"trace pmc lds rest"}
has() { [[ " $STAGES " == *" $1 "* ]]; }
if has trace; then
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/trace.log 2>&1
fi
if has pmc; then
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- $B > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- $B > $OUT/write.log 2>&1
fi
# LDS / SQ counters of both entry formats (three passes)
if has lds; then
tools/profile_lds.sh ${TAG}/pmc > $OUT/lds_counters.log 2>&1
fi
if has rest; then
# (no TCC_*_sum pass: rocprofv3 aborts on that counter set on this image and then hangs until killed — r02, 50 GPU-minutes)
python tools/fused_prof.py 20000000 value_format=2 > $OUT/timeline_code16.txt 2>&1
python tools/fused_prof.py 20000000 value_format=1 > $OUT/timeline_f64.txt 2>&1
tools/ubench/lds > $OUT/lds_ubench.log 2>&1
tools/ubench/stream > $OUT/stream_ubench.log 2>&1
# per-iteration cost outside the EM kernel, and what the library communicator adds (1 rank: host + launch side only)
for kt in 0; do for fc in "" "--force-comm"; do for r in 6250000 50000000; do
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-alt-layout --no-precision-sweep --no-reproducible-leg --kernel-timing $kt --rows $r $fc 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('rows $r kernel_timing $kt $fc: %.4f ms per EM iteration' % d['ms_per_step'])"
done; done; done > $OUT/comm_overhead.txt 2>&1
for r in 6250000 50000000; do
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-alt-layout --no-precision-sweep --no-reproducible-leg --kernel-timing 1 --rows $r 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('rows $r kernel_timing 1: %.4f ms per EM iteration, EM kernel %.4f ms' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
done >> $OUT/comm_overhead.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-alt-layout --no-precision-sweep --no-reproducible-leg --kernel-timing 0 --rows 6250000 > /dev/null 2>&1
python tools/iter_timeline.py $(ls $OUT/tl/*/*_kernel_trace.csv | head -1) 12 2 > $OUT/iter_timeline.txt 2>&1
python tools/time_setup.py > $OUT/time_setup.txt 2>&1
python tools/time_e2e.py > $OUT/time_e2e.txt 2>&1
python bench.py --steps 20 --warmup 3 > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log > $OUT/bench.json
fi
ls $OUT
