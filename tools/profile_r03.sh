#!/bin/bash
# Round-3 profile session on the GPU box -> gpurun_out/prof/r03/ (then tools/regen_profiles_r03.sh).  STAGES as in tools/profile.sh, plus
# "report" (kernel trace + HBM counters of the report pass and the whole run) and "short" (short-row sweep).
set -u
OUT=gpurun_out/prof/r03
mkdir -p $OUT
export TMPDIR=/tmp
STAGES=${STAGES:-"trace pmc lds rest report short"}
has() { [[ " $STAGES " == *" $1 "* ]]; }
S=""
for s in trace pmc lds rest; do has $s && S="$S $s"; done
[ -n "$S" ] && STAGES="$S" tools/profile.sh r03
if has report; then
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/e2e -- python tools/time_e2e.py > $OUT/time_e2e_traced.txt 2>&1
  python tools/kernel_table.py $OUT/e2e > $OUT/e2e_kernels.txt 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/e2e_fetch -- python tools/time_e2e.py 20000000 5 > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/e2e_write -- python tools/time_e2e.py 20000000 5 > /dev/null 2>&1
  python tools/pmc_summary.py $OUT k_report_rows > $OUT/report_pmc.txt 2>&1
  python tools/time_e2e.py > $OUT/time_e2e.txt 2>&1
  python tools/time_report.py > $OUT/time_report.txt 2>&1
  python tools/time_report.py 50000000 14 > $OUT/time_report_14.txt 2>&1
  python tools/time_choose.py > $OUT/time_choose.txt 2>&1
fi
if has short; then
  tools/sweep_short_r03.sh > $OUT/sweep_short.txt 2>&1
fi
ls $OUT
