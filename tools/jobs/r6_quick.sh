#!/bin/bash
# quick A/B of report_dbg values of the packed report kernel under a kernel trace:  DBGS="0 64 512" bash tools/jobs/r6_quick.sh
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r6_quick; rm -rf $O; mkdir -p $O
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/t -- python $GRAFT_REPO_ROOT/tools/time_report_final.py ${ROWS:-50000000} ${NNZ:-40} ${COLS:-30000} ${DBGS:-0} > $GRAFT_REPO_ROOT/$O/time_final.txt 2>&1 )
grep -v "amdgpu\|WARNING\|^W2026\|^E2026" $O/time_final.txt
python - <<PY
import csv, glob
rows = []
for p in glob.glob('$O/t/*/*_kernel_trace.csv'):
    for r in csv.DictReader(open(p)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if k.startswith('k_report_pack32') or k.startswith('k_report_hist') or k.startswith('k_report_rows'):
            rows.append((int(r['Start_Timestamp']), k, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
rows.sort()
print(' '.join('%s:%.0f' % (k.replace('k_report_', ''), us) for _, k, us in rows))
PY
