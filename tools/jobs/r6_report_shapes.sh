#!/bin/bash
# Round 6: the packed report kernel against the capacity kernel (report_dbg = 8) on other shapes: short rows (E = 8), K = 50k (pi*theta of the
# cold ids from L2), long rows
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r6_report_shapes; rm -rf $O; mkdir -p $O
for cfg in "50000000 18 30000" "50000000 10 30000" "20000000 100 50000" "50000000 40 50000" "10000000 200 30000"; do
  set -- $cfg
  echo "== rows $1, ~$2 per row, $3 loci"
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/t_$1_$2_$3 -- python $GRAFT_REPO_ROOT/tools/time_report_final.py $1 $2 $3 0 8 2>&1 | grep "report_dbg" )
  python tools/kernel_table.py $O/t_$1_$2_$3 k_report | grep "k_report_pack32\|k_report_rows\|k_report_hist\|k_report_slow\|^kernel"
done > $O/shapes.txt 2>&1
cat $O/shapes.txt
find $O -name "*.db" -delete
