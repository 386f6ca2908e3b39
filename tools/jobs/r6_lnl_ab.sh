#!/bin/bash
# Round 6: same-box A/B of the log-table lnl pass with (pi*theta, log pi*theta) packed into one 16-byte LDS entry
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r6_lnl_ab; rm -rf $O; mkdir -p $O
for round in 1 2 3; do
  for lib in build_ab/lib_base.so build_ab/lib_packed.so; do
    for fmt in 1 2; do
      echo "== $lib value_format=$fmt (1: fp64 entries <4,9,2,0>, 2: score codes <4,9,1,0>) round $round"
      TSEM_LIB=$PWD/$lib timeout 300 python tools/time_lnl.py value_format=$fmt 2>&1 | grep -v "amdgpu\|WARNING" | tail -3
    done
  done
done > $O/ab.txt 2>&1
cat $O/ab.txt
