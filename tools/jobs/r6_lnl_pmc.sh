#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r6_lnl_pmc; rm -rf $O; mkdir -p $O
for v in base packed; do
  ( cd /tmp && TSEM_LIB=$GRAFT_REPO_ROOT/build_ab/lib_$v.so timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/$v -- python $GRAFT_REPO_ROOT/tools/time_lnl.py value_format=2 > /dev/null 2>&1 )
done
for v in base packed; do mkdir -p $O/only_$v; mv $O/$v $O/only_$v/; echo "#### lib_$v.so (FZ_LT_PACKED=$([ $v = packed ] && echo 1 || echo 0)), tools/time_lnl.py value_format=2"; python tools/pmc_summary.py $O/only_$v "k_em_fused<4, 9"; done > $O/pmc.txt 2>&1; cat $O/pmc.txt
rm -rf $O/only_*
