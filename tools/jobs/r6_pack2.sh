#!/bin/bash
# Round 6: the packed fp32 report kernel — tests, timing experiments, kernel trace, PMC groups.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r6_pack2; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round6.py -x -q -k "packed or near_ties or bundled" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -5 $O/tests.log
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/t -- python $GRAFT_REPO_ROOT/tools/time_report_final.py 50000000 40 30000 0 8 128 256 272 288 304 > $GRAFT_REPO_ROOT/$O/time_final.txt 2>&1 )
grep -v "amdgpu\|WARNING" $O/time_final.txt; python tools/kernel_table.py $O/t k_report > $O/kernels.txt 2>&1; cat $O/kernels.txt
i=0
for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  ( cd /tmp && timeout 400 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/g$i -- python $GRAFT_REPO_ROOT/tools/time_report_final.py 50000000 40 30000 0 > $GRAFT_REPO_ROOT/$O/g$i.log 2>&1 ) || echo "group $i failed: $grp"
done
python tools/pmc_summary.py $O k_report_pack32 > $O/pmc.txt 2>&1; cat $O/pmc.txt
python tools/pmc_summary.py $O k_report_hist >> $O/pmc.txt 2>&1
find $O -name "*.db" -delete; rm -rf $O/g*/runc
