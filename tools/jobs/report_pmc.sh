# PMC counters of the report kernels (tools/time_report.py, 50M x 30k x ~40): one rocprofv3 --pmc pass per counter group
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/report_pmc; rm -rf $O; mkdir -p $O
i=0
for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVE_CYCLES" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  ( cd /tmp && timeout 400 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/g$i -- python $GRAFT_REPO_ROOT/tools/time_report.py > $GRAFT_REPO_ROOT/$O/g$i.log 2>&1 ) || echo "group $i failed: $grp"
done
python tools/pmc_summary.py $O k_report > gpurun_out/report_pmc_last.txt 2>&1
rm -rf $O/g*/runc
grep -A22 "k_report_rows<4, 16, false\|k_report_init_codes<4\|k_report_rows<4, 16, true" gpurun_out/report_pmc_last.txt
