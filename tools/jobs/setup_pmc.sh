# PMC counters of the set-up kernels (tools/time_setup.py, 50M x 30k x ~40): one rocprofv3 --pmc pass per counter group
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/setup_pmc; rm -rf $O; mkdir -p $O
i=0
for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVE_CYCLES" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/g$i -- python $GRAFT_REPO_ROOT/tools/time_setup.py > $GRAFT_REPO_ROOT/$O/g$i.log 2>&1 ) || echo "group $i failed: $grp"
done
for k in k_sb_fill_sorted k_sb_deconflict k_colsig k_row_partcounts k_rowstats k_block_greedy; do python tools/pmc_summary.py $O $k; done > gpurun_out/setup_pmc_last.txt 2>&1
rm -rf $O/g*/runc
cat gpurun_out/setup_pmc_last.txt
