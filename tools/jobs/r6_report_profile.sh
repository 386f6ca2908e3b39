#!/bin/bash
# Round 6 profile of the report passes: kernel table of tools/time_report.py (every kernel of the sweep: packed fp32 filter, capacity
# kernel, codes-only, generic), PMC groups + FETCH / WRITE of the packed kernel, end to end, the bench line with its report_pass block.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r6_report_profile; rm -rf $O; mkdir -p $O
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/rep_t -- python $GRAFT_REPO_ROOT/tools/time_report.py > $GRAFT_REPO_ROOT/$O/time_report.txt 2>&1 )
grep -v "amdgpu\|WARNING\|^W2026\|^E2026" $O/time_report.txt > $O/time_report_clean.txt; cat $O/time_report_clean.txt
python tools/kernel_table.py $O/rep_t k_report > $O/report_kernels.txt 2>&1; cat $O/report_kernels.txt
i=0
for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  ( cd /tmp && timeout 400 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/g$i -- python $GRAFT_REPO_ROOT/tools/time_report_final.py 50000000 40 30000 0 128 > $GRAFT_REPO_ROOT/$O/g$i.log 2>&1 ) || echo "group $i failed: $grp"
done
{ python tools/pmc_summary.py $O k_report_pack32; python tools/pmc_summary.py $O k_report_hist; } > $O/pmc.txt 2>&1; cat $O/pmc.txt
timeout 600 python tools/time_e2e.py 2>&1 | grep -v "amdgpu\|WARNING" > $O/time_e2e.txt; tail -12 $O/time_e2e.txt
python bench.py --steps 20 --warmup 5 > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log > $O/bench.json
python -c "
import json; d=json.loads(open('$O/bench.json').read()); print(json.dumps(d.get('report_pass'))); print(d['ms_per_step'], d['roofline']['frac'])"
find $O -name "*.db" -delete; rm -rf $O/g*/runc
