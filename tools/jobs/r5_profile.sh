#!/bin/bash
# Round-5 profile session on the GPU box: kernel stats + HBM counters of the bench (tools/profile.sh stages), the lnl passes
# (kernel time + VALU counter, log tables on / off), the report kernels (time + FETCH / WRITE), set-up, end to end, --use_likelihood.
set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/prof/r05; rm -rf $O; mkdir -p $O
STAGES="trace pmc lds" timeout 1500 bash tools/profile.sh r05 > $O/profile_sh.log 2>&1
python bench.py --steps 20 --warmup 5 > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log > $O/bench.json
python tools/profile_summary.py $O r05 > $O/summary.log 2>&1
# ---- lnl passes ----
R="cd /tmp && timeout 300 rocprofv3"
for fmt in 1 2; do for dbg in 0 8192; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/lnl_t_${fmt}_${dbg} -- python $GRAFT_REPO_ROOT/tools/time_lnl.py value_format=$fmt fused_dbg=$dbg > $GRAFT_REPO_ROOT/$O/lnl_t_${fmt}_${dbg}.log 2>&1 )
  ( cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/lnl_c_${fmt}_${dbg} -- python $GRAFT_REPO_ROOT/tools/time_lnl.py value_format=$fmt fused_dbg=$dbg > /dev/null 2>&1 )
done; done
{ for fmt in 1 2; do for dbg in 0 8192; do echo "== value_format=$fmt (1 fp64 entries, 2 score codes) fused_dbg=$dbg (8192: per-entry logarithm, MODE 1; 0: log tables, MODE 9)"; grep -v amdgpu $O/lnl_t_${fmt}_${dbg}.log | tail -3; python tools/kernel_table.py $O/lnl_t_${fmt}_${dbg} k_em_fused | tail -n +2; python tools/pmc_summary.py $O/lnl_c_${fmt}_${dbg}/.. k_em_fused 2>/dev/null | grep -A8 "lnl_c_${fmt}_${dbg}" | head -0; done; done; } > $O/lnl_passes.txt 2>&1
for fmt in 1 2; do for dbg in 0 8192; do echo "== counters value_format=$fmt fused_dbg=$dbg"; python - <<PY
import csv, glob, collections
v = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob('$O/lnl_c_${fmt}_${dbg}/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(p)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if k.startswith('k_em_fused'):
            v[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(v):
    print('  %-28s' % k, '  '.join('%s %.4g' % (c, sum(x) / len(x)) for c, x in sorted(v[k].items())))
PY
done; done >> $O/lnl_passes.txt 2>&1
# ---- report kernels ----
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/rep_t -- python $GRAFT_REPO_ROOT/tools/time_report.py > $GRAFT_REPO_ROOT/$O/time_report.txt 2>&1 )
( cd /tmp && timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/rep_f -- python $GRAFT_REPO_ROOT/tools/time_report.py 20000000 > /dev/null 2>&1 )
( cd /tmp && timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/rep_w -- python $GRAFT_REPO_ROOT/tools/time_report.py 20000000 > /dev/null 2>&1 )
{ python tools/kernel_table.py $O/rep_t k_report; echo; echo "## FETCH_SIZE / WRITE_SIZE (KB per launch, 20M rows x ~40: 0.8e9 entries, 3.28 GB algorithmic at 4 B per entry + 8 B + 4 B per row)"; python - <<PY
import csv, glob, collections
for d, c in (('rep_f', 'FETCH_SIZE'), ('rep_w', 'WRITE_SIZE')):
    v = collections.defaultdict(list)
    for p in glob.glob('$O/%s/*/*_counter_collection.csv' % d):
        for r in csv.DictReader(open(p)):
            k = r['Kernel_Name'].split('(')[0].replace('void ', '')
            if k.startswith('k_report') and r['Counter_Name'] == c:
                v[k].append(float(r['Counter_Value']))
    for k in sorted(v):
        print('%-12s %-36s n=%d avg %.5g KB' % (c, k, len(v[k]), sum(v[k]) / len(v[k])))
PY
} > $O/report_kernels.txt 2>&1
# ---- set-up, end to end, --use_likelihood, short rows ----
TSEM_TRACE=1 python tools/time_setup.py 2>&1 | grep -v amdgpu > $O/time_setup.txt
python tools/time_setup.py 2>&1 | grep -v amdgpu >> $O/time_setup.txt
python tools/time_setup_twice.py 2>&1 | grep -v amdgpu >> $O/time_setup.txt
python tools/time_e2e.py 2>&1 | grep -v "amdgpu\|WARNING" > $O/time_e2e.txt
python tools/time_use_likelihood.py 2>&1 | grep -v "amdgpu\|WARNING" > $O/use_likelihood.txt
python tools/time_whole_call.py 2>&1 | grep -v "amdgpu\|WARNING" > $O/whole_call.txt
bash tools/sweep_short_r03.sh > $O/sweep_short.txt 2>&1
# ---- the 8-GPU shard: per-iteration cost and phases ----
python bench.py --rows 6250000 --steps 40 --warmup 5 --no-cpu-baseline --no-alt-layout --no-precision-sweep --no-reproducible-leg --kernel-timing 0 2>/dev/null | tail -1 > $O/bench_shard.json
python bench.py --rows 6250000 --steps 40 --warmup 5 --no-cpu-baseline --no-alt-layout --no-precision-sweep --no-reproducible-leg --kernel-timing 0 --force-comm 2>/dev/null | tail -1 > $O/bench_shard_comm.json
# ---- capacity point: half of BASELINE config 5 on one GPU ----
python bench.py --rows 100000000 --cols 50000 --nnz-row 100 --value-format auto --steps 10 --warmup 2 --no-cpu-baseline --no-alt-layout --no-precision-sweep --no-reproducible-leg 2>$O/capacity.err | tail -1 > $O/capacity_point.json
find $O -name "*.db" -delete; du -sh $O
ls $O
