#!/bin/bash
# (an experiment that was reverted: `git apply tools/scratch/r06_merged_reduce_update.patch` first — without it fused_dbg bit 16 selects nothing and both legs run the
# three kernels; results: profiles/r06_fold_tail_ab.txt)
# k_reduce_update (column reduce + M-step in one kernel, single GPU) against k_colreduce + k_update (fused_dbg bit 16), same box:
# its tests, the time-out / chunk tests around it, then BASELINE configs 2, 3 and the headline  ->  gpurun_out/r6_merged/
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6_merged; export TMPDIR=/tmp
O=gpurun_out/r6_merged
timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round2.py tests/test_gpu_parity.py -x -q -m gpu -k "merged or timeout or chunk or overshoot or em_matches" > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -4 $O/pytest.txt
X="--no-cpu-baseline --no-alt-layout --no-reproducible-leg --no-precision-sweep"
run() { n=$1; shift; python bench.py "$@" $X > $O/$n.json 2> $O/$n.err; }
for dbg in 0 65536; do
  run config2_$dbg --config 2 --steps 400 --warmup 40 --fused-dbg $dbg
  run config2codes_$dbg --config 2 --value-format code16 --steps 400 --warmup 40 --fused-dbg $dbg
  run config3_$dbg --config 3 --steps 100 --warmup 20 --fused-dbg $dbg
done
run headline_0 --steps 20 --warmup 3
python - <<PY
import json
for n in ('config2', 'config2codes', 'config3', 'headline'):
    for dbg in (0, 65536):
        try:
            d = json.loads(open('$O/%s_%d.json' % (n, dbg)).read().strip().splitlines()[-1])
            c = d['check']
            print('%-14s dbg %5d  ms_per_step %.4f  kernel_ms %.4f  check %.17g %.17g %s ' % (n, dbg, d['ms_per_step'], d['roofline']['kernel_ms'], c['pi_weighted'], c['theta_weighted'], c.get('matches_embedded')),
                  {k: (round(v, 1) if isinstance(v, float) else v) for k, v in d['phase_us'].items() if k != 'how'})
        except Exception as e:
            print(n, dbg, 'failed', e)
PY
