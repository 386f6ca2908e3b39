#!/bin/bash
# (an experiment that was reverted: `git apply tools/scratch/r06_fold_tail.patch` first — without it fused_dbg bit 16 selects nothing and both legs run the
# three kernels; results: profiles/r06_fold_tail_ab.txt)
# the folded tail of the EM pass (FzFold) against the three-kernel iteration (fused_dbg bit 16), same box: parity tests of the goldens,
# then BASELINE configs 2, 3, the 8-GPU shard and the headline   ->  gpurun_out/r6_fold/
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6_fold; export TMPDIR=/tmp
O=gpurun_out/r6_fold
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round4.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -3 $O/pytest.txt
X="--no-cpu-baseline --no-alt-layout --no-reproducible-leg --no-precision-sweep"
run() {  # name, args...
  n=$1; shift
  python bench.py "$@" $X > $O/$n.json 2> $O/$n.err
}
for dbg in 0 65536; do
  run config2_$dbg --config 2 --steps 400 --warmup 40 --fused-dbg $dbg


  run shard_$dbg --rows 6250000 --steps 200 --warmup 20 --fused-dbg $dbg
  run headline_$dbg --steps 20 --warmup 3 --fused-dbg $dbg
done
python - <<PY
import json
for n in ('config2', 'shard', 'headline'):
    for dbg in (0, 65536):
        try:
            d = json.loads(open('$O/%s_%d.json' % (n, dbg)).read().strip().splitlines()[-1])
            print('%-14s dbg %5d  ms_per_step %.4f  kernel_ms %.4f  frac %.3f  check %s ' % (n, dbg, d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['check']),
                  {k: (round(v, 1) if isinstance(v, float) else v) for k, v in d['phase_us'].items() if k != 'how'})
        except Exception as e:
            print(n, dbg, 'failed', e)
PY
