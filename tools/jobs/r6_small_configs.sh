#!/bin/bash
# where an iteration's time goes on the small BASELINE configurations (2: 1M x 30k x ~20, 3: 10M x 30k x ~40) and on the 8-GPU shard
# (6.25M rows): the bench line's phase_us of each   ->  gpurun_out/r6_small/
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6_small; export TMPDIR=/tmp
O=gpurun_out/r6_small
X="--no-cpu-baseline --no-alt-layout --no-reproducible-leg --no-precision-sweep"
python bench.py --config 2 --steps 400 --warmup 40 $X > $O/config2.json 2> $O/config2.err
python bench.py --config 2 --value-format code16 --steps 400 --warmup 40 $X > $O/config2_codes.json 2>> $O/config2.err
python bench.py --config 3 --steps 100 --warmup 20 $X > $O/config3.json 2> $O/config3.err
python bench.py --rows 6250000 --steps 200 --warmup 20 $X > $O/shard.json 2> $O/shard.err
python - <<PY
import json
for n in ('config2', 'config2_codes', 'config3', 'shard'):
    try:
        d = json.loads(open('$O/%s.json' % n).read().strip().splitlines()[-1])
        print(n, 'ms_per_step %.4f' % d['ms_per_step'], 'kernel_ms %.4f' % d['roofline']['kernel_ms'], 'frac %.3f' % d['roofline']['frac'],
              {k: (round(v, 1) if isinstance(v, float) else v) for k, v in d['phase_us'].items() if k != 'how'})
    except Exception as e:
        print(n, 'failed', e)
PY
