#!/bin/bash
# the library's own layout choice (value_format auto = score codes, deconflict on, geometry by row length) and the fp64 layout at short
# rows, three repeats on one box: kernel ms and fraction of the HBM peak per EM pass
C="--steps 20 --warmup 3 --no-cpu-baseline --no-alt-layout --no-precision-sweep --no-reproducible-leg"
run() { python bench.py $C "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); l=d['config']['layout']
print('%-44s kernel %.3f ms  frac %.3f  R=%d P=%d geo=%d slow=%d' % ('$*', d['roofline']['kernel_ms'], d['roofline']['frac'], l['R'], l['P'], l['geometry'], l['slow_path']))"; }
for rep in 1 2 3; do
for nz in 10 14 18 20 28 40; do
  run --nnz-row $nz --value-format auto
  run --nnz-row $nz --value-format f64
done; done
