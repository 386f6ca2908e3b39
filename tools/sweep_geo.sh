#!/bin/bash
# geometry A/B at short rows: tools/sweep_geo.sh "<nnz values>" "<geometries>"
C="--steps 20 --warmup 3 --no-cpu-baseline --no-alt-layout --no-precision-sweep --no-reproducible-leg --value-format auto"
run() { python bench.py $C "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); l=d['config']['layout']
print('%-36s kernel %.3f ms  frac %.3f  R=%d P=%d geo=%d lds=%d slow=%d' % ('$*', d['roofline']['kernel_ms'], d['roofline']['frac'], l['R'], l['P'], l['geometry'], l['lds_bytes'], l['slow_path']))"; }
for rep in 1 2; do for nz in $1; do for g in $2; do run --nnz-row $nz --geometry $g; done; done; done
