#!/bin/bash
# order x geometry x format sweep of the fused EM kernel (kernel ms), the table behind the layout rules
C="--steps 12 --warmup 2 --no-cpu-baseline --no-alt-layout --no-precision-sweep --no-reproducible-leg"
run() { python bench.py $C "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); l=d['config']['layout']
print('%-70s kernel %.3f ms  frac %.3f  R=%d P=%d slow=%d' % ('$*', d['roofline']['kernel_ms'], d['roofline']['frac'], l['R'], l['P'], l['slow_path']))"; }
for nz in 10 20 40; do
 for f in code16 f64; do
  for so in 0 1; do
   for g in 0 2; do
     run --nnz-row $nz --value-format $f --sorted-fill $so --geometry $g
   done
  done
 done
done
for f in code16 f64; do for so in 0 1; do run --rows 20000000 --cols 50000 --nnz-row 100 --value-format $f --sorted-fill $so; done; done
for nz in 10 20 40; do run --nnz-row $nz --value-format auto; done
run --rows 20000000 --cols 50000 --nnz-row 100 --value-format auto
