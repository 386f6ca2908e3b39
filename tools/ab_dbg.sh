#!/bin/bash
# one library, several fused_dbg values (timing experiments): tools/ab_dbg.sh <lib> "<dbg values>" "<bench args>"
for round in 1 2; do
for D in $2; do
  for f in code16 f64; do
    TSEM_LIB=$PWD/$1 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-alt-layout --no-precision-sweep --no-reproducible-leg --value-format $f --fused-dbg $D $3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('dbg %-5s %-7s kernel %.3f ms  step %.3f ms  frac %.3f' % ('$D','$f',d['roofline']['kernel_ms'],d['ms_per_step'],d['roofline']['frac']))"
  done
done
done
