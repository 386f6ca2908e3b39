#!/bin/bash
# A/B of kernel builds on ONE box: tools/ab.sh "<lib1> <lib2> ..." "<bench args>"   (prints kernel ms per lib, 2 rounds)
for round in 1 2; do
for L in $1; do
  for f in code16 f64; do
    TSEM_LIB=$PWD/$L timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-alt-layout --no-precision-sweep --no-reproducible-leg --value-format $f $2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-40s %-7s kernel %.3f ms  step %.3f ms  frac %.3f  slow %d' % ('$L','$f',d['roofline']['kernel_ms'],d['ms_per_step'],d['roofline']['frac'],d['config']['layout']['slow_path']))"
  done
done
done
