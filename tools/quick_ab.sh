#!/bin/bash
# kernel ms of both entry formats on the bench workload (and optional extra bench args): tools/quick_ab.sh [args]
for f in code16 f64; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-alt-layout --no-precision-sweep --no-reproducible-leg --value-format $f "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-7s kernel %.3f ms  step %.3f ms  frac %.3f  slow %d' % ('$f',d['roofline']['kernel_ms'],d['ms_per_step'],d['roofline']['frac'],d['config']['layout']['slow_path']))"
done
