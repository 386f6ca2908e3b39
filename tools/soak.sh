#!/bin/bash
# many EM iterations back to back per layout / geometry: no watchdog time-out, no fallback, stable time per iteration
run() { python bench.py --steps $1 --warmup 3 --no-cpu-baseline --no-alt-layout --no-precision-sweep --no-reproducible-leg "${@:2}" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); l=d['config']['layout']
print('%-44s steps %5d  ms/iteration %.3f  EM kernel ms %.3f  fallbacks %d  tag misses in the last pass %d  geo %d R %d' % ('${*:2}', d['steps'], d['ms_per_step'], d['roofline']['kernel_ms'], l['fallbacks'], l['slow_path'], l['geometry'], l['R']))"; }
run 3000 --value-format f64
run 3000 --value-format auto
run 5000 --value-format auto --nnz-row 10
run 5000 --value-format auto --nnz-row 20
run 3000 --value-format auto --rows 20000000 --cols 50000 --nnz-row 100
run 8000 --value-format auto --rows 6250000
# teams of 7-8 with the 768-slot geometry (end of round 3)
run 4000 --value-format auto --rows 20000000 --cols 50000 --nnz-row 20
run 4000 --value-format auto --rows 20000000 --cols 50000 --nnz-row 40
run 3000 --value-format f64 --rows 20000000 --cols 45000 --nnz-row 30
# round 4: the split layout (K > 61 440: a row-sum pass and a scatter pass per iteration, no team exchange)
run 3000 --value-format auto --rows 10000000 --cols 100000 --nnz-row 40
run 3000 --value-format f64 --rows 4000000 --cols 122880 --nnz-row 100
run 3000 --value-format auto --rows 20000000 --cols 70000 --nnz-row 18
