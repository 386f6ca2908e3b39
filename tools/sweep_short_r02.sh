C="--steps 20 --warmup 3 --no-cpu-baseline --no-alt-layout --no-precision-sweep --no-reproducible-leg"
run() { python bench.py $C "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); l=d['config']['layout']
print('%-66s kernel %.3f ms  R=%d slow=%d' % ('$*', d['roofline']['kernel_ms'], l['R'], l['slow_path']))"; }
for rep in 1 2 3; do
for nz in 10 14 20 28; do
  run --nnz-row $nz --value-format code16 --sorted-fill 1 --geometry 2
  run --nnz-row $nz --value-format code16 --sorted-fill 1 --geometry 0
  run --nnz-row $nz --value-format code16 --sorted-fill 0 --geometry 2
  run --nnz-row $nz --value-format f64 --sorted-fill 0 --geometry 2
  run --nnz-row $nz --value-format f64 --sorted-fill 0 --geometry 0
done; done
