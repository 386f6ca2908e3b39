#!/bin/bash
# the rows of profiles/HISTORY.md section 9: shard sizes, config-5-like shard, uniform columns, two-pass, both entry formats
C="--steps 20 --warmup 3 --no-cpu-baseline --no-alt-layout"
for f in code16 f64; do
  echo "== value format $f"
  tools/sweep.sh "$C --value-format $f" --rows 50000000 25000000 12500000 6250000
  tools/sweep.sh "$C --value-format $f --cols 50000 --nnz-row 100" --rows 20000000
  tools/sweep.sh "$C --value-format $f --dist uniform" --rows 50000000
done
echo "== two-pass"
tools/sweep.sh "$C --em-kernel twopass" --rows 50000000
