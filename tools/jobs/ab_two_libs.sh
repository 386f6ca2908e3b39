# same-box A/B of two library builds on the bench's timed loop (code16 and fp64 layouts, 40 and 18 entries per row):
#   bash tools/jobs/ab_two_libs.sh build_ab/lib_base.so build_ab/lib_x.so
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() { TSEM_LIB=$PWD/$1 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-alt-layout --no-precision-sweep --no-reproducible-leg --value-format $2 ${@:3} 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-24s %-7s %-14s kernel %.3f ms  step %.3f ms' % ('$1','$2','${*:3}',d['roofline']['kernel_ms'],d['ms_per_step']))"; }
for round in 1 2; do
  for lib in "$@"; do
    run $lib auto
    run $lib auto --nnz-row 18
  done
done
