# BASELINE config 2 (1M x 30k x ~20) with forced row blocks: would shorter steps shrink the persistent kernel's 5 steps of pipeline fill?
# No — a step's fixed costs dominate (one MI355X, ms per iteration fp64 / codes):  auto (R 760, geometry 2) 0.0986 / 0.0852 |
# 384: 0.1025 / 0.0966 | 320: 0.1079 / 0.1054 | 256: 0.1174 / 0.1152 | 192: 0.1358 / 0.1340 | 128: 0.1611 / 0.1600.  The layout's own choice stands.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
X="--no-cpu-baseline --no-alt-layout --no-reproducible-leg --no-precision-sweep"
for br in 0 128 192 256 320 384; do
  for vf in f64 code16; do
    python bench.py --config 2 --value-format $vf --steps 400 --warmup 40 --block-rows $br $X 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); l = d['config']['layout']
print('block_rows $br $vf  ms_per_step %.4f kernel_ms %.4f  R %d geo %d nb %d P %d' % (d['ms_per_step'], d['roofline']['kernel_ms'], l['R'], l['geometry'], l['nb'], l['P']))"
  done
done
