#!/bin/bash
# usage: tools/sweep.sh "<common bench args>" "<varying flag>" v1 v2 ...
common="$1"; flag="$2"; shift 2
for v in "$@"; do
  python bench.py $common $flag $v 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
l=d['config']['layout']
print('$flag $v: %.3f ms/step  kernel %.3f ms  frac %.4f  %.1f Gnnz/s  R=%d nb=%d fused=%d pad=%.3f slow=%d maxsb=%d' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['nnz_per_sec']/1e9, l['R'], l['nb'], l['fused'], l['nnz_pad']/max(1,l['nnz_amb']), l['slow_path'], l['max_subblock']))"
done
