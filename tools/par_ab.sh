#!/bin/bash
# parity of a kernel build on the bench's 400k-row sample (oracle check inside bench.py): tools/par_ab.sh "<lib> ..."
for L in $1; do TSEM_LIB=$PWD/$L python bench.py --steps 3 --warmup 1 --no-precision-sweep --no-alt-layout ${2:-} 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); p=d['parity_on_sample']; print('$L', 'lnl', p['lnl_rel_delta'], 'pi', p['pi_max_rel_delta'], 'count mismatches', p['final_count_mismatches'], 'kernel ms', d['roofline']['kernel_ms'])"; done
