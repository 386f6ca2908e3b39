#!/usr/bin/env python3
"""Per-iteration timeline from a rocprofv3 --kernel-trace CSV: kernel durations and the gaps between them.
    python tools/iter_timeline.py <kernel_trace.csv> [first_iteration] [n]"""
import csv
import sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
ks = [(r['Kernel_Name'].split('(')[0].replace('void ', '')[:28], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
idx = [i for i, k in enumerate(ks) if k[0].startswith('k_em_fused')]
first = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
n = int(sys.argv[3]) if len(sys.argv) > 3 else 3
for a in idx[first:first + n]:
    b = idx[idx.index(a) + 1] if idx.index(a) + 1 < len(idx) else len(ks)
    t0 = ks[a][1]
    print('--- iteration (period %.1f us)' % ((ks[b][1] - t0) / 1e3 if b < len(ks) else float('nan')))
    prev_end = None
    for name, s, e in ks[a:b]:
        print('  %-28s start %8.1f us  dur %7.1f us  gap before %5.1f us' % (name, (s - t0) / 1e3, (e - s) / 1e3, ((s - prev_end) / 1e3) if prev_end else 0.0))
        prev_end = e
