#!/usr/bin/env python3
"""Per-kernel table of a rocprofv3 --kernel-trace CSV directory:  python tools/kernel_table.py <dir> [name-filter]"""
import csv, glob, sys
from collections import defaultdict
d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
agg, meta = defaultdict(list), {}
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if flt and flt not in k:
            continue
        agg[k].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
        meta[k] = (int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), r['Workgroup_Size_X'], r['LDS_Block_Size'],
                   int(r['VGPR_Count']) + int(r.get('Accum_VGPR_Count', 0) or 0), r['SGPR_Count'])
tot = sum(sum(v) for v in agg.values()) or 1
print('%-52s %6s %11s %11s %11s %11s %6s %6s %5s %7s %4s %4s' % ('kernel', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us', 'pct', 'grid', 'wg', 'lds_B', 'vgpr', 'sgpr'))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    g, wx, lds, vg, sg = meta[k]
    print('%-52s %6d %11.1f %11.1f %11.1f %11.1f %6.2f %6d %5s %7s %4d %4s' % (k[:52], len(v), sum(v) / 1e3, sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3, 100.0 * sum(v) / tot, g, wx, lds, vg, sg))
