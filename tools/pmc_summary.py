#!/usr/bin/env python3
"""Per-kernel averages of every counter found in the rocprofv3 --pmc CSVs under <dir>/*/ (tools/profile_lds.sh).

    python tools/pmc_summary.py gpurun_out/prof/<tag> [kernel-name-prefix]
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    src = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else 'k_em_fused'
    vals = defaultdict(lambda: defaultdict(list))
    durs = defaultdict(list)
    for path in sorted(glob.glob(os.path.join(src, '*', '*', '*_counter_collection.csv'))):
        for r in csv.DictReader(open(path)):
            k = r['Kernel_Name'].split('(')[0].replace('void ', '')
            if not k.startswith(want):
                continue
            vals[k][r['Counter_Name']].append(float(r['Counter_Value']))
            if 'End_Timestamp' in r:
                durs[k].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    for k in sorted(vals):
        d = durs[k]
        print('## %s   (%d profiled launches, avg %.1f us under the profiler)' % (k, len(d), sum(d) / max(1, len(d)) / 1e3))
        for c in sorted(vals[k]):
            v = vals[k][c]
            print('  %-28s avg %.6g  (n=%d)' % (c, sum(v) / len(v), len(v)))
        g = {c: sum(v) / len(v) for c, v in vals[k].items()}
        if 'SQ_LDS_IDX_ACTIVE' in g and g['SQ_LDS_IDX_ACTIVE']:
            print('  -> bank-conflict share of LDS-array cycles: %.3f' % (g.get('SQ_LDS_BANK_CONFLICT', 0) / g['SQ_LDS_IDX_ACTIVE']))
        if 'SQ_INSTS_LDS' in g and g['SQ_INSTS_LDS'] and 'SQ_LDS_IDX_ACTIVE' in g:
            print('  -> LDS-array cycles per LDS wave-instruction: %.2f' % (g['SQ_LDS_IDX_ACTIVE'] / g['SQ_INSTS_LDS']))
        print()


if __name__ == '__main__':
    main()
