#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd .db (kernel trace / PMC) as text for profiles/.

    python tools/rocpd_summary.py gpurun_out/prof/x_results.db [--skip-first N] > profiles/...txt
"""
import sqlite3
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    c = sqlite3.connect(path)
    rows = c.execute('select name, duration, grid_x, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, '
                     'sgpr_count, start from kernels order by start').fetchall()
    agg = defaultdict(list)
    meta = {}
    for name, dur, gx, wx, lds, vg, av, sg, st in rows:
        short = name.split('(')[0]
        agg[short].append(dur)
        meta[short] = (gx // max(1, wx), wx, lds, vg, av, sg)
    total = sum(sum(v) for v in agg.values())
    print('# rocprofv3 --kernel-trace summary of %s' % path)
    print('%-44s %6s %12s %12s %12s %12s %6s  %7s %5s %7s %4s %4s' % (
        'kernel', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us', 'pct', 'grid', 'wg', 'lds_B', 'vgpr', 'sgpr'))
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        g, wx, lds, vg, av, sg = meta[k]
        print('%-44s %6d %12.1f %12.1f %12.1f %12.1f %6.2f  %7d %5d %7d %4d %4d' % (
            k[:44], len(v), sum(v) / 1e3, sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3,
            100.0 * sum(v) / total, g, wx, lds, vg + av, sg))
    try:
        pm = c.execute('select * from pmc_events limit 1').fetchall()
    except sqlite3.Error:
        pm = []
    if pm:
        cols = [r[1] for r in c.execute("pragma table_info('pmc_events')")]
        print('\n# PMC events, columns: %s' % cols)
        q = c.execute('select name, counter_name, avg(value), count(*) from pmc_events group by name, counter_name')
        for name, cn, val, n in q:
            print('%-44s %-24s avg=%.6g n=%d' % (name.split('(')[0][:44], cn, val, n))


if __name__ == '__main__':
    main()
