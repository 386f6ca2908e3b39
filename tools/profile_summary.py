#!/usr/bin/env python3
"""Turn the CSVs written by tools/profile.sh into the text summaries kept under profiles/.

    python tools/profile_summary.py gpurun_out/prof/<tag> <round-prefix>      e.g. ... r01b r01

Writes profiles/<prefix>_fused_kernel_stats.txt, profiles/<prefix>_pmc_hbm_traffic_fused.txt and
profiles/pmc_traffic.json (HBM bytes per launch of the EM kernel, per entry format; FETCH_SIZE is
doubled on gfx950 as /opt/skills/guides/MI355X_MICROARCH.md prescribes for wide streaming reads).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    return name.split('(')[0].replace('void ', '')


WARMUP = 2          # launches of the EM kernel before bench.py's timed region (overwritten with the profiled run's own --warmup)


def kernel_table(path, steps=None):
    agg = defaultdict(list)
    meta = {}
    seen = defaultdict(int)
    for r in sorted(csv.DictReader(open(path)), key=lambda r: int(r['Start_Timestamp'])):
        k = short(r['Kernel_Name'])
        seen[k] += 1
        if k.startswith('k_em_fused') and (seen[k] <= WARMUP or (steps and seen[k] > WARMUP + steps)):
            continue                                        # the launches of bench.py's TIMED region only (not its warm-up, not the phase-timing /
                                                            # whole-call legs behind it): the summary then reproduces the line's roofline.kernel_ms
        agg[k].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
        meta[k] = (int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), r['Workgroup_Size_X'], r['LDS_Block_Size'],
                   int(r['VGPR_Count']) + int(r['Accum_VGPR_Count']), r['SGPR_Count'])
    tot = sum(sum(v) for v in agg.values())
    lines = ['%-40s %6s %12s %12s %12s %12s %6s  %7s %5s %7s %4s %4s' % (
        'kernel', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us', 'pct', 'grid', 'wg', 'lds_B', 'vgpr', 'sgpr')]
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        g, wx, lds, vg, sg = meta[k]
        lines.append('%-40s %6d %12.1f %12.1f %12.1f %12.1f %6.2f  %7d %5s %7s %4d %4s' % (
            k[:40], len(v), sum(v) / 1e3, sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3, 100.0 * sum(v) / tot,
            g, wx, lds, vg, sg))
    return lines, {k: sum(v) / len(v) / 1e3 for k, v in agg.items()}


def pmc_table(path, counter):
    agg = defaultdict(list)
    dur = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        k = short(r['Kernel_Name'])
        agg[k].append(float(r['Counter_Value']))
        dur[k].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    lines = ['%-40s %4s %16s %12s' % ('kernel', 'n', 'avg_' + counter + '(KB)', 'avg_dur_us')]
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:8]:
        lines.append('%-40s %4d %16.5g %12.1f' % (k[:40], len(v), sum(v) / len(v), sum(dur[k]) / len(dur[k]) / 1e3))
    return lines, {k: sum(v) / len(v) for k, v in agg.items()}


def main():
    src, prefix = sys.argv[1], sys.argv[2]
    one = lambda pat: sorted(glob.glob(os.path.join(src, pat)))[0]
    bench = json.load(open(os.path.join(src, 'bench.json')))
    global WARMUP
    steps = 6
    try:   # the profiled run's own result line (trace.log): its --steps / --warmup and the HIP-event kernel time to compare with
        own = json.loads([l for l in open(os.path.join(src, 'trace.log'), errors='replace') if l.startswith('{"metric"')][-1])
        steps, WARMUP = int(own['steps']), int(own['warmup'])
    except Exception:   # noqa: BLE001
        own = None
    cmd = 'python bench.py --steps %d --warmup %d --no-cpu-baseline --no-precision-sweep --no-reproducible-leg' % (steps, WARMUP)
    kl, avg = kernel_table(one('trace/*/*_kernel_trace.csv'), steps=steps)
    with open(os.path.join(ROOT, 'profiles', prefix + '_fused_kernel_stats.txt'), 'w') as f:
        f.write('# rocprofv3 --kernel-trace --stats -- %s   (default workload; the run times the headline fp64\n'
                '# layout k_em_fused<4, 0, 2, 0> and then the 2-byte-code layout k_em_fused<4, 0, 1, 0>;\n# template arguments: team size, mode (0 EM / 1 lnl), entry format, geometry)\n'
                '# k_em_fused rows: the launches of the TIMED region of each engine (the %d warm-up launches before it and the phase-timing / whole-call legs behind it are left out).  lds_B is the STATIC\n'
                '# allocation; the fused kernel allocates its LDS dynamically: %s B per workgroup (tsem.hip fz_lds_bytes).\n'
                % (cmd, WARMUP, bench['config']['layout'].get('lds_bytes', 'n/a')))
        f.write('\n'.join(kl) + '\n')
        if own:
            f.write('# the same run\'s own result line: roofline.kernel_ms %.4f (HIP events on the engine\'s stream around the timed launches), ms_per_step %.4f\n'
                    % (own['roofline']['kernel_ms'], own['ms_per_step']))
    fl, fetch = pmc_table(one('fetch/*/*_counter_collection.csv'), 'FETCH_SIZE')
    wl, write = pmc_table(one('write/*/*_counter_collection.csv'), 'WRITE_SIZE')
    runs = []
    notes = []
    for fmts, vb in (((2, 0), 8), ((1,), 2)):                # kernel FMT 2 / 0: fp64 entries, FMT 1: 2-byte codes
        name = next((n for f in fmts for n in fetch if n.startswith('k_em_fused<4, 0, %d' % f)), None)
        if name is None:
            continue
        rd = 2.0 * fetch[name] * 1024.0
        wr = write.get(name, 0.0) * 1024.0
        algo = (bench['roofline'] if vb == 8 else bench['code16_layout']['roofline'])['algo_bytes_per_launch']
        cfg = bench['config']
        runs.append({'workload': {'rows': cfg['rows'], 'cols': cfg['cols'], 'nnz_row': 40.0, 'dist': cfg['dist'],
                                  'n_gpus': 1, 'value_bytes': vb, 'kernel': name},
                     'traffic_bytes_per_launch': rd + wr, 'read_bytes': rd, 'write_bytes': wr,
                     'algorithmic_bytes': algo})
        notes.append('# %s: read = 2 * %.5g KB * 1024 = %.4g B ; write = %.4g B ; algorithmic = %d B ; traffic / algorithmic = %.3f'
                     % (name, fetch[name], rd, wr, algo, (rd + wr) / algo))
    with open(os.path.join(ROOT, 'profiles', prefix + '_pmc_hbm_traffic_fused.txt'), 'w') as f:
        f.write('## rocprofv3 --pmc FETCH_SIZE --kernel-trace -- %s\n' % cmd + '\n'.join(fl) + '\n\n')
        f.write('## rocprofv3 --pmc WRITE_SIZE --kernel-trace -- %s\n' % cmd + '\n'.join(wl) + '\n\n')
        f.write('# per EM pass; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced streaming read)\n')
        f.write('\n'.join(notes) + '\n')
    import datetime
    sys.path.insert(0, ROOT)
    from telescope_amd._lib import sources_fingerprint
    json.dump({'runs': runs,
               # what the numbers were measured on: bench.py prints this stamp beside `roofline.traffic` and says whether the
               # library's sources still are the ones measured (the constant must not go stale silently)
               'measured': {'date': datetime.date.today().isoformat(), 'sources_sha16': sources_fingerprint(),
                            'commit': os.environ.get('TSEM_COMMIT', 'unknown (the GPU box has no .git; see the commit that added this file)')},
               'source': 'profiles/%s_pmc_hbm_traffic_fused.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; '
                         'FETCH_SIZE x2 gfx950 correction)' % prefix},
              open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json'), 'w'), indent=1)
    print('\n'.join(kl[:8]))
    print('\n'.join(notes))


if __name__ == '__main__':
    main()
