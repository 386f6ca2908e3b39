"""`telescope resume` on the MI355X engine (reference: telescope/telescope_resume.py:28-232,
telescope/__main__.py:49-92).  Same option names and defaults, same log lines, same output files:
`<outdir>/<exp_tag>-run_stats.tsv` and `<outdir>/<exp_tag>-TE_counts.tsv`.

    python -m telescope_amd resume <checkpoint.npz> [--reassign_mode exclude] [--outdir .] ...

`assign` needs the BAM/GTF loader, which is outside the accelerated path.
"""
import argparse
import logging as lg
import os
import sys
from time import time

import numpy as np

VERSION = '1.0.3.1-mi355x'


def format_minutes(seconds):
    return '%d minutes and %d secs' % (seconds // 60, seconds % 60)


def build_parser():
    ap = argparse.ArgumentParser(prog='telescope', description='Telescope EM reassignment on MI355X')
    sub = ap.add_subparsers(dest='command')
    rs = sub.add_parser('resume', help='Resume from a checkpoint: EM + reports')
    g = rs.add_argument_group('Input Options')
    g.add_argument('checkpoint', help='Path to checkpoint file.')
    g = rs.add_argument_group('Reporting Options')
    g.add_argument('--quiet', action='store_true', help='Silence (most) output.')
    g.add_argument('--debug', action='store_true', help='Print debug messages.')
    g.add_argument('--logfile', type=argparse.FileType('a'), help='Log output to this file.')
    g.add_argument('--outdir', default='.', help='Output directory.')
    g.add_argument('--exp_tag', default='telescope', help='Experiment tag')
    g = rs.add_argument_group('Run Modes')
    g.add_argument('--reassign_mode', default='exclude',
                   choices=['exclude', 'choose', 'average', 'conf', 'unique'],
                   help='Reassignment mode for the final counts.')
    g.add_argument('--conf_prob', type=float, default=0.9,
                   help='Minimum probability for high confidence assignment.')
    g = rs.add_argument_group('Model Parameters')
    g.add_argument('--pi_prior', type=int, default=0, help='Prior on pi. Equivalent to adding n unique reads.')
    g.add_argument('--theta_prior', type=int, default=200000,
                   help='Prior on theta. Equivalent to adding n non-unique reads.')
    g.add_argument('--em_epsilon', type=float, default=1e-7, help='EM Algorithm Epsilon cutoff')
    g.add_argument('--max_iter', type=int, default=100, help='EM Algorithm maximum iterations')
    g.add_argument('--use_likelihood', action='store_true',
                   help='Use difference in log-likelihood as convergence criteria.')
    g.add_argument('--skip_em', action='store_true', help='Exits after loading the checkpoint.')
    g = rs.add_argument_group('Device')
    g.add_argument('--device', type=int, default=0, help='GPU index (single-process runs).')
    sub.add_parser('assign', help='(not available: needs the BAM/GTF loader of the reference)')
    return ap


class ResumeOptions(object):
    def __init__(self, args):
        self.__dict__.update(vars(args))
        self.version = VERSION
        if self.logfile is None:
            self.logfile = sys.stderr

    def outfile_path(self, suffix):
        return os.path.join(self.outdir, '%s-%s' % (self.exp_tag, suffix))

    def __str__(self):
        keys = ('checkpoint', 'quiet', 'debug', 'outdir', 'exp_tag', 'reassign_mode', 'conf_prob',
                'pi_prior', 'theta_prior', 'em_epsilon', 'max_iter', 'use_likelihood')
        lines = ['{:34}{}'.format('Version:', self.version)]
        lines += ['    {:30}{}'.format(k + ':', getattr(self, k)) for k in keys]
        return '\n'.join(lines)


def configure_logging(opts):
    """utils/__init__.py:84-104 — same format string."""
    level = lg.DEBUG if opts.debug else (lg.WARNING if opts.quiet else lg.INFO)
    fmt = '%(asctime)s %(levelname)-8s %(message)-60s (from %(funcName)s in %(filename)s:%(lineno)d)'
    lg.basicConfig(level=level, format=fmt, datefmt='%Y-%m-%d %H:%M:%S', stream=opts.logfile, force=True)


def run_resume(args):
    """telescope_resume.py:183-232."""
    from .likelihood import TelescopeLikelihood
    from .run_container import Telescope
    opts = ResumeOptions(args)
    configure_logging(opts)
    lg.info('\n{}\n'.format(opts))
    total_time = time()
    lg.info('Loading Telescope object from file...')
    ts = Telescope.load(opts.checkpoint)
    ts.opts = opts
    ts.print_summary(lg.INFO)
    if opts.skip_em:
        lg.info('Skipping EM...')
        return 0
    seed = ts.get_random_seed()
    lg.debug('Random seed: {}'.format(seed))
    np.random.seed(seed)
    ts_model = TelescopeLikelihood(ts.raw_scores, opts, device=opts.device)
    lg.info('Running Expectation-Maximization...')
    stime = time()
    ts_model.em(use_likelihood=opts.use_likelihood, loglev=lg.INFO)
    lg.info('EM completed in %s' % format_minutes(time() - stime))
    lg.info('Generating Report...')
    os.makedirs(opts.outdir, exist_ok=True)
    ts.output_report(ts_model, opts.outfile_path('run_stats.tsv'), opts.outfile_path('TE_counts.tsv'))
    lg.info('telescope resume complete (%s)' % format_minutes(time() - total_time))
    return 0


def main(argv=None):
    ap = build_parser()
    args = ap.parse_args(argv)
    if args.command == 'resume':
        return run_resume(args)
    if args.command == 'assign':
        ap.error('`assign` needs the BAM/GTF loader, which is outside this engine; run the reference '
                 '`telescope assign --skip_em` to write a checkpoint, then `resume` it here.')
    ap.print_help()
    return 2


if __name__ == '__main__':
    sys.exit(main())
