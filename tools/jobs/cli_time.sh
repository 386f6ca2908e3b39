cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
python -m telescope_amd resume tests/golden/resume_checkpoint.npz --outdir /tmp/o0 > /dev/null 2>&1
python - > gpurun_out/cli_time.txt 2>&1 <<'PY'
import os, subprocess, sys, time
for w in ('0', '1', '0', '1', '0', '1'):
    t = time.perf_counter()
    subprocess.run([sys.executable, '-m', 'telescope_amd', 'resume', 'tests/golden/resume_checkpoint.npz', '--outdir', '/tmp/o' + w],
                   env=dict(os.environ, TSEM_NO_WARM=w), capture_output=True)
    print('TSEM_NO_WARM=%s  `python -m telescope_amd resume` (bundled checkpoint) wall %.2f s' % (w, time.perf_counter() - t))
PY
cat gpurun_out/cli_time.txt
