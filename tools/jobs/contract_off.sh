cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
python tools/dbg_shard_tmp.py 2>&1 | grep "col-sum max\|columns off"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-precision-sweep --no-reproducible-leg 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('f64 step %.3f kernel %.3f | code16 step %.3f | whole_em_call %.2f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['code16_layout']['ms_per_step'], d['whole_em_call']['ms']))"
python tools/time_lnl_evolution.py rows=50000000 2>&1 | grep "after  20"
python tools/time_report.py 2>&1 | grep "kernel=1 cap=  0 wgs=2 wgs2=0\|codes only (thresh < 0) wgs2=0"
python tools/time_setup.py 2>&1 | grep "rowstats\|set_model"
python tools/time_use_likelihood_k50.py 2>&1 | grep -v amdgpu | tail -1
python - <<'PY'
import subprocess
print(subprocess.run('python tools/time_lnl.py 50000000 use_likelihood=1 2>&1 | grep -v amdgpu | tail -4', shell=True, capture_output=True, text=True).stdout)
PY
