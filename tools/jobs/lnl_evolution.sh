cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
{ for args in "rows=50000000" "rows=50000000 value_format=1" "rows=10000000 cols=50000 nnz_row=100" "rows=20000000 nnz_row=18"; do echo "## python tools/time_lnl_evolution.py $args"; python tools/time_lnl_evolution.py $args 2>&1 | grep -v amdgpu; done; echo "## python tools/time_use_likelihood_k50.py   (tl.em(use_likelihood=True), 10M x 50k x ~100: the lnl pass runs every iteration)"; python tools/time_use_likelihood_k50.py 2>&1 | grep -v amdgpu | tail -2; } > gpurun_out/lnl_evolution_last.txt 2>&1
cat gpurun_out/lnl_evolution_last.txt
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/suite_sel.log 2>&1; grep -n "passed\|failed" gpurun_out/suite_sel.log | tail -3
