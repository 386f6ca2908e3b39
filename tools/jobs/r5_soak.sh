cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 2400 bash tools/soak.sh > gpurun_out/r5_soak.txt 2>&1; cat gpurun_out/r5_soak.txt
timeout 900 python tools/soak_lnl.py 2>&1 | grep -v amdgpu > gpurun_out/r5_soak_lnl.txt; tail -12 gpurun_out/r5_soak_lnl.txt
