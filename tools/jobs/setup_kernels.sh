cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/prof_setup
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_setup -- python $GRAFT_REPO_ROOT/tools/time_setup.py > $GRAFT_REPO_ROOT/gpurun_out/time_setup_last.txt 2>&1 )
python tools/kernel_table.py gpurun_out/prof_setup > gpurun_out/setup_kernels_last.txt 2>&1
rm -rf gpurun_out/prof_setup
head -14 gpurun_out/setup_kernels_last.txt; grep -v "^{\|amdgpu" gpurun_out/time_setup_last.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "twin or golden or matches_reference or random_shapes or formats" 2>&1 | tail -3
