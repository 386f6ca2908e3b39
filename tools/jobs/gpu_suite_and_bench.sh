cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 2700 python -m pytest tests/ -x -q -m gpu > gpurun_out/r5_gpu_full2.log 2>&1; echo "rc=$?" >> gpurun_out/r5_gpu_full2.log
tail -8 gpurun_out/r5_gpu_full2.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5_smoke.log 2>&1; tail -2 gpurun_out/r5_smoke.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r5_bench_final.log 2> gpurun_out/r5_bench_final.err; tail -1 gpurun_out/r5_bench_final.log > gpurun_out/r5_bench_final.json; tail -c 600 gpurun_out/r5_bench_final.json
