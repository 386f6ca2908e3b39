set -x
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round3.py -x -q -m gpu -k "round5 or dry_run" > gpurun_out/r5_t1.log 2>&1; echo "rc=$?" >> gpurun_out/r5_t1.log
tail -5 gpurun_out/r5_t1.log
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-precision-sweep --no-reproducible-leg > gpurun_out/r5_bench_w3.json 2> gpurun_out/r5_bench_w3.err; tail -c 1500 gpurun_out/r5_bench_w3.json
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-precision-sweep --no-reproducible-leg --no-alt-layout > gpurun_out/r5_bench_w5.json 2> gpurun_out/r5_bench_w5.err
timeout 600 python bench.py --gpus 2 --one-device --rows 6000000 --steps 10 --warmup 2 --no-cpu-baseline --no-precision-sweep --no-reproducible-leg --no-alt-layout > gpurun_out/r5_bench_dry2.json 2> gpurun_out/r5_bench_dry2.err; tail -c 600 gpurun_out/r5_bench_dry2.err
