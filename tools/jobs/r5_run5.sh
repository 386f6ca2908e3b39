cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_round5.py -x -q -m gpu > gpurun_out/r5_t5.log 2>&1; echo "rc=$?" >> gpurun_out/r5_t5.log
tail -25 gpurun_out/r5_t5.log
timeout 600 python tools/time_report.py > gpurun_out/r5_time_report.txt 2>&1; grep -v amdgpu gpurun_out/r5_time_report.txt
timeout 600 python tools/time_report.py 50000000 18 > gpurun_out/r5_time_report18.txt 2>&1; grep -v amdgpu gpurun_out/r5_time_report18.txt | tail -4
