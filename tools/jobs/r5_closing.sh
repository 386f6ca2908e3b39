# closing measurements of round 5: set-up laps (plain + traced), end to end, full GPU suite, smoke, bench
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
TSEM_TRACE=1 timeout 300 python tools/time_setup.py 2>&1 | grep -v "^{\|amdgpu" > gpurun_out/setup_trace_last.txt
python tools/time_setup.py 2>&1 | grep -v "^{\|amdgpu" >> gpurun_out/setup_trace_last.txt
python tools/time_setup.py 2>&1 | grep -v "^{\|amdgpu" >> gpurun_out/setup_trace_last.txt
cat gpurun_out/setup_trace_last.txt
timeout 600 python tools/time_e2e.py 2>&1 | grep -v amdgpu > gpurun_out/time_e2e_last.txt; cat gpurun_out/time_e2e_last.txt
bash tools/jobs/gpu_suite_and_bench.sh
