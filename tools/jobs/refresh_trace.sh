# the bench command under rocprofv3 --kernel-trace --stats again (final tree), with the in-run kernel_ms of the same command beside it
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/prof/r05; rm -rf $O; mkdir -p $O
STAGES="trace" timeout 900 bash tools/profile.sh r05 > gpurun_out/profile_trace.log 2>&1
grep '^{"metric"' $O/trace.log | tail -1 > $O/bench.json
python tools/profile_summary.py $O r05x > gpurun_out/profile_trace_summary.log 2>&1
tail -3 gpurun_out/profile_trace_summary.log
find . -name "r05x_*" | head; for f in $(find . -name "r05x_fused_kernel_stats.txt"); do cp $f gpurun_out/r05x_fused_kernel_stats.txt; done
head -16 gpurun_out/r05x_fused_kernel_stats.txt
python -c "
import json; d=json.load(open('$O/bench.json')); print('same run: bench kernel_ms %.4f  ms_per_step %.4f  frac %.4f' % (d['roofline']['kernel_ms'], d['ms_per_step'], d['roofline']['frac']))"
rm -rf $O/trace
