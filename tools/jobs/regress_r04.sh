# the tools behind profiles/r04_large_k.txt, r04_group_sums.txt and r04_use_likelihood.txt on the current tree: a regression check of round 5 against round 4's numbers
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
{ echo "## python tools/time_large_k.py"; timeout 900 python tools/time_large_k.py 2>&1 | grep -v "amdgpu\|WARNING"; echo "## python tools/time_groups.py"; timeout 600 python tools/time_groups.py 2>&1 | grep -v "amdgpu\|WARNING"; } > gpurun_out/regress_r04_last.txt 2>&1
cat gpurun_out/regress_r04_last.txt
