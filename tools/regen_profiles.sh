#!/bin/bash
# profiles/<prefix>_* from the output of tools/profile.sh (+ tools/sweep_short_r03.sh) under gpurun_out/prof/<tag>:
#   tools/regen_profiles.sh <tag> <prefix>          e.g.  tools/regen_profiles.sh r02 r02
# Files with hand-written header lines keep them (the first '#' lines of the tracked file).
set -u
O=gpurun_out/prof/${1:-r02}; P=profiles/${2:-r02}
hdr() { grep -m${2:-1} '^#' "$1" 2>/dev/null; }            # first n comment lines of the tracked file
clean() { grep -v 'amdgpu.ids' "$1"; }
python tools/profile_summary.py $O ${2:-r02} | tail -2
{ hdr ${P}_lds_counters.txt 3; grep -v '^# ' $O/pmc/summary.txt | grep -v amdgpu.ids; } > /tmp/_p && mv /tmp/_p ${P}_lds_counters.txt
{ echo "# tools/fused_prof.py 20000000 value_format=2 (2-byte score codes, row order)"; clean $O/timeline_code16.txt; echo
  echo "# tools/fused_prof.py 20000000 value_format=1 (fp64 entries, row order: the bench headline)"; clean $O/timeline_f64.txt; } > ${P}_fused_timeline.txt
clean $O/lds_ubench.log > ${P}_lds_ubench.log
clean $O/stream_ubench.log > ${P}_stream_ubench.log
{ hdr ${P}_comm_overhead.txt 3; cat $O/comm_overhead.txt; echo; cat $O/iter_timeline.txt; } > /tmp/_p && mv /tmp/_p ${P}_comm_overhead.txt
[ -f $O/sweep.txt ] && { hdr ${P}_sweep.txt 1; cat $O/sweep.txt; } > /tmp/_p && mv /tmp/_p ${P}_sweep.txt
[ -f $O/sweep_short.txt ] && { hdr ${P}_sweep_short.txt 1; cat $O/sweep_short.txt; } > /tmp/_p && mv /tmp/_p ${P}_sweep_short.txt
cp $O/bench.json ${P}_bench.json
python -c "
import json; d=json.load(open('${P}_bench.json')); json.dump(d['precision_sweep'], open('${P}_precision_sweep.json','w'), indent=1)"
echo "not regenerated here (hand-annotated): ${P}_time_e2e.txt ${P}_time_setup.txt ${P}_ab_exchange.txt ${P}_host_upload.txt ${P}_soak.txt"
