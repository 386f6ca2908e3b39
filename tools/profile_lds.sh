#!/bin/bash
# LDS hardware counters of the fused EM kernel (both entry formats; the default bench run times the fp64
# layout and then the code16 layout).  Separate --pmc passes, kernel-trace only (no other trace domains).
#   tools/profile_lds.sh <tag> [bench args]
set -u
TAG=${1:-lds}; shift
OUT=gpurun_out/prof/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-precision-sweep --no-reproducible-leg $*"
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/lds -- $B > $OUT/lds.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/sq -- $B > $OUT/sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_ATOMIC_RETURN SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS --kernel-trace --output-format csv -d $OUT/lds2 -- $B > $OUT/lds2.log 2>&1
python tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
