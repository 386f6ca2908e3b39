#!/bin/bash
# build a kernel-experiment variant of the library: tools/build_variant.sh <out.so> [-DFLAG=..]...   (teams of 4 only: TSEM_FAST_BUILD)
out=$1; shift
python - "$out" "$@" <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from telescope_amd import _lib
_lib.build_library(force=True, extra_flags=['-DTSEM_FAST_BUILD'] + sys.argv[2:], out=os.path.abspath(sys.argv[1]))
PY
