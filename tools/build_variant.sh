#!/bin/bash
# build a kernel-experiment variant of the library: tools/build_variant.sh <out.so> [-DFLAG=..]...   (teams of 4 only: TSEM_FAST_BUILD)
out=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -DTSEM_FAST_BUILD "$@" \
  -Iinclude -Itelescope_amd/csrc -o $out telescope_amd/csrc/tsem.hip -ldl -lpthread
