// Fused-kernel instantiations for teams of 2 members (one of six translation units compiled in parallel, telescope_amd/_lib.py).
#include "tsem_fused_inst.h"

fz_fn tsem_fz_kernel_p2(int P, int mode, int fmt, int geo) {
#ifdef TSEM_FAST_BUILD                                     // kernel experiments (tools/ab.sh): teams of 4 only
  (void)P; (void)mode; (void)fmt; (void)geo;
  return nullptr;
#else
  switch (P) {
    case 2: return fz_pick<2>(mode, fmt, geo);
    default: return nullptr;
  }
#endif
}
