// Fused-kernel instantiations for teams of 5 members (one of eight such translation units compiled in parallel, telescope_amd/_lib.py).
#include "tsem_fused_inst.h"

fz_fn tsem_fz_kernel_p5(int P, int mode, int fmt, int geo) {
#ifdef TSEM_FAST_BUILD                                     // kernel experiments (tools/ab.sh): teams of 4 only
  (void)P; (void)mode; (void)fmt; (void)geo;
  return nullptr;
#else
  return P == 5 ? fz_pick<5>(mode, fmt, geo) : nullptr;
#endif
}
