// Fused-kernel instantiations for teams of 4 members (one of six translation units compiled in parallel, telescope_amd/_lib.py).
#include "tsem_fused_inst.h"

fz_fn tsem_fz_kernel_p4(int P, int mode, int fmt, int geo) {
  switch (P) {
    case 4: return fz_pick<4>(mode, fmt, geo);
    default: return nullptr;
  }
}
