// Instantiation of the fused kernel for the team sizes one translation unit is responsible for (included by tsem_fz_*.hip only).
#pragma once
#include <cstdint>
#include "tsem_common.h"
#include "tsem_device.h"
#include "tsem_fused.h"

typedef void (*fz_fn)(FusedArgs);
template <int P, int GEO> static fz_fn fz_pick2(int mode, int fmt) {
  if (mode == 4) {                                         // EM pass + the previous iteration's log-likelihood: needs the score table in LDS
#ifdef TSEM_NO_LAG
    return nullptr;
#else
    if constexpr (GEO != 3) { if (fmt == 1) return k_em_fused<P, 4, 1, GEO>; }   // (fp64 entries / geometry 3: the log1p finds no registers)
    return nullptr;
#endif
  }
  if (mode == 5 || mode == 7 || mode == 8) {               // split layout (parts of more than 7680 columns): teams of 5-8 only
#ifdef TSEM_NO_SPLIT
    return nullptr;
#else
    if constexpr (P > 4 && (GEO == 1 || GEO == 2)) {
      if (mode == 5) return fmt == 1 ? k_em_fused<P, 5, 1, GEO> : (fmt == 2 ? k_em_fused<P, 5, 2, GEO> : k_em_fused<P, 5, 0, GEO>);
      if (mode == 7) return fmt == 1 ? k_em_fused<P, 7, 1, GEO> : (fmt == 2 ? k_em_fused<P, 7, 2, GEO> : k_em_fused<P, 7, 0, GEO>);
      return fmt == 1 ? k_em_fused<P, 8, 1, GEO> : (fmt == 2 ? k_em_fused<P, 8, 2, GEO> : k_em_fused<P, 8, 0, GEO>);
    }
    return nullptr;
#endif
  }
  if (mode == 9) {                                         // the log-likelihood pass with log tables: needs the score table in LDS
    if (fmt == 1) return k_em_fused<P, 9, 1, GEO>;
    if (fmt == 2) return k_em_fused<P, 9, 2, GEO>;
    return nullptr;
  }
  if (mode >= 2) {                                         // exact (binned) column sums: needs the score table in LDS (formats 1, 2)
#ifdef TSEM_NO_REPRO
    return nullptr;
#else
    if (mode == 3) return fmt == 1 ? k_em_fused<P, 3, 1, GEO> : nullptr;   // both pieces in one pass: score codes only
    if (fmt == 1) return k_em_fused<P, 2, 1, GEO>;
    if (fmt == 2) return k_em_fused<P, 2, 2, GEO>;
    return nullptr;
#endif
  }
  if (fmt == 1) return mode ? k_em_fused<P, 1, 1, GEO> : k_em_fused<P, 0, 1, GEO>;
  if (fmt == 2) return mode ? k_em_fused<P, 1, 2, GEO> : k_em_fused<P, 0, 2, GEO>;
  return mode ? k_em_fused<P, 1, 0, GEO> : k_em_fused<P, 0, 0, GEO>;
}
template <int P> static fz_fn fz_pick(int mode, int fmt, int geo) {
  if constexpr (P > 4) return geo == 2 ? fz_pick2<P, 2>(mode, fmt) : fz_pick2<P, 1>(mode, fmt);   // teams of 5-8: 384 or 768 row slots
  else return geo == 3 ? fz_pick2<P, 3>(mode, fmt) : (geo == 2 ? fz_pick2<P, 2>(mode, fmt) : fz_pick2<P, 0>(mode, fmt));
}
