// libtelescope_em.so, report unit: passes over the canonical CSR after (or outside) the EM loop — z export, best hits,
// reassign x 6 (model.py:808-865), the streaming report pass (conf | exclude | average of one z, model.py:432-457),
// per-barcode sums (model.py:611-625), and the public estep / mstep / calculate_lnl on caller-supplied arrays.
#include "tsem_internal.h"
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include "tsem_npsum.h"
#include <atomic>

// ---- near-ties: rows whose integer outputs hang on the last bits of 1 / rowsum ---------------------------------------------------
// The masks compare z_j = fl(n_j r), r = fl(1 / rowsum), with each other (`==`, sparse_plus.py:125) and with conf_prob (`>=`,
// model.py:851).  The kernels below add a row's numerators in another order than scipy does (np.add.reduceat: tsem_npsum.h), so their
// r may differ from the reference's in its last bits.  That can change a mask only where two DIFFERENT numerators lie within a few
// ulp of each other at the row's maximum, or a z value lies that close to the threshold: with every other numerator more than
// `band` below the largest one, the best hits are the entries EQUAL to it whatever the last bits of r are.  Rows inside the band — a
// handful per matrix, none on most — get their sum in scipy's order from ONE lane (np_row_sum) and are counted (tsem_layout_info[31]).
// The band covers the distance between any two orders of adding up to `len` positive terms (2 len 2^-53) with room to spare.
constexpr double TS_NEAR_BAND = 0x1p-42;
__device__ __forceinline__ double near_band(int64_t len) { return fmax(TS_NEAR_BAND, (double)len * 0x1p-50); }
// the m terms of z's pattern in a row — every stored entry (`all`, the initial z) or the non-zero products (model.py:713-714 drops
// exact zeros before the sum) — added like np.add.reduceat adds them.  Term of the row's k-th stored entry: lut[raw[s + k]], times
// tab[column] unless tab is null (the initial z); the column is a CSR column id (col32) or a popularity id + toff (col16).
// NOT inlined: one lane runs it for a handful of rows, and its accumulators and stack must stay out of the row kernels' registers
// (inlined, k_rowpass went from 49 to 128 VGPRs + 1.5 KB of scratch).
struct NpRow {
  const double* lut; const uint16_t* raw; const int32_t* col32; const uint16_t* col16; const double* tab; uint32_t toff; int64_t s;
};
__device__ double np_row_sum(const NpRow& a, int64_t m, bool all) {
  struct Cur {
    const NpRow& a; int64_t k; bool all;
    __device__ double operator()() {
      for (;;) {
        const int64_t e = a.s + k++;
        double v = a.lut[a.raw[e]];
        if (a.tab) v = v * a.tab[a.col32 ? (uint32_t)a.col32[e] : (uint32_t)a.col16[e] + a.toff];
        if (all || v != 0.0) return v;
      }
    }
  };
  Cur c{a, 0, all};
  return m > 0 ? np_reduceat_sum(c, m) : 0.0;
}

// ============================================================================
// CSR row passes: z export, best-hit counts, reassign (model.py:808-865)
// ============================================================================
enum { RP_EXPORT_Z = 0, RP_BEST = 1, RP_REASSIGN = 2, RP_REPORT = 3 };

// Option "reproducible": values in [0, 2) — posteriors, shares of a tie — cut into a multiple of 2^-26 and the rest on the 2^-53
// grid: up to 2^26 of either add exactly in fp64, whatever order the atomics are served in; the two sums are added once at the end.
__device__ __forceinline__ void exact_split01(double v, double& hi, double& lo) {
  const double m1 = 100663296.0;                            // 1.5 * 2^26: ulp 2^-26
  hi = (v + m1) - m1;
  lo = ((v - hi) + 0.75) - 0.75;                            // ulp(0.75) = 2^-53
}
__global__ void k_check_offsets(int64_t n, const int32_t* __restrict__ rows, const int64_t* __restrict__ off, const int64_t* __restrict__ indptr,
                                uint32_t* __restrict__ bad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && off[i + 1] - off[i] != indptr[rows[i] + 1] - indptr[rows[i]]) *bad = 1u;
}
__global__ void k_add_lo(int64_t n, double* __restrict__ a, const double* __restrict__ lo) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] += lo[i];
}

struct RowPassArgs {
  int64_t N;
  int32_t K;
  const int64_t* indptr;
  const int32_t* indices;
  const uint16_t* raw;
  const double* lut;
  const double* pi;      // null => initial (c == 1)
  const double* theta;
  const double* zin;     // non-null: the caller's z (TSEM_Z_USER), aligned to the CSR pattern, NaN = no entry; used as is
  const double* cnat;    // pi[j] * theta[j] per column (natural order): ONE gather per entry of an ambiguous row instead of two
  int lut_len;           // the score table is staged in LDS ([lut_len] doubles in front of the hot slots)
  int method;
  double thresh;
  const int32_t* picks;
  double* zout;          // EXPORT_Z / mask
  int32_t* nbest;
  double* colsums;
  const int32_t* group;  // REASSIGN: optional row -> group map; colsums is then [g1 - g0][K], rows of other groups (or -1) are skipped
  int32_t g0 = 0, g1 = 0x7FFFFFFF;
  const int32_t* rowlist; int64_t nlist;   // REASSIGN / EXPORT_Z: optional list of rows to visit (picks[] is then indexed by list position)
  const int64_t* out_off = nullptr;        // with a row list: zout is COMPACT — the entries of list row i go to zout[out_off[i] ...] (tsem_rows_lookup)
  // REPORT: conf, exclude and average in ONE pass -> colsums[0..K), [K..2K), [2K..3K); best-hit counts -> nbest
  // REASSIGN without groups: the Hs most popular slots of every column part are summed in LDS per
  // workgroup and flushed once (global fp64 atomics: 22 G/s, 2 G/s on a popular column)
  const uint32_t* colmap; const int32_t* col_of_pc; int P, Kp, Hs;
  double* colsums_lo = nullptr;   // option "reproducible": the low pieces of every value (same shape as colsums; no LDS slots then)
  unsigned long long* exact_n = nullptr;   // counts the rows redone in the reference's order of additions (near-ties)
  // near-ties of a pass (see near_band): the pass proper (FIX = false) sets bit `idx` of flag_bits for a visited row it must not
  // decide with its own row sum and skips it; the same kernel with FIX = true then visits the flagged rows only, with the sum in the
  // reference's order.  Two kernels because that sum (one lane, eight accumulators, a stack for long rows) would cost the pass
  // proper its occupancy: 49 -> 128 VGPRs when it is inlined or called.
  uint32_t* flag_bits = nullptr;           // [ceil(n_visit / 512) * 16] words
  unsigned long long* flag_n = nullptr;    // [1] rows flagged by this pass
};

// METH >= 0 fixes the reassign method at compile time (the per-entry switch and the reductions a method does not
// need disappear: the pass is bound by instruction issue, ~300 per four rows); METH = -1 reads it from the arguments.
template <int MODE, int METH = -1, bool FIX = false>
__global__ __launch_bounds__(1024) void k_rowpass(RowPassArgs A) {
  const int method = METH >= 0 ? METH : A.method;
  extern __shared__ double rp_lds[];                       // [lut_len] score table | [P][Hs] hot slots (REASSIGN with A.Hs > 0)
  double* const lutS = rp_lds;
  double* const hot = rp_lds + A.lut_len;
  const int sub = threadIdx.x / RP_SUB, lane = threadIdx.x % RP_SUB, subs = blockDim.x / RP_SUB;
  const bool initial = (A.pi == nullptr);
  const int nhot1 = (MODE == RP_REASSIGN || MODE == RP_REPORT) ? A.P * A.Hs : 0;
  const int nhot = MODE == RP_REPORT ? 3 * nhot1 : nhot1;
  // The pass is bound by vector-memory INSTRUCTIONS (every gather touches 64 cache lines): the score table
  // comes from LDS and pi*theta from one precomputed table, 3 instead of 5 vector-memory instructions per round
  for (int t = threadIdx.x; t < A.lut_len; t += blockDim.x) lutS[t] = A.lut[t];
  for (int t = threadIdx.x; t < nhot; t += blockDim.x) hot[t] = 0.0;
  __syncthreads();
  const bool listed = (MODE == RP_REASSIGN || MODE == RP_EXPORT_Z) && A.rowlist;
  const int64_t n_visit = listed ? A.nlist : A.N;
  // FIX: the outer loop walks the flag words, 16 per group and step (one per lane), the inner loops their set bits
  const int64_t n_outer = FIX ? ((*A.flag_n != 0ull) ? (n_visit + 511) / 512 : 0) : n_visit;
  for (int64_t o = (int64_t)blockIdx.x * subs + sub; o < n_outer; o += (int64_t)gridDim.x * subs) {
   const uint32_t my_word = FIX ? A.flag_bits[o * 16 + lane] : 0u;
   uint32_t lanes = FIX ? (uint32_t)((__ballot(my_word != 0u) >> ((threadIdx.x & 63) / RP_SUB * RP_SUB)) & 0xFFFFull) : 1u;
   while (lanes) {
    const int wl = __ffs((int)lanes) - 1;
    lanes &= lanes - 1u;
    uint32_t bits = FIX ? (uint32_t)__shfl((int)my_word, wl, RP_SUB) : 1u;
   while (bits) {
    const int64_t idx = FIX ? (o * 16 + wl) * 32 + (__ffs((int)bits) - 1) : o;
    bits &= bits - 1u;
    const int64_t row = listed ? (int64_t)A.rowlist[idx] : idx;
    const int64_t s = A.indptr[row], e = A.indptr[row + 1];
    const int64_t zo = (listed && A.out_off) ? A.out_off[idx] - s : 0;      // where entry k of this row goes in zout: k + zo
    const bool amb = (e - s) > 1;
    // one value of report column m (0 for a plain reassign) for column `col`: popular columns in LDS, the rest global
    auto emit = [&](int m, int col, uint32_t cm, double val, int64_t grp_off) {
      if (A.colsums_lo) {
        double hi, lo;
        exact_split01(val, hi, lo);
        unsafeAtomicAdd(&A.colsums[(int64_t)m * A.K + grp_off + col], hi);
        if (lo != 0.0) unsafeAtomicAdd(&A.colsums_lo[(int64_t)m * A.K + grp_off + col], lo);
      } else if (nhot1 && (int)(cm & CM_SM) < A.Hs) lds_add(&hot[m * nhot1 + (cm >> CM_PS) * A.Hs + (cm & CM_SM)], val);
      else unsafeAtomicAdd(&A.colsums[(int64_t)m * A.K + grp_off + col], val);
    };
    auto numer = [&](int64_t k) -> double {
      if (A.zin) return A.zin[k];
      double q = A.lut_len ? lutS[A.raw[k]] : A.lut[A.raw[k]];
      if (initial) return q;
      int col = A.indices[k];
      double c = amb ? A.cnat[col] : A.pi[col];              // cnat[col] = pi[col] * theta[col]: the same product, formed once per column
      return q * c;
    };
    if (e - s <= 4 * RP_SUB) {
      // Rows of up to 64 entries (all of them, for alignment data): the numerators are computed once
      // and stay in registers for the row sum, the row maximum, the tie count and the output value.
      double n[4]; bool vld[4], inp[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t k = s + lane + i * RP_SUB;
        vld[i] = k < e;
        n[i] = vld[i] ? numer(k) : 0.0;
        inp[i] = vld[i] && (A.zin ? !isnan(n[i]) : (initial || n[i] != 0.0));
        if (A.zin && !inp[i]) n[i] = 0.0;
      }
      // same summation order as the long-row path below: lane-strided partial sums, then across lanes
      const double rs = recip0(sg_sum<RP_SUB>(((n[0] + n[1]) + n[2]) + n[3]));
      double r = A.zin ? 1.0 : rs;                          // the caller's z is used as is (model.py:837)
      int cnt = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) cnt += inp[i] ? 1 : 0;
      cnt = sg_sum_i<RP_SUB>(cnt);
      if (FIX) {                                            // a near-tie: the row sum in the reference's order (see near_band)
        double ex = 0.0;
        if (lane == 0) {
          const NpRow nr{A.lut, A.raw, A.indices, nullptr, initial ? nullptr : (amb ? A.cnat : A.pi), 0u, s};
          ex = np_row_sum(nr, cnt, initial);
          if (A.exact_n) atomicAdd(A.exact_n, 1ull);
        }
        r = recip0(__shfl(ex, 0, RP_SUB));
      } else if (MODE != RP_EXPORT_Z && !A.zin && A.flag_bits) {   // is it one?  Then it is left to the FIX launch
        double nmx = -1.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) if (inp[i]) nmx = fmax(nmx, n[i]);
        nmx = sg_max<RP_SUB>(nmx);
        const bool uses_thresh = MODE == RP_REPORT || method == TSEM_RA_CONF;
        const double lo = nmx * (1.0 - TS_NEAR_BAND), tb = A.thresh * TS_NEAR_BAND;
        int nf = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (inp[i] && ((n[i] >= lo && n[i] != nmx) || (uses_thresh && fabs(n[i] * r - A.thresh) <= tb))) nf = 1;
        if (sg_max_i<RP_SUB>(nf)) {
          if (lane == 0) { atomicOr(&A.flag_bits[idx >> 5], 1u << (idx & 31)); atomicAdd(A.flag_n, 1ull); }
          continue;
        }
      }
      double zmax = -1.0;
#pragma unroll
      for (int i = 0; i < 4; ++i) if (inp[i]) zmax = fmax(zmax, n[i] * r);
      zmax = sg_max<RP_SUB>(zmax);
      if (MODE == RP_EXPORT_Z) {
#pragma unroll
        for (int i = 0; i < 4; ++i) if (vld[i]) A.zout[s + lane + i * RP_SUB + zo] = inp[i] ? n[i] * r : -1.0;
        continue;
      }
      int nb = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) nb += (inp[i] && (n[i] * r) == zmax) ? 1 : 0;
      nb = sg_sum_i<RP_SUB>(nb);
      if (MODE == RP_BEST) {
        if (lane == 0) A.nbest[row] = cnt ? nb : 0;
        continue;
      }
      double vsum = 0.0;
      if (method == TSEM_RA_CONF || MODE == RP_REPORT) {
#pragma unroll
        for (int i = 0; i < 4; ++i) if (inp[i] && n[i] * r >= A.thresh) vsum += n[i] * r;
        vsum = sg_sum<RP_SUB>(vsum);
      }
      if (MODE == RP_REPORT) {                              // conf | exclude | average of model.py:839-856 from one set of numerators
        if (lane == 0 && A.nbest) A.nbest[row] = cnt ? nb : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const double z = n[i] * r;
          const bool best = inp[i] && (z == zmax);
          const double vc = (inp[i] && z >= A.thresh) ? z * recip0(vsum) : 0.0;
          if (vld[i] && (best || vc != 0.0)) {
            const int col = A.indices[s + lane + i * RP_SUB];
            const uint32_t cm = nhot1 ? A.colmap[col] : 0xFFFFFFFFu;
            if (vc != 0.0) emit(0, col, cm, vc, 0);
            if (best && nb == 1) emit(1, col, cm, 1.0, 0);
            if (best) emit(2, col, cm, 1.0 * recip0((double)nb), 0);
          }
        }
        continue;
      }
      const int pick = (method == TSEM_RA_CHOOSE && A.picks && nb > 1) ? A.picks[A.rowlist ? idx : row] : 0;
      const int64_t grp_off = A.group ? ((A.group[row] < A.g0 || A.group[row] >= A.g1) ? -1 : (int64_t)(A.group[row] - A.g0) * A.K) : 0;
      int base = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t k = s + lane + i * RP_SUB;
        const double z = n[i] * r;
        const bool best = inp[i] && (z == zmax);
        const unsigned long long bal = __ballot(best);
        const unsigned grp = (unsigned)((bal >> ((threadIdx.x & 63) / RP_SUB * RP_SUB)) & 0xFFFFull);
        const int ord = base + __popc(grp & ((1u << lane) - 1u));
        base += __popc(grp);
        double val = 0.0;
        switch (method) {
          case TSEM_RA_EXCLUDE: val = (best && nb == 1) ? 1.0 : 0.0; break;
          case TSEM_RA_CHOOSE:  val = (best && ord == pick) ? 1.0 : 0.0; break;
          case TSEM_RA_AVERAGE: val = best ? 1.0 * recip0((double)nb) : 0.0; break;
          case TSEM_RA_CONF:    val = (inp[i] && z >= A.thresh) ? z * recip0(vsum) : 0.0; break;
          case TSEM_RA_UNIQUE:  val = (inp[i] && !amb) ? ceil(z) : 0.0; break;
          case TSEM_RA_ALL:     val = (inp[i] && z > 0.0) ? 1.0 : 0.0; break;
        }
        if (vld[i]) {
          if (A.zout) A.zout[k + zo] = val;
          if (val != 0.0 && grp_off >= 0 && A.colsums) {
            const int col = A.indices[k];
            emit(0, col, nhot1 ? A.colmap[col] : 0xFFFFFFFFu, val, grp_off);
          }
        }
      }
      continue;
    }
    // sweep 1: row sum (and the largest numerator of z's pattern)
    double y = 0.0, nmx = -1.0;
    int cnt = 0;
    for (int64_t k = s + lane; k < e; k += RP_SUB) {
      const double v = numer(k);
      y += (A.zin && isnan(v)) ? 0.0 : v;
      if (A.zin ? !isnan(v) : (initial || v != 0.0)) { nmx = fmax(nmx, v); ++cnt; }
    }
    y = sg_sum<RP_SUB>(y);
    nmx = sg_max<RP_SUB>(nmx);
    cnt = sg_sum_i<RP_SUB>(cnt);
    double r = A.zin ? 1.0 : recip0(y);
    if (FIX) {                                              // a near-tie: the row sum in the reference's order (see near_band)
      double ex = 0.0;
      if (lane == 0) {
        const NpRow nr{A.lut, A.raw, A.indices, nullptr, initial ? nullptr : (amb ? A.cnat : A.pi), 0u, s};
        ex = np_row_sum(nr, cnt, initial);
        if (A.exact_n) atomicAdd(A.exact_n, 1ull);
      }
      r = recip0(__shfl(ex, 0, RP_SUB));
    } else if (MODE != RP_EXPORT_Z && !A.zin && A.flag_bits) {     // is it one?  Then it is left to the FIX launch
      const bool uses_thresh = MODE == RP_REPORT || method == TSEM_RA_CONF;
      const double band = near_band(e - s), lo = nmx * (1.0 - band), tb = A.thresh * band;
      int nf = 0;
      for (int64_t k = s + lane; k < e; k += RP_SUB) {
        const double v = numer(k);
        if ((initial || v != 0.0) && ((v >= lo && v != nmx) || (uses_thresh && fabs(v * r - A.thresh) <= tb))) nf = 1;
      }
      if (sg_max_i<RP_SUB>(nf)) {
        if (lane == 0) { atomicOr(&A.flag_bits[idx >> 5], 1u << (idx & 31)); atomicAdd(A.flag_n, 1ull); }
        continue;
      }
    }
    // sweep 2: row max over z's pattern
    double zmax = -1.0;
    for (int64_t k = s + lane; k < e; k += RP_SUB) {
      double n = numer(k);
      bool inpat = A.zin ? !isnan(n) : (initial || (n != 0.0));
      if (inpat) zmax = fmax(zmax, n * r);
    }
    zmax = sg_max<RP_SUB>(zmax);
    if (MODE == RP_EXPORT_Z) {
      for (int64_t k = s + lane; k < e; k += RP_SUB) {
        double n = numer(k);
        bool inpat = A.zin ? !isnan(n) : (initial || (n != 0.0));
        A.zout[k + zo] = inpat ? n * r : -1.0;   // -1 marks an entry the reference drops from z's pattern
      }
      continue;
    }
    // sweep 3: number of best hits (binmax, sparse_plus.py:117-129)
    int nb = 0;
    for (int64_t k = s + lane; k < e; k += RP_SUB) {
      double n = numer(k);
      bool inpat = A.zin ? !isnan(n) : (initial || (n != 0.0));
      if (inpat && (n * r) == zmax) ++nb;
    }
    nb = sg_sum_i<RP_SUB>(nb);
    if (MODE == RP_BEST) {
      if (lane == 0) A.nbest[row] = cnt ? nb : 0;
      continue;
    }
    // ---- reassign ----
    double vsum = 0.0;
    if (method == TSEM_RA_CONF || MODE == RP_REPORT) {
      for (int64_t k = s + lane; k < e; k += RP_SUB) {
        double n = numer(k);
        double z = n * r;
        if ((A.zin ? !isnan(n) : (initial || n != 0.0)) && z >= A.thresh) vsum += z;
      }
      vsum = sg_sum<RP_SUB>(vsum);
    }
    if (MODE == RP_REPORT) {
      if (lane == 0 && A.nbest) A.nbest[row] = cnt ? nb : 0;
      for (int64_t k = s + lane; k < e; k += RP_SUB) {
        const double n = numer(k);
        const bool inpat = A.zin ? !isnan(n) : (initial || n != 0.0);
        const double z = n * r;
        const bool best = inpat && (z == zmax);
        const double vc = (inpat && z >= A.thresh) ? z * recip0(vsum) : 0.0;
        if (best || vc != 0.0) {
          const int col = A.indices[k];
          const uint32_t cm = nhot1 ? A.colmap[col] : 0xFFFFFFFFu;
          if (vc != 0.0) emit(0, col, cm, vc, 0);
          if (best && nb == 1) emit(1, col, cm, 1.0, 0);
          if (best) emit(2, col, cm, 1.0 * recip0((double)nb), 0);
        }
      }
      continue;
    }
    const int pick = (method == TSEM_RA_CHOOSE && A.picks && nb > 1) ? A.picks[A.rowlist ? idx : row] : 0;
    const int64_t grp_off = A.group ? ((A.group[row] < A.g0 || A.group[row] >= A.g1) ? -1 : (int64_t)(A.group[row] - A.g0) * A.K) : 0;
    int base = 0;
    for (int64_t k0 = s; k0 < e; k0 += RP_SUB) {
      int64_t k = k0 + lane;
      bool valid = k < e;
      double n = valid ? numer(k) : 0.0;
      bool inpat = valid && (A.zin ? !isnan(n) : (initial || n != 0.0));
      double z = n * r;
      bool best = inpat && (z == zmax);
      unsigned long long bal = __ballot(best);
      unsigned grp = (unsigned)((bal >> ((threadIdx.x & 63) / RP_SUB * RP_SUB)) & 0xFFFFull);
      int ord = base + __popc(grp & ((1u << lane) - 1u));
      base += __popc(grp);
      double val = 0.0;
      switch (method) {
        case TSEM_RA_EXCLUDE: val = (best && nb == 1) ? 1.0 : 0.0; break;
        case TSEM_RA_CHOOSE:  val = (best && ord == pick) ? 1.0 : 0.0; break;
        case TSEM_RA_AVERAGE: val = best ? 1.0 * recip0((double)nb) : 0.0; break;
        case TSEM_RA_CONF:    val = (inpat && z >= A.thresh) ? z * recip0(vsum) : 0.0; break;
        case TSEM_RA_UNIQUE:  val = (inpat && !amb) ? ceil(z) : 0.0; break;
        case TSEM_RA_ALL:     val = (inpat && z > 0.0) ? 1.0 : 0.0; break;
      }
      if (valid) {
        if (A.zout) A.zout[k + zo] = val;
        if (val != 0.0 && grp_off >= 0 && A.colsums) {
          const int col = A.indices[k];
          emit(0, col, nhot1 ? A.colmap[col] : 0xFFFFFFFFu, val, grp_off);
        }
      }
    }
   }
   }
  }
  if (nhot) {
    __syncthreads();
    for (int t = threadIdx.x; t < nhot; t += blockDim.x) {
      const double v = hot[t];
      const int m = t / nhot1, tt = t % nhot1;
      if (v != 0.0) unsafeAtomicAdd(&A.colsums[(int64_t)m * A.K + A.col_of_pc[(tt / A.Hs) * A.Kp + tt % A.Hs]], v);
    }
  }
}

// ---- the report pass: conf | exclude | average of ONE z in one pass (model.py:432-457) ----------------------------
// Round 2's RP_REPORT ran at 0.07 of the HBM peak (20 ms for 11.8 GB at 50M x 40, profiles/r02_report_kernel_stats.txt):
// 16 lanes per row, one 4-byte + one 2-byte load per lane and sweep, every load behind the row pointers it depends on,
// nothing in flight while a row is computed, ~700 instructions per four rows, one pi*theta gather per entry from a
// 240 KB table in L2 (the vector cache takes ONE gather address per clock and CU: 2e9 gathers = 4 ms by themselves,
// measured: the same pass without them 3.3 ms), global atomics on popular columns.  What this kernel changes:
//   * it reads a 2-byte POPULARITY ID per entry (rid16, written while the layout is built: id = slot * P + part of the
//     column's place in the blocked layout, so small ids are popular columns) and the 2-byte score code: 4 B per entry
//     instead of 6, and the id indexes LDS tables directly — pi*theta of the HC most popular columns sits in LDS, only
//     the cold tail is gathered from L2; the winner's scatter needs no column-map lookup; everything is accumulated per
//     id and mapped back to columns by k_report_finish;
//   * a lane holds E = 16 (or 8) CONSECUTIVE entries of its row, a row takes G = 1 .. 16 lanes (capacity G x E, chosen
//     from the row-length histogram so that < 0.5 % of the rows overflow): the per-row work — butterflies, row
//     pointers, the winner's scatter, loop control — is paid once per 64 / G rows of a wave, the per-entry work is a
//     dozen instructions with no cross-lane step;
//   * row pointers are fetched two iterations ahead and the entries one iteration ahead of the row they belong to,
//     unconditionally (the entry arrays carry padding, rows past the end are clamped), so the next rows' loads are in
//     flight while a row is reduced; the group reductions are DPP butterflies on the VALU;
//   * a row has ONE winner in all but the tied rows: with conf_prob > 0.5 the entry with z >= conf_prob, if any, is the
//     unique best hit.  The lane that holds it does one 32-bit LDS counter increment (`exclude` and the `average` of
//     rows with one best hit are the same count) and one fp64 LDS add (`conf`); two-way ties (most of the 11 % tied rows of
//     the initial z) increment a second counter, average = n1 + n2 / 2 + the shares of the wider ties.  Ids beyond the LDS
//     slots use global atomics.  conf_prob <= 0.5 takes the general per-entry path.
// Rows longer than G x E entries are appended to a list and reduced by k_report_slow afterwards (same arithmetic); a
// caller-assigned z, score tables that do not fit LDS and layouts with more than 65536 slots stay on k_rowpass.  Integer
// outputs are exact; floats differ by summation order only.
struct ReportArgs {
  int64_t N, nnz;
  int32_t K, IDN;                      // ids 0 .. IDN-1 (= P * Kp)
  const int64_t* indptr;
  const uint16_t* rid;                 // [nnz + TS_ENTRY_PAD] popularity id of every entry's column
  const uint16_t* raw;                 // [nnz + TS_ENTRY_PAD] score codes
  const double* lut; int lut_len;      // staged in LDS (0 < lut_len <= 2048)
  const double* cnat2;                 // [2 IDN] by id: pi*theta | pi (ambiguous rows use the first half, unique rows the second); null => initial z
  double thresh;
  int32_t* nbest;                      // [N] number of best hits per row (0: empty pattern)
  double *g_conf, *g_n1, *g_n2, *g_avgt;   // [IDN] each, by id
  double *g_conf_lo = nullptr, *g_avgt_lo = nullptr;   // option "reproducible": low pieces (exact_split01); LDS then holds [Hs] more doubles
  int HC, Hs;                          // LDS slots: pi*theta of ids < HC; accumulators of ids < Hs
  int32_t* defer_rows; unsigned long long* defer_n;   // rows left to k_report_slow: too long for the lanes of a row, or (stored as ~row) a near-tie,
                                                      // whose sum k_report_slow forms in the reference's order of additions (near_band)
  unsigned long long* exact_n = nullptr;              // counts the latter
  int dbg;                             // timing experiments (wrong results): 1 drop the emits that miss the LDS slots, 2 the row-count stores, 4 the ties
  // per-GROUP sums (GM != 0; the per-barcode count matrix of scTelescope.output_report, model.py:611-625): rows of the groups
  // [g0, g1) add into a tile [g1 - g0][IDN] in HBM, by id — 32-bit counters for `exclude` (GM 1), doubles for `average` (GM 2) and
  // `conf` (GM 3); other rows are skipped (their length is taken as 0).  No LDS accumulators: (group, column) pairs do not repeat
  // within a workgroup the way columns do.
  const int32_t* group = nullptr; int32_t g0 = 0, g1 = 0;
  uint32_t* t_cnt = nullptr; double* t_val = nullptr;
};
// 16- / 8-byte loads at the natural alignment of their ELEMENTS (a row starts at any entry): plain vector types with
// a reduced alignment, so the compiler emits one global_load_dwordx4 / dwordx2 (the target allows unaligned access)
typedef unsigned int rr_u32x4_a2 __attribute__((ext_vector_type(4), aligned(2)));
typedef long long rr_i64x2_a8 __attribute__((ext_vector_type(2), aligned(8)));

// butterflies over aligned groups of G = 1 .. 16 lanes on the VALU: after the two quad permutes every lane of a quad
// holds the quad's total, the half-row mirror (lane i <-> 7 - i) then pairs the quads, the row mirror (i <-> 15 - i) the
// halves.  Every lane of a group must be active.
constexpr int RR_QP_X1 = 0xB1, RR_QP_X2 = 0x4E, RR_HALF_MIRROR = 0x141, RR_MIRROR = 0x140;
template <int G> __device__ __forceinline__ double rr_sum(double v) {
  if (G >= 2) v += fz_dpp_d<RR_QP_X1, 0xF>(v, v);
  if (G >= 4) v += fz_dpp_d<RR_QP_X2, 0xF>(v, v);
  if (G >= 8) v += fz_dpp_d<RR_HALF_MIRROR, 0xF>(v, v);
  if (G >= 16) v += fz_dpp_d<RR_MIRROR, 0xF>(v, v);
  return v;
}
template <int G> __device__ __forceinline__ double rr_max(double v) {
  if (G >= 2) v = fmax(v, fz_dpp_d<RR_QP_X1, 0xF>(v, v));
  if (G >= 4) v = fmax(v, fz_dpp_d<RR_QP_X2, 0xF>(v, v));
  if (G >= 8) v = fmax(v, fz_dpp_d<RR_HALF_MIRROR, 0xF>(v, v));
  if (G >= 16) v = fmax(v, fz_dpp_d<RR_MIRROR, 0xF>(v, v));
  return v;
}
template <int G> __device__ __forceinline__ int rr_max_i(int v) {
  if (G >= 2) v = max(v, fz_dpp_i<RR_QP_X1, 0xF>(v, v));
  if (G >= 4) v = max(v, fz_dpp_i<RR_QP_X2, 0xF>(v, v));
  if (G >= 8) v = max(v, fz_dpp_i<RR_HALF_MIRROR, 0xF>(v, v));
  if (G >= 16) v = max(v, fz_dpp_i<RR_MIRROR, 0xF>(v, v));
  return v;
}
template <int G> __device__ __forceinline__ int rr_sum_i(int v) {
  if (G >= 2) v += fz_dpp_i<RR_QP_X1, 0xF>(v, v);
  if (G >= 4) v += fz_dpp_i<RR_QP_X2, 0xF>(v, v);
  if (G >= 8) v += fz_dpp_i<RR_HALF_MIRROR, 0xF>(v, v);
  if (G >= 16) v += fz_dpp_i<RR_MIRROR, 0xF>(v, v);
  return v;
}

template <int GM>
struct ReportEmit {                                        // where a row's values go (both report kernels), by id; GM != 0: into the row's group (goff)
  const ReportArgs& A; double* hotF; uint32_t* hot1; uint32_t* hot2; int Hs; double* hotL;
  __device__ __forceinline__ void conf(uint32_t id, double v, int64_t goff) const {
    if (GM != 0) { if (GM == 3) unsafeAtomicAdd(&A.t_val[goff + id], v); return; }
    if (A.g_conf_lo) {
      double hi, lo;
      exact_split01(v, hi, lo);
      if ((int)id < Hs) { lds_add(&hotF[id], hi); if (lo != 0.0) lds_add(&hotL[id], lo); }
      else { unsafeAtomicAdd(&A.g_conf[id], hi); if (lo != 0.0) unsafeAtomicAdd(&A.g_conf_lo[id], lo); }
      return;
    }
    if ((int)id < Hs) lds_add(&hotF[id], v);
    else if (!(A.dbg & 1)) unsafeAtomicAdd(&A.g_conf[id], v);
  }
  __device__ __forceinline__ void one(uint32_t id, int64_t goff) const {           // the row's only best hit
    if (GM != 0) {
      if (GM == 1) atomicAdd(&A.t_cnt[goff + id], 1u);
      if (GM == 2) unsafeAtomicAdd(&A.t_val[goff + id], 1.0);
      return;
    }
    if ((int)id < Hs) atomicAdd(&hot1[id], 1u);
    else if (!(A.dbg & 1)) unsafeAtomicAdd(&A.g_n1[id], 1.0);
  }
  __device__ __forceinline__ void tie(uint32_t id, int nb, double share, int64_t goff) const {   // one of nb > 1 best hits
    if (GM != 0) { if (GM == 2) unsafeAtomicAdd(&A.t_val[goff + id], share); return; }
    if (nb == 2) {
      if ((int)id < Hs) atomicAdd(&hot2[id], 1u);
      else if (!(A.dbg & 1)) unsafeAtomicAdd(&A.g_n2[id], 1.0);
    } else if (A.g_avgt_lo) {
      double hi, lo;
      exact_split01(share, hi, lo);
      unsafeAtomicAdd(&A.g_avgt[id], hi);
      if (lo != 0.0) unsafeAtomicAdd(&A.g_avgt_lo[id], lo);
    } else {
      unsafeAtomicAdd(&A.g_avgt[id], share);
    }
  }
};

#include "tsem_report_pack.h"

// threads per workgroup: the pass over the final z needs ~92 VGPRs (pi*theta gathers in flight) and spills under the 128 of a 1024-thread
// workgroup (15.6 vs 5.3 ms); the pass over the initial z needs 68 and gains from 16 waves per CU instead of 8 (7.8 -> 6.9 ms with its tie list)
constexpr int rr_nt(bool init) { return init ? 1024 : 512; }
template <int G, int E, bool INIT, int GM = 0>
__global__ __launch_bounds__(rr_nt(INIT)) void k_report_rows(ReportArgs A) {
  static_assert(E == 8 || E == 16, "entries per lane");
  extern __shared__ double rr_lds[];   // [lut_len] score table | [HC] pi*theta | [Hs] conf (f64) | [Hs] single winners | [Hs] two-way ties (u32)
  double* const lutS = rr_lds;
  double* const cH = lutS + A.lut_len;
  double* const hotF = cH + A.HC;
  uint32_t* const hot1 = reinterpret_cast<uint32_t*>(hotF + A.Hs);
  uint32_t* const hot2 = hot1 + A.Hs;
  double* const hotL = reinterpret_cast<double*>(hot2 + A.Hs);   // (only with g_conf_lo)
  if (A.g_conf_lo) for (int t = threadIdx.x; t < A.Hs; t += blockDim.x) hotL[t] = 0.0;
  for (int t = threadIdx.x; t < A.lut_len; t += blockDim.x) lutS[t] = A.lut[t];
  if (!INIT) for (int t = threadIdx.x; t < A.HC; t += blockDim.x) cH[t] = A.cnat2[t];
  for (int t = threadIdx.x; t < A.Hs; t += blockDim.x) { hotF[t] = 0.0; hot1[t] = 0u; hot2[t] = 0u; }
  __syncthreads();
  const ReportEmit<GM> EM{A, hotF, hot1, hot2, A.Hs, hotL};
  const int gl = threadIdx.x % G, grp = threadIdx.x / G, ngrp = blockDim.x / G;
  const int64_t stride = (int64_t)gridDim.x * ngrp;
  const int64_t nit = (A.N + stride - 1) / stride;
  const bool one_winner = A.thresh > 0.51;                // an entry with z >= thresh is then the row's unique best hit
  struct Ip { int64_t s; int len; int64_t goff; };
  struct Ent { rr_u32x4_a2 id[E / 8]; rr_u32x4_a2 cd[E / 8]; };   // 8 ids / 8 codes per 16-byte word
  auto load_ip = [&](int64_t it) -> Ip {
    const int64_t row = it * stride + (int64_t)blockIdx.x * ngrp + grp;
    const int64_t rc = row < A.N ? row : A.N - 1;        // clamped, never branched around
    const rr_i64x2_a8 se = *reinterpret_cast<const rr_i64x2_a8*>(A.indptr + rc);   // indptr[rc], indptr[rc + 1]
    Ip r; r.s = se.x; r.len = row < A.N ? (int)min<int64_t>(se.y - se.x, 0x7FFFFFFF) : 0; r.goff = 0;
    if (GM != 0) {                                         // rows outside this tile's groups count as empty
      const int32_t g = A.group[rc];
      if (g < A.g0 || g >= A.g1) r.len = 0;
      r.goff = (int64_t)(g - A.g0) * A.IDN;
    }
    return r;
  };
  auto load_ent = [&](const Ip& p) -> Ent {
    // lanes past the row's end read the entries that follow it (the arrays carry TS_ENTRY_PAD entries of padding: never
    // out of bounds)
    const int64_t k = p.s + E * gl;
    Ent t;
#pragma unroll
    for (int q = 0; q < E / 8; ++q) {
      t.id[q] = *reinterpret_cast<const rr_u32x4_a2*>(A.rid + k + 8 * q);
      t.cd[q] = *reinterpret_cast<const rr_u32x4_a2*>(A.raw + k + 8 * q);
    }
    return t;
  };
  auto half = [](const rr_u32x4_a2* w, int j) -> uint32_t {          // 16-bit element j of the packed words
    const uint32_t x = w[j / 8][(j / 2) & 3];
    return (j & 1) ? x >> 16 : x & 0xFFFFu;
  };
  // prep: the lane's numerators (the gathers that depend on the entries); finish: everything else.  The loop issues the
  // NEXT rows' loads between the two, so that waiting for the gathers (loads return in order) does not wait for the
  // prefetch as well.
  struct Prep { double n[E]; };
  auto row_prep = [&](const Ip& p, const Ent& t) -> Prep {
    const int k0 = E * gl;
    const bool amb = p.len > 1;                           // ambiguous rows: pi*theta, unique rows: pi (model.py:706-714)
    Prep q;
#pragma unroll
    for (int j = 0; j < E; ++j) {
      const bool v = k0 + j < p.len;
      double x = lutS[v ? half(t.cd, j) : 0u];             // (lut[0] = expm1(0) = 0: a lane past the row's end holds zeros)
      if (!INIT) {
        const uint32_t id = v ? half(t.id, j) : 0u;
        const bool hot = amb && (int)id < A.HC;
        // UNCONDITIONAL gather: hot lanes read element 0 (one line for all of them) — a branch around the load would
        // cost the compiler its count of the loads in flight
        const double cg = A.cnat2[hot ? 0u : id + (amb ? 0u : (uint32_t)A.IDN)];
        const double cl = cH[hot ? id : 0u];
        x = x * (hot ? cl : cg);
      }
      q.n[j] = x;
    }
    return q;
  };
  auto row_finish = [&](int64_t row, const Ip& p, const Ent& t, const Prep& q) -> bool {   // true: a near-tie, left to k_report_slow
    const int k0 = E * gl;
    const double* n = q.n;
    // row sum and the largest numerator of z's pattern (INIT: every stored entry; else the non-zero products, model.py:720)
    double s = 0.0, nm = -1.0;
#pragma unroll
    for (int j = 0; j < E; ++j) {
      s += n[j];
      const bool in = INIT ? (k0 + j < p.len) : (n[j] != 0.0);
      nm = in ? fmax(nm, n[j]) : nm;
    }
    const double r = recip0(rr_sum<G>(s));
    if (GM == 4) {                                         // `all` per group (model.py:860-862): one for every entry with z > 0
#pragma unroll
      for (int j = 0; j < E; ++j) {
        const bool in = INIT ? (k0 + j < p.len) : (n[j] != 0.0);
        if (in && n[j] * r > 0.0) atomicAdd(&A.t_cnt[p.goff + half(t.id, j)], 1u);
      }
      return false;
    }
    nm = rr_max<G>(nm);
    const bool any = nm >= 0.0;
    const double zmax = any ? nm * r : -1.0;               // = max_j fl(n_j r): rounding is monotone
    // The best hits are the entries EQUAL to the largest numerator — unless another numerator lies within the band below it: then
    // the last bits of r decide which z values round together, and the row is redone in the reference's order (near_band).  One
    // butterfly for both counts (a row has at most 256 entries here).
    const double lo = nm * (1.0 - TS_NEAR_BAND);
    int nbl = 0, nnl = 0; uint32_t wid = 0u;
#pragma unroll
    for (int j = 0; j < E; ++j) {
      const bool in = INIT ? (k0 + j < p.len) : (n[j] != 0.0);
      const bool b = in && n[j] == nm;
      nbl += b ? 1 : 0;
      nnl += (in && n[j] >= lo) ? 1 : 0;
      wid = b ? half(t.id, j) : wid;
    }
    const int both = rr_sum_i<G>(nbl | (nnl << 16));
    const int nb = both & 0xFFFF;
    bool near = (both >> 16) != nb || fabs(zmax - A.thresh) <= A.thresh * TS_NEAR_BAND;
    if (!one_winner) {                                     // conf_prob <= 0.5: any entry of the row may sit on the threshold
      int nt = 0;
#pragma unroll
      for (int j = 0; j < E; ++j) {
        const bool in = INIT ? (k0 + j < p.len) : (n[j] != 0.0);
        nt |= (in && fabs(n[j] * r - A.thresh) <= A.thresh * TS_NEAR_BAND) ? 1 : 0;
      }
      near = near || rr_max_i<G>(nt) != 0;
    }
    if (near) return true;
    const int64_t goff = p.goff;
    if (GM == 0 && gl == 0 && row < A.N && !(A.dbg & 2)) A.nbest[row] = any ? nb : 0;
    if (one_winner) {
      if (nbl != 0 && nb == 1) {                           // this lane holds the row's only best hit
        EM.one(wid, goff);
        if (zmax >= A.thresh) { const double vc = zmax * recip0(zmax); if (vc != 0.0) EM.conf(wid, vc, goff); }   // vsum = the winner's z
      }
      if (__builtin_amdgcn_ballot_w64(nb > 1) != 0ull && !(A.dbg & 4)) {   // tied rows
        if (nbl == 1 && nb == 2) EM.tie(wid, 2, 0.5, goff);      // (the usual tie: two best hits, this lane holds one of them)
        if (__builtin_amdgcn_ballot_w64(nb > 1 && !(nbl == 1 && nb == 2) && nbl != 0) != 0ull) {
          const double share = 1.0 * recip0((double)nb);
#pragma unroll
          for (int j = 0; j < E; ++j) {
            const bool in = INIT ? (k0 + j < p.len) : (n[j] != 0.0);
            if (nb > 1 && !(nbl == 1 && nb == 2) && in && n[j] == nm) EM.tie(half(t.id, j), nb, share, goff);
          }
        }
      }
    } else {                                               // conf_prob <= 0.5: several entries of a row may pass the threshold
      double vs = 0.0;
#pragma unroll
      for (int j = 0; j < E; ++j) {
        const bool in = INIT ? (k0 + j < p.len) : (n[j] != 0.0);
        const double z = n[j] * r;
        if (in && z >= A.thresh) vs += z;
      }
      const double rv = recip0(rr_sum<G>(vs)), share = 1.0 * recip0((double)nb);
#pragma unroll
      for (int j = 0; j < E; ++j) {
        const bool in = INIT ? (k0 + j < p.len) : (n[j] != 0.0);
        const double z = n[j] * r;
        if (!in || !(z == zmax || z >= A.thresh)) continue;
        const uint32_t id = half(t.id, j);
        if (z >= A.thresh) { const double vc = z * rv; if (vc != 0.0) EM.conf(id, vc, goff); }
        if (z == zmax) { if (nb == 1) EM.one(id, goff); else EM.tie(id, nb, share, goff); }
      }
    }
    return false;
  };
  if (nit > 0) {
    Ip ip0 = load_ip(0), ip1 = load_ip(1);
    Ent e0 = load_ent(ip0);
    for (int64_t it = 0; it < nit; ++it) {
      const int64_t row = it * stride + (int64_t)blockIdx.x * ngrp + grp;
      const bool defer = ip0.len > G * E;                  // left to k_report_slow
      Ip cur = ip0;
      if (defer) cur.len = 0;
      const Prep q = row_prep(cur, e0);                    // gathers of this row first ...
      __builtin_amdgcn_sched_barrier(0);
      const Ent e1 = load_ent(ip1);                        // ... then the loads of the next rows: they have this row's
      const Ip ip2 = load_ip(it + 2);                      //     arithmetic to arrive in
      __builtin_amdgcn_sched_barrier(0);
      if (defer) {
        if (gl == 0) A.defer_rows[atomicAdd(A.defer_n, 1ull)] = (int32_t)row;
      } else if (row_finish(row, cur, e0, q)) {
        if (gl == 0 && row < A.N) A.defer_rows[atomicAdd(A.defer_n, 1ull)] = ~(int32_t)row;   // a near-tie
      }
      ip0 = ip1; ip1 = ip2; e0 = e1;
    }
  }
  if (GM != 0) return;
  __syncthreads();
  for (int t = threadIdx.x; t < A.Hs; t += blockDim.x) {
    const double v = hotF[t];
    const uint32_t c1 = hot1[t], c2 = hot2[t];
    if (v != 0.0) unsafeAtomicAdd(&A.g_conf[t], v);
    if (A.g_conf_lo) { const double l = hotL[t]; if (l != 0.0) unsafeAtomicAdd(&A.g_conf_lo[t], l); }
    if (c1) unsafeAtomicAdd(&A.g_n1[t], (double)c1);
    if (c2) unsafeAtomicAdd(&A.g_n2[t], (double)c2);
  }
}

// The report pass over the INITIAL z when the caller wants no `conf` column (output_report takes exclude, choose and average of the
// initial z, model.py:441-446; tsem_report_colsums with thresh < 0): z_ij = Q_ij / sum_j Q_ij with Q = lut[code], and lut is strictly
// increasing, so a row's best hits are the entries with the LARGEST SCORE CODE — integer work on the packed 16-bit codes, no score-table
// gather, no fp64 at all: ~7 VALU instructions per lane-entry where k_report_rows<.., INIT> spends ~20 and an LDS gather (its time
// follows its lane capacity, not its useful entries: cap 64 / 128 / 256 -> 5.1 / 8.6 / 15.9 ms, profiles/r03_time_report.txt — it is
// bound by instruction issue).  Same row -> lanes mapping, prefetch and emit rules as k_report_rows; rows longer than G x E go to
// k_report_slow<true>.  Needs: no stored score of 0 (code 0 marks the padding here), lut strictly increasing (checked by the host).
template <int G, int E>
__global__ __launch_bounds__(1024) void k_report_init_codes(ReportArgs A) {
  static_assert(E == 8 || E == 16, "entries per lane");
  extern __shared__ double rr_lds[];                       // [Hs] single winners | [Hs] two-way ties (u32 each)
  uint32_t* const hot1 = reinterpret_cast<uint32_t*>(rr_lds);
  uint32_t* const hot2 = hot1 + A.Hs;
  for (int t = threadIdx.x; t < A.Hs; t += blockDim.x) { hot1[t] = 0u; hot2[t] = 0u; }
  __syncthreads();
  const ReportEmit<0> EM{A, nullptr, hot1, hot2, A.Hs, nullptr};
  const int gl = threadIdx.x % G, grp = threadIdx.x / G, ngrp = blockDim.x / G;
  const int64_t stride = (int64_t)gridDim.x * ngrp;
  const int64_t nit = (A.N + stride - 1) / stride;
  constexpr int W = E / 2;                                 // packed words per lane
  struct Ip { int64_t s; int len; };
  struct Ent { rr_u32x4_a2 id[E / 8]; rr_u32x4_a2 cd[E / 8]; };
  auto load_ip = [&](int64_t it) -> Ip {
    const int64_t row = it * stride + (int64_t)blockIdx.x * ngrp + grp;
    const int64_t rc = row < A.N ? row : A.N - 1;        // clamped, never branched around
    const rr_i64x2_a8 se = *reinterpret_cast<const rr_i64x2_a8*>(A.indptr + rc);
    Ip r; r.s = se.x; r.len = row < A.N ? (int)min<int64_t>(se.y - se.x, 0x7FFFFFFF) : 0;
    return r;
  };
  auto load_ent = [&](const Ip& p) -> Ent {
    const int64_t k = p.s + E * gl;
    Ent t;
#pragma unroll
    for (int q = 0; q < E / 8; ++q) {
      t.id[q] = *reinterpret_cast<const rr_u32x4_a2*>(A.rid + k + 8 * q);
      t.cd[q] = *reinterpret_cast<const rr_u32x4_a2*>(A.raw + k + 8 * q);
    }
    return t;
  };
  auto row_do = [&](int64_t row, const Ip& p, const Ent& t) {
    const int nl = min(max(p.len - E * gl, 0), E);         // this lane's share of the row
    uint32_t c[W], mx = 0u;
#pragma unroll
    for (int w = 0; w < W; ++w) {                          // codes past the row's end -> 0 (never the largest: stored scores are >= 1)
      const int lim = nl - 2 * w;
      c[w] = t.cd[w / 4][w & 3] & (lim >= 2 ? 0xFFFFFFFFu : (lim == 1 ? 0x0000FFFFu : 0u));
      typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
      mx = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, mx), __builtin_bit_cast(u16x2, c[w])));   // v_pk_max_u16
    }
    const int mrow = rr_max_i<G>((int)max(mx & 0xFFFFu, mx >> 16));
    const bool any = mrow > 0;
    const uint32_t mm = (uint32_t)mrow * 0x10001u;
    uint32_t nz = 0u;                                      // bit w: the low code of word w differs from the maximum; bit 16 + w: the high code
#pragma unroll
    for (int w = 0; w < W; ++w) {
      const uint32_t x = c[w] ^ mm;
      const uint32_t z = ((x & 0x7FFF7FFFu) + 0x7FFF7FFFu) | x;   // bit 15 / 31 set <=> that half is non-zero
      nz |= ((z >> 15) & 0x10001u) << w;
    }
    constexpr uint32_t full = ((1u << W) - 1u) * 0x10001u;
    const uint32_t eq = any ? (~nz & full) : 0u;
    const int nbl = __popc(eq);
    const int nb = rr_sum_i<G>(nbl);
    if (gl == 0 && row < A.N && !(A.dbg & 2)) A.nbest[row] = any ? nb : 0;
    // the popularity id behind bit b of `eq` (a select chain over the lane's words: once per lane and row, not per entry)
    auto id_of = [&](int b) -> uint32_t {
      const int w = b & 15;
      uint32_t v = t.id[0][0];
#pragma unroll
      for (int q = 1; q < W; ++q) v = w == q ? t.id[q / 4][q & 3] : v;
      return b >= 16 ? v >> 16 : v & 0xFFFFu;
    };
    const int b0 = nbl ? __ffs((int)eq) - 1 : 0;
    const uint32_t id0 = id_of(b0);
    if (nb == 1 && nbl == 1) EM.one(id0, 0);
    if (__builtin_amdgcn_ballot_w64(nb > 1) != 0ull && !(A.dbg & 4)) {      // rows with several best hits
      const double share = 1.0 * recip0((double)nb);
      if (nb > 1 && nbl >= 1) EM.tie(id0, nb, nb == 2 ? 0.5 : share, 0);
      uint32_t rest = nb > 1 && nbl > 1 ? eq & (eq - 1u) : 0u;
      while (__builtin_amdgcn_ballot_w64(rest != 0u) != 0ull) {             // further best hits inside the same lane
        if (rest) { EM.tie(id_of(__ffs((int)rest) - 1), nb, nb == 2 ? 0.5 : share, 0); rest &= rest - 1u; }
      }
    }
  };
  if (nit > 0) {
    Ip ip0 = load_ip(0), ip1 = load_ip(1);
    Ent e0 = load_ent(ip0);
    for (int64_t it = 0; it < nit; ++it) {
      const int64_t row = it * stride + (int64_t)blockIdx.x * ngrp + grp;
      const bool defer = ip0.len > G * E;                  // left to k_report_slow
      Ip cur = ip0;
      if (defer) cur.len = 0;
      const Ent e1 = load_ent(ip1);                        // the next rows' loads go out before this row's arithmetic
      const Ip ip2 = load_ip(it + 2);
      __builtin_amdgcn_sched_barrier(0);
      if (defer) {
        if (gl == 0) A.defer_rows[atomicAdd(A.defer_n, 1ull)] = (int32_t)row;
      } else {
        row_do(row, cur, e0);
      }
      ip0 = ip1; ip1 = ip2; e0 = e1;
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < A.Hs; t += blockDim.x) {
    const uint32_t c1 = hot1[t], c2 = hot2[t];
    if (c1) unsafeAtomicAdd(&A.g_n1[t], (double)c1);
    if (c2) unsafeAtomicAdd(&A.g_n2[t], (double)c2);
  }
}

// `choose` over the INITIAL z for a LIST of (tied) rows, on the score codes alone (same premises as k_report_init_codes): the best hits of
// a row are its entries with the largest code, in CSR order; picks[i] is the ordinal of the chosen one (sparse_plus.py:140-154).  One
// 16-lane group per listed row, two sweeps over its 2-byte codes; the winner's column (col_of_id[rid]) gets +1.  The generic row pass
// (16 lanes per row over column ids + scores + the score table, fp64) took 3.5 ms for the 5.6M tied rows of the 50M-row matrix.
// The winners are counted per popularity id in LDS (32-bit counters for the Hs most popular ids, i.e. all of them up to 38k slots) and
// flushed once per workgroup: 5.6M global fp64 atomics straight onto the columns ran at the hot-column rate (2 G/s: 2.8 ms).
__global__ __launch_bounds__(1024) void k_choose_init_codes(int64_t n, const int32_t* __restrict__ rowlist, const int32_t* __restrict__ picks,
    const int64_t* __restrict__ indptr, const uint16_t* __restrict__ raw, const uint16_t* __restrict__ rid,
    const int32_t* __restrict__ col_of_id, double* __restrict__ colsums, int Hs) {
  constexpr int G = 16;
  extern __shared__ uint32_t ch_hot[];                     // [Hs]
  for (int t = threadIdx.x; t < Hs; t += blockDim.x) ch_hot[t] = 0u;
  __syncthreads();
  const int gl = threadIdx.x % G, grp = threadIdx.x / G, ngrp = blockDim.x / G;
  const int sh = (threadIdx.x & 63) / G * G;               // the group's first lane inside its wave
  for (int64_t i = (int64_t)blockIdx.x * ngrp + grp; i < n; i += (int64_t)gridDim.x * ngrp) {
    const int64_t row = rowlist[i];
    const int64_t s = indptr[row];
    const int len = (int)(indptr[row + 1] - s);
    int m = 0;
    for (int k = gl; k < len; k += G) m = max(m, (int)raw[s + k]);
    m = sg_max_i<G>(m);
    int want = picks ? picks[i] : 0, seen = 0;
    for (int k0 = 0; k0 < len; k0 += G) {                  // (all 16 lanes stay in the loop: the ballots need them)
      const bool hit = k0 + gl < len && (int)raw[s + k0 + gl] == m;
      const uint32_t mask = (uint32_t)(__ballot(hit) >> sh) & 0xFFFFu;
      const int before = seen + __popc(mask & ((1u << gl) - 1u));
      if (hit && before == want) {
        const uint32_t id = rid[s + k0 + gl];
        if ((int)id < Hs) atomicAdd(&ch_hot[id], 1u); else unsafeAtomicAdd(&colsums[col_of_id[id]], 1.0);
      }
      seen += __popc(mask);
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < Hs; t += blockDim.x) {
    const uint32_t c = ch_hot[t];
    if (c) unsafeAtomicAdd(&colsums[col_of_id[t]], (double)c);
  }
}

// the rows k_report_rows left: any length, sweeps of 16 entries, one 16-lane group per row
template <bool INIT, int GM = 0>
__global__ __launch_bounds__(256) void k_report_slow(ReportArgs A) {
  constexpr int G = 16;
  const ReportEmit<GM> EM{A, nullptr, nullptr, nullptr, 0, nullptr};  // (no LDS slots here: a handful of rows)
  const int gl = threadIdx.x % G, grp = threadIdx.x / G, ngrp = blockDim.x / G;
  const int64_t nd = (int64_t)*A.defer_n;
  for (int64_t d = (int64_t)blockIdx.x * ngrp + grp; d < nd; d += (int64_t)gridDim.x * ngrp) {
    const int32_t rcode = A.defer_rows[d];                 // ~row: k_report_rows found a near-tie
    const int64_t row = rcode < 0 ? ~rcode : rcode;
    const int64_t s = A.indptr[row];
    const int len = (int)(A.indptr[row + 1] - s);
    const uint32_t coff = len > 1 ? 0u : (uint32_t)A.IDN;
    const int64_t goff = GM != 0 ? (int64_t)(A.group[row] - A.g0) * A.IDN : 0;   // (only rows of the tile's groups are deferred)
    auto numer = [&](int64_t k) -> double {
      double x = A.lut[A.raw[s + k]];
      if (!INIT) x = x * A.cnat2[A.rid[s + k] + coff];
      return x;
    };
    double y = 0.0, nmx = -1.0; int cnt = 0;
    for (int k = gl; k < len; k += G) {
      const double n = numer(k);
      y += n;
      if (INIT || n != 0.0) { nmx = fmax(nmx, n); ++cnt; }
    }
    nmx = sg_max<G>(nmx); cnt = sg_sum_i<G>(cnt);
    double r = recip0(sg_sum<G>(y));
    if (GM == 4) {
      for (int k = gl; k < len; k += G) { const double n = numer(k); if ((INIT || n != 0.0) && n * r > 0.0) atomicAdd(&A.t_cnt[goff + A.rid[s + k]], 1u); }
      continue;
    }
    {                                                       // near-ties (found here or by k_report_rows): the row sum in the reference's order
      const double band = near_band(len), lo = nmx * (1.0 - band), tb = A.thresh * band;
      int nf = rcode < 0 ? 1 : 0;
      for (int k = gl; k < len; k += G) {
        const double n = numer(k);
        if ((INIT || n != 0.0) && ((n >= lo && n != nmx) || fabs(n * r - A.thresh) <= tb)) nf = 1;
      }
      if (sg_max_i<G>(nf)) {
        double ex = 0.0;
        if (gl == 0) {
          const NpRow nr{A.lut, A.raw, nullptr, A.rid, INIT ? nullptr : A.cnat2, coff, s};
          ex = np_row_sum(nr, cnt, INIT);
          if (A.exact_n) atomicAdd(A.exact_n, 1ull);
        }
        r = recip0(__shfl(ex, 0, G));
      }
    }
    double zmax = -1.0, vs = 0.0;
    for (int k = gl; k < len; k += G) {
      const double n = numer(k);
      if (INIT || n != 0.0) { const double z = n * r; zmax = fmax(zmax, z); if (z >= A.thresh) vs += z; }
    }
    zmax = sg_max<G>(zmax);
    const double vsum = sg_sum<G>(vs);
    int nb = 0;
    for (int k = gl; k < len; k += G) { const double n = numer(k); if ((INIT || n != 0.0) && n * r == zmax) ++nb; }
    nb = sg_sum_i<G>(nb);
    if (GM == 0 && gl == 0) A.nbest[row] = cnt ? nb : 0;
    const double share = 1.0 * recip0((double)nb);
    for (int k = gl; k < len; k += G) {
      const double n = numer(k);
      if (!(INIT || n != 0.0)) continue;
      const double z = n * r;
      if (!(z == zmax || z >= A.thresh)) continue;
      const uint32_t id = A.rid[s + k];
      if (z >= A.thresh) { const double vc = z * recip0(vsum); if (vc != 0.0) EM.conf(id, vc, goff); }
      if (z == zmax) { if (nb == 1) EM.one(id, goff); else EM.tie(id, nb, share, goff); }
    }
  }
}
// a tile of per-group sums by id -> [groups][K] doubles by column
__global__ __launch_bounds__(256) void k_group_finish(int64_t n, int IDN, int K, const int32_t* __restrict__ col_of_id,
                                                      const uint32_t* __restrict__ t_cnt, const double* __restrict__ t_val, double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t g = i / IDN;
  const int j = col_of_id[(int)(i - g * IDN)];
  if (j >= 0) out[g * K + j] = t_cnt ? (double)t_cnt[i] : t_val[i];
}
// `unique` per group (model.py:857-859: ceil(z) of the single-entry rows): nothing to normalise, one row per thread
__global__ __launch_bounds__(256) void k_group_unique(int64_t N, const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
    const uint16_t* __restrict__ raw, const double* __restrict__ lut, const double* __restrict__ pi /* null: initial z */,
    const int32_t* __restrict__ group, int32_t g0, int32_t g1, int K, double* __restrict__ out) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= N) return;
  const int32_t g = group[row];
  if (g < g0 || g >= g1) return;
  const int64_t s = indptr[row];
  if (indptr[row + 1] - s != 1) return;
  const int col = indices[s];
  double n = lut[raw[s]];
  if (pi) n = n * pi[col];                                 // unique rows use pi alone (model.py:714)
  if (!pi || n != 0.0) {                                   // z's pattern: every stored entry of the initial z, the non-zero products else
    const double v = ceil(n * recip0(n));
    if (v != 0.0) unsafeAtomicAdd(&out[(int64_t)(g - g0) * K + col], v);
  }
}
__global__ void k_check_groups(int64_t N, const int32_t* __restrict__ grp, int32_t n_groups, uint32_t* __restrict__ bad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N && grp[i] >= n_groups) atomicOr(bad, 1u);
}
// by id -> by column: out[0..K) = conf, out[K..2K) = exclude, out[2K..3K) = average = n1 + n2 / 2 + the wider ties' shares
__global__ void k_report_finish(int IDN, int K, const int32_t* __restrict__ col_of_id, const double* __restrict__ g_conf,
                                const double* __restrict__ g_n1, const double* __restrict__ g_n2, const double* __restrict__ g_avgt,
                                const double* __restrict__ g_conf_lo, const double* __restrict__ g_avgt_lo, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= IDN) return;
  const int j = col_of_id[i];
  if (j < 0) return;
  const double cf = g_conf_lo ? g_conf[i] + g_conf_lo[i] : g_conf[i], av = g_avgt_lo ? g_avgt[i] + g_avgt_lo[i] : g_avgt[i];
  out[j] = cf; out[K + j] = g_n1[i]; out[2 * (int64_t)K + j] = (g_n1[i] + 0.5 * g_n2[i]) + av;
}
__global__ void k_cnat2_id(int IDN, const int32_t* __restrict__ col_of_id, const double* __restrict__ pi, const double* __restrict__ theta,
                           double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= IDN) return;
  const int j = col_of_id[i];
  out[i] = j >= 0 ? pi[j] * theta[j] : 0.0; out[IDN + i] = j >= 0 ? pi[j] : 0.0;
}

// mstep(z) on caller-supplied z (model.py:724-742): colsums[j] = sum_i (z_ij * w_i) * Y_i
__global__ __launch_bounds__(256) void k_mstep_rows(RowPassArgs A, const double* __restrict__ zin) {
  const int sub = threadIdx.x / RP_SUB, lane = threadIdx.x % RP_SUB, subs = blockDim.x / RP_SUB;
  for (int64_t row = (int64_t)blockIdx.x * subs + sub; row < A.N; row += (int64_t)gridDim.x * subs) {
    const int64_t s = A.indptr[row], e = A.indptr[row + 1];
    if (e - s < 2) continue;
    int m = 0;
    for (int64_t k = s + lane; k < e; k += RP_SUB) m = max(m, (int)A.raw[k]);
    m = sg_max_i<RP_SUB>(m);
    const double w = A.lut[m];
    for (int64_t k = s + lane; k < e; k += RP_SUB) {
      double v = zin[k] * w;
      if (v != 0.0) unsafeAtomicAdd(&A.colsums[A.indices[k]], v);
    }
  }
}

// calculate_lnl(z, pi, theta) on caller-supplied z (model.py:744-760)
__global__ __launch_bounds__(256) void k_lnl_rows(RowPassArgs A, const double* __restrict__ zin,
                                                  double* __restrict__ part) {
  __shared__ double scratch[16];
  const int sub = threadIdx.x / RP_SUB, lane = threadIdx.x % RP_SUB, subs = blockDim.x / RP_SUB;
  double acc = 0.0;
  for (int64_t row = (int64_t)blockIdx.x * subs + sub; row < A.N; row += (int64_t)gridDim.x * subs) {
    const int64_t s = A.indptr[row], e = A.indptr[row + 1];
    const bool amb = (e - s) > 1;
    for (int64_t k = s + lane; k < e; k += RP_SUB) {
      int col = A.indices[k];
      double c = amb ? A.pi[col] * A.theta[col] : A.pi[col];
      double inner = A.lut[A.raw[k]] * c;
      double z = zin[k];
      if (inner != 0.0 && z != 0.0) acc += z * ts_log1p_pos(inner);
    }
  }
  double t = block_sum(acc, scratch);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// closed forms of mstep without committing (model.py:733-740)
__global__ void k_hats(int K, const double* __restrict__ ts, const double* __restrict__ pisum0, double theta_pw,
                       double theta_den, double pi_pw, double pi_den, double* __restrict__ pi_hat,
                       double* __restrict__ theta_hat) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= K) return;
  theta_hat[j] = (ts[j] + theta_pw) / theta_den;
  pi_hat[j] = ((pisum0[j] + ts[j]) + pi_pw) / pi_den;
}

typedef void (*RowKern)(RowPassArgs);
template <bool FIX> static RowKern rowpass_kern(int mode, int meth) {
  switch (mode) {
    case RP_EXPORT_Z: return k_rowpass<RP_EXPORT_Z, -1, false>;
    case RP_BEST:     return k_rowpass<RP_BEST, -1, FIX>;
    case RP_REPORT:   return k_rowpass<RP_REPORT, -1, FIX>;
    default: break;
  }
  switch (meth) {
    case TSEM_RA_EXCLUDE: return k_rowpass<RP_REASSIGN, TSEM_RA_EXCLUDE, FIX>;
    case TSEM_RA_CHOOSE:  return k_rowpass<RP_REASSIGN, TSEM_RA_CHOOSE, FIX>;
    case TSEM_RA_AVERAGE: return k_rowpass<RP_REASSIGN, TSEM_RA_AVERAGE, FIX>;
    case TSEM_RA_CONF:    return k_rowpass<RP_REASSIGN, TSEM_RA_CONF, FIX>;
    case TSEM_RA_UNIQUE:  return k_rowpass<RP_REASSIGN, TSEM_RA_UNIQUE, FIX>;
    case TSEM_RA_ALL:     return k_rowpass<RP_REASSIGN, TSEM_RA_ALL, FIX>;
    default:              return k_rowpass<RP_REASSIGN, -1, FIX>;
  }
}

extern "C" {

// ---------------------------------------------------------------------------
// results
// ---------------------------------------------------------------------------
__global__ void k_cnat(int K, const double* __restrict__ pi, const double* __restrict__ theta, double* __restrict__ out) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < K) out[j] = pi[j] * theta[j];
}
// pi*theta per column for the row passes (A.pi / A.theta must be set)
static int make_cnat(tsem_ctx* h, RowPassArgs& A) {
  if (!h->d_cnat) TSEM_ALLOC(h->d_cnat, h->K);
  k_cnat<<<cdiv64(h->K, 256), 256, 0, h->stream>>>(h->K, A.pi, A.theta, h->d_cnat);
  TSEM_HIP(hipGetLastError());
  A.cnat = h->d_cnat;
  return TSEM_OK;
}

// the counter of rows redone in the reference's order of additions (near-ties), zeroed when it is first needed
static int exact_counter(tsem_ctx* h, unsigned long long** out) {
  if (!h->d_exact_n) {
    TSEM_ALLOC(h->d_exact_n, 1);
    TSEM_HIP(hipMemsetAsync(h->d_exact_n, 0, sizeof(unsigned long long), h->stream));
  }
  *out = h->d_exact_n;
  return TSEM_OK;
}
// The kernels that read the CSR column ids — k_rowpass<*>, k_group_unique, k_mstep_rows, k_lnl_rows — call this right before they are
// launched; the streaming passes (k_report_rows, k_report_init_codes, k_choose_init_codes, the group tiles) never do, so that a
// matrix whose ids were dropped (option "drop_csr_indices") stays at 10 B per entry through a report.
static int with_indices(tsem_ctx* h, RowPassArgs& A) {
  if (int rc = tsem_ensure_indices(h)) return rc;
  A.indices = h->d_indices;
  return TSEM_OK;
}
// ... and this afterwards: ids that were rebuilt for one generic pass go again where the option drops them (the passes that need them
// are the exception on such a matrix: a rebuild is one 6 B-per-entry sweep)
static void redrop_indices(tsem_ctx* h) {
  if (h->d_indices && h->d_rid16 && h->d_col_of_id && (h->opt_drop_indices == 1 || (h->opt_drop_indices < 0 && h->nnz >= 4000000000ll))) {
    (void)hipStreamSynchronize(h->stream);
    dfree(h->d_indices);
  }
}

// The passes over the INITIAL z that work on the 2-byte score codes alone (k_report_init_codes, k_choose_init_codes): a row's best
// hits are its largest codes — true when the score table is strictly increasing and no stored score is 0 (code 0 is their padding).
// Option "report_kernel": 1 (default) streaming kernels incl. the codes-only ones; 2 streaming kernels WITHOUT the codes-only ones
// (timing comparisons); 0 the generic row pass for everything.
static bool codes_path_ok(const tsem_ctx* h) {
  return h->opt_report_kernel == 1 && h->lut_increasing && !h->opt_reproducible && h->have_rowstats && !h->has_zero_score &&
         h->d_rid16 && h->d_col_of_id;
}

struct IndicesGuard {                                      // at the top of an entry point: ids this call had to rebuild do not outlive it
  tsem_ctx* h; bool had;
  explicit IndicesGuard(tsem_ctx* c) : h(c), had(c && c->d_indices != nullptr) {}
  ~IndicesGuard() { if (h && !had) redrop_indices(h); }
};

// One generic row pass = the pass proper + the FIX launch over the rows it flagged as near-ties (RowPassArgs::flag_bits).  `meth`
// >= 0 picks the instantiation with the reassign method fixed at compile time (-1: read from A.method).  z export and a
// caller-assigned z decide nothing with a row sum: one launch.
static int launch_rowpass(tsem_ctx* h, int mode, int meth, int grid, int block, size_t lds, RowPassArgs& A) {
  const bool listed = (mode == RP_REASSIGN || mode == RP_EXPORT_Z) && A.rowlist;
  const int64_t n_visit = listed ? A.nlist : A.N;
  if (n_visit <= 0) return TSEM_OK;
  const bool fix = mode != RP_EXPORT_Z && !A.zin;
  A.flag_bits = nullptr; A.flag_n = nullptr;
  if (fix) {
    const int64_t words = (n_visit + 511) / 512 * 16;      // (+ 2 words: the counter behind the bits)
    if (h->flag_words < words || !h->d_flag_bits) {
      dfree(h->d_flag_bits); h->flag_words = 0;
      TSEM_ALLOC(h->d_flag_bits, words + 2);
      h->flag_words = words;
    }
    TSEM_HIP(hipMemsetAsync(h->d_flag_bits, 0, sizeof(uint32_t) * (words + 2), h->stream));
    A.flag_bits = h->d_flag_bits;
    A.flag_n = reinterpret_cast<unsigned long long*>(h->d_flag_bits + words);
  }
  const RowKern k0 = rowpass_kern<false>(mode, meth);
  if (lds > 48 * 1024) TSEM_HIP(hipFuncSetAttribute((const void*)k0, hipFuncAttributeMaxDynamicSharedMemorySize, TS_LDS_MAX - 1024));
  k0<<<grid, block, lds, h->stream>>>(A);
  TSEM_HIP(hipGetLastError());
  if (fix) {
    const RowKern k1 = rowpass_kern<true>(mode, meth);
    if (lds > 48 * 1024) TSEM_HIP(hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, TS_LDS_MAX - 1024));
    k1<<<std::min(grid, 128), block, lds, h->stream>>>(A);   // (returns at once when the pass flagged nothing)
    TSEM_HIP(hipGetLastError());
  }
  return TSEM_OK;
}

static int rowpass_args(tsem_ctx* h, int which, RowPassArgs& A) {
  if (int rc = exact_counter(h, &A.exact_n)) return rc;
  A.N = h->N; A.K = h->K; A.indptr = h->d_indptr; A.indices = h->d_indices; A.raw = h->d_raw; A.lut = h->d_lut;
  A.method = 0; A.thresh = 0; A.picks = nullptr; A.zout = nullptr; A.nbest = nullptr; A.colsums = nullptr; A.group = nullptr; A.rowlist = nullptr; A.nlist = 0; A.colmap = nullptr; A.col_of_pc = nullptr; A.P = 0; A.Kp = 0; A.Hs = 0;
  A.zin = nullptr; A.cnat = nullptr; A.lut_len = h->lut_len <= 2048 ? h->lut_len : 0;   // (larger tables stay in global memory)
  if (which == TSEM_Z_USER) {
    if (!h->d_user_z) TSEM_FAIL(TSEM_ERR_ARG, "TSEM_Z_USER without tsem_set_user_z");
    A.pi = h->d_pi; A.theta = h->d_theta; A.zin = h->d_user_z;
    return TSEM_OK;
  }
  if (which == TSEM_Z_INITIAL) { A.pi = nullptr; A.theta = nullptr; }
  else if (which == TSEM_Z_PREV) { A.pi = h->d_pi_prev; A.theta = h->d_theta_prev; }
  else if (which == TSEM_Z_CUR) { A.pi = h->d_pi; A.theta = h->d_theta; }
  else TSEM_FAIL(TSEM_ERR_ARG, "bad `which`");
  if (which != TSEM_Z_INITIAL && !h->have_model) TSEM_FAIL(TSEM_ERR_ARG, "model not set");
  if (A.pi) { if (int rc = make_cnat(h, A)) return rc; }
  return TSEM_OK;
}
int tsem_rowpass_grid(tsem_ctx* h) { return (int)std::min<int64_t>(8192, std::max<int64_t>(1, (h->N + 15) / 16)); }

static int export_z_with(tsem_ctx* h, RowPassArgs& A, double* z) {
  double* d_z = nullptr;
  TSEM_SCOPED(d_z);
  TSEM_ALLOC(d_z, h->nnz);
  A.zout = d_z;
  if (int rc = with_indices(h, A)) return rc;
  if (int rc = launch_rowpass(h, RP_EXPORT_Z, -1, tsem_rowpass_grid(h), 256, (size_t)A.lut_len * 8, A)) return rc;
  if (h->nnz) TSEM_HIP(hipMemcpyAsync(z, d_z, sizeof(double) * h->nnz, hipMemcpyDeviceToHost, h->stream));
  TSEM_HIP(hipStreamSynchronize(h->stream));
  return TSEM_OK;
}

int tsem_export_z(tsem_ctx* h, int which, double* z) {
  if (!h || !h->d_indptr || !z) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  IndicesGuard ig(h);
  RowPassArgs A;
  if (int rc = rowpass_args(h, which, A)) return rc;
  return export_z_with(h, A, z);
}

int tsem_set_user_z(tsem_ctx* h, const double* z) {
  if (!h || !h->d_indptr) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  if (!z) { dfree(h->d_user_z); return TSEM_OK; }
  TSEM_ALLOC(h->d_user_z, h->nnz);
  if (h->nnz) TSEM_HIP(hipMemcpy(h->d_user_z, z, sizeof(double) * h->nnz, hipMemcpyHostToDevice));
  return TSEM_OK;
}

int tsem_estep(tsem_ctx* h, const double* pi, const double* theta, double* z) {
  if (!h || !h->have_model || !pi || !theta || !z) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  TSEM_HIP(hipMemcpyAsync(h->d_tmp_pi, pi, sizeof(double) * h->K, hipMemcpyHostToDevice, h->stream));
  TSEM_HIP(hipMemcpyAsync(h->d_tmp_theta, theta, sizeof(double) * h->K, hipMemcpyHostToDevice, h->stream));
  IndicesGuard ig(h);
  RowPassArgs A;
  if (int rc = rowpass_args(h, TSEM_Z_CUR, A)) return rc;
  A.pi = h->d_tmp_pi; A.theta = h->d_tmp_theta;
  if (int rc = make_cnat(h, A)) return rc;
  return export_z_with(h, A, z);
}

int tsem_best_counts(tsem_ctx* h, int which, int32_t* nbest) {
  if (!h || !h->d_indptr || !nbest) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  RowPassArgs A;
  if (int rc = rowpass_args(h, which, A)) return rc;
  int32_t* d_nb = nullptr;
  TSEM_SCOPED(d_nb);
  TSEM_ALLOC(d_nb, h->N);
  A.nbest = d_nb;
  IndicesGuard ig(h);
  if (int rc = with_indices(h, A)) return rc;
  if (int rc = launch_rowpass(h, RP_BEST, -1, tsem_rowpass_grid(h), 256, (size_t)A.lut_len * 8, A)) return rc;
  if (h->N) TSEM_HIP(hipMemcpyAsync(nbest, d_nb, sizeof(int32_t) * h->N, hipMemcpyDeviceToHost, h->stream));
  TSEM_HIP(hipStreamSynchronize(h->stream));
  return TSEM_OK;
}

struct TiedRow {                                            // predicate of tsem_best_ties: rows with several best hits
  const int32_t* nb;
  __device__ bool operator()(const int32_t& i) const { return nb[i] > 1; }
};
__global__ void k_gather_i32(int64_t n, const int32_t* __restrict__ idx, const int32_t* __restrict__ src, int32_t* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = src[idx[i]];
}

int tsem_best_ties(tsem_ctx* h, int which, int64_t cap, int32_t* rows, int32_t* counts, int64_t* n_out) {
  if (!h || !h->d_indptr || !n_out || cap < 0 || (cap && (!rows || !counts))) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  *n_out = 0;
  if (h->N == 0) return TSEM_OK;
  RowPassArgs A;
  if (int rc = rowpass_args(h, which, A)) return rc;
  int32_t *d_nb = nullptr, *d_rows = nullptr, *d_cnt = nullptr;
  unsigned long long* d_n = nullptr;
  void* tmp = nullptr;
  TSEM_SCOPED(d_nb); TSEM_SCOPED(d_rows); TSEM_SCOPED(d_cnt); TSEM_SCOPED(d_n); TSEM_SCOPED(tmp);
  TSEM_ALLOC(d_nb, h->N); TSEM_ALLOC(d_rows, h->N); TSEM_ALLOC(d_n, 1);
  A.nbest = d_nb;
  IndicesGuard ig(h);
  if (int rc = with_indices(h, A)) return rc;
  if (int rc = launch_rowpass(h, RP_BEST, -1, tsem_rowpass_grid(h), 256, (size_t)A.lut_len * 8, A)) return rc;
  size_t tb = 0;
  TiedRow pred{d_nb};
  rocprim::counting_iterator<int32_t> first(0);
  TSEM_HIP(rocprim::select(nullptr, tb, first, d_rows, d_n, (size_t)h->N, pred, h->stream));
  TSEM_HIP(hipMalloc(&tmp, tb ? tb : 1));
  TSEM_HIP(rocprim::select(tmp, tb, first, d_rows, d_n, (size_t)h->N, pred, h->stream));
  unsigned long long n = 0;
  TSEM_HIP(hipMemcpyAsync(&n, d_n, 8, hipMemcpyDeviceToHost, h->stream));
  TSEM_HIP(hipStreamSynchronize(h->stream));
  *n_out = (int64_t)n;
  int rc = TSEM_OK;
  if ((int64_t)n > cap) {
    h->err = "tsem_best_ties: more tied rows than the caller's arrays hold (call again with the returned count)";
    rc = TSEM_ERR_ARG;
  } else if (n) {
    TSEM_ALLOC(d_cnt, n);
    k_gather_i32<<<cdiv64((int64_t)n, 256), 256, 0, h->stream>>>((int64_t)n, d_rows, d_nb, d_cnt);
    TSEM_HIP(hipMemcpyAsync(rows, d_rows, sizeof(int32_t) * n, hipMemcpyDeviceToHost, h->stream));
    TSEM_HIP(hipMemcpyAsync(counts, d_cnt, sizeof(int32_t) * n, hipMemcpyDeviceToHost, h->stream));
    TSEM_HIP(hipStreamSynchronize(h->stream));
  }
  return rc;
}

// option "reproducible": a second, zeroed buffer for the low pieces of the values a row pass sums (exact_split01);
// rowpass_lo_end adds it to the sums and frees it
static int rowpass_lo_begin(tsem_ctx* h, RowPassArgs& A, int64_t n, double** lo) {
  *lo = nullptr;
  if (!h->opt_reproducible || n <= 0) return TSEM_OK;
  TSEM_ALLOC(*lo, n);
  TSEM_HIP(hipMemsetAsync(*lo, 0, sizeof(double) * n, h->stream));
  A.colsums_lo = *lo;
  return TSEM_OK;
}
static int rowpass_lo_end(tsem_ctx* h, double* sums, double* lo, int64_t n) {
  if (!lo) return TSEM_OK;
  k_add_lo<<<cdiv64(n, 256), 256, 0, h->stream>>>(n, sums, lo);
  TSEM_HIP(hipGetLastError());
  TSEM_HIP(hipStreamSynchronize(h->stream));
  return TSEM_OK;                                          // (the caller's scope guard frees `lo`)
}

int tsem_reassign(tsem_ctx* h, int method, double thresh, int which, const int32_t* picks, double* colsums,
                  double* mask) {
  if (!h || !h->d_indptr || !colsums) return TSEM_ERR_ARG;
  if (method < TSEM_RA_EXCLUDE || method > TSEM_RA_ALL) TSEM_FAIL(TSEM_ERR_ARG, "bad reassign method");
  if (int rc = ensure_device(h)) return rc;
  RowPassArgs A;
  if (int rc = rowpass_args(h, which, A)) return rc;
  // Two of the report's columns do not depend on the posteriors and were counted while the matrix was set up
  // (option "report_shortcuts", default 1; the row pass gives the same numbers, tests/test_gpu_round2.py):
  //   all, initial=True (model.py:860-862 on Q.norm(1)): one per stored entry with a positive score = the column's
  //     entry count (every score > 0: z = q / rowsum > 0 for every entry);
  //   unique (model.py:857-859): ceil(z) over the single-entry rows = the column's number of such rows with a positive
  //     score — z = n * (1/n) in (0, 1] whenever n = q * pi_j is a normal positive number, which holds for the
  //     parameters the M-step produces (pi_j >= pisum0_j / W_tot > 1e-60 for a column that has such a row).
  if (!mask && h->opt_shortcuts && h->d_ucount && h->have_rowstats && which != TSEM_Z_USER) {
    if (method == TSEM_RA_ALL && which == TSEM_Z_INITIAL && !h->has_zero_score && h->d_colcount) {
      std::vector<unsigned long long> c(h->K);
      TSEM_HIP(hipMemcpy(c.data(), h->d_colcount, sizeof(unsigned long long) * h->K, hipMemcpyDeviceToHost));
      for (int j = 0; j < h->K; ++j) colsums[j] = (double)c[j];
      return TSEM_OK;
    }
    if (method == TSEM_RA_UNIQUE && (which == TSEM_Z_INITIAL || (which == TSEM_Z_CUR ? h->em_cur : h->em_prev))) {
      std::vector<uint32_t> c(h->K);
      TSEM_HIP(hipMemcpy(c.data(), h->d_ucount, sizeof(uint32_t) * h->K, hipMemcpyDeviceToHost));
      for (int j = 0; j < h->K; ++j) colsums[j] = (double)c[j];
      return TSEM_OK;
    }
  }
  A.method = method; A.thresh = thresh;
  IndicesGuard ig(h);
  if (int rc = with_indices(h, A)) return rc;
  double *d_cs = nullptr, *d_mask = nullptr;
  int32_t* d_picks = nullptr;
  TSEM_SCOPED(d_cs); TSEM_SCOPED(d_mask); TSEM_SCOPED(d_picks);
  TSEM_ALLOC(d_cs, h->K);
  TSEM_HIP(hipMemsetAsync(d_cs, 0, sizeof(double) * h->K, h->stream));
  if (mask) TSEM_ALLOC(d_mask, h->nnz);
  if (method == TSEM_RA_CHOOSE && picks) {
    TSEM_ALLOC(d_picks, h->N);
    if (h->N) TSEM_HIP(hipMemcpyAsync(d_picks, picks, sizeof(int32_t) * h->N, hipMemcpyHostToDevice, h->stream));
  }
  A.colsums = d_cs; A.zout = d_mask; A.picks = d_picks;
  double* d_lo = nullptr;
  TSEM_SCOPED(d_lo);                                    // (the low pieces of the exact sums: freed on every return path)
  if (int rc = rowpass_lo_begin(h, A, h->K, &d_lo)) return rc;
  if (h->N && h->d_colmap && h->d_col_of_pc && h->P > 0) {
    // hot slots of every part in LDS.  `all` emits one value per stored entry, so it wants as many slots as fit: one
    // 1024-thread workgroup per CU with ~150 KB of accumulators.  The other modes emit at most a few values per ROW
    // and the pass is bound by the latency of its dependent loads (row pointers -> entries), not by atomics: two
    // workgroups per CU (32 waves) with half the slots each (option "rowpass_wgs").
    const int wgs = (method == TSEM_RA_ALL || h->opt_rowpass_wgs < 2) ? 1 : 2;
    A.colmap = h->d_colmap; A.col_of_pc = h->d_col_of_pc; A.P = h->P; A.Kp = h->Kp;
    A.Hs = std::max(0, std::min(h->Kp, (int)((TS_LDS_MAX / wgs - 8192 / wgs - 1024 - A.lut_len * 8) / 8 / h->P)));
    if (int rc = launch_rowpass(h, RP_REASSIGN, method, h->n_cu * wgs, 1024, (size_t)(A.P * A.Hs + A.lut_len) * 8, A)) return rc;
  } else if (h->N) {
    if (int rc = launch_rowpass(h, RP_REASSIGN, -1, tsem_rowpass_grid(h), 256, (size_t)A.lut_len * 8, A)) return rc;
  }
  if (int rc = rowpass_lo_end(h, d_cs, d_lo, h->K)) return rc;
  TSEM_HIP(hipMemcpyAsync(colsums, d_cs, sizeof(double) * h->K, hipMemcpyDeviceToHost, h->stream));
  if (mask && h->nnz) TSEM_HIP(hipMemcpyAsync(mask, d_mask, sizeof(double) * h->nnz, hipMemcpyDeviceToHost, h->stream));
  TSEM_HIP(hipStreamSynchronize(h->stream));
  return TSEM_OK;
}

// One pass for the column sums output_report takes from one z (model.py:432-457): conf | exclude | average, and the rows
// with several best hits (the only rows `choose` treats differently from `exclude`) compacted in row order.
// packed report pass: a row with several best hits is a row the fp32 filter left to k_report_slow — the tied rows are picked from the
// deferred list (a few thousand entries) instead of a select over all N best-hit counts (0.5 ms at 5e7 rows); sorted afterwards
__global__ void k_ties_of_deferred(const unsigned long long* __restrict__ nd, const int32_t* __restrict__ defer_rows, const int32_t* __restrict__ nbest,
                                   int32_t* __restrict__ out, unsigned long long* __restrict__ n_out) {
  const int64_t n = (int64_t)*nd;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t rc = defer_rows[i];
    const int32_t row = rc < 0 ? ~rc : rc;
    if (nbest[row] > 1) out[atomicAdd(n_out, 1ull)] = row;
  }
}
// HIP loads a translation unit's code object at the first launch (or attribute query) of one of its kernels — ~10 ms for this unit,
// which used to sit at the head of the first report of a process (11 of its 14.5 ms at 50M rows).  build_layout calls this once the fill
// of the blocked layout is enqueued, next to the fused unit's load: the host loads while the device is busy.  (From tsem_em_chunk, behind
// eight enqueued iterations, the load was NOT hidden: em() grew by the same 10 ms.)
void tsem_report_preload(void) {
  static std::atomic<bool> done{false};
  if (done.exchange(true)) return;
  hipFuncAttributes a;
  (void)hipFuncGetAttributes(&a, (const void*)k_report_finish);
}

// the chunk table of k_report_pack32: built on the first report of a matrix (or when the entries per lane change), kept until the matrix goes
static int ensure_report_chunks(tsem_ctx* h, int E) {
  if (h->d_rep_chunks && h->rep_chunk_E == E) return TSEM_OK;
  if (h->d_rep_chunks) { (void)hipFree(h->d_rep_chunks); h->d_rep_chunks = nullptr; h->n_rep_chunks = 0; }
  unsigned long long* d_n = nullptr;
  TSEM_SCOPED(d_n);
  TSEM_ALLOC(d_n, 1);
  const unsigned grid = (unsigned)cdiv64(h->N, RC_TILE);
  unsigned long long n = 0, n2 = 0;
  TSEM_HIP(hipMemsetAsync(d_n, 0, sizeof(unsigned long long), h->stream));
  k_report_chunks<<<grid, 256, 0, h->stream>>>(h->N, h->d_indptr, E, nullptr, d_n, 0);          // count
  TSEM_HIP(hipGetLastError());
  TSEM_HIP(hipMemcpyAsync(&n, d_n, 8, hipMemcpyDeviceToHost, h->stream));
  TSEM_HIP(hipStreamSynchronize(h->stream));
  if (n == 0) TSEM_FAIL(TSEM_ERR_ARG, "report chunk table: no chunks");
  RpChunk* d = nullptr;
  if (hipMalloc((void**)&d, (size_t)n * sizeof(RpChunk)) != hipSuccess) TSEM_FAIL(TSEM_ERR_NOMEM, "report chunk table");
  hipError_t e = hipMemsetAsync(d_n, 0, sizeof(unsigned long long), h->stream);
  if (e == hipSuccess) { k_report_chunks<<<grid, 256, 0, h->stream>>>(h->N, h->d_indptr, E, d, d_n, (int64_t)n); e = hipGetLastError(); }
  if (e == hipSuccess) e = hipMemcpyAsync(&n2, d_n, 8, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e != hipSuccess || n2 != n) {
    (void)hipFree(d);
    if (e != hipSuccess) TSEM_HIP(e);
    TSEM_FAIL(TSEM_ERR_ARG, "report chunk table: the two packing launches disagree");
  }
  h->d_rep_chunks = d; h->n_rep_chunks = (int64_t)n; h->rep_chunk_E = E;
  return TSEM_OK;
}

int tsem_report_colsums(tsem_ctx* h, int which, double thresh, double* out3K, int64_t* n_ties) {
  if (!h || !h->d_indptr || !out3K) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  const bool want_conf = !(thresh < 0.0);                  // thresh < 0: no `conf` column wanted (its third of out3K stays 0)
  PhaseTimer pt(h->stream);                                // (TSEM_TRACE=1)
  if (!want_conf) thresh = 0.9;                            // (what the paths that compute it anyway use)
  IndicesGuard ig(h);
  RowPassArgs A;
  if (int rc = rowpass_args(h, which, A)) return rc;
  dfree(h->d_tie_rows); dfree(h->d_tie_cnt); h->n_ties = 0;
  if (n_ties) *n_ties = 0;
  const int K = h->K;
  double* d_cs = nullptr;
  TSEM_SCOPED(d_cs);
  TSEM_ALLOC(d_cs, 3 * (int64_t)K);
  TSEM_HIP(hipMemsetAsync(d_cs, 0, sizeof(double) * 3 * K, h->stream));
  if (h->N) {
    // per-row best-hit counts, the compacted tie list and the scan's scratch stay allocated between reports (two N-sized
    // vectors: allocating and freeing them cost more than the pass's kernels)
    if (!h->d_rep_nb) { TSEM_ALLOC(h->d_rep_nb, h->N); TSEM_ALLOC(h->d_rep_rows, h->N); TSEM_ALLOC(h->d_rep_n, 1); }
    int32_t *const d_nb = h->d_rep_nb, *const d_rows = h->d_rep_rows;
    unsigned long long* const d_n = h->d_rep_n;
    A.thresh = thresh; A.colsums = d_cs; A.nbest = d_nb;
    pt.lap("report: scratch");
    if (h->opt_report_kernel != 0 && which != TSEM_Z_USER && h->d_rid16 && h->d_col_of_id && A.lut_len > 0) {
      // the streaming report kernel (k_report_rows): lanes per row x entries per lane = the smallest capacity that
      // fewer than 0.5 % of the rows exceed (row-length histogram of tsem_rowstats); the rest goes to k_report_slow
      const int IDN = h->Kpad;
      ReportArgs R;
      R.N = h->N; R.nnz = h->nnz; R.K = K; R.IDN = IDN; R.indptr = h->d_indptr; R.rid = h->d_rid16; R.raw = h->d_raw;
      R.lut = h->d_lut; R.lut_len = A.lut_len; R.cnat2 = nullptr; R.thresh = thresh; R.nbest = d_nb; R.exact_n = A.exact_n;
      const bool init = A.pi == nullptr;
      // thresh < 0: the caller wants no `conf` column.  The initial z then needs no arithmetic at all — the best hits of a row are its
      // largest score codes (k_report_init_codes) — provided the score table is strictly increasing and no stored score is 0.
      const bool codes_only = init && !want_conf && codes_path_ok(h);
      double *d_g = nullptr, *d_c2 = nullptr;
      TSEM_SCOPED(d_g); TSEM_SCOPED(d_c2);
      const bool exact = h->opt_reproducible != 0;
      TSEM_ALLOC(d_g, 6 * (int64_t)IDN);
      TSEM_HIP(hipMemsetAsync(d_g, 0, sizeof(double) * 6 * IDN, h->stream));
      if (!init) {
        TSEM_ALLOC(d_c2, 2 * (int64_t)IDN);
        k_cnat2_id<<<cdiv64(IDN, 256), 256, 0, h->stream>>>(IDN, h->d_col_of_id, A.pi, A.theta, d_c2);
        R.cnat2 = d_c2;
      }
      R.g_conf = d_g; R.g_n1 = d_g + IDN; R.g_n2 = d_g + 2 * (int64_t)IDN; R.g_avgt = d_g + 3 * (int64_t)IDN;
      if (exact) { R.g_conf_lo = d_g + 4 * (int64_t)IDN; R.g_avgt_lo = d_g + 5 * (int64_t)IDN; }
      // the final z at conf_prob > 0.51: rows packed into the lanes without padding (tsem_report_pack.h)
      int lut_sq = 0;                                        // lut 2^-lut_sq: the largest score table entry in [2^40, 2^41)
      double lut_max = 0.0;
      for (double v : h->lut_host) lut_max = std::max(lut_max, v);
      if (lut_max > 0.0 && std::isfinite(lut_max)) lut_sq = std::ilogb(lut_max) - 40;
      const bool packed = !init && !exact && thresh > 0.51 && thresh <= 0.99999 &&   // (at 1.0 a unique row's z = fl(n fl(1 / n)) may miss the threshold)
                          h->lut0_zero && lut_max > 0.0 && std::isfinite(lut_max) &&
                          (int)h->lut_host.size() == R.lut_len && !(h->opt_report_dbg & 8) && h->opt_report_lanes == 0 && h->N < 0x7FFFFFFF;
      // LDS: one workgroup of rr_nt() threads per CU (or two, option rowpass_wgs).  The final z wants pi*theta of as many
      // ids as fit (8 B each) next to a few thousand accumulator slots (16 B each); the initial z has no pi*theta.
      const int wgs = h->opt_rowpass_wgs >= 2 && h->opt_report_wgs2 ? 2 : 1;
      const int lds_avail = TS_LDS_MAX / wgs - 2048 - R.lut_len * 8;
      const int slot_bytes = exact ? 24 : 16;                // conf (f64) + two counters (+ the low pieces)
      R.Hs = std::min(IDN, init ? lds_avail / slot_bytes : std::min(3072, lds_avail / slot_bytes / 4));
      R.HC = init ? 0 : std::max(0, std::min(IDN, (lds_avail - R.Hs * slot_bytes) / 8));
      const int cw = h->opt_report_wgs2 ? 2 : 1;             // (codes-only kernel: 32 VGPRs, so two 1024-thread workgroups fit a CU with half the slots each)
      if (codes_only) R.Hs = std::min(IDN, (TS_LDS_MAX / cw - 2048) / 8);   // two 32-bit counters per id, nothing else in LDS
      int cap = 256;                                       // G x E
      if (h->opt_report_lanes > 0) {
        cap = (int)h->opt_report_lanes;
      } else if (h->have_rowstats) {
        for (int q = 0; q < 6; ++q)
          if ((double)h->len_gt[q] <= 0.005 * (double)h->N) { cap = 8 << q; break; }
      }
      void (*rk)(ReportArgs) = nullptr;
#define RK(G_, E_) (codes_only ? k_report_init_codes<G_, E_> : (init ? k_report_rows<G_, E_, true> : k_report_rows<G_, E_, false>))
      if (cap <= 8) rk = RK(1, 8); else if (cap <= 16) rk = RK(1, 16); else if (cap <= 32) rk = RK(2, 16);
      else if (cap <= 64) rk = RK(4, 16); else if (cap <= 128) rk = RK(8, 16); else rk = RK(16, 16);
#undef RK
      TSEM_HIP(hipFuncSetAttribute((const void*)rk, hipFuncAttributeMaxDynamicSharedMemorySize, TS_LDS_MAX - 1024));
      R.dbg = (int)h->opt_report_dbg;
      R.defer_rows = d_rows; R.defer_n = d_n;               // (d_rows is the tie list later: the slow kernel is done with it by then)
      TSEM_HIP(hipMemsetAsync(d_n, 0, sizeof(unsigned long long), h->stream));
      h->rep_timed = h->opt_timing != 0; h->rep_kernel = packed ? 4 : (codes_only ? 3 : 2); h->rep_deferred = 0;
      if (h->rep_timed) {
        for (int q = 0; q < 2; ++q) if (!h->ev_rep[q]) TSEM_HIP(hipEventCreate(&h->ev_rep[q]));
      }
      if (packed) {
        // tsem_report_pack.h: fp32 tables (by id), the per-row winner records, the chunk table; LDS = the table of as many ids as fit
        // (all of them up to ~34 000 slots) + the score table + 16 B per lane of row slots
        // entries per lane: 16 amortise the per-chunk work (row map, scan, prefetch) over twice the entries; short rows fill 8 better
        const int pkE = (h->opt_report_dbg & 128) ? 8 : ((h->opt_report_dbg & 256) ? 16 : (h->nnz >= 24 * h->N ? 16 : 8));
        pt.lap("report: by-id tables");
        if (int rc = ensure_report_chunks(h, pkE)) return rc;
        pt.lap("report: chunk table");
        uint32_t* d_t32 = nullptr; float* d_l32 = nullptr;
        TSEM_SCOPED(d_t32); TSEM_SCOPED(d_l32);
        TSEM_ALLOC(d_t32, (int64_t)IDN + 1); TSEM_ALLOC(d_l32, R.lut_len);
        k_rp32_tables<<<cdiv64(std::max(IDN + 1, R.lut_len), 256), 256, 0, h->stream>>>(IDN, d_c2, R.lut_len, R.lut, lut_sq, d_t32, d_l32);
        constexpr int nwv = 16;
        const int words = (TS_LDS_MAX - 1024) / 4 - 8 - (pkE + 1) * (pkE / 2);   // LDS words for the table of the ids and the copies of the score table
        Rp32Args P;
        P.N = h->N; P.IDN = IDN; P.lut_len = R.lut_len; P.indptr = h->d_indptr; P.rid = h->d_rid16; P.raw = h->d_raw;
        P.HC = std::min(IDN, std::max(0, words - R.lut_len));
        P.lut_rep = 0;
        while (P.lut_rep < 5 && P.HC + 1 + (R.lut_len << (P.lut_rep + 1)) <= words) ++P.lut_rep;
        P.t32 = d_t32; P.l32 = d_l32; P.thresh = (float)thresh; P.win = reinterpret_cast<uint32_t*>(d_nb); P.defer_rows = d_rows; P.defer_n = d_n;
        P.chunks = h->d_rep_chunks; P.nchunks = h->n_rep_chunks; P.dbg = (int)h->opt_report_dbg;
        const size_t lds = (size_t)((pkE + 1) * (pkE / 2) + P.HC + 1 + (P.lut_len << P.lut_rep)) * 4;
        void (*pk)(Rp32Args) = pkE == 16 ? (P.HC < IDN ? k_report_pack32<16, true> : k_report_pack32<16, false>)
                                         : (P.HC < IDN ? k_report_pack32<8, true> : k_report_pack32<8, false>);
        TSEM_HIP(hipFuncSetAttribute((const void*)pk, hipFuncAttributeMaxDynamicSharedMemorySize, TS_LDS_MAX - 1024));
        if (h->rep_timed) TSEM_HIP(hipEventRecord(h->ev_rep[0], h->stream));
        pk<<<h->n_cu, nwv * 64, lds, h->stream>>>(P);
        if (h->rep_timed) TSEM_HIP(hipEventRecord(h->ev_rep[1], h->stream));
        TSEM_HIP(hipGetLastError());
        pt.lap("report: k_report_pack32");
        // the winners per id: the whole LDS for counters, the ids in windows
        const int W = std::min(IDN, (TS_LDS_MAX - 1024) / 8);
        TSEM_HIP(hipFuncSetAttribute((const void*)k_report_hist, hipFuncAttributeMaxDynamicSharedMemorySize, TS_LDS_MAX - 1024));
        for (int id0 = 0; id0 < IDN; id0 += W)
          k_report_hist<<<h->n_cu, 1024, (size_t)std::min(W, IDN - id0) * 8, h->stream>>>(h->N, reinterpret_cast<const uint32_t*>(d_nb), id0, std::min(W, IDN - id0), R.g_n1, R.g_conf);
      } else {
        if (h->rep_timed) TSEM_HIP(hipEventRecord(h->ev_rep[0], h->stream));
        if (codes_only) rk<<<h->n_cu * cw, 1024, (size_t)R.Hs * 8, h->stream>>>(R);
        else rk<<<h->n_cu * wgs, rr_nt(init), (size_t)R.lut_len * 8 + (size_t)R.HC * 8 + (size_t)R.Hs * slot_bytes, h->stream>>>(R);
        if (h->rep_timed) TSEM_HIP(hipEventRecord(h->ev_rep[1], h->stream));
      }
      TSEM_HIP(hipGetLastError());
      pt.lap("report: pass (+ histogram)");
      if (init) k_report_slow<true><<<h->n_cu * 2, 256, 0, h->stream>>>(R);
      else k_report_slow<false><<<h->n_cu * 2, 256, 0, h->stream>>>(R);
      TSEM_HIP(hipGetLastError());
      k_report_finish<<<cdiv64(IDN, 256), 256, 0, h->stream>>>(IDN, K, h->d_col_of_id, R.g_conf, R.g_n1, R.g_n2, R.g_avgt, R.g_conf_lo, R.g_avgt_lo, d_cs);
      TSEM_HIP(hipGetLastError());
      if (packed) {
        // the tied rows: among the deferred ones (the fp32 filter decides only rows with ONE best hit) — picked from the list, sorted
        unsigned long long nd = 0, nt = 0;
        TSEM_HIP(hipMemcpyAsync(&nd, d_n, 8, hipMemcpyDeviceToHost, h->stream));
        TSEM_HIP(hipMemcpyAsync(out3K, d_cs, sizeof(double) * 3 * K, hipMemcpyDeviceToHost, h->stream));
        TSEM_HIP(hipStreamSynchronize(h->stream));
        h->rep_deferred = (int64_t)nd;
        pt.lap("report: exact rows, by column");
        if (nd) {
          int32_t *d_unsorted = nullptr, *d_sorted = nullptr;
          unsigned long long* d_nt = nullptr;
          TSEM_SCOPED(d_unsorted); TSEM_SCOPED(d_sorted); TSEM_SCOPED(d_nt);
          TSEM_ALLOC(d_unsorted, nd); TSEM_ALLOC(d_nt, 1);
          TSEM_HIP(hipMemsetAsync(d_nt, 0, sizeof(unsigned long long), h->stream));
          k_ties_of_deferred<<<256, 256, 0, h->stream>>>(d_n, d_rows, d_nb, d_unsorted, d_nt);
          TSEM_HIP(hipGetLastError());
          TSEM_HIP(hipMemcpyAsync(&nt, d_nt, 8, hipMemcpyDeviceToHost, h->stream));
          TSEM_HIP(hipStreamSynchronize(h->stream));
          if (nt) {
            TSEM_ALLOC(d_sorted, nt);
            size_t tb = 0;
            TSEM_HIP(rocprim::radix_sort_keys(nullptr, tb, d_unsorted, d_sorted, (size_t)nt, 0, 32, h->stream));
            if (h->rep_tmp_bytes < tb || !h->d_rep_tmp) {
              if (h->d_rep_tmp) (void)hipFree(h->d_rep_tmp);
              h->d_rep_tmp = nullptr; h->rep_tmp_bytes = 0;
              TSEM_HIP(hipMalloc(&h->d_rep_tmp, tb ? tb : 1));
              h->rep_tmp_bytes = tb;
            }
            TSEM_HIP(rocprim::radix_sort_keys(h->d_rep_tmp, tb, d_unsorted, d_sorted, (size_t)nt, 0, 32, h->stream));
            TSEM_ALLOC(h->d_tie_rows, nt); TSEM_ALLOC(h->d_tie_cnt, nt);
            TSEM_HIP(hipMemcpyAsync(h->d_tie_rows, d_sorted, sizeof(int32_t) * nt, hipMemcpyDeviceToDevice, h->stream));
            k_gather_i32<<<cdiv64((int64_t)nt, 256), 256, 0, h->stream>>>((int64_t)nt, d_sorted, d_nb, h->d_tie_cnt);
            TSEM_HIP(hipStreamSynchronize(h->stream));
          }
        }
        pt.lap("report: tied rows");
        h->n_ties = (int64_t)nt;
        if (n_ties) *n_ties = (int64_t)nt;
        return TSEM_OK;
      }
      TSEM_HIP(hipStreamSynchronize(h->stream));
    } else if (h->d_colmap && h->d_col_of_pc && h->P > 0) {
      h->rep_timed = false; h->rep_kernel = 1; h->rep_deferred = 0;
      if (int rc = with_indices(h, A)) return rc;
      const int wgs = h->opt_rowpass_wgs < 2 ? 1 : 2;
      A.colmap = h->d_colmap; A.col_of_pc = h->d_col_of_pc; A.P = h->P; A.Kp = h->Kp;
      A.Hs = std::max(0, std::min(h->Kp, (int)((TS_LDS_MAX / wgs - 8192 / wgs - 1024 - A.lut_len * 8) / 8 / h->P / 3)));
      double* d_lo = nullptr;
      TSEM_SCOPED(d_lo);                                    // (the low pieces of the exact sums: freed on every return path)
      if (int rc = rowpass_lo_begin(h, A, 3 * (int64_t)K, &d_lo)) return rc;
      if (int rc = launch_rowpass(h, RP_REPORT, -1, h->n_cu * wgs, 1024, (size_t)(3 * A.P * A.Hs + A.lut_len) * 8, A)) return rc;
      if (int rc = rowpass_lo_end(h, d_cs, d_lo, 3 * (int64_t)K)) return rc;
    } else {
      if (int rc = with_indices(h, A)) return rc;
      double* d_lo = nullptr;
      TSEM_SCOPED(d_lo);                                    // (the low pieces of the exact sums: freed on every return path)
      if (int rc = rowpass_lo_begin(h, A, 3 * (int64_t)K, &d_lo)) return rc;
      if (int rc = launch_rowpass(h, RP_REPORT, -1, tsem_rowpass_grid(h), 256, (size_t)A.lut_len * 8, A)) return rc;
      if (int rc = rowpass_lo_end(h, d_cs, d_lo, 3 * (int64_t)K)) return rc;
    }
    TSEM_HIP(hipGetLastError());
    size_t tb = 0;
    TiedRow pred{d_nb};
    rocprim::counting_iterator<int32_t> first(0);
    TSEM_HIP(rocprim::select(nullptr, tb, first, d_rows, d_n, (size_t)h->N, pred, h->stream));
    if (h->rep_tmp_bytes < tb || !h->d_rep_tmp) {
      if (h->d_rep_tmp) (void)hipFree(h->d_rep_tmp);
      h->d_rep_tmp = nullptr; h->rep_tmp_bytes = 0;
      TSEM_HIP(hipMalloc(&h->d_rep_tmp, tb ? tb : 1));
      h->rep_tmp_bytes = tb;
    }
    void* const tmp = h->d_rep_tmp;
    TSEM_HIP(rocprim::select(tmp, tb, first, d_rows, d_n, (size_t)h->N, pred, h->stream));
    unsigned long long n = 0;
    TSEM_HIP(hipMemcpyAsync(&n, d_n, 8, hipMemcpyDeviceToHost, h->stream));
    TSEM_HIP(hipMemcpyAsync(out3K, d_cs, sizeof(double) * 3 * K, hipMemcpyDeviceToHost, h->stream));
    TSEM_HIP(hipStreamSynchronize(h->stream));
    if (n) {
      TSEM_ALLOC(h->d_tie_rows, n); TSEM_ALLOC(h->d_tie_cnt, n);
      TSEM_HIP(hipMemcpyAsync(h->d_tie_rows, d_rows, sizeof(int32_t) * n, hipMemcpyDeviceToDevice, h->stream));
      k_gather_i32<<<cdiv64((int64_t)n, 256), 256, 0, h->stream>>>((int64_t)n, d_rows, d_nb, h->d_tie_cnt);
      TSEM_HIP(hipStreamSynchronize(h->stream));
    }
    h->n_ties = (int64_t)n;
    if (n_ties) *n_ties = (int64_t)n;
  } else {
    for (int64_t j = 0; j < 3 * (int64_t)K; ++j) out3K[j] = 0.0;
  }
  return TSEM_OK;
}

int tsem_report_ties(tsem_ctx* h, int64_t cap, int32_t* rows, int32_t* counts) {
  if (!h || cap < 0) return TSEM_ERR_ARG;
  if (h->n_ties > cap) TSEM_FAIL(TSEM_ERR_ARG, "tsem_report_ties: the arrays are shorter than the tie count of the last tsem_report_colsums");
  if (h->n_ties == 0) return TSEM_OK;
  if (!rows || !counts) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  TSEM_HIP(hipMemcpyAsync(rows, h->d_tie_rows, sizeof(int32_t) * h->n_ties, hipMemcpyDeviceToHost, h->stream));
  TSEM_HIP(hipMemcpyAsync(counts, h->d_tie_cnt, sizeof(int32_t) * h->n_ties, hipMemcpyDeviceToHost, h->stream));
  TSEM_HIP(hipStreamSynchronize(h->stream));
  return TSEM_OK;
}

// The contribution of a LIST of rows to reassign(method).sum(0): `choose` = `exclude` + the picked entries of the tied
// rows (rows == NULL: the tie rows the last tsem_report_colsums left on the device; picks[i] belongs to list entry i).
int tsem_reassign_rows(tsem_ctx* h, int method, double thresh, int which, const int32_t* rows, const int32_t* picks,
                       int64_t n, double* colsums) {
  if (!h || !h->d_indptr || !colsums || n < 0) return TSEM_ERR_ARG;
  if (method < TSEM_RA_EXCLUDE || method > TSEM_RA_ALL) TSEM_FAIL(TSEM_ERR_ARG, "bad reassign method");
  if (!rows && n != h->n_ties) TSEM_FAIL(TSEM_ERR_ARG, "tsem_reassign_rows: rows == NULL needs n == the tie count of the last report");
  if (int rc = ensure_device(h)) return rc;
  for (int j = 0; j < h->K; ++j) colsums[j] = 0.0;
  if (n == 0) return TSEM_OK;
  if (rows)
    for (int64_t i = 0; i < n; ++i)
      if (rows[i] < 0 || rows[i] >= h->N) TSEM_FAIL(TSEM_ERR_ARG, "tsem_reassign_rows: row out of range");
  double* d_cs = nullptr;
  int32_t *d_rows = nullptr, *d_picks = nullptr;
  TSEM_SCOPED(d_cs); TSEM_SCOPED(d_rows); TSEM_SCOPED(d_picks);
  TSEM_ALLOC(d_cs, h->K);
  TSEM_HIP(hipMemsetAsync(d_cs, 0, sizeof(double) * h->K, h->stream));
  if (rows) {
    TSEM_ALLOC(d_rows, n);
    TSEM_HIP(hipMemcpyAsync(d_rows, rows, sizeof(int32_t) * n, hipMemcpyHostToDevice, h->stream));
  }
  if (method == TSEM_RA_CHOOSE && picks) {
    TSEM_ALLOC(d_picks, n);
    TSEM_HIP(hipMemcpyAsync(d_picks, picks, sizeof(int32_t) * n, hipMemcpyHostToDevice, h->stream));
  }
  // `choose` over the initial z: the picked best hit is the picks[i]-th entry with the row's largest score code — no score table,
  // no column ids (the popularity ids do), no floating point — under the premises of k_report_init_codes
  if (method == TSEM_RA_CHOOSE && which == TSEM_Z_INITIAL) {
    if (codes_path_ok(h)) {
      const int Hs = std::min(h->Kpad, (TS_LDS_MAX - 2048) / 4);
      const int grid = (int)std::min<int64_t>(h->n_cu, std::max<int64_t>(1, (n + 63) / 64));
      TSEM_HIP(hipFuncSetAttribute((const void*)k_choose_init_codes, hipFuncAttributeMaxDynamicSharedMemorySize, TS_LDS_MAX - 1024));
      k_choose_init_codes<<<grid, 1024, (size_t)Hs * 4, h->stream>>>(n, rows ? d_rows : h->d_tie_rows, d_picks, h->d_indptr, h->d_raw, h->d_rid16,
                                                                     h->d_col_of_id, d_cs, Hs);
      TSEM_HIP(hipGetLastError());
      TSEM_HIP(hipMemcpyAsync(colsums, d_cs, sizeof(double) * h->K, hipMemcpyDeviceToHost, h->stream));
      TSEM_HIP(hipStreamSynchronize(h->stream));
      return TSEM_OK;
    }
  }
  RowPassArgs A;
  if (int rc = rowpass_args(h, which, A)) return rc;
  A.method = method; A.thresh = thresh; A.colsums = d_cs; A.picks = d_picks;
  A.rowlist = rows ? d_rows : h->d_tie_rows; A.nlist = n;
  IndicesGuard ig(h);
  if (int rc = with_indices(h, A)) return rc;
  const int grid = (int)std::min<int64_t>(8192, std::max<int64_t>(1, (n + 15) / 16));
  double* d_lo = nullptr;
  TSEM_SCOPED(d_lo);                                    // (the low pieces of the exact sums: freed on every return path)
  if (int rc = rowpass_lo_begin(h, A, h->K, &d_lo)) return rc;
  if (int rc = launch_rowpass(h, RP_REASSIGN, -1, grid, 256, (size_t)A.lut_len * 8, A)) return rc;
  if (int rc = rowpass_lo_end(h, d_cs, d_lo, h->K)) return rc;
  TSEM_HIP(hipMemcpyAsync(colsums, d_cs, sizeof(double) * h->K, hipMemcpyDeviceToHost, h->stream));
  TSEM_HIP(hipStreamSynchronize(h->stream));
  return TSEM_OK;
}

// z[ridx, fidx] and reassign(method)[ridx, fidx] for the entries of a LIST of rows — what Telescope.update_sam asks per alignment
// (model.py:483,508-511) — without materialising the N x K matrices: two row passes over the listed rows only, compact outputs.
// out_off[i] = where list row i's entries start in z_out / mask_out (out_off[n] = their total; the caller knows the row lengths:
// it holds the CSR it loaded).  z_out: -1 where the reference drops the entry from z's pattern; mask_out: the assignment value.
// picks[i] (choose): ordinal of the chosen best hit of list row i.
int tsem_rows_lookup(tsem_ctx* h, int which, int method, double thresh, int64_t n, const int32_t* rows, const int32_t* picks,
                     const int64_t* out_off, double* z_out, double* mask_out) {
  if (!h || !h->d_indptr || n < 0 || (n && (!rows || !out_off)) || (!z_out && !mask_out)) return TSEM_ERR_ARG;
  if (method < TSEM_RA_EXCLUDE || method > TSEM_RA_ALL) TSEM_FAIL(TSEM_ERR_ARG, "bad reassign method");
  if (int rc = ensure_device(h)) return rc;
  if (n == 0) return TSEM_OK;
  RowPassArgs A;
  if (int rc = rowpass_args(h, which, A)) return rc;
  const int64_t total = out_off[n];
  for (int64_t i = 0; i < n; ++i)
    if (rows[i] < 0 || rows[i] >= h->N || out_off[i] < 0 || out_off[i + 1] < out_off[i]) TSEM_FAIL(TSEM_ERR_ARG, "tsem_rows_lookup: bad row or offset");
  int32_t *d_rows = nullptr, *d_picks = nullptr;
  int64_t* d_off = nullptr;
  double *d_z = nullptr, *d_m = nullptr;
  uint32_t* d_bad = nullptr;
  TSEM_SCOPED(d_rows); TSEM_SCOPED(d_picks); TSEM_SCOPED(d_off); TSEM_SCOPED(d_z); TSEM_SCOPED(d_m); TSEM_SCOPED(d_bad);
  TSEM_ALLOC(d_rows, n); TSEM_ALLOC(d_off, n + 1); TSEM_ALLOC(d_bad, 1);
  TSEM_HIP(hipMemcpyAsync(d_rows, rows, sizeof(int32_t) * n, hipMemcpyHostToDevice, h->stream));
  TSEM_HIP(hipMemcpyAsync(d_off, out_off, sizeof(int64_t) * (n + 1), hipMemcpyHostToDevice, h->stream));
  TSEM_HIP(hipMemsetAsync(d_bad, 0, 4, h->stream));
  k_check_offsets<<<cdiv64(n, 256), 256, 0, h->stream>>>(n, d_rows, d_off, h->d_indptr, d_bad);   // every row's slice holds exactly its entries
  uint32_t bad = 0;
  TSEM_HIP(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, h->stream));
  TSEM_HIP(hipStreamSynchronize(h->stream));
  if (bad) TSEM_FAIL(TSEM_ERR_ARG, "tsem_rows_lookup: out_off does not follow the listed rows' lengths");
  if (method == TSEM_RA_CHOOSE && picks) {
    TSEM_ALLOC(d_picks, n);
    TSEM_HIP(hipMemcpyAsync(d_picks, picks, sizeof(int32_t) * n, hipMemcpyHostToDevice, h->stream));
  }
  A.rowlist = d_rows; A.nlist = n; A.out_off = d_off;
  IndicesGuard ig(h);
  if (int rc = with_indices(h, A)) return rc;
  const int grid = (int)std::min<int64_t>(8192, std::max<int64_t>(1, (n + 15) / 16));
  if (z_out) {
    TSEM_ALLOC(d_z, total);
    A.zout = d_z;
    if (int rc = launch_rowpass(h, RP_EXPORT_Z, -1, grid, 256, (size_t)A.lut_len * 8, A)) return rc;
    if (total) TSEM_HIP(hipMemcpyAsync(z_out, d_z, sizeof(double) * total, hipMemcpyDeviceToHost, h->stream));
  }
  if (mask_out) {
    TSEM_ALLOC(d_m, total);
    A.zout = d_m; A.method = method; A.thresh = thresh; A.picks = d_picks; A.colsums = nullptr;
    if (int rc = launch_rowpass(h, RP_REASSIGN, -1, grid, 256, (size_t)A.lut_len * 8, A)) return rc;
    if (total) TSEM_HIP(hipMemcpyAsync(mask_out, d_m, sizeof(double) * total, hipMemcpyDeviceToHost, h->stream));
  }
  TSEM_HIP(hipStreamSynchronize(h->stream));
  return TSEM_OK;
}

// The row -> group map of the per-group sums (tsem_reassign_groups): copied to the device ONCE and kept until the next call / the
// next matrix (-1 = the row belongs to no group); range-checked on the device.  group_of_row == NULL drops it.
int tsem_set_groups(tsem_ctx* h, const int32_t* group_of_row, int32_t n_groups) {
  if (!h || !h->d_indptr || n_groups < 0) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  if (!group_of_row) { dfree(h->d_group); h->n_groups = 0; return TSEM_OK; }
  TSEM_ALLOC(h->d_group, h->N);
  h->n_groups = 0;
  if (h->N) {
    DevTmp bad;
    TSEM_TMP(bad, 4);
    TSEM_HIP(hipMemsetAsync(bad.p, 0, 4, h->stream));
    TSEM_HIP(hipMemcpyAsync(h->d_group, group_of_row, sizeof(int32_t) * h->N, hipMemcpyHostToDevice, h->stream));
    k_check_groups<<<cdiv64(h->N, 256), 256, 0, h->stream>>>(h->N, h->d_group, n_groups, bad.as<uint32_t>());
    TSEM_HIP(hipGetLastError());
    uint32_t b = 0;
    TSEM_HIP(hipMemcpyAsync(&b, bad.p, 4, hipMemcpyDeviceToHost, h->stream));
    TSEM_HIP(hipStreamSynchronize(h->stream));
    if (b) { dfree(h->d_group); TSEM_FAIL(TSEM_ERR_ARG, "group_of_row entry out of range"); }
  }
  h->n_groups = n_groups;
  return TSEM_OK;
}

// Per-group column sums of the assignment matrix (scTelescope.output_report, model.py:611-625: row g of the result is
// reassign(method)[rows of group g, :].sum(0)).  The groups are processed in TILES of at most `group_tile_bytes` (1 GB) of output
// on the device — n_groups x K doubles need not fit anywhere but in the caller's `out` — one pass over the matrix per tile:
//   exclude | average | conf (conf_prob > 0.5) of the model's z: the streaming report kernel (k_report_rows, 4 B per entry), the
//     row's winner added to its group's line of the tile by a global atomic (32-bit counters for `exclude`);
//   unique | all | choose, a caller-assigned z, conf_prob <= 0.5, option "reproducible": the generic row pass over the CSR.
int tsem_reassign_groups(tsem_ctx* h, int method, double thresh, int which, const int32_t* picks,
                         const int32_t* group_of_row, int32_t n_groups, double* out) {
  if (!h || !h->d_indptr || !out || n_groups < 0) return TSEM_ERR_ARG;
  if (method < TSEM_RA_EXCLUDE || method > TSEM_RA_ALL) TSEM_FAIL(TSEM_ERR_ARG, "bad reassign method");
  if (int rc = ensure_device(h)) return rc;
  if (group_of_row) { if (int rc = tsem_set_groups(h, group_of_row, n_groups)) return rc; }
  else if (!h->d_group || h->n_groups != n_groups) TSEM_FAIL(TSEM_ERR_ARG, "tsem_reassign_groups: no group map (tsem_set_groups) for this number of groups");
  IndicesGuard ig(h);
  RowPassArgs A;
  if (int rc = rowpass_args(h, which, A)) return rc;
  A.method = method; A.thresh = thresh;
  const int K = h->K, IDN = h->Kpad;
  if (n_groups == 0 || K == 0) return TSEM_OK;
  const int64_t budget = h->opt_group_tile > 0 ? h->opt_group_tile : ((int64_t)1 << 30);
  const bool stream_ok = (method == TSEM_RA_EXCLUDE || method == TSEM_RA_AVERAGE || method == TSEM_RA_ALL || (method == TSEM_RA_CONF && thresh > 0.51)) &&
                         h->opt_report_kernel != 0 && which != TSEM_Z_USER && h->d_rid16 && h->d_col_of_id && A.lut_len > 0 &&
                         !h->opt_reproducible && h->N > 0;
  // the budget covers BOTH copies of a tile the streaming kernel needs (by column for the caller, by id for the kernel)
  const int tile = (int)std::max<int64_t>(1, std::min<int64_t>(n_groups, budget / (((int64_t)K + (stream_ok ? IDN : 0)) * 8)));
  // scratch kept between calls: the tile by column (what the caller gets) and, for the streaming kernel, the tile by id
  const size_t out_bytes = (size_t)tile * K * 8, id_bytes = stream_ok ? (size_t)tile * IDN * 8 : 0;
  if (h->gtile_bytes < out_bytes + id_bytes || !h->d_gtile) {
    if (h->d_gtile) (void)hipFree(h->d_gtile);
    h->d_gtile = nullptr; h->gtile_bytes = 0;
    if (hipMalloc(&h->d_gtile, out_bytes + id_bytes) != hipSuccess) TSEM_FAIL(TSEM_ERR_NOMEM, "hipMalloc failed (per-group tile)");
    h->gtile_bytes = out_bytes + id_bytes;
  }
  double* const d_out = static_cast<double*>(h->d_gtile);
  void* const d_id = static_cast<char*>(h->d_gtile) + out_bytes;
  DevTmp t_picks, t_c2, t_lo;
  if (method == TSEM_RA_CHOOSE && picks) {
    TSEM_TMP(t_picks, sizeof(int32_t) * h->N);
    if (h->N) TSEM_HIP(hipMemcpyAsync(t_picks.p, picks, sizeof(int32_t) * h->N, hipMemcpyHostToDevice, h->stream));
  }
  ReportArgs R;
  void (*rk)(ReportArgs) = nullptr;
  void (*rs)(ReportArgs) = nullptr;
  const bool init = A.pi == nullptr;
  const int gm = method == TSEM_RA_EXCLUDE ? 1 : (method == TSEM_RA_AVERAGE ? 2 : (method == TSEM_RA_ALL ? 4 : 3));
  size_t lds = 0;
  if (stream_ok) {
    R.N = h->N; R.nnz = h->nnz; R.K = K; R.IDN = IDN; R.indptr = h->d_indptr; R.rid = h->d_rid16; R.raw = h->d_raw;
    R.lut = h->d_lut; R.lut_len = A.lut_len; R.cnat2 = nullptr; R.thresh = method == TSEM_RA_CONF ? thresh : 0.9; R.nbest = nullptr;
    R.g_conf = R.g_n1 = R.g_n2 = R.g_avgt = nullptr; R.dbg = 0; R.exact_n = A.exact_n;
    if (!init) {
      TSEM_TMP(t_c2, sizeof(double) * 2 * IDN);
      k_cnat2_id<<<cdiv64(IDN, 256), 256, 0, h->stream>>>(IDN, h->d_col_of_id, A.pi, A.theta, t_c2.as<double>());
      R.cnat2 = t_c2.as<double>();
    }
    const int lds_avail = TS_LDS_MAX - 2048 - R.lut_len * 8;
    R.Hs = 0;
    R.HC = init ? 0 : std::max(0, std::min(IDN, lds_avail / 8));
    lds = (size_t)R.lut_len * 8 + (size_t)R.HC * 8;
    int cap = 256;                                         // G x E: the smallest capacity that fewer than 0.5 % of the rows exceed
    if (h->opt_report_lanes > 0) cap = (int)h->opt_report_lanes;
    else if (h->have_rowstats)
      for (int q = 0; q < 6; ++q)
        if ((double)h->len_gt[q] <= 0.005 * (double)h->N) { cap = 8 << q; break; }
#define RKG(G_, E_) (gm == 1 ? (init ? k_report_rows<G_, E_, true, 1> : k_report_rows<G_, E_, false, 1>) : \
                     gm == 2 ? (init ? k_report_rows<G_, E_, true, 2> : k_report_rows<G_, E_, false, 2>) : \
                     gm == 4 ? (init ? k_report_rows<G_, E_, true, 4> : k_report_rows<G_, E_, false, 4>) : \
                               (init ? k_report_rows<G_, E_, true, 3> : k_report_rows<G_, E_, false, 3>))
    if (cap <= 8) rk = RKG(1, 8); else if (cap <= 16) rk = RKG(1, 16); else if (cap <= 32) rk = RKG(2, 16);
    else if (cap <= 64) rk = RKG(4, 16); else if (cap <= 128) rk = RKG(8, 16); else rk = RKG(16, 16);
#undef RKG
    rs = gm == 1 ? (init ? k_report_slow<true, 1> : k_report_slow<false, 1>) :
         gm == 2 ? (init ? k_report_slow<true, 2> : k_report_slow<false, 2>) :
         gm == 4 ? (init ? k_report_slow<true, 4> : k_report_slow<false, 4>) : (init ? k_report_slow<true, 3> : k_report_slow<false, 3>);
    TSEM_HIP(hipFuncSetAttribute((const void*)rk, hipFuncAttributeMaxDynamicSharedMemorySize, TS_LDS_MAX - 1024));
    if (!h->d_rep_nb) { TSEM_ALLOC(h->d_rep_nb, h->N); TSEM_ALLOC(h->d_rep_rows, h->N); TSEM_ALLOC(h->d_rep_n, 1); }
    // (d_rep_rows / d_rep_n are only scratch here: the tie list of the last tsem_report_colsums lives in its own d_tie_rows / d_tie_cnt,
    //  which tsem_reassign_rows(rows = NULL) and `choose` still use after this call — ADVICE r4)
    R.defer_rows = h->d_rep_rows; R.defer_n = h->d_rep_n; R.group = h->d_group;
  } else if (h->opt_reproducible) {
    TSEM_TMP(t_lo, out_bytes);
  }
  for (int g0 = 0; g0 < n_groups; g0 += tile) {
    const int g1 = std::min(n_groups, g0 + tile);
    const int64_t n_out = (int64_t)(g1 - g0) * K;
    TSEM_HIP(hipMemsetAsync(d_out, 0, sizeof(double) * n_out, h->stream));
    if (stream_ok) {
      const int64_t n_id = (int64_t)(g1 - g0) * IDN;
      TSEM_HIP(hipMemsetAsync(d_id, 0, ((gm == 1 || gm == 4) ? 4 : 8) * (size_t)n_id, h->stream));
      TSEM_HIP(hipMemsetAsync(R.defer_n, 0, sizeof(unsigned long long), h->stream));
      R.g0 = g0; R.g1 = g1; R.t_cnt = (gm == 1 || gm == 4) ? static_cast<uint32_t*>(d_id) : nullptr; R.t_val = (gm == 1 || gm == 4) ? nullptr : static_cast<double*>(d_id);
      rk<<<h->n_cu, rr_nt(init), lds, h->stream>>>(R);
      TSEM_HIP(hipGetLastError());
      rs<<<h->n_cu * 2, 256, 0, h->stream>>>(R);
      TSEM_HIP(hipGetLastError());
      k_group_finish<<<cdiv64(n_id, 256), 256, 0, h->stream>>>(n_id, IDN, K, h->d_col_of_id, R.t_cnt, R.t_val, d_out);
      TSEM_HIP(hipGetLastError());
    } else if (h->N && method == TSEM_RA_UNIQUE && which != TSEM_Z_USER && !h->opt_reproducible) {
      if (int rc = with_indices(h, A)) return rc;
      k_group_unique<<<cdiv64(h->N, 256), 256, 0, h->stream>>>(h->N, h->d_indptr, h->d_indices, h->d_raw, h->d_lut, A.pi, h->d_group, g0, g1, K, d_out);
      TSEM_HIP(hipGetLastError());
    } else if (h->N) {
      if (int rc = with_indices(h, A)) return rc;
      A.colsums = d_out; A.picks = t_picks.as<int32_t>(); A.group = h->d_group; A.g0 = g0; A.g1 = g1;
      if (t_lo.p) { TSEM_HIP(hipMemsetAsync(t_lo.p, 0, sizeof(double) * n_out, h->stream)); A.colsums_lo = t_lo.as<double>(); }
      if (int rc = launch_rowpass(h, RP_REASSIGN, -1, tsem_rowpass_grid(h), 256, (size_t)A.lut_len * 8, A)) return rc;
      if (t_lo.p) { k_add_lo<<<cdiv64(n_out, 256), 256, 0, h->stream>>>(n_out, d_out, t_lo.as<double>()); TSEM_HIP(hipGetLastError()); }
    }
    TSEM_HIP(hipMemcpyAsync(out + (int64_t)g0 * K, d_out, sizeof(double) * n_out, hipMemcpyDeviceToHost, h->stream));
    TSEM_HIP(hipStreamSynchronize(h->stream));
  }
  return TSEM_OK;
}

int tsem_mstep(tsem_ctx* h, const double* z, double* pi_hat, double* theta_hat) {
  if (!h || !h->have_model || !z || !pi_hat || !theta_hat) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  RowPassArgs A;
  if (int rc = rowpass_args(h, TSEM_Z_CUR, A)) return rc;
  double *d_z = nullptr, *d_cs = nullptr;
  TSEM_SCOPED(d_z); TSEM_SCOPED(d_cs);
  TSEM_ALLOC(d_z, h->nnz);
  TSEM_ALLOC(d_cs, h->K);
  if (h->nnz) TSEM_HIP(hipMemcpyAsync(d_z, z, sizeof(double) * h->nnz, hipMemcpyHostToDevice, h->stream));
  TSEM_HIP(hipMemsetAsync(d_cs, 0, sizeof(double) * h->K, h->stream));
  A.colsums = d_cs;
  IndicesGuard ig(h);
  if (int rc = with_indices(h, A)) return rc;
  if (h->N) k_mstep_rows<<<tsem_rowpass_grid(h), 256, 0, h->stream>>>(A, d_z);
  if (tsem_comm_on(h)) {                                         // row-sharded: thetasum over all ranks (model.py:731)
    if (int rc = tsem_comm_allreduce_dev(h->comm, d_cs, (size_t)h->K, 0, h->stream, h->err)) return rc;
  }
  const double tpw = h->theta_prior * h->w_max, ppw = h->pi_prior * h->w_max;
  k_hats<<<cdiv64(h->K, 256), 256, 0, h->stream>>>(h->K, d_cs, h->d_pisum0, tpw, h->W_amb + tpw * h->K, ppw,
                                                  h->W_tot + ppw * h->K, h->d_tmp_pi, h->d_tmp_theta);
  TSEM_HIP(hipGetLastError());
  TSEM_HIP(hipMemcpyAsync(pi_hat, h->d_tmp_pi, sizeof(double) * h->K, hipMemcpyDeviceToHost, h->stream));
  TSEM_HIP(hipMemcpyAsync(theta_hat, h->d_tmp_theta, sizeof(double) * h->K, hipMemcpyDeviceToHost, h->stream));
  TSEM_HIP(hipStreamSynchronize(h->stream));
  return TSEM_OK;
}

int tsem_calc_lnl(tsem_ctx* h, const double* z, const double* pi, const double* theta, double* lnl) {
  if (!h || !h->have_model || !z || !pi || !theta || !lnl) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  RowPassArgs A;
  if (int rc = rowpass_args(h, TSEM_Z_CUR, A)) return rc;
  double* d_z = nullptr;
  TSEM_SCOPED(d_z);
  TSEM_ALLOC(d_z, h->nnz);
  if (h->nnz) TSEM_HIP(hipMemcpyAsync(d_z, z, sizeof(double) * h->nnz, hipMemcpyHostToDevice, h->stream));
  TSEM_HIP(hipMemcpyAsync(h->d_tmp_pi, pi, sizeof(double) * h->K, hipMemcpyHostToDevice, h->stream));
  TSEM_HIP(hipMemcpyAsync(h->d_tmp_theta, theta, sizeof(double) * h->K, hipMemcpyHostToDevice, h->stream));
  A.pi = h->d_tmp_pi; A.theta = h->d_tmp_theta;
  IndicesGuard ig(h);
  if (int rc = with_indices(h, A)) return rc;
  int grid = std::min(4096, tsem_rowpass_grid(h));
  if (h->N) k_lnl_rows<<<grid, 256, 0, h->stream>>>(A, d_z, h->d_lnl_part);
  if (int rc = tsem_sum_parts(h, h->d_lnl_part, h->N ? grid : 0, h->d_lnl_part, 0, h->d_lnl_part + 8000)) return rc;
  TSEM_HIP(hipGetLastError());
  if (tsem_comm_on(h)) {
    if (int rc = tsem_comm_allreduce_dev(h->comm, h->d_lnl_part + 8000, 1, 0, h->stream, h->err)) return rc;
  }
  TSEM_HIP(hipMemcpyAsync(lnl, h->d_lnl_part + 8000, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  TSEM_HIP(hipStreamSynchronize(h->stream));
  return TSEM_OK;
}


}  // extern "C"
