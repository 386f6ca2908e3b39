// libtelescope_em.so — MI355X (gfx950 / CDNA4) engine for Telescope's EM
// reassignment path.  C ABI in include/telescope_em.h.
//
// Data layout in HBM:
//   * canonical CSR of uint16 raw scores (indptr int64, indices int32, raw u16)
//     + the Q lookup table lut[r] = expm1(r/max*100) (model.py:653);
//   * for the EM hot loop, the AMBIGUOUS rows (Y_i = 1, model.py:679) re-laid
//     as a column-partitioned blocked COO ("PCOO"): columns are dealt by
//     popularity into P parts of Kp <= 7680 columns so that one part's
//     pi*theta table AND its fp64 column accumulators fit in LDS; rows are cut
//     in blocks of R; sub-block (b,p) is a contiguous run of
//     {fp64 Q value, u32 (local row << 16 | local col)} = 12 B per entry.
//   Global fp64 atomics reach only ~22 G/s on MI355X (2 G/s on hot columns)
//   while LDS gathers / ds_add_f64 keep up with the 5.8 TB/s HBM stream
//   (profiles/r01_primitives_ubench.log), hence every per-entry gather and
//   scatter of the hot loop goes through LDS.
#include <hip/hip_runtime.h>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include <dlfcn.h>
#include <link.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <numeric>

#include "tsem_common.h"

static std::string g_create_err;

// ============================================================================
// small device helpers
// ============================================================================
#include "tsem_device.h"

template <int W>
__device__ __forceinline__ double sg_sum(double v) {
#pragma unroll
  for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, W);
  return v;
}
template <int W>
__device__ __forceinline__ double sg_max(double v) {
#pragma unroll
  for (int o = W / 2; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, W));
  return v;
}
template <int W>
__device__ __forceinline__ int sg_sum_i(int v) {
#pragma unroll
  for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, W);
  return v;
}
template <int W>
__device__ __forceinline__ int sg_max_i(int v) {
#pragma unroll
  for (int o = W / 2; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, W));
  return v;
}

// block-wide sum of one double per thread; result valid in thread 0
__device__ __forceinline__ double block_sum(double v, double* scratch /* >= 16 doubles */) {
  v = sg_sum<64>(v);
  int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) {
    int nw = (blockDim.x + 63) >> 6;
    for (int i = 0; i < nw; ++i) t += scratch[i];
  }
  return t;
}


// ============================================================================
// synthetic generator (bit-exact twin of telescope_amd/synthetic.py)
// ============================================================================
__global__ void k_gen_len(int64_t row_begin, int64_t n, int32_t K, const uint32_t* __restrict__ cdf,
                          int cdf_len, uint64_t seed, uint32_t uniq_thresh, int64_t* __restrict__ lens) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  uint64_t row = (uint64_t)(row_begin + t);
  uint32_t h = (uint32_t)(ts_hash3(seed ^ TS_SALT_LEN, row, 0) >> 32);
  // searchsorted(cdf, h, side='right') == #{cdf[i] <= h}
  int lo = 0, hi = cdf_len;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (cdf[mid] <= h) lo = mid + 1; else hi = mid;
  }
  int len = lo;
  int cap = min(255, K - 1);
  len = max(1, min(len, cap));
  if (uniq_thresh) {
    uint32_t hu = (uint32_t)(ts_hash3(seed ^ TS_SALT_UNIQ, row, 0) >> 32);
    if (hu < uniq_thresh) len = 1;
  }
  lens[t] = len;
}

__global__ void k_gen_rows(int64_t row_begin, int64_t n, int32_t K, uint64_t seed, int dist,
                           const int64_t* __restrict__ indptr, int32_t* __restrict__ indices,
                           uint16_t* __restrict__ raw) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  uint64_t row = (uint64_t)(row_begin + t);
  int64_t s = indptr[t];
  int len = (int)(indptr[t + 1] - s);
  int32_t cols[256];
  bool has0 = (uint32_t)(ts_hash3(seed ^ TS_SALT_COL0, row, 0) >> 32) < 214748364u;  // int(0.05*2^32)
  for (int k = 0; k < len; ++k) {
    if (k == 0 && has0) { cols[0] = 0; continue; }
    for (int attempt = 0;; ++attempt) {
      uint64_t h = ts_hash3(seed, row, (uint64_t)(k + 256 * attempt));
      double u = (double)(h >> 11) * (1.0 / 9007199254740992.0);
      if (dist == 1) u = __dmul_rn(__dmul_rn(u, u), u);
      int32_t cand = 1 + (int32_t)floor(__dmul_rn((double)(K - 1), u));
      bool dup = false;
      for (int q = 0; q < k; ++q) dup |= (cols[q] == cand);
      if (!dup) { cols[k] = cand; break; }
    }
  }
  // insertion sort ascending
  for (int i = 1; i < len; ++i) {
    int32_t v = cols[i];
    int j = i - 1;
    while (j >= 0 && cols[j] > v) { cols[j + 1] = cols[j]; --j; }
    cols[j + 1] = v;
  }
  for (int p = 0; p < len; ++p) {
    indices[s + p] = cols[p];
    raw[s + p] = (uint16_t)(139 + (ts_hash3(seed ^ TS_SALT_SCORE, row, (uint64_t)p) % 162ull));
  }
}

// ============================================================================
// row statistics (model.py:679-699)
// ============================================================================
// One 16-lane group per row.  Outputs: per-row (len>=2 ? max raw code : 0),
// flags, per-WG partial sums of w (total / ambiguous), global max code,
// pisum0[col] += Q for unique rows — EXACTLY, so that the result does not depend on the order of the atomics (round 3; the
// fp64 atomics this replaced made pi differ in the last bit from run to run): Q is cut into pieces on PIS_LEVELS fixed grids
// 26 bits apart, from the largest score-table value down past the last mantissa bit of the smallest; a level's sum of up to
// 2^26 pieces is exact in fp64, k_pisum_finish adds the levels in a fixed order.  A 53-bit Q has pieces on 3-4 levels.
constexpr int PIS_LEVELS = 9, PIS_W = 26;
constexpr int RS_SUB = 16;
// G lanes per row, sixteen consecutive scores per lane (two 16-byte loads), G from the mean row length: the round-2 shape
// (16 lanes per row, one 2-byte load per lane and step) read the scores at 1 TB/s: 3.8 ms at 2e9 entries.
typedef unsigned int rs_u32x4_a2 __attribute__((ext_vector_type(4), aligned(2)));
template <int G>
__global__ __launch_bounds__(256) void k_rowstats(int64_t N, const int64_t* __restrict__ indptr,
    const int32_t* __restrict__ indices, const uint16_t* __restrict__ raw,
    const double* __restrict__ lut, uint16_t* __restrict__ row_code, uint8_t* __restrict__ row_class,
    double* __restrict__ wsum_part /* [grid][2] */, uint32_t* __restrict__ maxcode,
    double* __restrict__ pis_lv /* [PIS_LEVELS][K] */, int pis_e0 /* biased exponent of a power of two above every Q */,
    uint32_t* __restrict__ ucount /* [K] unique rows with a positive score per column; [K] = 1 if any stored score is 0 */,
    int K, unsigned long long* __restrict__ len_gt /* [6] rows longer than 8, 16, 32, 64, 128, 256 entries */) {
  __shared__ double scratch[16];
  constexpr int E = 16;
  const int gl = threadIdx.x % G, grp = threadIdx.x / G, ngrp = blockDim.x / G;
  double wt = 0.0, wa = 0.0;
  int mymax = 0;
  unsigned lg[6] = {0, 0, 0, 0, 0, 0};
  for (int64_t row = (int64_t)blockIdx.x * ngrp + grp; row < N; row += (int64_t)gridDim.x * ngrp) {
    const int64_t s = indptr[row];
    const int len = (int)(indptr[row + 1] - s);
    int m = 0;
    bool zero = false;
    if (gl == 0) {
#pragma unroll
      for (int q = 0; q < 6; ++q) lg[q] += len > (8 << q) ? 1u : 0u;
    }
    for (int k0 = E * gl; k0 < len; k0 += E * G) {           // (the array carries TS_ENTRY_PAD entries of padding)
      rs_u32x4_a2 cd[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) cd[q] = *reinterpret_cast<const rs_u32x4_a2*>(raw + s + k0 + 8 * q);
#pragma unroll
      for (int j = 0; j < E; ++j) {
        const uint32_t w = cd[j / 8][(j / 2) & 3];
        const int r = (int)((j & 1) ? w >> 16 : w & 0xFFFFu);
        if (k0 + j < len) { m = max(m, r); zero |= r == 0; }
      }
    }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, G));
    if (zero) ucount[K] = 1u;                              // (a stored score of 0: the shortcuts of tsem_reassign do not apply)
    if (gl == 0) {
      double w = (len > 0) ? lut[m] : 0.0;
      wt += w;
      if (len > 1) wa += w;
      row_code[row] = (uint16_t)m;
      row_class[row] = (len > 1) ? 2 : (len == 1 ? 1 : 0);
      mymax = max(mymax, m);
      if (len == 1) {
        double r = lut[raw[s]];
        const int col = indices[s];
        for (int lv = 0; lv < PIS_LEVELS && r != 0.0; ++lv) {
          const int eb = pis_e0 - PIS_W * lv;               // pieces of this level: |piece| <= 2^(eb-1023), multiples of 2^(eb-1023-PIS_W)
          if (eb + 52 - PIS_W < 1) break;                   // (below the normal range: nothing of a finite score table gets here)
          const double mm = __hiloint2double((int)(((uint32_t)(eb + 52 - PIS_W) << 20) | 0x80000u), 0);
          const double piece = (r + mm) - mm;
          if (piece != 0.0) unsafeAtomicAdd(&pis_lv[(size_t)lv * K + col], piece);
          r -= piece;
        }
        if (raw[s]) atomicAdd(&ucount[col], 1u);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const int t = sg_sum_i<64>((int)lg[q]);
    if ((threadIdx.x & 63) == 0 && t) atomicAdd(&len_gt[q], (unsigned long long)t);
  }
  double bt = block_sum(wt, scratch);
  double ba = block_sum(wa, scratch);
  int bm = sg_max_i<64>(mymax);
  if ((threadIdx.x & 63) == 0 && bm > 0) atomicMax(maxcode, (uint32_t)bm);
  if (threadIdx.x == 0) { wsum_part[2 * blockIdx.x] = bt; wsum_part[2 * blockIdx.x + 1] = ba; }
}

// ============================================================================
// layout build
// ============================================================================
__global__ void k_class_flags(int64_t N, const uint8_t* __restrict__ cls, int32_t* __restrict__ famb,
                              int32_t* __restrict__ funi) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) { famb[i] = cls[i] == 2; funi[i] = cls[i] == 1; }
}

__global__ void k_compact_rows(int64_t N, const uint8_t* __restrict__ cls, const int32_t* __restrict__ samb,
    const int32_t* __restrict__ suni, const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
    const uint16_t* __restrict__ raw, const uint16_t* __restrict__ row_code,
    int32_t* __restrict__ amb_row, uint16_t* __restrict__ amb_wcode, int32_t* __restrict__ uni_col,
    uint16_t* __restrict__ uni_code) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  if (cls[i] == 2) { int a = samb[i]; amb_row[a] = (int32_t)i; amb_wcode[a] = row_code[i]; }
  else if (cls[i] == 1) { int u = suni[i]; int64_t s = indptr[i]; uni_col[u] = indices[s]; uni_code[u] = raw[s]; }
}

// per-column entry count and order-independent signature sum_i hash(row_i, raw_ij)
// over ALL rows, LDS-privatised over a window of SIG_WIN columns per sweep.
// Counts order columns by popularity; (count, hash) identifies exact twin
// columns (same rows, same scores) whose parameters the reference keeps
// bit-identical (it accumulates every column in row order).
constexpr int SIG_WIN = 18432;      // 8 B of LDS per column: 147 KB
// One 64-bit LDS atomic per entry: the low half counts the column's entries, the high half sums a 32-bit hash of
// (global row, score) modulo 2^32 (a workgroup sees fewer than 2^32 entries of a column, so the halves never mix).
// Twins must agree on the count and on the hash sum of every rank — and k_update still only ties two columns whose
// accumulated sums agree to 1e-12, so a 32-bit signature is a filter, not the proof.  (Round 1: a 32-bit counter and
// a 64-bit hash, 12 B per column: three passes over the matrix at K = 30k instead of two, and two atomics per entry:
// 29 -> 14 ms at 2e9 entries.)
// Round 3: G lanes per row, SIXTEEN consecutive entries per lane (two 16-byte loads of column ids... four, and two of
// scores), the row half of the hash formed once per lane — the round-2 kernel (16 lanes per row, one 4-byte and one
// 2-byte load per lane and step, both hash rounds per entry) took 9.9 ms per sweep at 2e9 entries, bound by instruction
// issue like the row pass it resembled.  Same signature values.
typedef unsigned int cs_u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned int cs_u32x4_a2 __attribute__((ext_vector_type(4), aligned(2)));
template <int G>
__global__ __launch_bounds__(1024) void k_colsig(int64_t N, int64_t row_offset, const int64_t* __restrict__ indptr,
    const int32_t* __restrict__ indices, const uint16_t* __restrict__ raw, int col_base, int K,
    unsigned long long* __restrict__ counts, unsigned long long* __restrict__ hashes) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned long long* hh = reinterpret_cast<unsigned long long*>(smem);
  for (int t = threadIdx.x; t < SIG_WIN; t += blockDim.x) hh[t] = 0;
  __syncthreads();
  constexpr int E = 16;
  const int gl = threadIdx.x % G, grp = threadIdx.x / G, ngrp = blockDim.x / G;
  for (int64_t i = (int64_t)blockIdx.x * ngrp + grp; i < N; i += (int64_t)gridDim.x * ngrp) {
    const int64_t s = indptr[i];
    const int len = (int)(indptr[i + 1] - s);
    const uint64_t hrow = ts_mix64(0x7715ull ^ ((uint64_t)(row_offset + i) * TS_GOLDEN));   // the row half of ts_hash3
    for (int k0 = E * gl; k0 < len; k0 += E * G) {         // (the arrays carry TS_ENTRY_PAD entries of padding)
      cs_u32x4_a4 ix[4]; cs_u32x4_a2 cd[2];
#pragma unroll
      for (int q = 0; q < 4; ++q) ix[q] = *reinterpret_cast<const cs_u32x4_a4*>(indices + s + k0 + 4 * q);
#pragma unroll
      for (int q = 0; q < 2; ++q) cd[q] = *reinterpret_cast<const cs_u32x4_a2*>(raw + s + k0 + 8 * q);
#pragma unroll
      for (int j = 0; j < E; ++j) {
        const int c = (int)ix[j / 4][j & 3] - col_base;
        if (k0 + j < len && c >= 0 && c < SIG_WIN) {
          const uint32_t w = cd[j / 8][(j / 2) & 3];
          const uint64_t r = (j & 1) ? w >> 16 : w & 0xFFFFu;
          atomicAdd(&hh[c], (ts_mix64(hrow ^ (r * TS_M1)) & 0xFFFFFFFF00000000ull) | 1ull);
        }
      }
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < SIG_WIN; t += blockDim.x)
    if (hh[t] && col_base + t < K) {
      atomicAdd(&counts[col_base + t], hh[t] & 0xFFFFFFFFull);
      atomicAdd(&hashes[col_base + t], hh[t] >> 32);
    }
}

// entries of each ambiguous row per column part, packed 8 x 16 bit in two words (fused layout, P <= 8); the popularity
// ids of the report pass are written on the way (the column map is gathered here anyway).  G lanes per row, sixteen
// consecutive entries per lane (round 2: 16 lanes per row, one entry per lane and step: 8.3 ms at 2e9 entries).
// LM: the column map (4 B per column) in LDS, one 1024-thread workgroup per CU — 2e9 gathers of a 120 KB table through the vector
// cache (about one address per clock and CU) were most of this kernel's 8.1 ms at 2e9 entries; LDS serves 16+ lanes per clock.
template <int G, bool LM>
__global__ __launch_bounds__(LM ? 1024 : 256) void k_row_partcounts(int64_t N_amb, const int32_t* __restrict__ amb_row,
    const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices, const uint32_t* __restrict__ colmap_g, int K,
    unsigned long long* __restrict__ out /* [N_amb][2] */, uint16_t* __restrict__ rid /* popularity ids (k_report_rows) or null */, int P) {
  constexpr int E = 16;
  extern __shared__ uint32_t pc_cm[];                      // LM: [K]
  if (LM) {
    for (int t = threadIdx.x; t < K; t += blockDim.x) pc_cm[t] = colmap_g[t];
    __syncthreads();
  }
  const uint32_t* const colmap = LM ? pc_cm : colmap_g;
  const int gl = threadIdx.x % G, grp = threadIdx.x / G, ngrp = blockDim.x / G;
  for (int64_t a = (int64_t)blockIdx.x * ngrp + grp; a < N_amb; a += (int64_t)gridDim.x * ngrp) {
    const int64_t i = amb_row[a];
    const int64_t s = indptr[i];
    const int len = (int)(indptr[i + 1] - s);
    unsigned long long lo = 0, hi = 0;                     // 4 x 16-bit counters each (a lane sees at most 16 entries per step; rows < 65536)
    for (int k0 = E * gl; k0 < len; k0 += E * G) {
      cs_u32x4_a4 ix[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) ix[q] = *reinterpret_cast<const cs_u32x4_a4*>(indices + s + k0 + 4 * q);
      uint32_t cm[E];
#pragma unroll
      for (int j = 0; j < E; ++j) cm[j] = colmap[k0 + j < len ? ix[j / 4][j & 3] : 0u];
      uint32_t idv[E];
#pragma unroll
      for (int j = 0; j < E; ++j) {
        const uint32_t p = cm[j] >> 16;
        idv[j] = (cm[j] & 0x1FFFu) * P + p;
        if (k0 + j < len) {
          const unsigned long long one = 1ull << (16 * (p & 3));
          if (p < 4) lo += one; else hi += one;
        }
      }
      if (rid) {
        if (k0 + E <= len) {                               // a full lane: two 16-byte stores instead of sixteen 2-byte ones
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            cs_u32x4_a2 w;
#pragma unroll
            for (int t = 0; t < 4; ++t) w[t] = idv[8 * q + 2 * t] | (idv[8 * q + 2 * t + 1] << 16);
            *reinterpret_cast<cs_u32x4_a2*>(rid + s + k0 + 8 * q) = w;
          }
        } else {
#pragma unroll
          for (int j = 0; j < E; ++j) if (k0 + j < len) rid[s + k0 + j] = (uint16_t)idv[j];
        }
      }
    }
    // sums over the group (the packed 16-bit fields cannot carry into each other: a row has fewer than 65536 entries)
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) { lo += __shfl_xor(lo, o, G); hi += __shfl_xor(hi, o, G); }
    if (gl == 0) { out[2 * a] = lo; out[2 * a + 1] = hi; }
  }
}

// Block boundaries of the fused layout: a block takes consecutive ambiguous rows until one part would
// exceed `cap` entries (or R rows).  The rule is sequential, so the rows are cut into chunks of L rows
// (>= 256 blocks each: the forced break at a chunk end costs ~0.2 % more blocks) and one WAVE walks
// each chunk 64 rows at a time: lane prefix sums of the per-part counts, then the first lane that does
// not fit starts the next block.  pass 0 counts the blocks of a chunk, pass 1 (after an exclusive scan
// of the counts) writes their first rows.  flags[0]: a single row overflows the tile (-> two-pass).
__global__ __launch_bounds__(64) void k_block_greedy(int64_t na, int P, int R, int cap, int64_t L, int pass,
    const unsigned long long* __restrict__ pc, int64_t* __restrict__ cnt, const int64_t* __restrict__ off,
    int64_t* __restrict__ bstart, int* __restrict__ flags) {
  const int64_t ch = blockIdx.x;
  const int lane = threadIdx.x;
  const int64_t a0 = ch * L, a1 = min(na, a0 + L);
  if (a0 >= a1) return;
  int c[8] = {0, 0, 0, 0, 0, 0, 0, 0}, rows = 0;         // the open block so far (uniform across the wave)
  int64_t n = 1;
  const int64_t o = pass ? off[ch] : 0;
  if (pass && lane == 0) bstart[o] = a0;
  for (int64_t t0 = a0; t0 < a1; t0 += 64) {
    const bool v = t0 + lane < a1;
    const unsigned long long lo = v ? pc[2 * (t0 + lane)] : 0ull, hi = v ? pc[2 * (t0 + lane) + 1] : 0ull;
    int S[8];
    bool too_big = false;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      S[q] = q < P ? (int)(((q < 4 ? lo : hi) >> (16 * (q & 3))) & 0xFFFF) : 0;
      too_big |= S[q] > cap;
    }
    if (too_big) flags[0] = 1;
#pragma unroll
    for (int q = 0; q < 8; ++q) {                          // inclusive prefix over the lanes
      if (q < P) {
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(S[q], d, 64); if (lane >= d) S[q] += t; }
      }
    }
    int sub[8] = {0, 0, 0, 0, 0, 0, 0, 0};                 // prefix just before the open block's first lane of this tile
    int s = 0;                                             // that lane (0: the block continues from earlier tiles)
    for (;;) {
      bool bad = false;
#pragma unroll
      for (int q = 0; q < 8; ++q) bad |= c[q] + S[q] - sub[q] > cap;
      bad |= rows + lane - s + 1 > R;
      bad &= v && lane >= s && !(rows == 0 && lane == s);  // the first row of a block always goes in
      const unsigned long long m = __ballot(bad);
      if (!m) break;
      const int b = __ffsll((long long)m) - 1;             // first row that does not fit: it starts the next block
      if (pass && lane == 0) bstart[o + n] = t0 + b;
      ++n;
#pragma unroll
      for (int q = 0; q < 8; ++q) { sub[q] = b > 0 ? __shfl(S[q], b - 1, 64) : 0; c[q] = 0; }
      rows = 0; s = b;
    }
    const int last = (int)min<int64_t>(63, a1 - t0 - 1);
#pragma unroll
    for (int q = 0; q < 8; ++q) c[q] += __shfl(S[q], last, 64) - sub[q];
    rows += last - s + 1;
  }
  if (!pass && lane == 0) cnt[ch] = n;
}
// sub-block sizes from the per-row part counts the block boundaries were computed from (one wave per block)
__global__ __launch_bounds__(64) void k_sb_count_pc(int64_t nb, int P, const int64_t* __restrict__ bstart,
    const unsigned long long* __restrict__ pc, int64_t* __restrict__ sb_cnt) {
  const int64_t b = blockIdx.x;
  int c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int64_t a = bstart[b] + threadIdx.x; a < bstart[b + 1]; a += 64) {
    const unsigned long long lo = pc[2 * a], hi = pc[2 * a + 1];
#pragma unroll
    for (int q = 0; q < 8; ++q) c[q] += (int)(((q < 4 ? lo : hi) >> (16 * (q & 3))) & 0xFFFF);
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int t = sg_sum_i<64>(c[q]);
    if (threadIdx.x == 0 && q < P) sb_cnt[b * P + q] = t;
  }
}
__global__ void k_fixed_blocks(int64_t nb, int R, int64_t na, int64_t* __restrict__ bstart) {
  int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b <= nb) bstart[b] = min(na, b * R);
}

// rows of block b are the compact ambiguous rows [bstart[b], bstart[b+1]); the rest of its R slots are holes
__global__ __launch_bounds__(256) void k_make_slots(int64_t nb, int R, const int64_t* __restrict__ bstart,
    const int32_t* __restrict__ amb_row, const uint16_t* __restrict__ wcode_c, int32_t* __restrict__ slot_row,
    uint16_t* __restrict__ slot_wcode) {
  int64_t b = blockIdx.x;
  const int64_t a0 = bstart[b], n = bstart[b + 1] - a0;
  for (int lr = threadIdx.x; lr < R; lr += blockDim.x) {
    const bool v = lr < n;
    slot_row[b * R + lr] = v ? amb_row[a0 + lr] : -1;
    slot_wcode[b * R + lr] = v ? wcode_c[a0 + lr] : (uint16_t)0;
  }
}

// per (row block, part) entry counts — one WG per block
__global__ __launch_bounds__(256) void k_sb_count(int64_t N_amb, int R, int P, const int32_t* __restrict__ amb_row,
    const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices, const uint32_t* __restrict__ colmap,
    int64_t* __restrict__ sb_cnt) {
  __shared__ uint32_t cnt[64];
  if (threadIdx.x < 64) cnt[threadIdx.x] = 0;
  __syncthreads();
  int64_t b = blockIdx.x;
  const int sub = threadIdx.x / RS_SUB, lane = threadIdx.x % RS_SUB, subs = blockDim.x / RS_SUB;
  for (int lr = sub; lr < R; lr += subs) {
    int64_t i = amb_row[b * R + lr];                  // row slot -> CSR row, -1 = hole
    if (i < 0) continue;
    int64_t s = indptr[i], e = indptr[i + 1];
    for (int64_t k = s + lane; k < e; k += RS_SUB) atomicAdd(&cnt[colmap[indices[k]] >> 16], 1u);
  }
  __syncthreads();
  if (threadIdx.x < P) sb_cnt[b * P + threadIdx.x] = cnt[threadIdx.x];
}

__global__ __launch_bounds__(256) void k_sb_fill(int64_t N_amb, int R, int P, const int32_t* __restrict__ amb_row,
    const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices, const uint16_t* __restrict__ raw,
    const double* __restrict__ lut, const uint32_t* __restrict__ colmap, const int64_t* __restrict__ sb_off,
    double* __restrict__ pval, uint16_t* __restrict__ pcode, uint32_t* __restrict__ prc) {
  __shared__ uint32_t cur[64];
  if (threadIdx.x < 64) cur[threadIdx.x] = 0;
  __syncthreads();
  int64_t b = blockIdx.x;
  const int sub = threadIdx.x / RS_SUB, lane = threadIdx.x % RS_SUB, subs = blockDim.x / RS_SUB;
  for (int lr = sub; lr < R; lr += subs) {
    int64_t i = amb_row[b * R + lr];
    if (i < 0) continue;
    int64_t s = indptr[i], e = indptr[i + 1];
    for (int64_t k = s + lane; k < e; k += RS_SUB) {
      uint32_t cm = colmap[indices[k]];
      uint32_t p = cm >> 16;
      // Entries of one row are handed consecutive tickets; writing ticket t of a
      // sub-block to slot (t % S) * L + t / S (S strands of L slots) puts them S..L
      // slots apart, so the lanes of one wave instruction hit different rows and the
      // LDS row-sum atomics do not serialise on one address.
      const int64_t base = sb_off[b * P + p];
      const uint32_t L = (uint32_t)((sb_off[b * P + p + 1] - base) / TS_STRANDS);
      const uint32_t t = atomicAdd(&cur[p], 1u);
      int64_t pos = base + (int64_t)(t % TS_STRANDS) * L + t / TS_STRANDS;
      if (pcode) pcode[pos] = raw[k];
      else pval[pos] = lut[raw[k]];
      prc[pos] = ((uint32_t)lr << 16) | ((cm & 0x1FFFu) + (t & ((1u << ((cm >> 13) & 7u)) - 1u)));   // hot column: deal over its slots
    }
  }
}

// Fused layout: the entries of a sub-block are stored densely in ROW order (any order inside a row), so
// a thread's four consecutive entries and its neighbours' mostly share a row and the row sums can be
// reduced in registers / across lanes instead of one LDS atomic per entry (tsem_fused.h, phase 1).
// The padding at the end of a sub-block repeats the last row with value 0.
constexpr int FILL_MAX_RP = 768 * 8;                       // row slots x parts of a block the row-order fill can take (1152 x 4, 768 x 8)
// Round 3 (second pass over this kernel, 13.9 ms at 2e9 entries): it was bound by the LATENCY of three dependent loads
// per row (row slot -> row pointers -> entries, then one more round trip per 16 entries of the row) with 16 rows in
// flight per workgroup.  Now the row pointers of the whole block go to LDS in one parallel sweep, and a 16-lane group
// loads the first 64 ids and scores of its NEXT row before it places the current one (unconditional loads: the arrays
// carry TS_ENTRY_PAD entries of padding), so a group waits for memory about once per row instead of four times.
__host__ __device__ inline size_t fill_lds_bytes(int R, int P) { return (size_t)(((R * P + 1) & ~1) * 4) + (size_t)R * 12; }
__global__ __launch_bounds__(256) void k_sb_fill_sorted(int64_t N_amb, int R, int P, const int32_t* __restrict__ amb_row,
    const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices, const uint16_t* __restrict__ raw,
    const double* __restrict__ lut, const uint32_t* __restrict__ colmap, const int64_t* __restrict__ sb_off,
    double* __restrict__ pval, uint16_t* __restrict__ pcode, uint32_t* __restrict__ prc,
    const int64_t* __restrict__ bstart, const unsigned long long* __restrict__ pc,
    const uint16_t* __restrict__ rid /* popularity ids (slot * P + part) instead of the column-map gather, or null */,
    uint32_t magicP /* ceil(2^32 / P) */, int nsplit /* ids below this may be split columns */, const uint8_t* __restrict__ lgtab) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fl_lds[];
  uint32_t* const cnt = reinterpret_cast<uint32_t*>(fl_lds);                   // [row slot][part]: counts, then write cursors
  int64_t* const rstart = reinterpret_cast<int64_t*>(cnt + ((R * P + 1) & ~1));   // [row slot] first entry of the row in the CSR
  int32_t* const rlen = reinterpret_cast<int32_t*>(rstart + R);                // [row slot] its length (0: empty slot)
  __shared__ uint32_t total[8], lastrow[8];
  __shared__ int64_t sbase[8];
  const int64_t b = blockIdx.x;
  const int sub = threadIdx.x / RS_SUB, lane = threadIdx.x % RS_SUB, subs = blockDim.x / RS_SUB;
  for (int t = threadIdx.x; t < R * P; t += blockDim.x) cnt[t] = 0;
  for (int lr = threadIdx.x; lr < R; lr += blockDim.x) {
    const int64_t i = amb_row[b * R + lr];
    const int64_t s0 = i >= 0 ? indptr[i] : 0;
    rstart[lr] = s0; rlen[lr] = i >= 0 ? (int32_t)(indptr[i + 1] - s0) : 0;
  }
  if (threadIdx.x < P) sbase[threadIdx.x] = sb_off[b * P + threadIdx.x];
  __syncthreads();
  if (pc) {                                                // the per-row part counts are already known
    const int64_t a0 = bstart[b], n = bstart[b + 1] - a0;
    for (int lr = threadIdx.x; lr < n; lr += blockDim.x) {
      const unsigned long long lo = pc[2 * (a0 + lr)], hi = pc[2 * (a0 + lr) + 1];
      for (int q = 0; q < P; ++q) cnt[lr * P + q] = (uint32_t)(((q < 4 ? lo : hi) >> (16 * (q & 3))) & 0xFFFF);
    }
  } else {
    for (int lr = sub; lr < R; lr += subs) {
      const int64_t s0 = rstart[lr];
      for (int k = lane; k < rlen[lr]; k += RS_SUB) atomicAdd(&cnt[lr * P + (colmap[indices[s0 + k]] >> 16)], 1u);
    }
  }
  __syncthreads();
  if (threadIdx.x < P) {                                   // exclusive scan down the rows, one thread per part
    uint32_t run = 0, last = 0;
    for (int lr = 0; lr < R; ++lr) {
      const uint32_t c = cnt[lr * P + threadIdx.x];
      cnt[lr * P + threadIdx.x] = run;
      if (c) last = (uint32_t)lr;
      run += c;
    }
    total[threadIdx.x] = run; lastrow[threadIdx.x] = last;
  }
  __syncthreads();
  // A row's entries keep their CSR order inside each part: the position of an entry = the row's cursor for its part
  // (read-only after the scan) + the number of earlier entries of the row in that part, counted with wave ballots over
  // the 16 lanes that walk the row — no LDS atomic per entry (2e9 of them with a return value were half of this
  // kernel's time), and the layout is the same from run to run.
  const int sgbase = (threadIdx.x & 63) / RS_SUB * RS_SUB;
  const uint32_t below = (1u << lane) - 1u;
  // one step of 16 entries: column-map word cm of this lane's entry (0xFFFF.... part for lanes past the row's end), its score
  auto place = [&](int lr, uint32_t (&run)[8], bool valid, uint32_t cm, uint32_t code) {
    const uint32_t p = valid ? cm >> 16 : 0xFFFFu;
    uint32_t t = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (q < P) {
        const uint32_t m = (uint32_t)(__ballot(p == (uint32_t)q) >> sgbase) & 0xFFFFu;
        if (p == (uint32_t)q) t = run[q] + __popc(m & below);
        run[q] += __popc(m);
      }
    }
    if (valid) {
      t += cnt[lr * P + p];
      const int64_t pos = sbase[p] + t;
      if (pcode) pcode[pos] = (uint16_t)code;
      else pval[pos] = lut[code];
      prc[pos] = ((uint32_t)lr << 16) | ((cm & 0x1FFFu) + (t & ((1u << ((cm >> 13) & 7u)) - 1u)));   // hot column: deal over its slots
    }
  };
  auto cm_of_id = [&](uint32_t id) -> uint32_t {           // the ids k_row_partcounts wrote: a coalesced 2-byte read instead of a gather
    const uint32_t slot = P == 1 ? id : __umulhi(id, magicP);   // id / P, exact for 16-bit ids and 2 <= P <= 8 (ceil(2^32 / 1) does not fit 32 bits)
    const uint32_t lg = (int)id < nsplit ? (uint32_t)lgtab[id] : 0u;
    return ((id - slot * (uint32_t)P) << 16) | (lg << 13) | slot;
  };
  if (rid) {
    constexpr int PF = 4;                                  // steps of a row loaded ahead (64 entries)
    uint32_t nid[PF], nrw[PF];
    auto load_row = [&](int lr) {
      const int64_t s0 = rstart[lr < R ? lr : R - 1];
#pragma unroll
      for (int j = 0; j < PF; ++j) { nid[j] = rid[s0 + 16 * j + lane]; nrw[j] = raw[s0 + 16 * j + lane]; }
    };
    load_row(sub);
    for (int lr = sub; lr < R; lr += subs) {
      uint32_t cid[PF], crw[PF];
#pragma unroll
      for (int j = 0; j < PF; ++j) { cid[j] = nid[j]; crw[j] = nrw[j]; }
      load_row(lr + subs);                                 // (past the block's last row: the last row again, unused)
      const int len = rlen[lr];
      const int64_t s0 = rstart[lr];
      uint32_t run[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int j = 0; j < PF; ++j)
        if (16 * j < len) place(lr, run, 16 * j + lane < len, cm_of_id(cid[j]), crw[j]);   // (uniform over the 16 lanes)
      for (int k0 = 16 * PF; k0 < len; k0 += RS_SUB) {      // the rest of a long row
        const bool valid = k0 + lane < len;
        const uint32_t id = valid ? (uint32_t)rid[s0 + k0 + lane] : 0u;
        place(lr, run, valid, cm_of_id(id), valid ? (uint32_t)raw[s0 + k0 + lane] : 0u);
      }
    }
  } else {
    for (int lr = sub; lr < R; lr += subs) {
      const int len = rlen[lr];
      const int64_t s0 = rstart[lr];
      uint32_t run[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int k0 = 0; k0 < len; k0 += RS_SUB) {           // (all 16 lanes stay in the loop: the ballots need them)
        const bool valid = k0 + lane < len;
        place(lr, run, valid, valid ? colmap[indices[s0 + k0 + lane]] : 0u, valid ? (uint32_t)raw[s0 + k0 + lane] : 0u);
      }
    }
  }
  for (int p = 0; p < P; ++p) {                            // padding: value 0 (buffers are zero-filled), row = last row
    const int64_t base = sbase[p], end = sb_off[b * P + p + 1];
    for (int64_t pos = base + total[p] + threadIdx.x; pos < end; pos += blockDim.x) prc[pos] = lastrow[p] << 16;
  }
}

// Conflict-aware entry order INSIDE the rows of the row-ordered layout (score codes).  The column scatter of the
// fused kernel (ds_add_f64 into the part's accumulators) is served in four groups of 16 lanes, each at 2 clk x the
// largest number of lanes whose slots agree modulo 16 (tools/ubench/lds.hip: 16 slots distinct mod 32 but pairwise
// equal mod 16 cost 16 clk, distinct mod 16 cost 8, random 24.5); SQ_LDS_BANK_CONFLICT is a third of the LDS-array
// cycles of the pass (profiles/r02_lds_counters.txt).  One such group and instruction covers the entries at
// positions = j (mod 4) of a WINDOW of 64 consecutive entries (16 lanes x 4 entries; sub-blocks are padded to 64,
// so windows never straddle them).  Any order inside a row is valid, so every window is walked once and each
// position takes, among itself and the next two entries of the same row, the one whose slot class is rarest so
// far in its (window, j) bin: mean worst multiplicity 3.2 -> 2.3 (two look-ahead entries already give what a search
// over the whole row gives; the bound for a fixed row order is ~2.1, a window holds ~8 entries of its most popular
// class).  One thread per window, windows are independent; the walk is fully unrolled, so every index is static and
// the 64 four-bit classes, the row boundaries, the four 16-counter histograms and the permutation itself are
// packed in registers — no LDS, no memory access, no divergent loop.  The permutation is applied while copying
// the window into fresh arrays.  The padding at the end of a sub-block (code 0) stays where it is.
constexpr int DC_NT = 256;
// Round 3: the window lives in REGISTERS (64 packed row/column words + 64 codes) and the chosen entry is swapped into
// place with selects — every position is a compile-time constant of the unrolled walk.  The round-2 version built a
// permutation and applied it with 128 scattered loads per window afterwards: 8192 vector-cache address cycles per
// wave against 1536 for reading the window, which was most of its 14 ms at 2e9 entries.
__global__ __launch_bounds__(DC_NT) void k_sb_deconflict(int64_t n_win, const uint32_t* __restrict__ prc_in, const uint16_t* __restrict__ code_in,
                                                          uint32_t* __restrict__ prc_out, uint16_t* __restrict__ code_out) {
  for (int64_t w = (int64_t)blockIdx.x * DC_NT + threadIdx.x; w < n_win; w += (int64_t)gridDim.x * DC_NT) {
    const int64_t base = w * 64;
    const uint4* pin = reinterpret_cast<const uint4*>(prc_in + base);
    const uint2* cin = reinterpret_cast<const uint2*>(code_in + base);
    uint32_t P[64], Cd[64];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const uint4 a = pin[q];
      const uint2 cd = cin[q];
      P[4 * q] = a.x; P[4 * q + 1] = a.y; P[4 * q + 2] = a.z; P[4 * q + 3] = a.w;
      Cd[4 * q] = cd.x & 0xFFFFu; Cd[4 * q + 1] = cd.x >> 16; Cd[4 * q + 2] = cd.y & 0xFFFFu; Cd[4 * q + 3] = cd.y >> 16;
    }
    unsigned long long cont = 0ull;                        // bit i: entry i continues the row of entry i-1 (and neither is padding)
#pragma unroll
    for (int i = 1; i < 64; ++i)
      if (Cd[i] != 0u && Cd[i - 1] != 0u && (P[i] >> 16) == (P[i - 1] >> 16)) cont |= 1ull << i;
    unsigned long long h[4] = {0ull, 0ull, 0ull, 0ull};   // per instruction slot j: 16 four-bit counters, one per class
#pragma unroll
    for (int pos = 0; pos < 64; ++pos) {
      const int j = pos & 3;
      const uint32_t c0 = P[pos] & 15u;
      uint32_t best = 0u, cb = c0, lb = (uint32_t)(h[j] >> (c0 * 4u)) & 15u;
      if (pos + 1 < 64) {
        const bool ok1 = (cont >> (pos + 1)) & 1ull;
        const uint32_t c1 = P[pos + 1] & 15u;
        const uint32_t l1 = (uint32_t)(h[j] >> (c1 * 4u)) & 15u;
        if (ok1 && l1 < lb) { best = 1u; cb = c1; lb = l1; }
        if (pos + 2 < 64) {
          const bool ok2 = ok1 && ((cont >> (pos + 2)) & 1ull);
          const uint32_t c2 = P[pos + 2] & 15u;
          const uint32_t l2 = (uint32_t)(h[j] >> (c2 * 4u)) & 15u;
          if (ok2 && l2 < lb) { best = 2u; cb = c2; lb = l2; }
        }
        // swap entry pos with entry pos + best (same row: the continuation bits stay valid)
        const uint32_t tp = P[pos], tc = Cd[pos];
        if (pos + 2 < 64) {
          P[pos] = best == 1u ? P[pos + 1] : (best == 2u ? P[pos + 2] : tp);
          Cd[pos] = best == 1u ? Cd[pos + 1] : (best == 2u ? Cd[pos + 2] : tc);
          P[pos + 2] = best == 2u ? tp : P[pos + 2];
          Cd[pos + 2] = best == 2u ? tc : Cd[pos + 2];
        } else {
          P[pos] = best == 1u ? P[pos + 1] : tp;
          Cd[pos] = best == 1u ? Cd[pos + 1] : tc;
        }
        P[pos + 1] = best == 1u ? tp : P[pos + 1];
        Cd[pos + 1] = best == 1u ? tc : Cd[pos + 1];
      }
      h[j] += (unsigned long long)(lb < 15u ? 1u : 0u) << (cb * 4u);
    }
    uint4* pout = reinterpret_cast<uint4*>(prc_out + base);
    uint2* cout = reinterpret_cast<uint2*>(code_out + base);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      pout[q] = make_uint4(P[4 * q], P[4 * q + 1], P[4 * q + 2], P[4 * q + 3]);
      cout[q] = make_uint2(Cd[4 * q] | (Cd[4 * q + 1] << 16), Cd[4 * q + 2] | (Cd[4 * q + 3] << 16));
    }
  }
}

// ============================================================================
// EM hot loop — two-pass form (phase 1: partial row sums; phase 2: scatter)
// ============================================================================
// WG = (part p, stripe g).  LDS: ctab[Kp] | y[R]
template <int NT>
__global__ __launch_bounds__(NT) void k_phase1(int P, int Kp, int R, int64_t b0, int64_t nb, int G, int64_t N_amb_pad,
    const int64_t* __restrict__ sb_off, const double* __restrict__ pval, const uint32_t* __restrict__ prc,
    const double* __restrict__ ctab, double* __restrict__ ypart, const uint32_t* __restrict__ ctl) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (ctl && __hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;   // the run has stopped (tsem_em_chunk)
  double* c = reinterpret_cast<double*>(smem);
  double* y = c + Kp;
  const int p = blockIdx.x % P, g = blockIdx.x / P;
  for (int t = threadIdx.x; t < Kp; t += NT) c[t] = ctab[p * Kp + t];
  for (int t = threadIdx.x; t < R; t += NT) y[t] = 0.0;
  __syncthreads();
  for (int64_t b = b0 + g; b < nb; b += G) {
    const int64_t q0 = sb_off[b * P + p] >> 2, q1 = sb_off[b * P + p + 1] >> 2;
    for (int64_t q = q0 + threadIdx.x; q < q1; q += NT) {
      uint4 rc = reinterpret_cast<const uint4*>(prc)[q];
      double2 v0 = reinterpret_cast<const double2*>(pval)[2 * q];
      double2 v1 = reinterpret_cast<const double2*>(pval)[2 * q + 1];
      lds_add(&y[rc.x >> 16], v0.x * c[rc.x & 0xFFFF]);
      lds_add(&y[rc.y >> 16], v0.y * c[rc.y & 0xFFFF]);
      lds_add(&y[rc.z >> 16], v1.x * c[rc.z & 0xFFFF]);
      lds_add(&y[rc.w >> 16], v1.y * c[rc.w & 0xFFFF]);
    }
    __syncthreads();
    double* out = ypart + (int64_t)p * N_amb_pad + b * R;
    for (int t = threadIdx.x; t < R; t += NT) { out[t] = y[t]; y[t] = 0.0; }
    __syncthreads();
  }
}

// LDS: ctab[Kp] | acc[Kp] | s[R].  thetasum partials -> partial[g][p*Kp + l]
template <int NT>
__global__ __launch_bounds__(NT) void k_phase2_em(int P, int Kp, int R, int64_t b0, int64_t nb, int G, int accumulate, int64_t N_amb_pad,
    const int64_t* __restrict__ sb_off, const double* __restrict__ pval, const uint32_t* __restrict__ prc,
    const double* __restrict__ ctab, const double* __restrict__ ypart, const uint16_t* __restrict__ wcode,
    const double* __restrict__ lut, double* __restrict__ partial, const uint32_t* __restrict__ ctl) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (ctl && __hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
  double* c = reinterpret_cast<double*>(smem);
  double* acc = c + Kp;
  double* s = acc + Kp;
  const int p = blockIdx.x % P, g = blockIdx.x / P;
  {
    const double* prev = partial + (int64_t)g * (P * Kp) + p * Kp;
    for (int t = threadIdx.x; t < Kp; t += NT) { c[t] = ctab[p * Kp + t]; acc[t] = accumulate ? prev[t] : 0.0; }
  }
  for (int64_t b = b0 + g; b < nb; b += G) {
    __syncthreads();
    for (int t = threadIdx.x; t < R; t += NT) {
      int64_t a = b * R + t;
      double ys = 0.0;
      for (int pp = 0; pp < P; ++pp) ys += ypart[(int64_t)pp * N_amb_pad + a];
      // z = n * recip0(rowsum) (sparse_plus.py:52), weighted by w_i (model.py:730)
      s[t] = recip0(ys) * lut[wcode[a]];
    }
    __syncthreads();
    const int64_t q0 = sb_off[b * P + p] >> 2, q1 = sb_off[b * P + p + 1] >> 2;
    for (int64_t q = q0 + threadIdx.x; q < q1; q += NT) {
      uint4 rc = reinterpret_cast<const uint4*>(prc)[q];
      double2 v0 = reinterpret_cast<const double2*>(pval)[2 * q];
      double2 v1 = reinterpret_cast<const double2*>(pval)[2 * q + 1];
      lds_add(&acc[rc.x & 0xFFFF], (v0.x * c[rc.x & 0xFFFF]) * s[rc.x >> 16]);
      lds_add(&acc[rc.y & 0xFFFF], (v0.y * c[rc.y & 0xFFFF]) * s[rc.y >> 16]);
      lds_add(&acc[rc.z & 0xFFFF], (v1.x * c[rc.z & 0xFFFF]) * s[rc.z >> 16]);
      lds_add(&acc[rc.w & 0xFFFF], (v1.y * c[rc.w & 0xFFFF]) * s[rc.w >> 16]);
    }
  }
  __syncthreads();
  double* out = partial + (int64_t)g * (P * Kp) + p * Kp;
  for (int t = threadIdx.x; t < Kp; t += NT) out[t] = acc[t];
}

// lnl over ambiguous rows: sum z(prev) * log1p(Q * c_cur)   (model.py:755-758)
// LDS: c_old[Kp] | c_new[Kp] | r[R]
template <int NT>
__global__ __launch_bounds__(NT) void k_phase2_lnl(int P, int Kp, int R, int64_t nb, int G, int64_t N_amb_pad,
    const int64_t* __restrict__ sb_off, const double* __restrict__ pval, const uint32_t* __restrict__ prc,
    const double* __restrict__ ctab_old, const double* __restrict__ ctab_new, const double* __restrict__ ypart,
    double* __restrict__ lnl_part) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double scratch[16];
  double* co = reinterpret_cast<double*>(smem);
  double* cn = co + Kp;
  double* r = cn + Kp;
  const int p = blockIdx.x % P, g = blockIdx.x / P;
  for (int t = threadIdx.x; t < Kp; t += NT) { co[t] = ctab_old[p * Kp + t]; cn[t] = ctab_new[p * Kp + t]; }
  double acc = 0.0;
  for (int64_t b = g; b < nb; b += G) {
    __syncthreads();
    for (int t = threadIdx.x; t < R; t += NT) {
      int64_t a = b * R + t;
      double ys = 0.0;
      for (int pp = 0; pp < P; ++pp) ys += ypart[(int64_t)pp * N_amb_pad + a];
      r[t] = recip0(ys);
    }
    __syncthreads();
    const int64_t e0 = sb_off[b * P + p], e1 = sb_off[b * P + p + 1];
    for (int64_t e = e0 + threadIdx.x; e < e1; e += NT) {
      uint32_t rc = prc[e];
      double v = pval[e];
      double z = (v * co[rc & 0xFFFF]) * r[rc >> 16];
      if (z != 0.0) acc += z * ts_log1p_pos(v * cn[rc & 0xFFFF]);
    }
  }
  double t = block_sum(acc, scratch);
  if (threadIdx.x == 0) lnl_part[blockIdx.x] = t;
}

// lnl over unique rows: z = n * recip0(n), n = Q*pi_prev; inner = Q*pi_cur
__global__ __launch_bounds__(256) void k_lnl_unique(int64_t N_uni, const int32_t* __restrict__ ucol,
    const uint16_t* __restrict__ ucode, const double* __restrict__ lut, const double* __restrict__ pi_old,
    const double* __restrict__ pi_new, double* __restrict__ lnl_part) {
  __shared__ double scratch[16];
  double acc = 0.0;
  for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < N_uni; u += (int64_t)gridDim.x * blockDim.x) {
    int col = ucol[u];
    double q = lut[ucode[u]];
    double n = q * pi_old[col];
    if (n != 0.0) {
      double z = n * recip0(n);
      if (z != 0.0) acc += z * ts_log1p_pos(q * pi_new[col]);
    }
  }
  double t = block_sum(acc, scratch);
  if (threadIdx.x == 0) lnl_part[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void k_sum_parts(const double* __restrict__ a, int na, const double* __restrict__ b,
                                                   int nbb, double* __restrict__ out) {
  __shared__ double scratch[16];
  double acc = 0.0;
  for (int t = threadIdx.x; t < na; t += blockDim.x) acc += a[t];
  for (int t = threadIdx.x; t < nbb; t += blockDim.x) acc += b[t];
  double t = block_sum(acc, scratch);
  if (threadIdx.x == 0) out[0] = t;
}

// red[col] = sum_g partial[g][pc]   (fixed order -> deterministic given partials).  A block handles 64
// slots x 4 interleaved slices of the team axis, so the strided reads of one slot overlap instead of
// forming a chain of G dependent loads.  `sync` (fused kernel): only the teams that formed wrote
// their slice — G = sum over XCDs of floor(tickets / P).
// red[K] = 1 when this rank's fused pass raised its watchdog error word (the flag is summed with the column
// sums by the all-reduce, so every rank's update kernel sees that SOME rank failed), red[K+1] = 0.
__global__ __launch_bounds__(256) void k_colreduce(int Kpad, int G, const double* __restrict__ partial,
                            const int32_t* __restrict__ col_of_pc, const uint32_t* __restrict__ colmap,
                            double* __restrict__ red, int K, const uint32_t* __restrict__ sync, int P,
                            const uint32_t* __restrict__ ctl, unsigned long long* __restrict__ fz_xchg, int64_t fz_xchg_n) {
  // the fused kernel's exchange ring must be zero at its next launch: cleared here, by the ~950 blocks of the
  // kernel that follows every fused pass (no launch of its own, no fence: the next fused launch is a kernel boundary away)
  if (fz_xchg) {
    typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
    u64x2_t* const x2 = reinterpret_cast<u64x2_t*>(fz_xchg);                  // (hipMalloc alignment; the ring holds an even number of words)
    const u64x2_t zz = {0ull, 0ull};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < fz_xchg_n / 2; i += (int64_t)gridDim.x * blockDim.x)
      x2[i] = zz;
  }
  // 32 slots x 8 interleaved slices of the team axis per block: 8 independent loads per thread in flight (the
  // kernel is latency-bound: 64 teams x 30k slots = 15 MB; 13.6 us with 4 slices of 16 loads, r01 profile)
  constexpr int NS = 8, NC = 32;
  __shared__ double part[NS][NC];
  if (ctl && __hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;   // the run has stopped
  if (sync) {
    int t = 0;
    for (int x = 0; x < 8; ++x) t += (int)(sync[x] / (uint32_t)P);
    G = min(G, t);
  }
  const int pcl = threadIdx.x % NC, slice = threadIdx.x / NC;
  const int pc = blockIdx.x * NC + pcl;
  if (blockIdx.x == 0 && threadIdx.x == 0) { red[K] = (sync && sync[9]) ? 1.0 : 0.0; red[K + 1] = 0.0; }
  const int col = pc < Kpad ? col_of_pc[pc] : -1;        // -1: padding, or a secondary slot of a split column
  double s = 0.0;
  if (col >= 0) {
    const int copies = 1 << ((colmap[col] >> 13) & 7u);
    if (copies == 1) {
#pragma unroll 8
      for (int g = slice; g < G; g += NS) s += partial[(int64_t)g * Kpad + pc];
    } else {
      for (int g = slice; g < G; g += NS)
        for (int c = 0; c < copies; ++c) s += partial[(int64_t)g * Kpad + pc + c];
    }
  }
  part[slice][pcl] = s;
  __syncthreads();
  if (slice == 0 && col >= 0)                            // fixed order -> deterministic given the partials
    red[col] = ((part[0][pcl] + part[1][pcl]) + (part[2][pcl] + part[3][pcl])) + ((part[4][pcl] + part[5][pcl]) + (part[6][pcl] + part[7][pcl]));
}

__global__ void k_keep_err(const uint32_t* sync, uint32_t* errlog) { errlog[0] |= sync[9]; errlog[1] = sync[10]; }

// Loop control of tsem_em_chunk, evaluated on the device so that the host need not synchronise every
// iteration: ctl[0] = stop flag (0 run, 1 converged, 2 a rank's EM pass timed out), ctl[1] = iterations
// committed since the host last cleared it.
struct UpdCtl {
  uint32_t* ctl;          // null: legacy stepwise use (always commit unless the error flag is up)
  double eps;             // converged = diff_est < eps (model.py:792) unless use_lnl
  int use_lnl;            // convergence is decided by k_lnl_check instead (model.py:785-789)
  double* pi_first;       // non-null: also store the new parameters here (pi_init / theta_init, model.py:776-778)
  double* theta_first;
};

// M-step closed forms (model.py:733-740) + per-block partials of diff_est (model.py:781)
__global__ __launch_bounds__(256) void k_update(UpdCtl C, int K, const double* __restrict__ red,
    const double* __restrict__ pisum0, double theta_pw, double theta_den, double pi_pw, double pi_den,
    double* __restrict__ pi, double* __restrict__ theta, double* __restrict__ pi_prev,
    double* __restrict__ theta_prev, const uint32_t* __restrict__ colmap, int Kp,
    double* __restrict__ ctab, double* __restrict__ ctab_prev, const int32_t* __restrict__ twin_rep,
    double* __restrict__ diff_part, double* __restrict__ diff_out, uint32_t* __restrict__ done,
    uint32_t* __restrict__ fz_sync, uint32_t* __restrict__ fz_errlog, unsigned long long* __restrict__ fz_xchg, int64_t fz_xchg_n) {
  __shared__ double scratch[16];
  __shared__ bool last;
  // the fused kernel's sync words and exchange ring must be zero at its next launch: do it here (this
  // kernel runs once per EM pass, after the pass) instead of three memsets in front of every launch
  if (fz_sync) {
    (void)fz_xchg; (void)fz_xchg_n;                        // (the ring itself is cleared by k_colreduce)
    if (blockIdx.x == 0 && threadIdx.x < 16) {            // FZ_SYNC_WORDS
      if (threadIdx.x == 9) atomicOr(&fz_errlog[0], fz_sync[9]);          // keep the error word / miss counter for the host
      if (threadIdx.x == 10) fz_errlog[1] = fz_sync[10];
      fz_sync[threadIdx.x] = 0u;
    }
  }
  if (C.ctl && __hip_atomic_load(C.ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;   // the run has stopped: nothing to commit
  // some rank's fused pass timed out (flag summed by the all-reduce): the column sums are incomplete, so NO
  // rank commits; the last block raises stop = 2 and the host redoes the iteration (tsem_em_chunk)
  const bool failed = red[K] > 0.0;
  double d = 0.0;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < K && !failed; j += gridDim.x * blockDim.x) {
    // exact twin columns share one accumulation (see k_colsig) as long as their
    // sums agree to rounding, i.e. their parameters are still symmetric
    const int jr = twin_rep[j];
    double ts = red[j];
    if (jr != j) {
      double tr = red[jr];
      if (fabs(ts - tr) <= 1e-12 * fmax(fabs(ts), fabs(tr))) ts = tr;
    }
    double th = (ts + theta_pw) / theta_den;
    double ps = pisum0[j] + ts;
    double ph = (ps + pi_pw) / pi_den;
    double po = pi[j], to = theta[j];
    d += fabs(ph - po);
    pi_prev[j] = po; theta_prev[j] = to;
    pi[j] = ph; theta[j] = th;
    if (C.pi_first) { C.pi_first[j] = ph; C.theta_first[j] = th; }
    const uint32_t cm = colmap[j];
    const int pc = (int)(cm >> 16) * Kp + (int)(cm & 0x1FFFu), copies = 1 << ((cm >> 13) & 7u);
    const double cold = ctab[pc], cnew = ph * th;
    for (int c = 0; c < copies; ++c) { ctab_prev[pc + c] = cold; ctab[pc + c] = cnew; }
  }
  double t = block_sum(d, scratch);
  if (threadIdx.x == 0) {
    diff_part[blockIdx.x] = t;
    __threadfence();
    last = atomicAdd(done, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (last) {                                              // fixed order -> deterministic
    __threadfence();
    double v = 0.0;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += blockDim.x) v += __hip_atomic_load(&diff_part[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    double tt = block_sum(v, scratch);
    if (threadIdx.x == 0) {
      *done = 0u;
      if (failed) {
        if (C.ctl) C.ctl[0] = 2u;
        *diff_out = -1.0;                                  // (legacy stepwise hosts: negative = not committed)
      } else {
        *diff_out = tt;
        if (C.ctl) {
          C.ctl[1] += 1u;
          if (!C.use_lnl && tt < C.eps) C.ctl[0] = 1u;     // model.py:792
        }
      }
    }
  }
}

// use_likelihood convergence test (model.py:785-789): lred[0] = all-reduced log-likelihood of the iteration just
// committed, lred[1] > 0 when some rank's lnl pass timed out.
__global__ void k_lnl_check(uint32_t* ctl, double* ctld, const double* __restrict__ lred, double eps,
                            double* __restrict__ lnl_out) {
  if (ctl[0]) return;
  if (lred[1] > 0.0) { ctl[0] = 3u; return; }
  const double l = lred[0];
  *lnl_out = l;
  if (fabs(l - ctld[0]) < eps) ctl[0] = 1u;
  ctld[0] = l;
}
// lnl partial + error flag of this rank into the two lnl reduce slots
__global__ void k_lnl_slots(const double* __restrict__ red_lnl, const uint32_t* __restrict__ sync, double* __restrict__ lred) {
  lred[0] = *red_lnl;
  lred[1] = (sync && sync[9]) ? 1.0 : 0.0;
}

__global__ void k_make_ctab(int K, const double* __restrict__ pi, const double* __restrict__ theta,
                            const uint32_t* __restrict__ colmap, int Kp, double* __restrict__ ctab) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= K) return;
  const uint32_t cm = colmap[j];
  const int pc = (int)(cm >> 16) * Kp + (int)(cm & 0x1FFFu), copies = 1 << ((cm >> 13) & 7u);
  for (int c = 0; c < copies; ++c) ctab[pc + c] = pi[j] * theta[j];
}

__global__ void k_row_weights(int64_t n, const uint16_t* __restrict__ code, const double* __restrict__ lut,
                              double* __restrict__ w) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) w[i] = lut[code[i]];
}

__global__ void k_fill(double* p, int64_t n, double v) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

#include "tsem_fused.h"
static_assert(FZ_SYNC_WORDS == 16, "k_update clears 16 sync words");

// ============================================================================
// CSR row passes: z export, best-hit counts, reassign (model.py:808-865)
// ============================================================================
enum { RP_EXPORT_Z = 0, RP_BEST = 1, RP_REASSIGN = 2, RP_REPORT = 3 };
constexpr int RP_SUB = 16;

// Option "reproducible": values in [0, 2) — posteriors, shares of a tie — cut into a multiple of 2^-26 and the rest on the 2^-53
// grid: up to 2^26 of either add exactly in fp64, whatever order the atomics are served in; the two sums are added once at the end.
__device__ __forceinline__ void exact_split01(double v, double& hi, double& lo) {
  const double m1 = 100663296.0;                            // 1.5 * 2^26: ulp 2^-26
  hi = (v + m1) - m1;
  lo = ((v - hi) + 0.75) - 0.75;                            // ulp(0.75) = 2^-53
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
  const int32_t* group;  // REASSIGN: optional row -> group map; colsums is then [n_groups][K]
  const int32_t* rowlist; int64_t nlist;   // REASSIGN: optional list of rows to visit (picks[] is then indexed by list position)
  // REPORT: conf, exclude and average in ONE pass -> colsums[0..K), [K..2K), [2K..3K); best-hit counts -> nbest
  // REASSIGN without groups: the Hs most popular slots of every column part are summed in LDS per
  // workgroup and flushed once (global fp64 atomics: 22 G/s, 2 G/s on a popular column)
  const uint32_t* colmap; const int32_t* col_of_pc; int P, Kp, Hs;
  double* colsums_lo = nullptr;   // option "reproducible": the low pieces of every value (same shape as colsums; no LDS slots then)
};

// METH >= 0 fixes the reassign method at compile time (the per-entry switch and the reductions a method does not
// need disappear: the pass is bound by instruction issue, ~300 per four rows); METH = -1 reads it from the arguments.
template <int MODE, int METH = -1>
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
  const int64_t n_visit = (MODE == RP_REASSIGN && A.rowlist) ? A.nlist : A.N;
  for (int64_t idx = (int64_t)blockIdx.x * subs + sub; idx < n_visit; idx += (int64_t)gridDim.x * subs) {
    const int64_t row = (MODE == RP_REASSIGN && A.rowlist) ? (int64_t)A.rowlist[idx] : idx;
    const int64_t s = A.indptr[row], e = A.indptr[row + 1];
    const bool amb = (e - s) > 1;
    // one value of report column m (0 for a plain reassign) for column `col`: popular columns in LDS, the rest global
    auto emit = [&](int m, int col, uint32_t cm, double val, int64_t grp_off) {
      if (A.colsums_lo) {
        double hi, lo;
        exact_split01(val, hi, lo);
        unsafeAtomicAdd(&A.colsums[(int64_t)m * A.K + grp_off + col], hi);
        if (lo != 0.0) unsafeAtomicAdd(&A.colsums_lo[(int64_t)m * A.K + grp_off + col], lo);
      } else if (nhot1 && (int)(cm & 0x1FFFu) < A.Hs) lds_add(&hot[m * nhot1 + (cm >> 16) * A.Hs + (cm & 0x1FFFu)], val);
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
      const double r = A.zin ? 1.0 : rs;                    // the caller's z is used as is (model.py:837)
      double zmax = -1.0; int cnt = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) if (inp[i]) { zmax = fmax(zmax, n[i] * r); ++cnt; }
      zmax = sg_max<RP_SUB>(zmax);
      cnt = sg_sum_i<RP_SUB>(cnt);
      if (MODE == RP_EXPORT_Z) {
#pragma unroll
        for (int i = 0; i < 4; ++i) if (vld[i]) A.zout[s + lane + i * RP_SUB] = inp[i] ? n[i] * r : -1.0;
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
      const int64_t grp_off = A.group ? (A.group[row] < 0 ? -1 : (int64_t)A.group[row] * A.K) : 0;
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
          if (A.zout) A.zout[k] = val;
          if (val != 0.0 && grp_off >= 0) {
            const int col = A.indices[k];
            emit(0, col, nhot1 ? A.colmap[col] : 0xFFFFFFFFu, val, grp_off);
          }
        }
      }
      continue;
    }
    // sweep 1: row sum
    double y = 0.0;
    for (int64_t k = s + lane; k < e; k += RP_SUB) { const double v = numer(k); y += (A.zin && isnan(v)) ? 0.0 : v; }
    y = sg_sum<RP_SUB>(y);
    const double r = A.zin ? 1.0 : recip0(y);
    // sweep 2: row max over z's pattern
    double zmax = -1.0;
    int cnt = 0;
    for (int64_t k = s + lane; k < e; k += RP_SUB) {
      double n = numer(k);
      bool inpat = A.zin ? !isnan(n) : (initial || (n != 0.0));
      if (inpat) { zmax = fmax(zmax, n * r); ++cnt; }
    }
    zmax = sg_max<RP_SUB>(zmax);
    cnt = sg_sum_i<RP_SUB>(cnt);
    if (MODE == RP_EXPORT_Z) {
      for (int64_t k = s + lane; k < e; k += RP_SUB) {
        double n = numer(k);
        bool inpat = A.zin ? !isnan(n) : (initial || (n != 0.0));
        A.zout[k] = inpat ? n * r : -1.0;   // -1 marks an entry the reference drops from z's pattern
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
    const int64_t grp_off = A.group ? (A.group[row] < 0 ? -1 : (int64_t)A.group[row] * A.K) : 0;
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
        if (A.zout) A.zout[k] = val;
        if (val != 0.0 && grp_off >= 0) {
          const int col = A.indices[k];
          emit(0, col, nhot1 ? A.colmap[col] : 0xFFFFFFFFu, val, grp_off);
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
  int32_t* defer_rows; unsigned long long* defer_n;   // rows left to k_report_slow
  int dbg;                             // timing experiments (wrong results): 1 drop the emits that miss the LDS slots, 2 the row-count stores, 4 the ties
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
template <int G> __device__ __forceinline__ int rr_sum_i(int v) {
  if (G >= 2) v += fz_dpp_i<RR_QP_X1, 0xF>(v, v);
  if (G >= 4) v += fz_dpp_i<RR_QP_X2, 0xF>(v, v);
  if (G >= 8) v += fz_dpp_i<RR_HALF_MIRROR, 0xF>(v, v);
  if (G >= 16) v += fz_dpp_i<RR_MIRROR, 0xF>(v, v);
  return v;
}

struct ReportEmit {                                        // where a row's values go (both report kernels), by id
  const ReportArgs& A; double* hotF; uint32_t* hot1; uint32_t* hot2; int Hs; double* hotL;
  __device__ __forceinline__ void conf(uint32_t id, double v) const {
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
  __device__ __forceinline__ void one(uint32_t id) const {           // the row's only best hit
    if ((int)id < Hs) atomicAdd(&hot1[id], 1u);
    else if (!(A.dbg & 1)) unsafeAtomicAdd(&A.g_n1[id], 1.0);
  }
  __device__ __forceinline__ void tie(uint32_t id, int nb, double share) const {   // one of nb > 1 best hits
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

// threads per workgroup: the pass over the final z needs ~92 VGPRs (pi*theta gathers in flight) and spills under the 128 of a 1024-thread
// workgroup (15.6 vs 5.3 ms); the pass over the initial z needs 68 and gains from 16 waves per CU instead of 8 (7.8 -> 6.9 ms with its tie list)
constexpr int rr_nt(bool init) { return init ? 1024 : 512; }
template <int G, int E, bool INIT>
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
  const ReportEmit EM{A, hotF, hot1, hot2, A.Hs, hotL};
  const int gl = threadIdx.x % G, grp = threadIdx.x / G, ngrp = blockDim.x / G;
  const int64_t stride = (int64_t)gridDim.x * ngrp;
  const int64_t nit = (A.N + stride - 1) / stride;
  const bool one_winner = A.thresh > 0.51;                // an entry with z >= thresh is then the row's unique best hit
  struct Ip { int64_t s; int len; };
  struct Ent { rr_u32x4_a2 id[E / 8]; rr_u32x4_a2 cd[E / 8]; };   // 8 ids / 8 codes per 16-byte word
  auto load_ip = [&](int64_t it) -> Ip {
    const int64_t row = it * stride + (int64_t)blockIdx.x * ngrp + grp;
    const int64_t rc = row < A.N ? row : A.N - 1;        // clamped, never branched around
    const rr_i64x2_a8 se = *reinterpret_cast<const rr_i64x2_a8*>(A.indptr + rc);   // indptr[rc], indptr[rc + 1]
    Ip r; r.s = se.x; r.len = row < A.N ? (int)min<int64_t>(se.y - se.x, 0x7FFFFFFF) : 0;
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
  auto row_finish = [&](int64_t row, const Ip& p, const Ent& t, const Prep& q) {
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
    nm = rr_max<G>(nm);
    const bool any = nm >= 0.0;
    const double zmax = any ? nm * r : -1.0;               // = max_j fl(n_j r): rounding is monotone
    int nbl = 0; uint32_t wid = 0u;
#pragma unroll
    for (int j = 0; j < E; ++j) {
      const bool in = INIT ? (k0 + j < p.len) : (n[j] != 0.0);
      const bool b = in && (n[j] * r == zmax);
      nbl += b ? 1 : 0;
      wid = b ? half(t.id, j) : wid;
    }
    const int nb = rr_sum_i<G>(nbl);
    if (gl == 0 && row < A.N && !(A.dbg & 2)) A.nbest[row] = any ? nb : 0;
    if (one_winner) {
      if (nbl != 0 && nb == 1) {                           // this lane holds the row's only best hit
        EM.one(wid);
        if (zmax >= A.thresh) { const double vc = zmax * recip0(zmax); if (vc != 0.0) EM.conf(wid, vc); }   // vsum = the winner's z
      }
      if (__builtin_amdgcn_ballot_w64(nb > 1) != 0ull && !(A.dbg & 4)) {   // tied rows
        if (nbl == 1 && nb == 2) EM.tie(wid, 2, 0.5);      // (the usual tie: two best hits, this lane holds one of them)
        if (__builtin_amdgcn_ballot_w64(nb > 1 && !(nbl == 1 && nb == 2) && nbl != 0) != 0ull) {
          const double share = 1.0 * recip0((double)nb);
#pragma unroll
          for (int j = 0; j < E; ++j) {
            const bool in = INIT ? (k0 + j < p.len) : (n[j] != 0.0);
            if (nb > 1 && !(nbl == 1 && nb == 2) && in && n[j] * r == zmax) EM.tie(half(t.id, j), nb, share);
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
        if (z >= A.thresh) { const double vc = z * rv; if (vc != 0.0) EM.conf(id, vc); }
        if (z == zmax) { if (nb == 1) EM.one(id); else EM.tie(id, nb, share); }
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
      const Prep q = row_prep(cur, e0);                    // gathers of this row first ...
      __builtin_amdgcn_sched_barrier(0);
      const Ent e1 = load_ent(ip1);                        // ... then the loads of the next rows: they have this row's
      const Ip ip2 = load_ip(it + 2);                      //     arithmetic to arrive in
      __builtin_amdgcn_sched_barrier(0);
      if (defer) {
        if (gl == 0) A.defer_rows[atomicAdd(A.defer_n, 1ull)] = (int32_t)row;
      } else {
        row_finish(row, cur, e0, q);
      }
      ip0 = ip1; ip1 = ip2; e0 = e1;
    }
  }
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

// the rows k_report_rows left: any length, sweeps of 16 entries, one 16-lane group per row
template <bool INIT>
__global__ __launch_bounds__(256) void k_report_slow(ReportArgs A) {
  constexpr int G = 16;
  const ReportEmit EM{A, nullptr, nullptr, nullptr, 0, nullptr};  // (no LDS slots here: a handful of rows)
  const int gl = threadIdx.x % G, grp = threadIdx.x / G, ngrp = blockDim.x / G;
  const int64_t nd = (int64_t)*A.defer_n;
  for (int64_t d = (int64_t)blockIdx.x * ngrp + grp; d < nd; d += (int64_t)gridDim.x * ngrp) {
    const int64_t row = A.defer_rows[d];
    const int64_t s = A.indptr[row];
    const int len = (int)(A.indptr[row + 1] - s);
    const uint32_t coff = len > 1 ? 0u : (uint32_t)A.IDN;
    auto numer = [&](int k) -> double {
      double x = A.lut[A.raw[s + k]];
      if (!INIT) x = x * A.cnat2[A.rid[s + k] + coff];
      return x;
    };
    double y = 0.0;
    for (int k = gl; k < len; k += G) y += numer(k);
    const double r = recip0(sg_sum<G>(y));
    double zmax = -1.0, vs = 0.0; int cnt = 0;
    for (int k = gl; k < len; k += G) {
      const double n = numer(k);
      if (INIT || n != 0.0) { const double z = n * r; zmax = fmax(zmax, z); ++cnt; if (z >= A.thresh) vs += z; }
    }
    zmax = sg_max<G>(zmax); cnt = sg_sum_i<G>(cnt);
    const double vsum = sg_sum<G>(vs);
    int nb = 0;
    for (int k = gl; k < len; k += G) { const double n = numer(k); if ((INIT || n != 0.0) && n * r == zmax) ++nb; }
    nb = sg_sum_i<G>(nb);
    if (gl == 0) A.nbest[row] = cnt ? nb : 0;
    const double share = 1.0 * recip0((double)nb);
    for (int k = gl; k < len; k += G) {
      const double n = numer(k);
      if (!(INIT || n != 0.0)) continue;
      const double z = n * r;
      if (!(z == zmax || z >= A.thresh)) continue;
      const uint32_t id = A.rid[s + k];
      if (z >= A.thresh) { const double vc = z * recip0(vsum); if (vc != 0.0) EM.conf(id, vc); }
      if (z == zmax) { if (nb == 1) EM.one(id); else EM.tie(id, nb, share); }
    }
  }
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
// popularity ids: id = slot * P + part of the column's first slot in the blocked layout (popular columns come first in
// every part, so small ids are popular columns); cnat2 by id
__global__ __launch_bounds__(256) void k_rid16_rows(int64_t N, const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
    const uint32_t* __restrict__ colmap, int P, int only_short, uint16_t* __restrict__ rid) {
  const int sub = threadIdx.x / RS_SUB, lane = threadIdx.x % RS_SUB, subs = blockDim.x / RS_SUB;
  for (int64_t i = (int64_t)blockIdx.x * subs + sub; i < N; i += (int64_t)gridDim.x * subs) {
    const int64_t s = indptr[i], e = indptr[i + 1];
    if (only_short && e - s > 1) continue;                 // (the ambiguous rows were written by k_row_partcounts)
    for (int64_t k = s + lane; k < e; k += RS_SUB) { const uint32_t cm = colmap[indices[k]]; rid[k] = (uint16_t)((cm & 0x1FFFu) * P + (cm >> 16)); }
  }
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

// ---- reduced-precision EM pass (BASELINE config 3: the fp32 leg of the tolerance sweep) -------------
// A DIAGNOSTIC, not a product path: the E-step products, the row sums, the posteriors and the column sums are all
// fp32.  Q = expm1(100 s / max) reaches 2.7e43 > FLT_MAX, so the score table is scaled by 2^-64
// (exact) before rounding to fp32, and pi*theta by 1 / max_j(pi*theta) (z is invariant under both); products
// that still underflow are lost — which is what the sweep is there to measure.  One 16-lane group per row of the
// canonical CSR, fp32 global atomics for the column sums.
constexpr int F32_SHIFT = 64;
__global__ void k_cmax(int K, const double* __restrict__ pi, const double* __restrict__ theta, unsigned long long* __restrict__ out) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  double v = j < K ? pi[j] * theta[j] : 0.0;
  v = sg_max<64>(v);
  if ((threadIdx.x & 63) == 0) atomicMax(out, (unsigned long long)__double_as_longlong(v));   // non-negative doubles order like integers
}
__global__ void k_make_c32(int K, const double* __restrict__ pi, const double* __restrict__ theta,
                           const unsigned long long* __restrict__ cmax, float* __restrict__ c32) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= K) return;
  const double m = __longlong_as_double((long long)*cmax);
  c32[j] = (float)((pi[j] * theta[j]) / (m > 0.0 ? m : 1.0));
}
__global__ void k_lut32(int n, const double* __restrict__ lut, float* __restrict__ lut32) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) lut32[i] = (float)ldexp(lut[i], -F32_SHIFT);
}
__global__ __launch_bounds__(256) void k_em_rows_f32(int64_t N, const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
    const uint16_t* __restrict__ raw, const float* __restrict__ lut32, const float* __restrict__ c32, float* __restrict__ colsums) {
  const int sub = threadIdx.x / RP_SUB, lane = threadIdx.x % RP_SUB, subs = blockDim.x / RP_SUB;
  for (int64_t row = (int64_t)blockIdx.x * subs + sub; row < N; row += (int64_t)gridDim.x * subs) {
    const int64_t s = indptr[row], e = indptr[row + 1];
    if (e - s < 2) continue;                               // unique rows feed pi through pisum0 only (model.py:699)
    float y = 0.f, w = 0.f;
    for (int64_t k = s + lane; k < e; k += RP_SUB) { const float q = lut32[raw[k]]; y += q * c32[indices[k]]; w = fmaxf(w, q); }
#pragma unroll
    for (int o = RP_SUB / 2; o > 0; o >>= 1) { y += __shfl_xor(y, o, RP_SUB); w = fmaxf(w, __shfl_xor(w, o, RP_SUB)); }
    float r = 1.f / y;
    if (isinf(r)) r = 0.f;                                 // recip0, sparse_plus.py:16-22
    for (int64_t k = s + lane; k < e; k += RP_SUB) {
      const float v = ((lut32[raw[k]] * c32[indices[k]]) * r) * w;
      if (v != 0.f) unsafeAtomicAdd(&colsums[indices[k]], v);
    }
  }
}
__global__ void k_red_from_f32(int K, const float* __restrict__ colsums, double* __restrict__ red) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < K) red[j] = ldexp((double)colsums[j], F32_SHIFT);
  if (j == 0) { red[K] = 0.0; red[K + 1] = 0.0; }
}

// ---- csr_matrix_plus primitives on fp64 CSR --------------------------------
__global__ __launch_bounds__(256) void k_norm_rows(int64_t N, const int64_t* __restrict__ indptr,
                                                   const double* __restrict__ data, double* __restrict__ out) {
  const int sub = threadIdx.x / RP_SUB, lane = threadIdx.x % RP_SUB, subs = blockDim.x / RP_SUB;
  for (int64_t row = (int64_t)blockIdx.x * subs + sub; row < N; row += (int64_t)gridDim.x * subs) {
    int64_t s = indptr[row], e = indptr[row + 1];
    double y = 0.0;
    for (int64_t k = s + lane; k < e; k += RP_SUB) y += data[k];
    y = sg_sum<RP_SUB>(y);
    double r = recip0(y);
    for (int64_t k = s + lane; k < e; k += RP_SUB) out[k] = data[k] * r;
  }
}
__global__ __launch_bounds__(256) void k_binmax_rows(int64_t N, int32_t K, const int64_t* __restrict__ indptr,
                                                     const double* __restrict__ data, int8_t* __restrict__ out) {
  const int sub = threadIdx.x / RP_SUB, lane = threadIdx.x % RP_SUB, subs = blockDim.x / RP_SUB;
  for (int64_t row = (int64_t)blockIdx.x * subs + sub; row < N; row += (int64_t)gridDim.x * subs) {
    int64_t s = indptr[row], e = indptr[row + 1];
    bool any = false;
    double m = 0.0;
    for (int64_t k = s + lane; k < e; k += RP_SUB) { m = any ? fmax(m, data[k]) : data[k]; any = true; }
    // combine: lanes without entries must not contribute
    double mm = any ? m : -INFINITY;
    mm = sg_max<RP_SUB>(mm);
    if ((e - s) < K) mm = fmax(mm, 0.0);  // implicit zeros take part in max(1)
    for (int64_t k = s + lane; k < e; k += RP_SUB) out[k] = (data[k] == mm) ? 1 : 0;
  }
}

// whole-matrix reductions for csr_matrix_plus.norm() / scale() (sparse_plus.py:46-48, 93-95)
__global__ __launch_bounds__(256) void k_reduce_all(const double* __restrict__ v, int64_t n, int want_max,
                                                    double* __restrict__ part) {
  __shared__ double scratch[16];
  double acc = want_max ? -INFINITY : 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    acc = want_max ? fmax(acc, v[i]) : acc + v[i];
  if (want_max) {
    acc = sg_max<64>(acc);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { double m = scratch[0]; for (int i = 1; i < (int)(blockDim.x >> 6); ++i) m = fmax(m, scratch[i]); part[blockIdx.x] = m; }
  } else {
    double t = block_sum(acc, scratch);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
  }
}
__global__ void k_scale_all(const double* __restrict__ v, int64_t n, double f, double* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = v[i] * f;
}
__global__ __launch_bounds__(256) void k_scale_rows(int64_t N, int32_t K, const int64_t* __restrict__ indptr,
                                                    const double* __restrict__ data, double* __restrict__ out) {
  const int sub = threadIdx.x / RP_SUB, lane = threadIdx.x % RP_SUB, subs = blockDim.x / RP_SUB;
  for (int64_t row = (int64_t)blockIdx.x * subs + sub; row < N; row += (int64_t)gridDim.x * subs) {
    int64_t s = indptr[row], e = indptr[row + 1];
    double m = -INFINITY;
    for (int64_t k = s + lane; k < e; k += RP_SUB) m = fmax(m, data[k]);
    m = sg_max<RP_SUB>(m);
    if ((e - s) < K) m = fmax(m, 0.0);                 // implicit zeros take part in max(1)
    double r = recip0(m);
    for (int64_t k = s + lane; k < e; k += RP_SUB) out[k] = data[k] * r;
  }
}

// ============================================================================
// host side
// ============================================================================
template <typename T>
static int dalloc(tsem_ctx* h, T** p, size_t n) {
  if (*p) { (void)hipFree(*p); *p = nullptr; }
  if (n == 0) n = 1;
  hipError_t e = hipMalloc((void**)p, n * sizeof(T));
  if (e != hipSuccess) {
    h->err = std::string("hipMalloc(") + std::to_string(n * sizeof(T)) + " B): " + hipGetErrorString(e);
    return TSEM_ERR_NOMEM;
  }
  return TSEM_OK;
}
#define TSEM_ALLOC(ptr, n) do { int rc_ = dalloc(h, &(ptr), (size_t)(n)); if (rc_) return rc_; } while (0)

template <typename T>
static void dfree(T*& p) { if (p) { (void)hipFree(p); p = nullptr; } }

// option "reproducible": the slots' bounds as a run finds them: 2^E > the largest fragment weight >= every contribution w * z
// (refined column by column, see k_bin_check).  Called when parameters are set from outside, so that a run's bits depend on its
// starting point only, not on what the context computed before.
static int bin_reset(tsem_ctx* h) {
  if (!h->d_ebias) return TSEM_OK;
  int e2 = 0;
  (void)std::frexp(h->w_max > 0 ? h->w_max : 1.0, &e2);    // w_max = m * 2^e2, m in [0.5, 1)  ->  w_max < 2^e2
  std::vector<uint16_t> eb((size_t)h->Kpad, (uint16_t)std::min(2000, std::max(64, e2 + 1023)));
  TSEM_HIP(hipStreamSynchronize(h->stream));
  TSEM_HIP(hipMemcpy(h->d_ebias, eb.data(), sizeof(uint16_t) * h->Kpad, hipMemcpyHostToDevice));
  TSEM_HIP(hipMemset(h->d_ovf, 0, (size_t)h->Kpad));
  TSEM_HIP(hipMemset(h->d_ehist, 0, sizeof(int16_t) * (2 * (size_t)h->K + 2)));
  return TSEM_OK;
}


static inline int cdiv64(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// TSEM_TRACE=1: host wall clock of the set-up phases on stderr (each lap synchronises the stream; a diagnostic, not a product path)
struct PhaseTimer {
  bool on; hipStream_t s; std::chrono::steady_clock::time_point t;
  explicit PhaseTimer(hipStream_t st) : on(getenv("TSEM_TRACE") != nullptr), s(st), t(std::chrono::steady_clock::now()) {}
  void lap(const char* name) {
    if (!on) return;
    (void)hipStreamSynchronize(s);
    const auto n = std::chrono::steady_clock::now();
    fprintf(stderr, "[tsem] %-34s %8.3f ms\n", name, std::chrono::duration<double, std::milli>(n - t).count());
    t = n;
  }
};

static int ensure_device(tsem_ctx* h) {
  TSEM_HIP(hipSetDevice(h->device));
  return TSEM_OK;
}

// The fused kernel's instantiations (team size x mode x entry format x geometry: ~200 kernels, most of the library's build time)
// live in six translation units compiled in parallel (tsem_fz_p1.hip ... tsem_fz_p78.hip, tsem_fused_inst.h); each exports one
// look-up function.  The host launches through the pointer.
typedef void (*fz_fn)(FusedArgs);
fz_fn tsem_fz_kernel_p1(int P, int mode, int fmt, int geo);
fz_fn tsem_fz_kernel_p2(int P, int mode, int fmt, int geo);
fz_fn tsem_fz_kernel_p3(int P, int mode, int fmt, int geo);
fz_fn tsem_fz_kernel_p4(int P, int mode, int fmt, int geo);
fz_fn tsem_fz_kernel_p56(int P, int mode, int fmt, int geo);
fz_fn tsem_fz_kernel_p78(int P, int mode, int fmt, int geo);
static int fz_fmt(const tsem_ctx* h) { return h->fmt_code ? 1 : (h->fmt_wcode ? 2 : 0); }
static fz_fn fz_kernel(int P, int mode, int fmt, int geo) {
  switch (P) {
    case 1: return tsem_fz_kernel_p1(P, mode, fmt, geo); case 2: return tsem_fz_kernel_p2(P, mode, fmt, geo);
    case 3: return tsem_fz_kernel_p3(P, mode, fmt, geo); case 4: return tsem_fz_kernel_p4(P, mode, fmt, geo);
    case 5: case 6: return tsem_fz_kernel_p56(P, mode, fmt, geo);
    case 7: case 8: return tsem_fz_kernel_p78(P, mode, fmt, geo);
    default: return nullptr;
  }
}

// code16 entry format: only with the fused kernel, and only while the score table is small enough to
// sit in LDS beside the column tables (uint16 scores allow 65536 entries; alignments give a few hundred).
// With the round-2 exchange (branch-free, partner loads after the combine for short rows) codes in row order win
// at every row length measured — 10 / 14 / 20 / 28 / 40 / 100 entries per row: 1.47 / 1.61 / 1.98 / 2.55 / 3.60 /
// 3.76 ms against 1.71 / 1.96 / 2.57 / 3.26 / 4.23 / 4.39 ms with fp64 entries (profiles/r02_sweep_short.txt).
static bool fz_wants_codes(const tsem_ctx* h) {
  return h->opt_format != 1 && h->lut_len > 0 && h->lut_len <= 2048;
}
static size_t fz_lds_bytes(const tsem_ctx* h, bool codes) {
  return (size_t)((h->exact_single ? 3 : 2) * h->Kp + (fz_yr(h->geo) + 2) * h->R) * 8 + 192 + 512 + (codes ? (size_t)h->lut_len * 8 : 0) +
         std::max<size_t>(h->opt_reproducible ? (size_t)h->Kp * 2 + 16 : 0, FZ_LOGTAB * 16 + 16);   // (+ the slots' exponent table | the lnl pass's log table)
}

static void free_layout(tsem_ctx* h) {
  dfree(h->d_ebias); dfree(h->d_ovf); dfree(h->d_red_hi); dfree(h->d_binflag); dfree(h->d_ehist);
  dfree(h->d_colmap); dfree(h->d_col_of_pc); dfree(h->d_rid16); dfree(h->d_col_of_id); dfree(h->d_sb_off); dfree(h->d_pval); dfree(h->d_pcode); dfree(h->d_prc);
  dfree(h->d_ypart); dfree(h->d_partial); dfree(h->d_xchg); dfree(h->d_xflags); dfree(h->d_fz_aux); h->fz_clean = false; dfree(h->d_fpartial); dfree(h->d_fpartial2); dfree(h->d_amb_w); dfree(h->d_sb_q32);
  h->fused_launched = false;
}
static void free_matrix(tsem_ctx* h) {
  dfree(h->d_indptr); dfree(h->d_indices); dfree(h->d_raw); dfree(h->d_lut);
  dfree(h->d_amb_row); dfree(h->d_amb_wcode); dfree(h->d_amb_wcode_c); dfree(h->d_slot_row); dfree(h->d_uni_col); dfree(h->d_uni_code);
  dfree(h->d_pisum0); dfree(h->d_twin_rep); dfree(h->d_ucount); dfree(h->d_colcount);
  free_layout(h);
  dfree(h->d_pi); dfree(h->d_theta); dfree(h->d_pi_prev); dfree(h->d_theta_prev);
  dfree(h->d_ctab); dfree(h->d_ctab_prev); dfree(h->d_red_own); dfree(h->d_tmp_pi); dfree(h->d_tmp_theta);
  dfree(h->d_c32); dfree(h->d_cs32); dfree(h->d_lut32); dfree(h->d_cnat);
  dfree(h->d_ctl); dfree(h->d_ctld); dfree(h->d_lnls); dfree(h->d_pi_first); dfree(h->d_theta_first); dfree(h->d_user_z);
  dfree(h->d_tie_rows); dfree(h->d_tie_cnt); h->n_ties = 0;
  dfree(h->d_rep_nb); dfree(h->d_rep_rows); dfree(h->d_rep_n);
  if (h->d_rep_tmp) { (void)hipFree(h->d_rep_tmp); h->d_rep_tmp = nullptr; h->rep_tmp_bytes = 0; }
  h->first_pending = false;
  h->d_red = nullptr;
  h->have_rowstats = h->have_model = false;
  h->N = h->nnz = 0; h->K = 0;
  h->max_code = -1;
}

extern "C" {

int tsem_create(tsem_ctx** out, int device) {
  if (!out) return TSEM_ERR_ARG;
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    g_create_err = std::string("no usable HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    return TSEM_ERR_HIP;
  }
  if (device < 0 || device >= ndev) { g_create_err = "device index out of range"; return TSEM_ERR_ARG; }
  e = hipSetDevice(device);
  if (e != hipSuccess) { g_create_err = std::string("hipSetDevice: ") + hipGetErrorString(e); return TSEM_ERR_HIP; }
  tsem_ctx* h = new tsem_ctx();
  h->device = device;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) h->n_cu = prop.multiProcessorCount;
  if (hipMalloc((void**)&h->d_diffs, TS_DIFF_RING * sizeof(double)) != hipSuccess ||
      hipMalloc((void**)&h->d_lnl_part, 16384 * sizeof(double)) != hipSuccess ||
      hipMalloc((void**)&h->d_maxcode, 64) != hipSuccess ||
      hipMalloc((void**)&h->d_xerr, 64) != hipSuccess) {
    g_create_err = "hipMalloc failed in tsem_create";
    delete h;
    return TSEM_ERR_NOMEM;
  }
  (void)hipMemset(h->d_xerr, 0, 64);
  *out = h;
  return TSEM_OK;
}

void tsem_destroy(tsem_ctx* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  free_matrix(h);
  dfree(h->d_diffs); dfree(h->d_lnl_part); dfree(h->d_maxcode); dfree(h->d_xerr);
  for (auto& ev : h->ev) (void)hipEventDestroy(ev);
  delete h;
}

const char* tsem_last_error(const tsem_ctx* h) { return h ? h->err.c_str() : g_create_err.c_str(); }

int tsem_set_stream(tsem_ctx* h, void* s) {
  if (!h) return TSEM_ERR_ARG;
  h->stream = (hipStream_t)s;
  return TSEM_OK;
}

int tsem_set_option(tsem_ctx* h, const char* key, int64_t v) {
  if (!h || !key) return TSEM_ERR_ARG;
  std::string k(key);
  if (k == "em_kernel") h->em_kernel = (int)v;
  else if (k == "block_rows") h->opt_R = v;
  else if (k == "parts") h->opt_P = v;
  else if (k == "row_offset") h->row_offset = v;
  else if (k == "chunk_blocks") h->opt_chunk = v;
  else if (k == "fused_dbg") h->opt_dbg = v;
  else if (k == "value_format") h->opt_format = v;
  else if (k == "hot_split") h->opt_hot_split = v;
  else if (k == "geometry") h->opt_geo = v;
  else if (k == "sorted_fill") h->opt_sorted = v;
  else if (k == "deconflict") h->opt_deconflict = v;
  else if (k == "reproducible") h->opt_reproducible = v;
  else if (k == "em_precision") h->opt_precision = v;
  else if (k == "kernel_timing") h->opt_timing = v;
  else if (k == "report_shortcuts") h->opt_shortcuts = v;
  else if (k == "rowpass_wgs") h->opt_rowpass_wgs = v;
  else if (k == "report_kernel") h->opt_report_kernel = v;   // 0: the generic row pass (k_rowpass<RP_REPORT>) instead of k_report_rows
  else if (k == "report_wgs2") h->opt_report_wgs2 = v;
  else if (k == "report_dbg") h->opt_report_dbg = v;
  else if (k == "report_lanes") h->opt_report_lanes = v;     // capacity (lanes per row x entries per lane) of k_report_rows: 8 .. 256 (0 = from the row lengths)
  else if (k == "issue_early") h->opt_issue = v;       // (kept for old scripts; the exchange has one order now)
  else if (k == "fused_prof") {
    if (v && !h->d_prof) { if (hipMalloc((void**)&h->d_prof, 64 * 16 * 8) != hipSuccess) return TSEM_ERR_NOMEM; }
    if (h->d_prof) (void)hipMemset(h->d_prof, 0, 64 * 16 * 8);
  }
  else TSEM_FAIL(TSEM_ERR_ARG, "unknown option " + k);
  return TSEM_OK;
}

int tsem_synchronize(tsem_ctx* h) {
  if (!h) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  TSEM_HIP(hipStreamSynchronize(h->stream));
  return TSEM_OK;
}

static int set_lut(tsem_ctx* h, const double* lut, int32_t lut_len) {
  if (!lut || lut_len <= 0 || lut_len > 65536) TSEM_FAIL(TSEM_ERR_ARG, "lut must have 1..65536 entries");
  h->lut_len = lut_len;
  dfree(h->d_lut32); dfree(h->d_c32); dfree(h->d_cs32);      // (the fp32 diagnostic tables follow the score table)
  h->lut_host.assign(lut, lut + lut_len);
  TSEM_ALLOC(h->d_lut, lut_len);
  TSEM_HIP(hipMemcpy(h->d_lut, lut, sizeof(double) * lut_len, hipMemcpyHostToDevice));
  return TSEM_OK;
}

__global__ __launch_bounds__(256) void k_check_csr(int64_t N, int32_t K, int32_t lut_len, const int64_t* __restrict__ indptr,
                                                   const int32_t* __restrict__ indices, const uint16_t* __restrict__ raw,
                                                   uint32_t* __restrict__ bad) {
  const int64_t nnz = indptr[N];
  uint32_t f = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int64_t i = t0; i < N; i += stride) {
    const int64_t a = indptr[i], b = indptr[i + 1];
    if (b < a || a < 0 || b > nnz) f |= 1u;
  }
  // canonical form (column ids strictly increasing inside a row): every position where the ids do not increase must
  // be the first entry of a row -> count such positions over the entries and over the row starts, compare
  unsigned long long desc = 0;
  for (int64_t k = t0; k < nnz; k += stride) {
    if ((uint32_t)indices[k] >= (uint32_t)K) f |= 2u;
    if ((int32_t)raw[k] >= lut_len) f |= 4u;
    if (k > 0 && indices[k] <= indices[k - 1]) ++desc;
  }
  for (int64_t i = t0; i < N; i += stride) {
    const int64_t a = indptr[i], b = indptr[i + 1];
    if (a > 0 && a < b && a < nnz && indices[a] <= indices[a - 1]) --desc;     // (wraps; the sum over all threads is what counts)
  }
  if (desc) atomicAdd(reinterpret_cast<unsigned long long*>(bad + 2), desc);
  if (f) atomicOr(bad, f);
}

int tsem_load_scores(tsem_ctx* h, int64_t n_rows, int32_t n_cols, const int64_t* indptr, const int32_t* indices,
                     const uint16_t* raw, const double* lut, int32_t lut_len) {
  if (!h) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  if (n_rows < 0 || n_cols <= 0 || !indptr) TSEM_FAIL(TSEM_ERR_ARG, "bad matrix dimensions");
  if (n_rows >= (int64_t)INT32_MAX) TSEM_FAIL(TSEM_ERR_ARG, "more than 2^31-1 rows per rank is not supported");
  const int64_t nnz = indptr[n_rows];
  if (indptr[0] != 0 || nnz < 0) TSEM_FAIL(TSEM_ERR_ARG, "indptr must start at 0");
  if (nnz && (!indices || !raw)) TSEM_FAIL(TSEM_ERR_ARG, "null entry arrays");
  free_matrix(h);
  if (lut) {
    if (int rc = set_lut(h, lut, lut_len)) return rc;
  } else {                                                   // the table follows (tsem_max_score -> tsem_set_lut)
    dfree(h->d_lut); h->lut_host.clear(); h->lut_len = 0;
    lut_len = 65536;
  }
  h->N = n_rows; h->K = n_cols; h->nnz = nnz;
  TSEM_ALLOC(h->d_indptr, n_rows + 1);
  TSEM_ALLOC(h->d_indices, nnz + TS_ENTRY_PAD);           // (k_report_rows reads whole lanes of 16 entries past a row's end)
  TSEM_ALLOC(h->d_raw, nnz + TS_ENTRY_PAD);
  // plain hipMemcpy from the caller's pageable arrays: 55 GB/s on the GPU box (tools/time_host_upload.py; a pipeline
  // through pinned staging buffers filled by 8 host threads was slower: 37 GB/s)
  TSEM_HIP(hipMemcpy(h->d_indptr, indptr, sizeof(int64_t) * (n_rows + 1), hipMemcpyHostToDevice));
  if (nnz) {
    TSEM_HIP(hipMemcpy(h->d_indices, indices, sizeof(int32_t) * nnz, hipMemcpyHostToDevice));
    TSEM_HIP(hipMemcpy(h->d_raw, raw, sizeof(uint16_t) * nnz, hipMemcpyHostToDevice));
  }
  // The host arrays are checked on the DEVICE, after the copy (the host loop of round 1 over the entries took twice as
  // long as the copy itself: 97 of 142 ms at 4e8 entries): row pointers non-decreasing, column ids in [0, K) and strictly
  // increasing inside a row (canonical CSR: z, masks and the tie order of `choose` are aligned to it), scores inside the table.
  uint32_t* d_bad = nullptr;                                  // [0] flags, [2..3] 64-bit count (see k_check_csr)
  TSEM_ALLOC(d_bad, 4);
  TSEM_HIP(hipMemsetAsync(d_bad, 0, 4 * sizeof(uint32_t), h->stream));
  k_check_csr<<<2048, 256, 0, h->stream>>>(n_rows, n_cols, lut_len, h->d_indptr, h->d_indices, h->d_raw, d_bad);
  uint32_t badw[4] = {0, 0, 0, 0};
  TSEM_HIP(hipMemcpyAsync(badw, d_bad, sizeof(badw), hipMemcpyDeviceToHost, h->stream));
  TSEM_HIP(hipStreamSynchronize(h->stream));
  uint32_t bad = badw[0] | ((badw[2] | badw[3]) && !(badw[0] & 1u) ? 8u : 0u);
  (void)hipFree(d_bad);
  if (bad) {
    free_matrix(h);
    h->N = 0; h->nnz = 0;
    if (bad & 1u) TSEM_FAIL(TSEM_ERR_ARG, "indptr must be non-decreasing");
    if (bad & 2u) TSEM_FAIL(TSEM_ERR_ARG, "column index out of range");
    if (bad & 4u) TSEM_FAIL(TSEM_ERR_ARG, "raw score exceeds lookup table");
    TSEM_FAIL(TSEM_ERR_ARG, "not a canonical CSR: column ids must be strictly increasing inside every row");
  }
  return TSEM_OK;
}

int tsem_generate(tsem_ctx* h, int64_t row_begin, int64_t row_end, int32_t n_cols, const uint32_t* len_cdf,
                  int32_t cdf_len, uint64_t seed, int32_t dist, double uniq_frac) {
  if (!h) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  int64_t n = row_end - row_begin;
  if (n < 0 || n_cols < 2 || !len_cdf || cdf_len <= 0) TSEM_FAIL(TSEM_ERR_ARG, "bad generator arguments");
  if (n >= (int64_t)INT32_MAX) TSEM_FAIL(TSEM_ERR_ARG, "more than 2^31-1 rows per rank is not supported");
  free_matrix(h);
  h->N = n; h->K = n_cols;
  uint32_t* d_cdf = nullptr;
  TSEM_ALLOC(d_cdf, cdf_len);
  TSEM_HIP(hipMemcpy(d_cdf, len_cdf, sizeof(uint32_t) * cdf_len, hipMemcpyHostToDevice));
  TSEM_ALLOC(h->d_indptr, n + 1);
  TSEM_HIP(hipMemsetAsync(h->d_indptr, 0, sizeof(int64_t), h->stream));
  uint32_t uth = 0;
  if (uniq_frac > 0) {
    double t = uniq_frac * 4294967296.0;
    uth = t >= 4294967295.0 ? 4294967295u : (uint32_t)t;
  }
  if (n) {
    k_gen_len<<<cdiv64(n, 256), 256, 0, h->stream>>>(row_begin, n, n_cols, d_cdf, cdf_len, seed, uth, h->d_indptr + 1);
    size_t tb = 0;
    TSEM_HIP(rocprim::inclusive_scan(nullptr, tb, h->d_indptr + 1, h->d_indptr + 1, (size_t)n, rocprim::plus<int64_t>(), h->stream));
    void* tmp = nullptr;
    TSEM_HIP(hipMalloc(&tmp, tb ? tb : 1));
    TSEM_HIP(rocprim::inclusive_scan(tmp, tb, h->d_indptr + 1, h->d_indptr + 1, (size_t)n, rocprim::plus<int64_t>(), h->stream));
    TSEM_HIP(hipStreamSynchronize(h->stream));
    (void)hipFree(tmp);
  }
  int64_t nnz = 0;
  TSEM_HIP(hipMemcpy(&nnz, h->d_indptr + n, sizeof(int64_t), hipMemcpyDeviceToHost));
  h->nnz = nnz;
  TSEM_ALLOC(h->d_indices, nnz + TS_ENTRY_PAD);
  TSEM_ALLOC(h->d_raw, nnz + TS_ENTRY_PAD);
  if (n) {
    k_gen_rows<<<cdiv64(n, 128), 128, 0, h->stream>>>(row_begin, n, n_cols, seed, dist, h->d_indptr, h->d_indices, h->d_raw);
    TSEM_HIP(hipGetLastError());
    TSEM_HIP(hipStreamSynchronize(h->stream));
  }
  (void)hipFree(d_cdf);
  return TSEM_OK;
}

__global__ void k_max_u16(const uint16_t* __restrict__ v, int64_t n, uint32_t* __restrict__ out) {
  uint32_t m = 0;
  // eight scores per 16-byte load (hipMalloc alignment); the tail one by one
  const uint4* v4 = reinterpret_cast<const uint4*>(v);
  const int64_t n8 = n / 8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    const uint4 w = v4[i];
    const uint32_t a = max(max(w.x & 0xFFFFu, w.x >> 16), max(w.y & 0xFFFFu, w.y >> 16));
    const uint32_t b = max(max(w.z & 0xFFFFu, w.z >> 16), max(w.w & 0xFFFFu, w.w >> 16));
    m = max(m, max(a, b));
  }
  for (int64_t i = n8 * 8 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    m = max(m, (uint32_t)v[i]);
  m = (uint32_t)sg_max_i<64>((int)m);
  if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

int tsem_max_score(tsem_ctx* h, int32_t* max_score) {
  if (!h || !h->d_indptr || !max_score) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  if (h->max_code < 0) {                                     // (the matrix does not change between load / generate calls)
    TSEM_HIP(hipMemsetAsync(h->d_maxcode, 0, 4, h->stream));
    if (h->nnz) k_max_u16<<<1024, 256, 0, h->stream>>>(h->d_raw, h->nnz, h->d_maxcode);
    uint32_t m = 0;
    TSEM_HIP(hipMemcpyAsync(&m, h->d_maxcode, 4, hipMemcpyDeviceToHost, h->stream));
    TSEM_HIP(hipStreamSynchronize(h->stream));
    h->max_code = (int32_t)m;
  }
  *max_score = h->max_code;
  return TSEM_OK;
}

int tsem_set_lut(tsem_ctx* h, const double* lut, int32_t lut_len) {
  if (!h || !h->d_indptr) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  int32_t m = 0;
  if (int rc = tsem_max_score(h, &m)) return rc;
  if (m >= lut_len) TSEM_FAIL(TSEM_ERR_ARG, "lookup table shorter than the largest raw score");
  h->have_rowstats = h->have_model = false;
  return set_lut(h, lut, lut_len);
}

int tsem_dims(tsem_ctx* h, int64_t* n_rows, int32_t* n_cols, int64_t* nnz) {
  if (!h) return TSEM_ERR_ARG;
  if (n_rows) *n_rows = h->N;
  if (n_cols) *n_cols = h->K;
  if (nnz) *nnz = h->nnz;
  return TSEM_OK;
}

int tsem_export_csr(tsem_ctx* h, int64_t* indptr, int32_t* indices, uint16_t* raw) {
  if (!h || !h->d_indptr) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  TSEM_HIP(hipStreamSynchronize(h->stream));
  if (indptr) TSEM_HIP(hipMemcpy(indptr, h->d_indptr, sizeof(int64_t) * (h->N + 1), hipMemcpyDeviceToHost));
  if (indices && h->nnz) TSEM_HIP(hipMemcpy(indices, h->d_indices, sizeof(int32_t) * h->nnz, hipMemcpyDeviceToHost));
  if (raw && h->nnz) TSEM_HIP(hipMemcpy(raw, h->d_raw, sizeof(uint16_t) * h->nnz, hipMemcpyDeviceToHost));
  return TSEM_OK;
}

// Column parts (the tables of one part must fit LDS), rows per block and the fused kernel's geometry, from
// the row statistics and the options; called again when a handle falls back to the two-pass kernels.
static int choose_geometry(tsem_ctx* h) {
  const int K = h->K;
  const int64_t na = h->N_amb, nu = h->N_uni;
  {
    // column parts (tables of one part must fit LDS) and rows per block
    // option "reproducible" = 1: both pieces of the exact sums in ONE pass if three tables per part fit the LDS with at most 8
    // parts (score codes; 26 B of LDS per column); else — or with "reproducible" = 2 — two passes over two tables
    h->exact_single = false;
    if (h->opt_reproducible == 1 && h->em_kernel != TSEM_EMK_TWOPASS && h->opt_format != 1 && h->lut_len > 0 && h->lut_len <= 2048) {
      const int p3 = h->opt_P > 0 ? (int)h->opt_P : (K + TS_MAX_KP3 - 64 - 1) / (TS_MAX_KP3 - 64);
      // teams of 5-8 have ONE geometry (384 row slots): worth it only when the rows are long enough to fill their register
      // tiles (20M x 30k: 100 per row 9.4 -> 6.3 ms per iteration, 18 per row 2.4 -> 3.3; K = 15k, teams of 4: 3.9 -> 2.6 at
      // 40 per row, 1.9 -> 1.4 at 18; profiles/r03_reproducible.txt)
      const double ml = na > 0 ? (double)(h->nnz - nu) / (double)na : 0.0;
      // (one pass costs ~1.25 default passes on a full tile, two passes cost 2: worth it down to tiles ~2/3 full — K = 30k, 40 per
      //  row, teams of 7: 3.58 against 3.79 ms per iteration)
      const bool long_enough = p3 <= 4 || h->opt_P > 0 || ml * fz_rmax(2) >= 0.62 * 1.05 * fz_cap(1) * p3;
      if (p3 >= 1 && p3 <= FZ_MAX_P && (K + p3 - 1) / p3 + 64 <= TS_MAX_KP3 && long_enough) h->exact_single = true;
    }
    const int max_kp = h->exact_single ? TS_MAX_KP3 - 64 : TS_MAX_KP;
    int P = h->opt_P > 0 ? (int)h->opt_P : (K + max_kp - 1) / max_kp;
    if (P < 1) P = 1;
    if (h->opt_P <= 0 && h->em_kernel != TSEM_EMK_TWOPASS && P < FZ_MAX_P && na > 0) {
      // Teams never span XCDs, so floor(cpx / P) * P of an XCD's cpx CUs work: 28 of 32 for teams of 7.
      // One more member per team is worth it when it puts >= 10 % more CUs to work and the rows are
      // long enough to fill the register tiles of the larger team (measured: K = 50k, 100 nnz/row,
      // P 7 -> 8: fp64 5.80 -> 5.48 ms, codes 4.82 -> 4.28 ms; K = 38k, 40 nnz/row is better off at P = 5).
      const int cpx = std::max(1, h->n_cu / 8);
      auto util = [&](int p) { return (double)(cpx / p * p) / cpx; };
      const double mean_len = (double)(h->nnz - nu) / (double)na;
      for (int p2 = P + 1; p2 <= FZ_MAX_P; ++p2)
        if (util(p2) >= util(P) + 0.10 && mean_len * fz_rmax(2) >= 1.05 * fz_cap(1) * p2) { P = p2; break; }
    }
    if (P > 64) TSEM_FAIL(TSEM_ERR_ARG, "more than 64 column parts (K > 491520) is not supported");
    int Kp = (K + P - 1) / P;
    if (Kp > TS_MAX_KP) TSEM_FAIL(TSEM_ERR_ARG, "parts option leaves more than 7680 columns per part");
    // spare accumulator slots per part for very popular columns (build_layout splits them)
    h->hot_extra = h->opt_hot_split ? std::min(64, (h->exact_single ? TS_MAX_KP3 : TS_MAX_KP) - Kp) : 0;
    Kp += h->hot_extra;
    h->P = P; h->Kp = Kp; h->Kpad = P * Kp;
    h->use_fused = (h->em_kernel != TSEM_EMK_TWOPASS) && P <= FZ_MAX_P;   // AUTO: fused when the layout allows it
    int R = 2048;
    h->geo = P > 4 ? 1 : 0;
    h->run_len_est = na > 0 ? (double)(h->nnz - nu) / (double)na / P : 0.0;   // entries per ambiguous row and part
    if (h->use_fused && na > 0) {
      // size blocks so a member's sub-block (~R*len/P entries) fills ~85 % of its register tile
      double mean_len = (double)(h->nnz - nu) / (double)na;
      // row SLOTS per block: ~7 % above the average a register tile takes, so blocks end on the
      // tile's capacity, not on R (the exchange cost depends on R, hence not more than needed)
      // geometry: teams of 5-8 have one; smaller teams switch to three exchange waves when the rows
      // are so short that 512 row slots cannot fill the register tile and the pass is bound by the
      // exchange (fp64 entries; with score codes the 14th data wave is worth more)
      // (profiles/r02_sweep_short.txt, 50M rows, both entry formats: the third exchange wave pays once the tile
      // needs more than ~1.25x the 512 row slots of geometry 0 — 20 entries per row: codes 2.21 -> 1.98 ms, fp64
      // 2.63 -> 2.57; 28 per row: codes 2.55 -> 2.66, fp64 equal)
      // teams of 5-8 (round 3): 768 row slots (geometry 2: two row pairs per exchange lane, (P - 1) x 2 partner values in its
      // registers: 118-124 VGPRs, no spill) when 384 cannot fill the register tiles
      h->geo = P > 4 ? ((1.07 * fz_cap(1) * P / std::max(2.0, mean_len) > 1.25 * fz_rmax(1)) ? 2 : 1)
                     : ((1.07 * fz_cap(0) * P / std::max(2.0, mean_len) > 1.25 * fz_rmax(0)) ? 2 : 0);
      // rows so short that 768 of them cannot fill the tile either: geometry 3 (32 B of LDS per row slot instead of 48)
      // (profiles/r03_sweep_short.txt: 8 / 10 / 12 entries per row 1.27 / 1.31 / 1.39 -> 1.18 / 1.23 / 1.34 ms, 14 equal, 16 and more slower:
      //  the exchange of a step grows with its row slots)
      if (P <= 4 && h->geo == 2 && 1.07 * fz_cap(2) * P / std::max(2.0, mean_len) > 1.4 * fz_rmax(2)) h->geo = 3;
      if (h->opt_geo >= 0 && P <= 4) h->geo = (h->opt_geo == 2 || h->opt_geo == 3) ? (int)h->opt_geo : 0;
      if (h->opt_geo >= 0 && P > 4) h->geo = h->opt_geo == 2 ? 2 : 1;
      double r = 1.07 * fz_cap(h->geo) * P / std::max(2.0, mean_len);
      const int lut_bytes = (h->lut_len > 0 && h->lut_len <= 2048) ? h->lut_len * 8 : 0;   // the score table shares LDS with the rings
      int rmax = std::min(fz_rmax(h->geo), (TS_LDS_MAX - 2560 - (h->exact_single ? 3 : 2) * Kp * 8 - lut_bytes - std::max(h->opt_reproducible ? Kp * 2 + 16 : 0, FZ_LOGTAB * 16 + 16)) / ((fz_yr(h->geo) + 2) * 8));
      rmax = std::min(rmax, FILL_MAX_RP / P);                // (k_sb_fill_sorted keeps R x P counters in LDS)
      R = (int)std::min<double>(r, rmax);
      R = std::max(64, (R + 63) / 64 * 64);
      R = std::min(R, rmax / 8 * 8);
    }
    if (h->opt_R > 0) R = (int)h->opt_R;
    h->R = R;
  }
  if (h->R > 65536 || h->R < 64) TSEM_FAIL(TSEM_ERR_ARG, "block_rows must be in [64, 65536]");
  h->nb = (na + h->R - 1) / h->R;
  h->N_amb_pad = std::max<int64_t>(1, h->nb) * h->R;
  return TSEM_OK;
}

// ---------------------------------------------------------------------------
// rowstats: classes, weights, local sums; compacts ambiguous / unique rows
// ---------------------------------------------------------------------------
__global__ void k_pisum_finish(int K, const double* __restrict__ lv, double* __restrict__ pisum0) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= K) return;
  double t = 0.0;
  for (int l = PIS_LEVELS - 1; l >= 0; --l) t += lv[(size_t)l * K + j];   // small to large
  pisum0[j] = t;
}

int tsem_rowstats(tsem_ctx* h, double* stats3, double* pisum0, uint64_t* col_count, uint64_t* col_hash) {
  if (!h || !h->d_indptr) return TSEM_ERR_ARG;
  if (!h->d_lut || h->lut_len <= 0) TSEM_FAIL(TSEM_ERR_ARG, "no score table: call tsem_set_lut after tsem_generate");
  if (int rc = ensure_device(h)) return rc;
  const int64_t N = h->N;
  const int K = h->K;
  uint16_t* d_code = nullptr; uint8_t* d_cls = nullptr; double* d_wpart = nullptr;
  int32_t *d_fa = nullptr, *d_fu = nullptr;
  TSEM_ALLOC(d_code, N); TSEM_ALLOC(d_cls, N);
  const int grid = (int)std::min<int64_t>(4096, std::max<int64_t>(1, (N + 15) / 16));
  TSEM_ALLOC(d_wpart, 2 * grid);
  TSEM_ALLOC(h->d_pisum0, K);
  TSEM_ALLOC(h->d_ucount, K + 1);
  TSEM_HIP(hipMemsetAsync(h->d_ucount, 0, sizeof(uint32_t) * (K + 1), h->stream));
  double* d_pis_lv = nullptr;
  TSEM_ALLOC(d_pis_lv, (size_t)PIS_LEVELS * K);
  TSEM_HIP(hipMemsetAsync(d_pis_lv, 0, sizeof(double) * PIS_LEVELS * K, h->stream));
  int pis_e2 = 0;
  (void)std::frexp(h->lut_host[h->lut_len - 1] > 0 ? h->lut_host[h->lut_len - 1] : 1.0, &pis_e2);   // Q < 2^e2 (the table is increasing)
  TSEM_HIP(hipMemsetAsync(h->d_maxcode, 0, 4, h->stream));
  TSEM_HIP(hipMemsetAsync(d_wpart, 0, sizeof(double) * 2 * grid, h->stream));
  unsigned long long* d_lg = nullptr;
  TSEM_ALLOC(d_lg, 8);
  TSEM_HIP(hipMemsetAsync(d_lg, 0, 64, h->stream));
  if (N) {
    const double mean_len = (double)h->nnz / (double)N;    // lanes per row x 16 entries >= ~1.5 mean row lengths
    const int G = mean_len * 1.5 <= 16 ? 1 : mean_len * 1.5 <= 32 ? 2 : mean_len * 1.5 <= 64 ? 4 : mean_len * 1.5 <= 128 ? 8 : 16;
    auto rk = G == 1 ? k_rowstats<1> : G == 2 ? k_rowstats<2> : G == 4 ? k_rowstats<4> : G == 8 ? k_rowstats<8> : k_rowstats<16>;
    rk<<<grid, 256, 0, h->stream>>>(N, h->d_indptr, h->d_indices, h->d_raw, h->d_lut, d_code, d_cls,
                                    d_wpart, h->d_maxcode, d_pis_lv, pis_e2 + 1023, h->d_ucount, K, d_lg);
  }
  k_pisum_finish<<<cdiv64(K, 256), 256, 0, h->stream>>>(K, d_pis_lv, h->d_pisum0);
  TSEM_HIP(hipGetLastError());
  TSEM_HIP(hipMemcpyAsync(h->len_gt, d_lg, 6 * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
  std::vector<double> wpart(2 * grid);
  uint32_t maxcode = 0;
  TSEM_HIP(hipMemcpyAsync(wpart.data(), d_wpart, sizeof(double) * 2 * grid, hipMemcpyDeviceToHost, h->stream));
  TSEM_HIP(hipMemcpyAsync(&maxcode, h->d_maxcode, 4, hipMemcpyDeviceToHost, h->stream));
  TSEM_HIP(hipStreamSynchronize(h->stream));
  double wt = 0, wa = 0;
  for (int i = 0; i < grid; ++i) { wt += wpart[2 * i]; wa += wpart[2 * i + 1]; }
  if (stats3) { stats3[0] = wt; stats3[1] = wa; stats3[2] = (N && h->nnz) ? h->lut_host[maxcode] : 0.0; }
  if (pisum0) TSEM_HIP(hipMemcpy(pisum0, h->d_pisum0, sizeof(double) * K, hipMemcpyDeviceToHost));

  // column signatures (popularity + twin detection)
  if (col_count && col_hash) {
    unsigned long long *d_cnt = nullptr, *d_hash = nullptr;
    TSEM_ALLOC(d_cnt, K); TSEM_ALLOC(d_hash, K);
    TSEM_HIP(hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long) * K, h->stream));
    TSEM_HIP(hipMemsetAsync(d_hash, 0, sizeof(unsigned long long) * K, h->stream));
    if (N) {
      const int lds = SIG_WIN * 8;
      // lanes per row from the row-length histogram taken a moment ago: 16 entries per lane, the smallest group that
      // takes 99.5 % of the rows in one step (longer rows loop)
      int cap = 256;
      for (int q = 1; q < 6; ++q)
        if ((double)h->len_gt[q] <= 0.005 * (double)N) { cap = 8 << q; break; }
      void (*ck)(int64_t, int64_t, const int64_t*, const int32_t*, const uint16_t*, int, int, unsigned long long*, unsigned long long*) =
          cap <= 16 ? k_colsig<1> : cap <= 32 ? k_colsig<2> : cap <= 64 ? k_colsig<4> : cap <= 128 ? k_colsig<8> : k_colsig<16>;
      TSEM_HIP(hipFuncSetAttribute((const void*)ck, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      int g2 = (int)std::min<int64_t>(h->n_cu, std::max<int64_t>(1, (N + 63) / 64));
      for (int base = 0; base < K; base += SIG_WIN)
        ck<<<g2, 1024, lds, h->stream>>>(N, h->row_offset, h->d_indptr, h->d_indices, h->d_raw, base, K, d_cnt, d_hash);
      TSEM_HIP(hipGetLastError());
    }
    TSEM_HIP(hipMemcpyAsync(col_count, d_cnt, sizeof(uint64_t) * K, hipMemcpyDeviceToHost, h->stream));
    TSEM_HIP(hipMemcpyAsync(col_hash, d_hash, sizeof(uint64_t) * K, hipMemcpyDeviceToHost, h->stream));
    TSEM_HIP(hipStreamSynchronize(h->stream));
    dfree(h->d_colcount);
    h->d_colcount = d_cnt;                                 // LOCAL stored entries per column: reassign('all', initial) of this rank
    (void)hipFree(d_hash);
  }
  // compact ambiguous and unique rows
  TSEM_ALLOC(d_fa, N + 1); TSEM_ALLOC(d_fu, N + 1);
  int32_t na = 0, nu = 0;
  TSEM_HIP(hipMemsetAsync(d_fa + N, 0, 4, h->stream));
  TSEM_HIP(hipMemsetAsync(d_fu + N, 0, 4, h->stream));
  if (N) {
    k_class_flags<<<cdiv64(N, 256), 256, 0, h->stream>>>(N, d_cls, d_fa, d_fu);
    size_t tb = 0;
    TSEM_HIP(rocprim::exclusive_scan(nullptr, tb, d_fa, d_fa, (int32_t)0, (size_t)(N + 1), rocprim::plus<int32_t>(), h->stream));
    void* tmp = nullptr;
    TSEM_HIP(hipMalloc(&tmp, tb ? tb : 1));
    TSEM_HIP(rocprim::exclusive_scan(tmp, tb, d_fa, d_fa, (int32_t)0, (size_t)(N + 1), rocprim::plus<int32_t>(), h->stream));
    TSEM_HIP(rocprim::exclusive_scan(tmp, tb, d_fu, d_fu, (int32_t)0, (size_t)(N + 1), rocprim::plus<int32_t>(), h->stream));
    TSEM_HIP(hipMemcpyAsync(&na, d_fa + N, 4, hipMemcpyDeviceToHost, h->stream));
    TSEM_HIP(hipMemcpyAsync(&nu, d_fu + N, 4, hipMemcpyDeviceToHost, h->stream));
    TSEM_HIP(hipStreamSynchronize(h->stream));
    (void)hipFree(tmp);
  }
  h->N_amb = na; h->N_uni = nu;
  if (int rc = choose_geometry(h)) return rc;
  TSEM_ALLOC(h->d_amb_row, na);
  TSEM_ALLOC(h->d_amb_wcode_c, na);                       // per compact row; build_layout makes the slot copy
  TSEM_ALLOC(h->d_uni_col, nu);
  TSEM_ALLOC(h->d_uni_code, nu);
  if (N)
    k_compact_rows<<<cdiv64(N, 256), 256, 0, h->stream>>>(N, d_cls, d_fa, d_fu, h->d_indptr, h->d_indices, h->d_raw,
                                                         d_code, h->d_amb_row, h->d_amb_wcode_c, h->d_uni_col,
                                                         h->d_uni_code);
  TSEM_HIP(hipGetLastError());
  TSEM_HIP(hipStreamSynchronize(h->stream));
  (void)hipFree(d_code); (void)hipFree(d_cls); (void)hipFree(d_wpart); (void)hipFree(d_fa); (void)hipFree(d_fu); (void)hipFree(d_lg); (void)hipFree(d_pis_lv);
  h->have_rowstats = true;
  return TSEM_OK;
}

// ---------------------------------------------------------------------------
// layout: column partition by popularity, blocked COO of ambiguous rows
// ---------------------------------------------------------------------------
static int build_layout(tsem_ctx* h) {
  PhaseTimer pt(h->stream);
  const int K = h->K;
  const int64_t na = h->N_amb;
  free_layout(h);
  h->nnz_amb = 0;
  // 1. column popularity: global entry counts handed in by set_model
  const std::vector<uint64_t>& counts = h->col_count;
  // 2. parts: deal columns by popularity so every part carries ~equal nnz
  const int P = h->P, Kp = h->Kp;
  std::vector<int> order(K);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return counts[a] > counts[b]; });
  // colmap[j] = part << 16 | log2(copies) << 13 | first slot.  A column that holds a large share of its
  // part's entries would serialise the LDS scatter (64 f lanes of every ds_add_f64 on ONE address:
  // the hottest column of a Zipf-like matrix, or Telescope's `__no_feature`, reaches 8-way), so it
  // gets 2..16 consecutive slots; k_sb_fill deals its entries over them, k_colreduce adds them up.
  std::vector<uint32_t> colmap(K);
  std::vector<int32_t> col_of_pc(h->Kpad, -1);
  // Columns go, most popular first, to the part that holds the fewest entries so far (and still has
  // a free slot): every member of a team then streams the same number of entries per row block, so
  // the register tiles of all parts fill evenly and no member waits for a heavier one.
  std::vector<double> part_nnz(P, 0.0);
  std::vector<int> part_of(K), ncols(P, 0);
  const int percap = h->Kp - h->hot_extra;                 // plain columns per part
  for (int rank = 0; rank < K; ++rank) {
    int best = -1;
    for (int p = 0; p < P; ++p)
      if (ncols[p] < percap && (best < 0 || part_nnz[p] < part_nnz[best])) best = p;
    part_of[rank] = best;
    ncols[best] += 1;
    part_nnz[best] += (double)counts[order[rank]];
  }
  std::vector<int> cursor(P, 0), spare(P, h->hot_extra);
  h->n_hot_cols = 0;
  for (int rank = 0; rank < K; ++rank) {
    const int j = order[rank], p = part_of[rank];
    const double lanes = 64.0 * (double)counts[j] / std::max(1.0, part_nnz[p]);
    int lg = 0;
    while (lg < 4 && lanes / (1 << lg) > 1.5 && (2 << lg) - 1 <= spare[p]) ++lg;
    spare[p] -= (1 << lg) - 1;
    if (lg) h->n_hot_cols += 1;
    colmap[j] = ((uint32_t)p << 16) | ((uint32_t)lg << 13) | (uint32_t)cursor[p];
    col_of_pc[p * Kp + cursor[p]] = j;                     // the first slot owns the column; the others stay -1
    cursor[p] += 1 << lg;
  }
  pt.lap("layout: column map (host)");
  TSEM_ALLOC(h->d_colmap, K);
  TSEM_ALLOC(h->d_col_of_pc, h->Kpad);
  TSEM_HIP(hipMemcpy(h->d_colmap, colmap.data(), sizeof(uint32_t) * K, hipMemcpyHostToDevice));
  TSEM_HIP(hipMemcpy(h->d_col_of_pc, col_of_pc.data(), sizeof(int32_t) * h->Kpad, hipMemcpyHostToDevice));
  // popularity ids for the report pass (k_report_rows): id = slot * P + part, 2 bytes per stored entry
  const bool want_rid = h->Kpad <= 65536 && h->opt_report_kernel != 0 && h->nnz > 0;
  if (want_rid) {
    std::vector<int32_t> col_of_id(h->Kpad, -1);
    for (int p = 0; p < P; ++p)
      for (int sl = 0; sl < Kp; ++sl) col_of_id[sl * P + p] = col_of_pc[p * Kp + sl];
    TSEM_ALLOC(h->d_col_of_id, h->Kpad);
    TSEM_HIP(hipMemcpy(h->d_col_of_id, col_of_id.data(), sizeof(int32_t) * h->Kpad, hipMemcpyHostToDevice));
    TSEM_ALLOC(h->d_rid16, h->nnz + TS_ENTRY_PAD);
  }
  pt.lap("layout: maps to the device, rid16 alloc");
  // 3. row blocks.  Two-pass layout: R rows each.  Fused layout: as many consecutive rows as the
  //    register tile takes (no part may exceed FZ_CAP entries, at most R rows) — rows per block vary,
  //    every block still owns R row SLOTS (holes at the end), so all kernels keep b*R+lr indexing.
  const int R = h->R;
  int64_t nb = 0;
  int64_t* d_bs = nullptr;                                 // first compact row of every block, [nb + 1]
  unsigned long long* d_pc = nullptr;                      // per-row part counts (fused layout only)
  bool rid_amb_done = false;
  if (h->use_fused && na > 0 && P <= FZ_MAX_P) {
    TSEM_ALLOC(d_pc, 2 * na);
    {
      int capc = 256;                                      // lanes per row x 16 entries, from the row-length histogram
      for (int q = 1; q < 6; ++q)
        if ((double)h->len_gt[q] <= 0.005 * (double)h->N) { capc = 8 << q; break; }
      const int G = capc <= 16 ? 1 : capc <= 32 ? 2 : capc <= 64 ? 4 : capc <= 128 ? 8 : 16;
      if ((size_t)K * 4 <= (size_t)TS_LDS_MAX - 2048 && na >= 65536) {   // the column map fits LDS (K <= 40k) and the matrix is worth a 120 KB preload per CU
        auto pk = G == 1 ? k_row_partcounts<1, true> : G == 2 ? k_row_partcounts<2, true> : G == 4 ? k_row_partcounts<4, true>
                : G == 8 ? k_row_partcounts<8, true> : k_row_partcounts<16, true>;
        TSEM_HIP(hipFuncSetAttribute((const void*)pk, hipFuncAttributeMaxDynamicSharedMemorySize, TS_LDS_MAX - 1024));
        pk<<<h->n_cu, 1024, (size_t)K * 4, h->stream>>>(na, h->d_amb_row, h->d_indptr, h->d_indices, h->d_colmap, K, d_pc, h->d_rid16, P);
      } else {
        const unsigned grid = (unsigned)std::min<int64_t>(65535, (na + 256 / G - 1) / (256 / G));
        auto pk = G == 1 ? k_row_partcounts<1, false> : G == 2 ? k_row_partcounts<2, false> : G == 4 ? k_row_partcounts<4, false>
                : G == 8 ? k_row_partcounts<8, false> : k_row_partcounts<16, false>;
        pk<<<grid, 256, 0, h->stream>>>(na, h->d_amb_row, h->d_indptr, h->d_indices, h->d_colmap, K, d_pc, h->d_rid16, P);
      }
    }
    TSEM_HIP(hipGetLastError());
    rid_amb_done = true;
    const int cap = fz_cap(h->geo) - TS_STRANDS * 4;          // sub-blocks are padded to TS_STRANDS*4 entries
    // chunk length: >= 64 blocks' worth of rows (the forced break at a chunk end costs ~0.8 % more blocks; round 2 used 256 blocks' worth,
    // 484 sequential waves for 47M rows: 2 x 2.5 ms; four times as many waves walk a quarter each)
    const int64_t L = std::max<int64_t>((int64_t)R * 64, (na + 16383) / 16384);
    const int64_t nch = (na + L - 1) / L;
    int64_t *d_cnt = nullptr, *d_off = nullptr;
    int* d_flag = nullptr;
    TSEM_ALLOC(d_cnt, nch + 1); TSEM_ALLOC(d_off, nch + 1); TSEM_ALLOC(d_flag, 1);
    TSEM_HIP(hipMemsetAsync(d_flag, 0, sizeof(int), h->stream));
    TSEM_HIP(hipMemsetAsync(d_cnt, 0, sizeof(int64_t) * (nch + 1), h->stream));
    k_block_greedy<<<(unsigned)nch, 64, 0, h->stream>>>(na, P, R, cap, L, 0, d_pc, d_cnt, nullptr, nullptr, d_flag);
    TSEM_HIP(hipGetLastError());
    {
      size_t tb = 0;
      TSEM_HIP(rocprim::exclusive_scan(nullptr, tb, d_cnt, d_off, (int64_t)0, (size_t)(nch + 1), rocprim::plus<int64_t>(), h->stream));
      void* tmp = nullptr;
      TSEM_HIP(hipMalloc(&tmp, tb ? tb : 1));
      TSEM_HIP(rocprim::exclusive_scan(tmp, tb, d_cnt, d_off, (int64_t)0, (size_t)(nch + 1), rocprim::plus<int64_t>(), h->stream));
      int flag = 0;
      TSEM_HIP(hipMemcpyAsync(&nb, d_off + nch, sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
      TSEM_HIP(hipMemcpyAsync(&flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, h->stream));
      TSEM_HIP(hipStreamSynchronize(h->stream));
      (void)hipFree(tmp);
      if (flag) { h->use_fused = false; nb = 0; }          // one row overflows the register tile
    }
    if (h->use_fused) {
      TSEM_ALLOC(d_bs, nb + 1);
      k_block_greedy<<<(unsigned)nch, 64, 0, h->stream>>>(na, P, R, cap, L, 1, d_pc, d_cnt, d_off, d_bs, d_flag);
      TSEM_HIP(hipGetLastError());
      TSEM_HIP(hipMemcpyAsync(d_bs + nb, &na, sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
      TSEM_HIP(hipStreamSynchronize(h->stream));
    }
    (void)hipFree(d_cnt); (void)hipFree(d_off); (void)hipFree(d_flag);
    if (!h->use_fused) { (void)hipFree(d_pc); d_pc = nullptr; }
  }
  if (!d_bs) {                                             // two-pass layout: R rows per block
    nb = (na + R - 1) / R;
    TSEM_ALLOC(d_bs, nb + 1);
    k_fixed_blocks<<<cdiv64(nb + 1, 256), 256, 0, h->stream>>>(nb, R, na, d_bs);
    TSEM_HIP(hipGetLastError());
  }
  if (h->d_rid16 && h->N) {                                // the rows k_row_partcounts did not visit (all of them without the fused layout)
    k_rid16_rows<<<(unsigned)std::min<int64_t>(65535, (h->N + 15) / 16), 256, 0, h->stream>>>(
        h->N, h->d_indptr, h->d_indices, h->d_colmap, P, rid_amb_done ? 1 : 0, h->d_rid16);
    TSEM_HIP(hipGetLastError());
  }
  h->nb = nb;
  h->N_amb_pad = std::max<int64_t>(1, nb) * R;
  {
    TSEM_ALLOC(h->d_slot_row, h->N_amb_pad);
    TSEM_ALLOC(h->d_amb_wcode, h->N_amb_pad);
    if (nb) k_make_slots<<<(unsigned)nb, 256, 0, h->stream>>>(nb, R, d_bs, h->d_amb_row, h->d_amb_wcode_c,
                                                             h->d_slot_row, h->d_amb_wcode);
    else TSEM_HIP(hipMemsetAsync(h->d_amb_wcode, 0, sizeof(uint16_t) * h->N_amb_pad, h->stream));
    TSEM_HIP(hipGetLastError());
  }
  pt.lap("layout: part counts, blocks, slots");
  // 4. sub-block sizes -> offsets
  std::vector<int64_t> sb(nb * P + 1, 0);
  if (nb) {
    int64_t* d_cnt = nullptr;
    TSEM_ALLOC(d_cnt, nb * P);
    if (d_pc) k_sb_count_pc<<<(unsigned)nb, 64, 0, h->stream>>>(nb, P, d_bs, d_pc, d_cnt);
    else k_sb_count<<<(unsigned)nb, 256, 0, h->stream>>>(na, R, P, h->d_slot_row, h->d_indptr, h->d_indices, h->d_colmap, d_cnt);
    TSEM_HIP(hipGetLastError());
    TSEM_HIP(hipMemcpyAsync(sb.data(), d_cnt, sizeof(int64_t) * nb * P, hipMemcpyDeviceToHost, h->stream));
    TSEM_HIP(hipStreamSynchronize(h->stream));
    (void)hipFree(d_cnt);
  }
  if (h->use_fused) {
    int64_t mx = 0;
    for (int64_t i = 0; i < nb * P; ++i) mx = std::max(mx, sb[i]);
    h->max_subblock = mx;
    if (mx > fz_cap(h->geo)) h->use_fused = false;
  }
  int64_t off = 0;
  for (int64_t i = 0; i < nb * P; ++i) {   // sub-blocks padded to TS_STRANDS*4 entries (strand-transposed order)
    h->nnz_amb += sb[i];
    int64_t c = (sb[i] + (TS_STRANDS * 4 - 1)) / (TS_STRANDS * 4) * (TS_STRANDS * 4);
    sb[i] = off; off += c;
  }
  sb[nb * P] = off;
  h->nnz_pad = off;
  TSEM_ALLOC(h->d_sb_off, nb * P + 1);
  TSEM_HIP(hipMemcpy(h->d_sb_off, sb.data(), sizeof(int64_t) * (nb * P + 1), hipMemcpyHostToDevice));
  if (h->use_fused && (off >> 2) < 0xFFFFFFFFll) {
    std::vector<uint32_t> q32(nb * P + 2, 0);
    for (int64_t i = 0; i <= nb * P; ++i) q32[i] = (uint32_t)(sb[i] >> 2);
    TSEM_ALLOC(h->d_sb_q32, nb * P + 2);
    TSEM_HIP(hipMemcpy(h->d_sb_q32, q32.data(), sizeof(uint32_t) * (nb * P + 2), hipMemcpyHostToDevice));
  } else if (h->use_fused) {
    h->use_fused = false;
  }
  if (h->use_fused && (R > fz_rmax(h->geo) || (R & 1) || fz_lds_bytes(h, false) > (size_t)TS_LDS_MAX - 1024)) h->use_fused = false;
  h->fmt_code = h->use_fused && fz_wants_codes(h) && fz_lds_bytes(h, true) <= (size_t)TS_LDS_MAX - 1024;
  h->fmt_wcode = h->use_fused && !h->fmt_code && h->lut_len > 0 && h->lut_len <= 2048 &&
                 fz_lds_bytes(h, true) <= (size_t)TS_LDS_MAX - 1024;
  if (h->opt_format == 2 && !h->fmt_code)
    TSEM_FAIL(TSEM_ERR_ARG, "value_format=codes needs the fused kernel and a score table of at most 2048 entries");
  if (h->opt_reproducible && !(h->use_fused && (h->fmt_code || h->fmt_wcode)))
    TSEM_FAIL(TSEM_ERR_ARG, "reproducible mode needs the fused kernel (at most 8 column parts, every row within the register tile) and a score "
                            "table of at most 2048 entries");
  pt.lap("layout: sub-block offsets");
  TSEM_ALLOC(h->d_prc, off);
  TSEM_HIP(hipMemsetAsync(h->d_prc, 0, sizeof(uint32_t) * std::max<int64_t>(1, off), h->stream));
  if (h->fmt_code) {
    TSEM_ALLOC(h->d_pcode, off);
    TSEM_HIP(hipMemsetAsync(h->d_pcode, 0, sizeof(uint16_t) * std::max<int64_t>(1, off), h->stream));   // code 0 -> Q = 0
  } else {
    TSEM_ALLOC(h->d_pval, off);
    TSEM_HIP(hipMemsetAsync(h->d_pval, 0, sizeof(double) * std::max<int64_t>(1, off), h->stream));
  }
  pt.lap("layout: entry buffers (alloc + zero)");
  // Row order (row sums reduced in registers, a tenth of the LDS atomics) for every fused layout.  Score codes: 40
  // entries per row at P = 4 4.44 -> 3.59 ms, 20 per row 2.30 -> 1.92, 10 per row 1.65 -> 1.40.  fp64 entries
  // were indifferent to it while the exchange wave stalled behind the memory pipe (round 1: 4.62 against 4.57 ms);
  // since the exchange is one generation per step, the step ends when the LDS queue has drained, and less LDS work
  // shortens it for them too: 40 per row 4.33 -> 4.14 ms (0.73 of the HBM peak), teams of 8 4.42 -> 4.20, 20 per
  // row 2.70 -> 2.48 (profiles/r02_sweep.txt, r02_sweep_short.txt).
  h->sorted_layout = h->use_fused && R * P <= FILL_MAX_RP &&   // (the fill kernel keeps R x P counters in LDS)
                     (h->opt_sorted >= 0 ? h->opt_sorted != 0 : true);
  if (nb && h->sorted_layout) {
    // the popularity ids stand in for the column-map gather when every row's ids are written (they are: k_row_partcounts +
    // k_rid16_rows above) and the split columns' ids fit the small table
    const uint16_t* rid_fill = nullptr;
    uint8_t* d_lgtab = nullptr;
    int nsplit = 0;
    if (h->d_rid16 && P <= 8) {
      std::vector<uint8_t> lgt;
      for (int j = 0; j < K; ++j) {
        const uint32_t cm = colmap[j], lg = (cm >> 13) & 7u;
        if (lg) { const uint32_t id = (cm & 0x1FFFu) * P + (cm >> 16); if (id >= lgt.size()) lgt.resize(id + 1, 0); lgt[id] = (uint8_t)lg; }
      }
      nsplit = (int)lgt.size();
      if (nsplit <= 4096) {
        TSEM_ALLOC(d_lgtab, std::max(1, nsplit));
        if (nsplit) TSEM_HIP(hipMemcpyAsync(d_lgtab, lgt.data(), nsplit, hipMemcpyHostToDevice, h->stream));
        TSEM_HIP(hipStreamSynchronize(h->stream));         // (lgt is a local)
        rid_fill = h->d_rid16;
      }
    }
    const uint32_t magicP = (uint32_t)((0x100000000ull + (uint64_t)P - 1) / (uint64_t)P);
    k_sb_fill_sorted<<<(unsigned)nb, 256, fill_lds_bytes(R, P), h->stream>>>(na, R, P, h->d_slot_row, h->d_indptr, h->d_indices, h->d_raw, h->d_lut,
                                                         h->d_colmap, h->d_sb_off, h->d_pval, h->d_pcode, h->d_prc,
                                                         d_pc ? d_bs : nullptr, d_pc, rid_fill, magicP, nsplit, d_lgtab);
    TSEM_HIP(hipGetLastError());
    if (d_lgtab) { TSEM_HIP(hipStreamSynchronize(h->stream)); (void)hipFree(d_lgtab); }
    TSEM_HIP(hipGetLastError());
    // (reproducible mode keeps the row order: a row's entries in a sub-block then form ONE run, which ends in at most two LDS
    // atomics on its row sum — two additions commute, three need not)
    if (h->fmt_code && h->opt_deconflict != 0 && !h->opt_reproducible && off >= 64) {
      const int64_t n_win = off / 64;
      uint32_t* prc2 = nullptr; uint16_t* code2 = nullptr;
      TSEM_ALLOC(prc2, off); TSEM_ALLOC(code2, off);
      k_sb_deconflict<<<(unsigned)std::min<int64_t>(n_win / DC_NT + 1, (int64_t)h->n_cu * 32), DC_NT, 0, h->stream>>>(
          n_win, h->d_prc, h->d_pcode, prc2, code2);
      TSEM_HIP(hipGetLastError());
      TSEM_HIP(hipStreamSynchronize(h->stream));
      (void)hipFree(h->d_prc); (void)hipFree(h->d_pcode);
      h->d_prc = prc2; h->d_pcode = code2;
    }
  } else if (nb) {
    k_sb_fill<<<(unsigned)nb, 256, 0, h->stream>>>(na, R, P, h->d_slot_row, h->d_indptr, h->d_indices, h->d_raw, h->d_lut,
                                                  h->d_colmap, h->d_sb_off, h->d_pval, h->d_pcode, h->d_prc);
    TSEM_HIP(hipGetLastError());
  }
  TSEM_HIP(hipStreamSynchronize(h->stream));
  (void)hipFree(d_bs);
  if (d_pc) (void)hipFree(d_pc);
  pt.lap("layout: fill + conflict-aware order");
  if (!h->use_fused) TSEM_ALLOC(h->d_ypart, (int64_t)P * h->N_amb_pad);   // partial row sums of the two-pass kernels
  // launch geometry
  const size_t lds1 = (size_t)(Kp + R) * 8, lds2 = (size_t)(2 * Kp + R) * 8;
  if (lds2 > (size_t)TS_LDS_MAX - 1024) TSEM_FAIL(TSEM_ERR_ARG, "LDS budget exceeded (reduce block_rows)");
  int w1 = std::max(1, std::min(4, (int)(TS_LDS_MAX / lds1)));   // 512-thread WGs per CU
  int w2 = std::max(1, std::min(2, (int)(TS_LDS_MAX / lds2)));   // 1024-thread WGs per CU
  h->G1 = (int)std::max<int64_t>(1, std::min<int64_t>(nb, (int64_t)h->n_cu * w1 / P));
  h->G2 = (int)std::max<int64_t>(1, std::min<int64_t>(nb, (int64_t)h->n_cu * w2 / P));
  if (!h->use_fused) TSEM_ALLOC(h->d_partial, (int64_t)h->G2 * h->Kpad);
  if (h->use_fused) {
    {
      h->fz_grid = h->n_cu;
      h->fz_teams = std::max(1, h->fz_grid / P);
      TSEM_ALLOC(h->d_fpartial, (int64_t)h->fz_teams * h->Kpad);
      TSEM_ALLOC(h->d_xchg, (int64_t)h->fz_teams * FZ_XS * P * R);
      TSEM_ALLOC(h->d_xflags, FZ_SYNC_WORDS);
      TSEM_HIP(hipMemset(h->d_xflags, 0, sizeof(uint32_t) * FZ_SYNC_WORDS));
      if (!h->fmt_code && !h->fmt_wcode) {                 // fp64 row weights; otherwise the kernel reads d_amb_wcode
        TSEM_ALLOC(h->d_amb_w, h->N_amb_pad);
        k_row_weights<<<cdiv64(h->N_amb_pad, 256), 256, 0, h->stream>>>(h->N_amb_pad, h->d_amb_wcode, h->d_lut, h->d_amb_w);
      }
      for (int mode = 0; mode < 2; ++mode)
        TSEM_HIP(hipFuncSetAttribute((const void*)fz_kernel(P, mode, fz_fmt(h), h->geo),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, TS_LDS_MAX - 1024));
      if (h->opt_reproducible) {
        if (h->exact_single && !fz_kernel(P, 3, fz_fmt(h), h->geo)) h->exact_single = false;   // (not score codes after all: fz_lds_bytes then counts two tables again)
        if (h->exact_single) TSEM_ALLOC(h->d_fpartial2, (int64_t)h->fz_teams * h->Kpad);
        fz_fn f2 = fz_kernel(P, h->exact_single ? 3 : 2, fz_fmt(h), h->geo);
        if (!f2) TSEM_FAIL(TSEM_ERR_ARG, "reproducible mode needs the fused kernel with a score table of at most 2048 entries");
        TSEM_HIP(hipFuncSetAttribute((const void*)f2, hipFuncAttributeMaxDynamicSharedMemorySize, TS_LDS_MAX - 1024));
        TSEM_ALLOC(h->d_ebias, h->Kpad); TSEM_ALLOC(h->d_ovf, h->Kpad); TSEM_ALLOC(h->d_red_hi, K + 2); TSEM_ALLOC(h->d_binflag, 4);
        TSEM_ALLOC(h->d_ehist, 2 * (size_t)K + 2);
        if (int rc = bin_reset(h)) return rc;
      }
    }
  }
  TSEM_HIP(hipFuncSetAttribute((const void*)k_phase1<512>, hipFuncAttributeMaxDynamicSharedMemorySize, TS_LDS_MAX));
  TSEM_HIP(hipFuncSetAttribute((const void*)k_phase2_em<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, TS_LDS_MAX));
  TSEM_HIP(hipFuncSetAttribute((const void*)k_phase2_lnl<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, TS_LDS_MAX - 1024));
  TSEM_HIP(hipStreamSynchronize(h->stream));
  pt.lap("layout: launch buffers, attributes");
  return TSEM_OK;
}

int tsem_set_model(tsem_ctx* h, const double* stats3, const double* pisum0, const uint64_t* col_count,
                   const uint64_t* col_hash, double pi_prior, double theta_prior) {
  if (!h || !h->have_rowstats || !stats3 || !pisum0 || !col_count || !col_hash) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  PhaseTimer pt0(h->stream);
  const int K = h->K;
  h->col_count.assign(col_count, col_count + K);
  {  // exact twin columns -> representative = smallest column index of the class
    struct Sig { uint64_t count, hash; int col; };            // (sorted in place)
    std::vector<Sig> sg((size_t)K);
    for (int j = 0; j < K; ++j) sg[j] = Sig{col_count[j], col_hash[j], j};
    std::sort(sg.begin(), sg.end(), [](const Sig& a, const Sig& b) {
      if (a.count != b.count) return a.count < b.count;
      if (a.hash != b.hash) return a.hash < b.hash;
      return a.col < b.col;
    });
    std::vector<int> ord(K);
    for (int j = 0; j < K; ++j) ord[j] = sg[j].col;
    std::vector<int32_t> rep(K);
    h->n_twin_cols = 0;
    for (int i = 0; i < K;) {
      int j = i;
      while (j + 1 < K && col_count[ord[j + 1]] == col_count[ord[i]] && col_hash[ord[j + 1]] == col_hash[ord[i]]) ++j;
      for (int t = i; t <= j; ++t) rep[ord[t]] = (col_count[ord[i]] == 0) ? ord[t] : ord[i];
      if (j > i && col_count[ord[i]] != 0) h->n_twin_cols += (j - i + 1);
      i = j + 1;
    }
    TSEM_ALLOC(h->d_twin_rep, K);
    TSEM_HIP(hipMemcpy(h->d_twin_rep, rep.data(), sizeof(int32_t) * K, hipMemcpyHostToDevice));
    h->twin_rep_host = rep;
  }
  h->W_tot = stats3[0]; h->W_amb = stats3[1]; h->w_max = stats3[2];
  h->pi_prior = pi_prior; h->theta_prior = theta_prior;
  {
    std::vector<double> ps(pisum0, pisum0 + K);
    for (int j = 0; j < K; ++j) {   // twins: identical unique-row sums up to atomics order
      int r = h->twin_rep_host[j];
      if (r != j && std::fabs(ps[j] - ps[r]) <= 1e-12 * std::max(std::fabs(ps[j]), std::fabs(ps[r]))) ps[j] = ps[r];
    }
    TSEM_HIP(hipMemcpy(h->d_pisum0, ps.data(), sizeof(double) * K, hipMemcpyHostToDevice));
  }
  pt0.lap("set_model: twins, pisum0 (host)");
  if (int rc = build_layout(h)) return rc;
  pt0.lap("set_model: build_layout total");
  TSEM_ALLOC(h->d_pi, K); TSEM_ALLOC(h->d_theta, K); TSEM_ALLOC(h->d_pi_prev, K); TSEM_ALLOC(h->d_theta_prev, K);
  TSEM_ALLOC(h->d_tmp_pi, K); TSEM_ALLOC(h->d_tmp_theta, K);
  TSEM_ALLOC(h->d_ctab, h->Kpad); TSEM_ALLOC(h->d_ctab_prev, h->Kpad);
  if (!h->d_red) {
    TSEM_ALLOC(h->d_red_own, K + 2);
    h->d_red = h->d_red_own; h->red_count = K + 2;
  } else if (h->red_count < K + 2) {
    TSEM_FAIL(TSEM_ERR_ARG, "bound reduce buffer is smaller than K+2 doubles");
  }
  TSEM_HIP(hipMemsetAsync(h->d_ctab, 0, sizeof(double) * h->Kpad, h->stream));
  TSEM_HIP(hipMemsetAsync(h->d_ctab_prev, 0, sizeof(double) * h->Kpad, h->stream));
  const double init = 1.0 / (double)K;   // model.py:667,673
  k_fill<<<cdiv64(K, 256), 256, 0, h->stream>>>(h->d_pi, K, init);
  k_fill<<<cdiv64(K, 256), 256, 0, h->stream>>>(h->d_theta, K, init);
  k_fill<<<cdiv64(K, 256), 256, 0, h->stream>>>(h->d_pi_prev, K, init);
  k_fill<<<cdiv64(K, 256), 256, 0, h->stream>>>(h->d_theta_prev, K, init);
  k_make_ctab<<<cdiv64(K, 256), 256, 0, h->stream>>>(K, h->d_pi, h->d_theta, h->d_colmap, h->Kp, h->d_ctab);
  k_make_ctab<<<cdiv64(K, 256), 256, 0, h->stream>>>(K, h->d_pi, h->d_theta, h->d_colmap, h->Kp, h->d_ctab_prev);
  TSEM_HIP(hipGetLastError());
  TSEM_HIP(hipStreamSynchronize(h->stream));
  h->have_model = true;
  h->lnl_prev_seed = INFINITY;                             // model.py:683
  h->em_cur = h->em_prev = true;                           // pi = theta = 1/K
  return TSEM_OK;
}

int tsem_set_params(tsem_ctx* h, const double* pi, const double* theta) {
  if (!h || !h->have_model || !pi || !theta) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  h->em_prev = h->em_cur; h->em_cur = false;               // arbitrary values (possibly 0): no shortcut in tsem_reassign
  const int K = h->K;
  TSEM_HIP(hipMemcpyAsync(h->d_pi_prev, h->d_pi, sizeof(double) * K, hipMemcpyDeviceToDevice, h->stream));
  TSEM_HIP(hipMemcpyAsync(h->d_theta_prev, h->d_theta, sizeof(double) * K, hipMemcpyDeviceToDevice, h->stream));
  TSEM_HIP(hipMemcpyAsync(h->d_ctab_prev, h->d_ctab, sizeof(double) * h->Kpad, hipMemcpyDeviceToDevice, h->stream));
  TSEM_HIP(hipMemcpyAsync(h->d_pi, pi, sizeof(double) * K, hipMemcpyHostToDevice, h->stream));
  TSEM_HIP(hipMemcpyAsync(h->d_theta, theta, sizeof(double) * K, hipMemcpyHostToDevice, h->stream));
  k_make_ctab<<<cdiv64(K, 256), 256, 0, h->stream>>>(K, h->d_pi, h->d_theta, h->d_colmap, h->Kp, h->d_ctab);
  TSEM_HIP(hipGetLastError());
  TSEM_HIP(hipStreamSynchronize(h->stream));
  return bin_reset(h);
}

int tsem_get_params(tsem_ctx* h, int which, double* pi, double* theta) {
  if (!h || !h->have_model) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  TSEM_HIP(hipStreamSynchronize(h->stream));
  if (which == TSEM_Z_FIRST && !h->d_pi_first) TSEM_FAIL(TSEM_ERR_ARG, "no EM run yet: pi_init / theta_init are not set");
  const double* sp = which == TSEM_Z_FIRST ? h->d_pi_first : (which == TSEM_Z_PREV ? h->d_pi_prev : h->d_pi);
  const double* st = which == TSEM_Z_FIRST ? h->d_theta_first : (which == TSEM_Z_PREV ? h->d_theta_prev : h->d_theta);
  if (pi) TSEM_HIP(hipMemcpy(pi, sp, sizeof(double) * h->K, hipMemcpyDeviceToHost));
  if (theta) TSEM_HIP(hipMemcpy(theta, st, sizeof(double) * h->K, hipMemcpyDeviceToHost));
  return TSEM_OK;
}

int tsem_reduce_buffer(tsem_ctx* h, void** dptr, int64_t* count) {
  if (!h || !h->have_model) return TSEM_ERR_ARG;
  if (dptr) *dptr = h->d_red;
  if (count) *count = h->K + 2;
  return TSEM_OK;
}

int tsem_bind_reduce_buffer(tsem_ctx* h, void* dptr, int64_t count) {
  if (!h || !dptr) return TSEM_ERR_ARG;
  if (h->K && count < h->K + 2) TSEM_FAIL(TSEM_ERR_ARG, "reduce buffer needs K+2 doubles");
  h->d_red = (double*)dptr;
  h->red_count = count;
  return TSEM_OK;
}

// ---------------------------------------------------------------------------
// EM pass / update / lnl
// ---------------------------------------------------------------------------
static int launch_phase1(tsem_ctx* h, const double* ctab, int64_t b0 = 0, int64_t b1 = -1, bool em = false) {
  if (h->nb == 0) return TSEM_OK;
  if (b1 < 0) b1 = h->nb;
  const size_t lds1 = (size_t)(h->Kp + h->R) * 8;
  k_phase1<512><<<h->G1 * h->P, 512, lds1, h->stream>>>(h->P, h->Kp, h->R, b0, b1, h->G1, h->N_amb_pad, h->d_sb_off,
                                                        h->d_pval, h->d_prc, ctab, h->d_ypart, em ? h->d_ctl : nullptr);
  TSEM_HIP(hipGetLastError());
  return TSEM_OK;
}

static int begin_timing(tsem_ctx* h, hipEvent_t** pair) {
  *pair = nullptr;
  // option "kernel_timing" = n: HIP events around every n-th EM pass (0 = never, default 1).  An event pair costs
  // the stream several microseconds per iteration (profiles/r02_comm_overhead.txt): the Python host switches it
  // off for em(), bench.py samples every 4th launch of the timed region.
  if (h->opt_timing <= 0 || (h->em_launches % h->opt_timing) != 0) return TSEM_OK;
  if (h->ev_used + 2 > 8192) return TSEM_OK;
  while (h->ev.size() < h->ev_used + 2) {
    hipEvent_t e;
    TSEM_HIP(hipEventCreate(&e));
    h->ev.push_back(e);
  }
  *pair = &h->ev[h->ev_used];
  h->ev_used += 2;
  h->em_timed += 1;
  TSEM_HIP(hipEventRecord((*pair)[0], h->stream));
  return TSEM_OK;
}

// One launch of the persistent fused kernel.  mode 0: EM pass (column sums of w*z into d_fpartial);
// mode 1: log-likelihood of the ambiguous rows (one partial per workgroup into d_lnl_part).
static int launch_fused(tsem_ctx* h, int mode, hipEvent_t* pair, int bin = 0) {
  if (!h->fz_clean) {                                      // (k_update leaves them zero after every EM pass)
    if (h->fused_launched && h->d_fz_aux)                  // keep the error word of a launch nobody cleaned up after
      k_keep_err<<<1, 1, 0, h->stream>>>(h->d_xflags, h->d_fz_aux + 2);
    TSEM_HIP(hipMemsetAsync(h->d_xflags, 0, sizeof(uint32_t) * FZ_SYNC_WORDS, h->stream));
    if (h->P > 1) TSEM_HIP(hipMemsetAsync(h->d_xchg, 0, sizeof(double) * (size_t)h->fz_teams * FZ_XS * h->P * h->R, h->stream));
  }
  h->fz_clean = false;
  if (mode == 1) TSEM_HIP(hipMemsetAsync(h->d_lnl_part, 0, sizeof(double) * (size_t)h->fz_grid, h->stream));   // teams that do not form write nothing
  FusedArgs A;
  A.P = h->P; A.Kp = h->Kp; A.R = h->R; A.nb = h->nb; A.N_amb_pad = h->N_amb_pad;
  A.sb_off = h->d_sb_off; A.sb_q32 = h->d_sb_q32; A.pval = h->d_pval; A.prc = h->d_prc;
  const bool lnl = mode == 1;                               // mode 2 is an EM pass (exact column sums), not the lnl pass
  A.ctab = lnl ? h->d_ctab_prev : h->d_ctab; A.ctab2 = h->d_ctab; A.lnl_out = h->d_lnl_part; A.lnl_mode = lnl ? 1 : 0;
  A.wrow = h->d_amb_w; A.partial = h->d_fpartial; A.xchg = h->d_xchg; A.sorted = h->sorted_layout ? 1 : 0;
  A.sync = h->d_xflags;
  A.prof = mode ? nullptr : h->d_prof; A.prof_blocks = A.prof ? 64 : 0; A.dbg = (int)h->opt_dbg;
  A.ctl = h->d_ctl;
  A.ebias = h->d_ebias; A.bin = bin; A.ovf = h->d_ovf; A.partial2 = h->d_fpartial2;

  A.pcode = h->d_pcode; A.lut = h->d_lut; A.lut_len = fz_fmt(h) ? h->lut_len : 0; A.wcode = h->d_amb_wcode;
  const size_t ldsf = fz_lds_bytes(h, fz_fmt(h) != 0);
  if (lnl && h->fz_grid > 4096) TSEM_FAIL(TSEM_ERR_ARG, "fused lnl: more workgroups than partial slots");
  fz_fn fn = fz_kernel(h->P, mode, fz_fmt(h), h->geo);
  if (!fn) TSEM_FAIL(TSEM_ERR_ARG, mode >= 2 ? "reproducible mode needs the fused kernel with a score table of at most 2048 entries"
                                               : "fused kernel supports at most 8 column parts");
  if (pair) TSEM_HIP(hipEventRecord(pair[0], h->stream));   // time the kernel, not the memsets
  fn<<<h->fz_grid, FZ_NT, ldsf, h->stream>>>(A);
  TSEM_HIP(hipGetLastError());
  h->fused_launched = true;
  return TSEM_OK;
}

static int rowpass_grid(tsem_ctx* h);
// the fp32 diagnostic pass (option "em_precision" = 1): same outputs as tsem_em_pass, fp32 arithmetic
static int em_pass_f32(tsem_ctx* h) {
  const int K = h->K;
  if (!h->d_c32) {
    TSEM_ALLOC(h->d_c32, K); TSEM_ALLOC(h->d_cs32, K); TSEM_ALLOC(h->d_lut32, h->lut_len);
    k_lut32<<<cdiv64(h->lut_len, 256), 256, 0, h->stream>>>(h->lut_len, h->d_lut, h->d_lut32);
  }
  unsigned long long* cmax = reinterpret_cast<unsigned long long*>(h->d_lnl_part + 12000);
  TSEM_HIP(hipMemsetAsync(cmax, 0, 8, h->stream));
  TSEM_HIP(hipMemsetAsync(h->d_cs32, 0, sizeof(float) * K, h->stream));
  k_cmax<<<cdiv64(K, 256), 256, 0, h->stream>>>(K, h->d_pi, h->d_theta, cmax);
  k_make_c32<<<cdiv64(K, 256), 256, 0, h->stream>>>(K, h->d_pi, h->d_theta, cmax, h->d_c32);
  if (h->N) k_em_rows_f32<<<rowpass_grid(h), 256, 0, h->stream>>>(h->N, h->d_indptr, h->d_indices, h->d_raw, h->d_lut32, h->d_c32, h->d_cs32);
  k_red_from_f32<<<cdiv64(K, 256), 256, 0, h->stream>>>(K, h->d_cs32, h->d_red);
  TSEM_HIP(hipGetLastError());
  return TSEM_OK;
}

// option "reproducible".  Every contribution to a column sum is cut into a high and a low piece on a per-slot grid (tsem_fused.h,
// phase 2), one pass each; sums of such pieces are exact in fp64, so they do not depend on the order of the LDS atomics.  The grid
// hangs on the slot's bound 2^E (ebias = E + 1023): pieces are multiples of 2^(E-30) and 2^(E-60).  All decisions below are functions
// of exact sums, hence identical in every run.
//   k_bin_check, after the high pass: a contribution reached the bound -> raise it;  the high sum lies more than BIN_SLACK bits
//     under what the bound was chosen for (or is zero although the column has entries and a non-zero pi * theta) -> lower it.  Either
//     way the high pass is repeated (one pass lost, the low pass has not run yet).
//   k_bin_finish, after the low pass: S = high + low, and the bound of the NEXT iteration = 16 x the sum this column is expected to
//     have then: columns that fall by d bits per iteration (linear EM convergence) or by d, 2d, 4d, ... bits (a column dying under a
//     zero prior) are followed, so that later iterations rarely repeat a pass.
// Worst-case relative error of a column sum: (entries of the column) x 2^-(57 - BIN_SLACK); typically the fp64 rounding of S.
constexpr int BIN_SLACK = 16;
__device__ __forceinline__ int bin_slot(uint32_t cm, int Kp, int* copies) {
  *copies = 1 << ((cm >> 13) & 7u);
  return (int)(cm >> 16) * Kp + (int)(cm & 0x1FFFu);
}
__global__ void k_bin_check(int K, const double* __restrict__ red, const uint32_t* __restrict__ colmap, int Kp,
                            const unsigned long long* __restrict__ colcount, const uint32_t* __restrict__ ucount,
                            const double* __restrict__ pi, const double* __restrict__ theta,
                            uint16_t* __restrict__ ebias, uint8_t* __restrict__ ovf, uint32_t* __restrict__ flag) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= K || red[K] > 0.0) return;                       // (a pass that timed out: the sums mean nothing, the update kernel refuses them)
  int copies;
  const int pc = bin_slot(colmap[j], Kp, &copies);
  const int eb = ebias[pc];
  bool over = false;
  for (int c = 0; c < copies; ++c) { over |= ovf[pc + c] != 0; ovf[pc + c] = 0; }
  const double S = red[j];
  int eb_new = eb;
  if (over) eb_new = eb + 12;
  else if (S == 0.0) {
    // nothing arrived: too coarse a grid, unless the column has no entry in a row of several (those of single-entry rows feed pi
    // through pisum0, not through this pass) or pi * theta is zero
    if (colcount && colcount[j] > (ucount ? ucount[j] : 0u) && pi[j] * theta[j] != 0.0 && eb > 120) eb_new = eb - 24;
  }
  else {
    // (a high sum a few grid steps large says little about S: move by at most 24 bits and keep 6 bits in hand)
    const int ex4 = (int)((__double2hiint(S) >> 20) & 0x7FF) + 4;
    if (eb - ex4 > BIN_SLACK) eb_new = max(ex4 + 6, eb - 24);
    else if (ex4 - eb > 20) eb_new = ex4;                     // a sum 2^16 bounds large: more would not be exact (53 - 30 bits of room)
  }
  eb_new = min(2000, max(64, eb_new));
  if (eb_new != eb) {
    for (int c = 0; c < copies; ++c) ebias[pc + c] = (uint16_t)eb_new;
    atomicOr(flag, 1u);
  }
}
__global__ void k_bin_finish(int K, const double* __restrict__ red_hi, double* __restrict__ red, const uint32_t* __restrict__ colmap, int Kp,
                             uint16_t* __restrict__ ebias, uint8_t* __restrict__ ovf, int16_t* __restrict__ hist) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j == 0) red[K] = fmax(red[K], red_hi[K]);             // a time-out of either pass
  if (j >= K) return;
  const double S = red_hi[j] + red[j];
  red[j] = S;
  int copies;
  const int pc = bin_slot(colmap[j], Kp, &copies);
  for (int c = 0; c < copies; ++c) ovf[pc + c] = 0;          // (the low pass sees the same contributions as the high pass: nothing new)
  if (S == 0.0) return;                                       // no information: keep the bound
  const int ex4 = (int)((__double2hiint(S) >> 20) & 0x7FF) + 4;
  const int eprev = hist[j], dprev = hist[K + j];
  const int drop = eprev ? eprev - ex4 : 0;
  int pred = drop;
  if (drop >= 4 && dprev >= 2) pred = min(drop * drop / dprev, 2 * drop + 2);
  pred = max(-10, min(40, pred));
  const int shift = pred > 3 ? pred - 3 : (pred < 0 ? pred : 0);
  const int eb_new = min(2000, max(64, ex4 - shift));
  for (int c = 0; c < copies; ++c) ebias[pc + c] = (uint16_t)eb_new;
  hist[j] = (int16_t)ex4; hist[K + j] = (int16_t)max(-1000, min(1000, drop));
}

int tsem_em_pass(tsem_ctx* h) {
  if (!h || !h->have_model) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  if (h->opt_precision == 1) return em_pass_f32(h);
  hipEvent_t* pair = nullptr;
  if (int rc = begin_timing(h, &pair)) return rc;
  bool fused_done = false;
  if (h->nb > 0 && h->use_fused && h->opt_reproducible) {
    // two exact passes (high and low pieces of every contribution); the high pass is repeated while a column's bound has to move:
    // the first iteration of a run takes a few repeats (the bounds start at the largest fragment weight), later ones rarely any
    if (h->d_ctl) {                                          // a chunk that has stopped: nothing to compute (the passes would return at once)
      uint32_t st = 0;
      TSEM_HIP(hipMemcpyAsync(&st, h->d_ctl, 4, hipMemcpyDeviceToHost, h->stream));
      TSEM_HIP(hipStreamSynchronize(h->stream));
      if (st) { h->em_launches += 1; return TSEM_OK; }
    }
    auto reduce = [&](const double* partial) -> int {
      k_colreduce<<<cdiv64(h->Kpad, 32), 256, 0, h->stream>>>(h->Kpad, h->fz_teams, partial, h->d_col_of_pc, h->d_colmap, h->d_red, h->K,
                                                              h->d_xflags, h->P, h->d_ctl,
                                                              h->P > 1 ? reinterpret_cast<unsigned long long*>(h->d_xchg) : nullptr,
                                                              h->P > 1 ? (int64_t)h->fz_teams * FZ_XS * h->P * h->R : 0);
      // (k_colreduce cleared the exchange ring; the sync words are cleared by the memset of the next launch)
      TSEM_HIP(hipGetLastError());
      return TSEM_OK;
    };
    // exact_single: ONE launch leaves the team partials of both pieces (d_fpartial: high, d_fpartial2: low)
    auto pass = [&](int bin, hipEvent_t* ev) -> int {
      if (h->exact_single) {
        if (bin == 1) { if (int rc = launch_fused(h, 3, ev, 0)) return rc; }
        return reduce(bin == 1 ? h->d_fpartial : h->d_fpartial2);
      }
      if (int rc = launch_fused(h, 2, ev, bin)) return rc;
      return reduce(h->d_fpartial);
    };
    for (int attempt = 0;; ++attempt) {
      if (int rc = pass(1, attempt == 0 ? pair : nullptr)) return rc;
      TSEM_HIP(hipMemsetAsync(h->d_binflag, 0, 4, h->stream));
      k_bin_check<<<cdiv64(h->K, 256), 256, 0, h->stream>>>(h->K, h->d_red, h->d_colmap, h->Kp, h->d_colcount, h->d_ucount, h->d_pi, h->d_theta,
                                                            h->d_ebias, h->d_ovf, h->d_binflag);
      TSEM_HIP(hipGetLastError());
      uint32_t redo = 0;
      TSEM_HIP(hipMemcpyAsync(&redo, h->d_binflag, 4, hipMemcpyDeviceToHost, h->stream));
      TSEM_HIP(hipStreamSynchronize(h->stream));
      if (!redo || attempt >= 40) break;                     // (40 x 24 bits: from the largest weight down to the smallest normal number)
      h->n_bin_repeats += 1;
    }
    TSEM_HIP(hipMemcpyAsync(h->d_red_hi, h->d_red, sizeof(double) * (h->K + 2), hipMemcpyDeviceToDevice, h->stream));
    if (int rc = pass(2, nullptr)) return rc;
    k_bin_finish<<<cdiv64(h->K, 256), 256, 0, h->stream>>>(h->K, h->d_red_hi, h->d_red, h->d_colmap, h->Kp, h->d_ebias, h->d_ovf, h->d_ehist);
    TSEM_HIP(hipGetLastError());
    if (pair) TSEM_HIP(hipEventRecord(pair[1], h->stream));
    h->em_launches += 1;
    return TSEM_OK;
  } else if (h->nb > 0 && h->use_fused) {
    if (int rc = launch_fused(h, 0, pair)) return rc;
    fused_done = true;
  } else if (h->nb > 0) {
    const size_t lds2 = (size_t)(2 * h->Kp + h->R) * 8;
    const int64_t chunk = h->opt_chunk > 0 ? h->opt_chunk : h->nb;
    for (int64_t b0 = 0; b0 < h->nb; b0 += chunk) {
      const int64_t b1 = std::min(h->nb, b0 + chunk);
      if (int rc = launch_phase1(h, h->d_ctab, b0, b1, true)) return rc;
      k_phase2_em<1024><<<h->G2 * h->P, 1024, lds2, h->stream>>>(h->P, h->Kp, h->R, b0, b1, h->G2, b0 > 0 ? 1 : 0,
          h->N_amb_pad, h->d_sb_off, h->d_pval, h->d_prc, h->d_ctab, h->d_ypart, h->d_amb_wcode, h->d_lut, h->d_partial, h->d_ctl);
    }
    TSEM_HIP(hipGetLastError());
  }
  if (pair) TSEM_HIP(hipEventRecord(pair[1], h->stream));
  h->em_launches += 1;
  if (fused_done) {
    k_colreduce<<<cdiv64(h->Kpad, 32), 256, 0, h->stream>>>(h->Kpad, h->fz_teams, h->d_fpartial, h->d_col_of_pc, h->d_colmap, h->d_red, h->K,
                                                            h->d_xflags, h->P, h->d_ctl,
                                                            h->P > 1 ? reinterpret_cast<unsigned long long*>(h->d_xchg) : nullptr,
                                                            h->P > 1 ? (int64_t)h->fz_teams * FZ_XS * h->P * h->R : 0);
  } else if (h->nb > 0) {
    k_colreduce<<<cdiv64(h->Kpad, 32), 256, 0, h->stream>>>(h->Kpad, h->G2, h->d_partial, h->d_col_of_pc, h->d_colmap, h->d_red, h->K,
                                                            nullptr, h->P, h->d_ctl, nullptr, 0);
  } else {
    TSEM_HIP(hipMemsetAsync(h->d_red, 0, sizeof(double) * (h->K + 2), h->stream));
  }
  TSEM_HIP(hipGetLastError());
  return TSEM_OK;
}

static int launch_update(tsem_ctx* h, double* d_diff_slot, bool chunked = false, double eps = 0.0, int use_lnl = 0) {
  const double tpw = h->theta_prior * h->w_max, ppw = h->pi_prior * h->w_max;   // model.py:696-697
  const double tden = h->W_amb + tpw * h->K, pden = h->W_tot + ppw * h->K;      // model.py:732,738
  const int nblk = std::min(1024, cdiv64(h->K, 256));
  double* part = h->d_lnl_part + 10000;   // scratch for the per-block |pi_hat - pi| partials
  if (!h->d_fz_aux) {
    TSEM_ALLOC(h->d_fz_aux, 4);
    TSEM_HIP(hipMemsetAsync(h->d_fz_aux, 0, sizeof(uint32_t) * 4, h->stream));
  }
  const bool clean = h->use_fused && h->d_xflags && h->fused_launched;
  UpdCtl C;
  C.ctl = chunked ? h->d_ctl : nullptr; C.eps = eps; C.use_lnl = use_lnl;
  C.pi_first = h->first_pending ? h->d_pi_first : nullptr; C.theta_first = h->first_pending ? h->d_theta_first : nullptr;
  k_update<<<nblk, 256, 0, h->stream>>>(C, h->K, h->d_red, h->d_pisum0, tpw, tden, ppw, pden, h->d_pi, h->d_theta,
                                        h->d_pi_prev, h->d_theta_prev, h->d_colmap, h->Kp, h->d_ctab, h->d_ctab_prev,
                                        h->d_twin_rep, part, d_diff_slot, h->d_fz_aux,
                                        clean ? h->d_xflags : nullptr, h->d_fz_aux + 2,
                                        reinterpret_cast<unsigned long long*>(h->d_xchg),
                                        h->P > 1 ? (int64_t)h->fz_teams * FZ_XS * h->P * h->R : 0);
  TSEM_HIP(hipGetLastError());
  if (clean) h->fz_clean = true;
  h->em_prev = h->em_cur; h->em_cur = true;                // (a skipped update — stop flag, time-out — leaves older, equally valid M-step values)
  return TSEM_OK;
}

// the error word of the fused kernel: live, or what k_update / k_keep_err saved before zeroing it.  Clears it.
static int take_fused_error(tsem_ctx* h, uint32_t* word) {
  *word = 0;
  if (!h->d_xflags || !h->fused_launched) return TSEM_OK;
  uint32_t ee[2] = {0, 0}, kept[2] = {0, 0};
  TSEM_HIP(hipMemcpyAsync(ee, h->d_xflags + 9, 8, hipMemcpyDeviceToHost, h->stream));
  if (h->d_fz_aux) TSEM_HIP(hipMemcpyAsync(kept, h->d_fz_aux + 2, 8, hipMemcpyDeviceToHost, h->stream));
  TSEM_HIP(hipStreamSynchronize(h->stream));
  *word = ee[0] | kept[0];
  h->last_slow_path = h->fz_clean ? kept[1] : ee[1];
  if (*word) {
    TSEM_HIP(hipMemsetAsync(h->d_xflags + 9, 0, 4, h->stream));
    if (h->d_fz_aux) TSEM_HIP(hipMemsetAsync(h->d_fz_aux + 2, 0, 4, h->stream));
  }
  return TSEM_OK;
}

int tsem_fallback_twopass(tsem_ctx* h) {
  if (!h || !h->have_model) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  if (!h->use_fused) return TSEM_OK;
  TSEM_HIP(hipStreamSynchronize(h->stream));
  uint32_t e = 0;
  (void)take_fused_error(h, &e);
  h->em_kernel = TSEM_EMK_TWOPASS;
  h->opt_format = 1;                                       // the two-pass kernels read fp64 entries
  h->opt_dbg &= ~(int64_t)(32 | 64);
  if (int rc = choose_geometry(h)) return rc;
  if (int rc = build_layout(h)) return rc;
  // the permuted pi*theta tables follow the new column map
  TSEM_ALLOC(h->d_ctab, h->Kpad); TSEM_ALLOC(h->d_ctab_prev, h->Kpad);
  TSEM_HIP(hipMemsetAsync(h->d_ctab, 0, sizeof(double) * h->Kpad, h->stream));
  TSEM_HIP(hipMemsetAsync(h->d_ctab_prev, 0, sizeof(double) * h->Kpad, h->stream));
  k_make_ctab<<<cdiv64(h->K, 256), 256, 0, h->stream>>>(h->K, h->d_pi, h->d_theta, h->d_colmap, h->Kp, h->d_ctab);
  k_make_ctab<<<cdiv64(h->K, 256), 256, 0, h->stream>>>(h->K, h->d_pi_prev, h->d_theta_prev, h->d_colmap, h->Kp, h->d_ctab_prev);
  TSEM_HIP(hipGetLastError());
  TSEM_HIP(hipStreamSynchronize(h->stream));
  h->n_fallbacks += 1;
  fprintf(stderr, "libtelescope_em: the persistent EM kernel could not keep its workgroups co-resident (watchdog code %u); "
                  "continuing with the two-pass kernels\n", e);
  return TSEM_OK;
}

int tsem_recover_timeout(tsem_ctx* h, int32_t* switched) {
  if (!h || !h->have_model) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  if (switched) *switched = 0;
  uint32_t mine = 0;
  if (int rc = take_fused_error(h, &mine)) return rc;
  if (mine && h->use_fused) {
    if (int rc = tsem_fallback_twopass(h)) return rc;
    if (switched) *switched = 1;
  }
  return TSEM_OK;
}

int tsem_em_update(tsem_ctx* h, double* diff_est) {
  if (!h || !h->have_model) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  if (int rc = launch_update(h, h->d_diffs)) return rc;
  h->first_pending = false;
  if (diff_est) {
    TSEM_HIP(hipMemcpyAsync(diff_est, h->d_diffs, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    TSEM_HIP(hipStreamSynchronize(h->stream));
    // negative: the (all-reduced) error flag was up, so no rank committed this iteration
    if (*diff_est < 0.0)
      TSEM_FAIL(TSEM_ERR_TIMEOUT, "EM pass: the hand-off watchdog of the fused kernel fired on some rank; parameters "
                                  "left untouched (tsem_fallback_twopass, then redo the pass)");
  }
  return TSEM_OK;
}

static int launch_lnl(tsem_ctx* h) {
  int na = 0, nu = 0;
  if (h->nb > 0 && h->use_fused) {
    if (int rc = launch_fused(h, 1, nullptr)) return rc;
    na = h->fz_grid;
  } else if (h->nb > 0) {
    if (int rc = launch_phase1(h, h->d_ctab_prev)) return rc;
    const size_t lds2 = (size_t)(2 * h->Kp + h->R) * 8;
    na = h->G2 * h->P;
    k_phase2_lnl<1024><<<na, 1024, lds2, h->stream>>>(h->P, h->Kp, h->R, h->nb, h->G2, h->N_amb_pad, h->d_sb_off,
        h->d_pval, h->d_prc, h->d_ctab_prev, h->d_ctab, h->d_ypart, h->d_lnl_part);
    TSEM_HIP(hipGetLastError());
  }
  if (h->N_uni > 0) {
    nu = (int)std::min<int64_t>(2048, (h->N_uni + 255) / 256);
    k_lnl_unique<<<nu, 256, 0, h->stream>>>(h->N_uni, h->d_uni_col, h->d_uni_code, h->d_lut, h->d_pi_prev, h->d_pi,
                                           h->d_lnl_part + 4096);
    TSEM_HIP(hipGetLastError());
  }
  k_sum_parts<<<1, 256, 0, h->stream>>>(h->d_lnl_part, na, h->d_lnl_part + 4096, nu, h->d_red + h->K);
  TSEM_HIP(hipGetLastError());
  return TSEM_OK;
}

int tsem_lnl_pass(tsem_ctx* h) {
  if (!h || !h->have_model) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  return launch_lnl(h);
}

int tsem_read_reduce(tsem_ctx* h, double* out, int64_t offset, int64_t count) {
  if (!h || !h->have_model || !out || offset < 0 || offset + count > h->K + 2) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  TSEM_HIP(hipMemcpyAsync(out, h->d_red + offset, sizeof(double) * count, hipMemcpyDeviceToHost, h->stream));
  TSEM_HIP(hipStreamSynchronize(h->stream));
  return TSEM_OK;
}

// ---------------------------------------------------------------------------
// collectives: RCCL resolved at RUN time, or the in-process transport
// ---------------------------------------------------------------------------
// RCCL is not linked.  A torch process already carries a librccl (torch/lib/librccl.so, loaded with torch); linking a
// second one by DT_NEEDED made the copy that serves this library's calls depend on load order.
// Now ONE copy is chosen deliberately: the librccl that is already mapped into the process if there is one (so the
// library and torch.distributed share a single RCCL — one set of IPC handles, one topology detection), otherwise
// librccl.so.1 from the loader's search path / /opt/rocm/lib.  A box without RCCL can still load and run the library
// on one GPU; tsem_comm_library_info reports which copy and version is in use (it goes into the bench line).
struct NcclApi {
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
  bool ok = false;
  int version = 0;
  std::string path, others, err;
};
static int nccl_phdr_cb(struct dl_phdr_info* info, size_t, void* data) {
  auto* v = static_cast<std::vector<std::string>*>(data);
  if (info->dlpi_name && strstr(info->dlpi_name, "librccl")) v->push_back(info->dlpi_name);
  return 0;
}
static NcclApi* nccl_api() {
  static NcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    std::vector<std::string> loaded;
    dl_iterate_phdr(nccl_phdr_cb, &loaded);
    void* hnd = nullptr;
    if (!loaded.empty()) {
      hnd = dlopen(loaded[0].c_str(), RTLD_NOW | RTLD_NOLOAD);
      if (hnd) api.path = loaded[0];
      for (size_t i = 1; i < loaded.size(); ++i) api.others += (api.others.empty() ? "" : ", ") + loaded[i];
    }
    const char* cands[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (int i = 0; i < 3 && !hnd; ++i) {
      hnd = dlopen(cands[i], RTLD_NOW | RTLD_LOCAL);
      if (hnd) {
        api.path = cands[i];
        Dl_info di;
        void* sym = dlsym(hnd, "ncclAllReduce");
        if (sym && dladdr(sym, &di) && di.dli_fname) api.path = di.dli_fname;
      }
    }
    if (!hnd) { api.err = std::string("librccl is not available: ") + (dlerror() ? dlerror() : "dlopen failed"); return; }
    auto need = [&](const char* name) -> void* {
      void* p = dlsym(hnd, name);
      if (!p && api.err.empty()) api.err = std::string("librccl (") + api.path + ") does not export " + name;
      return p;
    };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(need("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(need("ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(need("ncclCommDestroy"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(need("ncclAllReduce"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(need("ncclGetErrorString"));
    api.GetVersion = reinterpret_cast<decltype(api.GetVersion)>(need("ncclGetVersion"));
    if (!api.err.empty()) return;
    if (api.GetVersion(&api.version) != ncclSuccess) api.version = 0;
    // the id / enum layout this file was compiled against (rccl.h of the ROCm image) is the 2.x ABI
    if (api.version && api.version / 10000 != NCCL_MAJOR) {
      api.err = "librccl (" + api.path + ") has major version " + std::to_string(api.version / 10000) + ", this library was built for " +
                std::to_string(NCCL_MAJOR);
      return;
    }
    api.ok = true;
  });
  return &api;
}

// ---- in-process transport -------------------------------------------------------------------------------------
// Several handles on ONE device, each driven by its own host thread of one process, run the protocol of a row-sharded
// job: same tsem_em_chunk, same reduce-buffer layout, same device-side stop flag and error slot — only the all-reduce
// itself is different.  Rank r copies its vector into its slot, records an event; after a HOST rendezvous (every rank has
// recorded) each rank makes its stream wait for the peers' events and sums the slots in rank order, so all ranks get the
// same bits.  Two slot generations alternate; a slot is rewritten only after the peers' sums of two collectives ago
// have completed (their `done` events).  For tests of the N > 1 path on a one-GPU box, and for hosts that time-slice
// one GPU between several engines; a multi-GPU job uses RCCL.
constexpr int TS_LOCAL_MAXW = 8;
struct tsem_local_group {
  int world = 1, device = 0, refs = 0;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  uint64_t gen = 0;
  bool broken = false;
  double timeout_s = 120.0;
  void* slot[TS_LOCAL_MAXW][2] = {};
  size_t slot_bytes[TS_LOCAL_MAXW][2] = {};
  hipEvent_t ready[TS_LOCAL_MAXW][2] = {}, done[TS_LOCAL_MAXW][2] = {};
  bool done_rec[TS_LOCAL_MAXW][2] = {};
  bool taken[TS_LOCAL_MAXW] = {};
};
// host rendezvous of the group's ranks; false: a peer did not arrive in time (or the group broke earlier)
static bool local_rendezvous(tsem_local_group* g) {
  std::unique_lock<std::mutex> lk(g->mu);
  if (g->broken) return false;
  const uint64_t my = g->gen;
  if (++g->arrived == g->world) { g->arrived = 0; ++g->gen; g->cv.notify_all(); return true; }
  const bool ok = g->cv.wait_for(lk, std::chrono::duration<double>(g->timeout_s), [&] { return g->gen != my || g->broken; });
  if (!ok || g->broken) { g->broken = true; g->cv.notify_all(); return false; }
  return true;
}
// dtype / op as in tsem_comm_allreduce_host: 0 f64 sum, 1 u64 sum, 2 f64 max, 3 i64 max
struct LocalSlots { const void* p[TS_LOCAL_MAXW]; };
__global__ __launch_bounds__(256) void k_local_reduce(void* __restrict__ out, LocalSlots S, int world, int64_t n, int dtype) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (dtype == 0 || dtype == 2) {
    double v = static_cast<const double*>(S.p[0])[i];
    for (int q = 1; q < world; ++q) { const double t = static_cast<const double*>(S.p[q])[i]; v = dtype == 0 ? v + t : fmax(v, t); }
    static_cast<double*>(out)[i] = v;
  } else if (dtype == 1) {
    unsigned long long v = static_cast<const unsigned long long*>(S.p[0])[i];
    for (int q = 1; q < world; ++q) v += static_cast<const unsigned long long*>(S.p[q])[i];
    static_cast<unsigned long long*>(out)[i] = v;
  } else {
    long long v = static_cast<const long long*>(S.p[0])[i];
    for (int q = 1; q < world; ++q) v = max(v, static_cast<const long long*>(S.p[q])[i]);
    static_cast<long long*>(out)[i] = v;
  }
}
static int local_allreduce(tsem_comm* c, void* buf, size_t count, int dtype, hipStream_t s, std::string& err) {
  tsem_local_group* g = c->local;
  const int r = c->rank, par = (int)(c->epoch & 1);
  c->epoch += 1;
  const size_t bytes = count * 8;
  auto fail = [&](const std::string& m) { err = m; std::lock_guard<std::mutex> lk(g->mu); g->broken = true; g->cv.notify_all(); return TSEM_ERR_HIP; };
#define LOC_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)
  // my slot of this generation is free once the peers' sums of two collectives ago are done
  for (int q = 0; q < g->world; ++q)
    if (q != r && g->done_rec[q][par]) LOC_HIP(hipStreamWaitEvent(s, g->done[q][par], 0));
  if (g->slot_bytes[r][par] < bytes) {
    for (int q = 0; q < g->world; ++q)
      if (q != r && g->done_rec[q][par]) LOC_HIP(hipEventSynchronize(g->done[q][par]));
    if (g->slot[r][par]) LOC_HIP(hipFree(g->slot[r][par]));
    g->slot[r][par] = nullptr; g->slot_bytes[r][par] = 0;
    LOC_HIP(hipMalloc(&g->slot[r][par], std::max<size_t>(bytes, 4096)));
    g->slot_bytes[r][par] = std::max<size_t>(bytes, 4096);
  }
  if (bytes) LOC_HIP(hipMemcpyAsync(g->slot[r][par], buf, bytes, hipMemcpyDeviceToDevice, s));
  LOC_HIP(hipEventRecord(g->ready[r][par], s));
  if (!local_rendezvous(g)) { err = "in-process communicator: a peer rank did not reach the collective (time-out or an earlier failure)"; return TSEM_ERR_TIMEOUT; }
  LocalSlots S;
  for (int q = 0; q < g->world; ++q) {
    if (q != r) LOC_HIP(hipStreamWaitEvent(s, g->ready[q][par], 0));
    S.p[q] = g->slot[q][par];
  }
  if (count) k_local_reduce<<<(unsigned)((count + 255) / 256), 256, 0, s>>>(buf, S, g->world, (int64_t)count, dtype);
  LOC_HIP(hipGetLastError());
  LOC_HIP(hipEventRecord(g->done[r][par], s));
  g->done_rec[r][par] = true;
#undef LOC_HIP
  return TSEM_OK;
}

static bool comm_on(const tsem_ctx* h) { return h->comm && h->comm->active(); }
// in-place all-reduce of `count` 8-byte words on device memory, on stream s, over whichever transport the communicator has
static int comm_allreduce_dev(tsem_comm* c, void* buf, size_t count, int dtype, hipStream_t s, std::string& err) {
  if (!c || !c->active()) return TSEM_OK;
  if (c->local) return local_allreduce(c, buf, count, dtype, s, err);
  NcclApi* N = nccl_api();
  const ncclDataType_t dt = dtype == 1 ? ncclUint64 : (dtype == 3 ? ncclInt64 : ncclDouble);
  const ncclRedOp_t op = dtype >= 2 ? ncclMax : ncclSum;
  const ncclResult_t r = N->AllReduce(buf, buf, count, dt, op, c->nccl, s);
  if (r != ncclSuccess) { err = std::string("ncclAllReduce: ") + N->GetErrorString(r); return TSEM_ERR_HIP; }
  return TSEM_OK;
}

// the per-iteration exchange: ONE sum all-reduce of the per-locus column sums + the error flag
static int comm_allreduce_red(tsem_ctx* h, int64_t offset, int64_t count) {
  if (!comm_on(h)) return TSEM_OK;
  return comm_allreduce_dev(h->comm, h->d_red + offset, (size_t)count, 0, h->stream, h->err);
}

static int ensure_ctl(tsem_ctx* h) {
  if (h->d_ctl) return TSEM_OK;
  TSEM_ALLOC(h->d_ctl, 8); TSEM_ALLOC(h->d_ctld, 8); TSEM_ALLOC(h->d_lnls, TS_DIFF_RING);
  TSEM_HIP(hipMemsetAsync(h->d_ctl, 0, 32, h->stream));
  TSEM_HIP(hipMemsetAsync(h->d_ctld, 0, 64, h->stream));
  return TSEM_OK;
}

// the lnl of the iteration just committed, all-reduced, into d_ctld[1] (value) / d_ctld[2] (error flag)
static int enqueue_lnl_reduce(tsem_ctx* h) {
  if (int rc = launch_lnl(h)) return rc;
  k_lnl_slots<<<1, 1, 0, h->stream>>>(h->d_red + h->K, (h->use_fused && h->nb > 0) ? h->d_xflags : nullptr, h->d_ctld + 1);
  TSEM_HIP(hipGetLastError());
  if (comm_on(h)) { if (int rc = comm_allreduce_dev(h->comm, h->d_ctld + 1, 2, 0, h->stream, h->err)) return rc; }
  return TSEM_OK;
}

int tsem_em_chunk(tsem_ctx* h, int32_t n_max, double epsilon, int32_t use_likelihood, int32_t first,
                  int32_t* n_done, int32_t* stopped, double* diffs_out, double* lnls_out) {
  if (!h || !h->have_model || n_max < 0 || n_max > TS_DIFF_RING) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  if (int rc = ensure_ctl(h)) return rc;
  if (first) {
    // model.py:786 compares the first iteration's lnl with self.lnl as the previous run left it (inf on a fresh model,
    // model.py:683): tsem_set_prev_lnl / the end of tsem_em_run keep that value for the next run
    const double seed = h->lnl_prev_seed;
    TSEM_HIP(hipMemcpy(h->d_ctld, &seed, sizeof(double), hipMemcpyHostToDevice));
    if (!h->d_pi_first) { TSEM_ALLOC(h->d_pi_first, h->K); TSEM_ALLOC(h->d_theta_first, h->K); }
    h->first_pending = true;
  }
  int done = 0, retries = 0;
  bool stop = false, lnl_pending = false;
  // (an lnl pass that timed out in the LAST iteration of the chunk is redone before returning: its value is the
  // iteration's log-likelihood and may end the run — ADVICE r2)
  while ((done < n_max || lnl_pending) && !stop) {
    // enqueue everything that is left; the device stops itself
    TSEM_HIP(hipMemsetAsync(h->d_ctl, 0, 8, h->stream));
    const int base = done, want = n_max - done;
    if (lnl_pending) {                                     // the lnl pass of the last committed iteration timed out
      if (int rc = enqueue_lnl_reduce(h)) return rc;
      k_lnl_check<<<1, 1, 0, h->stream>>>(h->d_ctl, h->d_ctld, h->d_ctld + 1, epsilon, h->d_lnls + base - 1);
      TSEM_HIP(hipGetLastError());
    }
    for (int i = 0; i < want; ++i) {
      if (int rc = tsem_em_pass(h)) return rc;
      if (int rc = comm_allreduce_red(h, 0, h->K + 2)) return rc;
      if (int rc = launch_update(h, h->d_diffs + base + i, true, epsilon, use_likelihood)) return rc;
      h->first_pending = false;                            // (a failed first update is redone below with the flag restored)
      if (use_likelihood) {
        if (int rc = enqueue_lnl_reduce(h)) return rc;
        k_lnl_check<<<1, 1, 0, h->stream>>>(h->d_ctl, h->d_ctld, h->d_ctld + 1, epsilon, h->d_lnls + base + i);
        TSEM_HIP(hipGetLastError());
      }
    }
    uint32_t ctl[2] = {0, 0};
    TSEM_HIP(hipMemcpyAsync(ctl, h->d_ctl, 8, hipMemcpyDeviceToHost, h->stream));
    TSEM_HIP(hipStreamSynchronize(h->stream));
    done = base + (int)ctl[1];
    lnl_pending = false;
    if (ctl[0] == 1u) { stop = true; break; }
    if (ctl[0] == 2u || ctl[0] == 3u) {                    // some rank's fused pass timed out: nobody committed that step
      if (++retries > 3) TSEM_FAIL(TSEM_ERR_TIMEOUT, "EM pass: repeated hand-off time-outs");
      uint32_t mine = 0;
      if (int rc = take_fused_error(h, &mine)) return rc;
      if (mine) { if (int rc = tsem_fallback_twopass(h)) return rc; }
      if (first && done == 0 && ctl[0] == 2u) h->first_pending = true;
      lnl_pending = ctl[0] == 3u;
      continue;
    }
    if (h->use_fused && !comm_on(h)) {                     // belt and braces: an error word the flags did not carry.  (Row-sharded
      uint32_t mine = 0;                                   //  runs rely on slot K alone: failing on ONE rank would leave the others in the next collective.)
      if (int rc = take_fused_error(h, &mine)) return rc;
      if (mine) TSEM_FAIL(TSEM_ERR_TIMEOUT, "fused EM kernel: hand-off watchdog fired (code " + std::to_string(mine) + ")");
    }
  }
  if (n_done) *n_done = done;
  if (stopped) *stopped = stop ? 1 : 0;
  if (done && diffs_out) TSEM_HIP(hipMemcpy(diffs_out, h->d_diffs, sizeof(double) * done, hipMemcpyDeviceToHost));
  if (done && lnls_out && use_likelihood) TSEM_HIP(hipMemcpy(lnls_out, h->d_lnls, sizeof(double) * done, hipMemcpyDeviceToHost));
  TSEM_HIP(hipMemsetAsync(h->d_ctl, 0, 8, h->stream));     // passes launched outside a chunk must not see a stale stop flag
  return TSEM_OK;
}

int tsem_em_steps(tsem_ctx* h, int32_t n, double* diffs_out) {
  int32_t done = 0;
  return tsem_em_chunk(h, n, 0.0, 0, 0, &done, nullptr, diffs_out, nullptr);
}

// the final log-likelihood (model.py:800-801), all-reduced; redone on the two-pass kernels after a time-out
static int final_lnl(tsem_ctx* h, double* lnl) {
  if (int rc = ensure_ctl(h)) return rc;
  for (int attempt = 0;; ++attempt) {
    if (int rc = enqueue_lnl_reduce(h)) return rc;
    double v[2] = {0, 0};
    TSEM_HIP(hipMemcpyAsync(v, h->d_ctld + 1, 16, hipMemcpyDeviceToHost, h->stream));
    TSEM_HIP(hipStreamSynchronize(h->stream));
    if (v[1] > 0.0) {
      if (attempt >= 2) TSEM_FAIL(TSEM_ERR_TIMEOUT, "lnl pass: repeated hand-off time-outs");
      uint32_t mine = 0;
      if (int rc = take_fused_error(h, &mine)) return rc;
      if (mine) { if (int rc = tsem_fallback_twopass(h)) return rc; }
      continue;
    }
    *lnl = v[0];
    return TSEM_OK;
  }
}

int tsem_final_lnl(tsem_ctx* h, double* lnl) {
  if (!h || !h->have_model || !lnl) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  return final_lnl(h, lnl);
}

int tsem_em_run(tsem_ctx* h, double epsilon, int32_t max_iter, int32_t use_likelihood, int32_t* n_iter,
                int32_t* converged, double* lnl_out, double* diffs, double* lnls, double* pi_init,
                double* theta_init) {
  if (!h || !h->have_model) return TSEM_ERR_ARG;
  int inum = 0;
  bool conv = false;
  double lnl = INFINITY;
  do {                                        // model.py:771-797: at least one iteration
    const int want = std::max(1, std::min(8, max_iter - inum));
    int32_t done = 0, stopped = 0;
    std::vector<double> d(want), l(want);
    if (int rc = tsem_em_chunk(h, want, epsilon, use_likelihood, inum == 0, &done, &stopped, d.data(), l.data())) return rc;
    for (int i = 0; i < done; ++i) {
      if (diffs && inum + i < std::max(1, max_iter)) diffs[inum + i] = d[i];
      if (lnls && use_likelihood && inum + i < std::max(1, max_iter)) lnls[inum + i] = l[i];
      if (use_likelihood) lnl = l[i];
    }
    inum += done;
    conv = stopped != 0;
  } while (!conv && inum < max_iter);
  if (pi_init || theta_init) { if (int rc = tsem_get_params(h, TSEM_Z_FIRST, pi_init, theta_init)) return rc; }
  if (!use_likelihood) {                      // model.py:800-801
    if (int rc = final_lnl(h, &lnl)) return rc;
  }
  h->lnl_prev_seed = lnl;                     // what the next run's first lnl is compared with (model.py:786)
  if (n_iter) *n_iter = inum;
  if (converged) *converged = conv ? 1 : 0;
  if (lnl_out) *lnl_out = lnl;
  return TSEM_OK;
}

int tsem_set_prev_lnl(tsem_ctx* h, double lnl) {
  if (!h) return TSEM_ERR_ARG;
  h->lnl_prev_seed = lnl;
  return TSEM_OK;
}

// ---------------------------------------------------------------------------
// communicator (RCCL over xGMI; one per process / GPU)
// ---------------------------------------------------------------------------
static thread_local std::string g_comm_err;            // (per host thread: the in-process transport runs one rank per thread)
const char* tsem_comm_last_error(void) { return g_comm_err.c_str(); }

int tsem_comm_library_info(char* buf, int32_t cap) {
  if (!buf || cap <= 0) return TSEM_ERR_ARG;
  NcclApi* N = nccl_api();
  std::string t;
  if (N->ok) {
    t = "rccl " + std::to_string(N->version / 10000) + "." + std::to_string(N->version / 100 % 100) + "." + std::to_string(N->version % 100) +
        " (" + N->path + ")";
    if (!N->others.empty()) t += "; other copies mapped: " + N->others;
  } else {
    t = "rccl unavailable: " + N->err;
  }
  snprintf(buf, (size_t)cap, "%s", t.c_str());
  return N->ok ? TSEM_OK : TSEM_ERR_HIP;
}

int tsem_comm_unique_id(void* id128) {
  if (!id128) return TSEM_ERR_ARG;
  static_assert(sizeof(ncclUniqueId) == TSEM_COMM_ID_BYTES, "ncclUniqueId size");
  NcclApi* N = nccl_api();
  if (!N->ok) { g_comm_err = N->err; return TSEM_ERR_HIP; }
  ncclUniqueId id;
  ncclResult_t r = N->GetUniqueId(&id);
  if (r != ncclSuccess) { g_comm_err = std::string("ncclGetUniqueId: ") + N->GetErrorString(r); return TSEM_ERR_HIP; }
  memcpy(id128, &id, sizeof(id));
  return TSEM_OK;
}

int tsem_comm_create(tsem_comm** out, int device, const void* id128, int rank, int world) {
  if (!out || !id128 || world < 1 || rank < 0 || rank >= world) return TSEM_ERR_ARG;
  *out = nullptr;
  NcclApi* N = nccl_api();
  if (!N->ok) { g_comm_err = N->err; return TSEM_ERR_HIP; }
  if (hipSetDevice(device) != hipSuccess) { g_comm_err = "hipSetDevice failed"; return TSEM_ERR_HIP; }
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  tsem_comm* c = new tsem_comm();
  c->device = device; c->rank = rank; c->world = world;
  ncclResult_t r = N->CommInitRank(&c->nccl, world, id, rank);
  if (r != ncclSuccess) { g_comm_err = std::string("ncclCommInitRank: ") + N->GetErrorString(r); delete c; return TSEM_ERR_HIP; }
  *out = c;
  return TSEM_OK;
}

int tsem_comm_local_group(tsem_local_group** out, int device, int world) {
  if (!out || world < 1 || world > TS_LOCAL_MAXW) return TSEM_ERR_ARG;
  *out = nullptr;
  if (hipSetDevice(device) != hipSuccess) { g_comm_err = "hipSetDevice failed"; return TSEM_ERR_HIP; }
  tsem_local_group* g = new tsem_local_group();
  g->world = world; g->device = device;
  for (int q = 0; q < world; ++q)
    for (int par = 0; par < 2; ++par)
      if (hipEventCreateWithFlags(&g->ready[q][par], hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&g->done[q][par], hipEventDisableTiming) != hipSuccess) {
        g_comm_err = "hipEventCreate failed"; delete g; return TSEM_ERR_HIP;
      }
  *out = g;
  return TSEM_OK;
}

void tsem_comm_local_group_destroy(tsem_local_group* g) {
  if (!g) return;
  (void)hipSetDevice(g->device);
  (void)hipDeviceSynchronize();
  for (int q = 0; q < g->world; ++q)
    for (int par = 0; par < 2; ++par) {
      if (g->slot[q][par]) (void)hipFree(g->slot[q][par]);
      if (g->ready[q][par]) (void)hipEventDestroy(g->ready[q][par]);
      if (g->done[q][par]) (void)hipEventDestroy(g->done[q][par]);
    }
  delete g;
}

int tsem_comm_create_local(tsem_comm** out, tsem_local_group* g, int rank) {
  if (!out || !g || rank < 0 || rank >= g->world) return TSEM_ERR_ARG;
  *out = nullptr;
  {
    std::lock_guard<std::mutex> lk(g->mu);
    if (g->taken[rank]) { g_comm_err = "tsem_comm_create_local: this rank of the group is taken"; return TSEM_ERR_ARG; }
    g->taken[rank] = true;
  }
  tsem_comm* c = new tsem_comm();
  c->device = g->device; c->rank = rank; c->world = g->world; c->local = g;
  *out = c;
  return TSEM_OK;
}

void tsem_comm_destroy(tsem_comm* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->d_stage) (void)hipFree(c->d_stage);
  if (c->nccl) (void)nccl_api()->CommDestroy(c->nccl);
  if (c->local) { std::lock_guard<std::mutex> lk(c->local->mu); c->local->taken[c->rank] = false; }
  delete c;
}

int tsem_comm_attach(tsem_ctx* h, tsem_comm* c) {
  if (!h) return TSEM_ERR_ARG;
  if (c && c->device != h->device) TSEM_FAIL(TSEM_ERR_ARG, "communicator and handle are on different devices");
  h->comm = c;
  return TSEM_OK;
}

int tsem_comm_allreduce(tsem_ctx* h, int64_t offset, int64_t count) {
  if (!h || !h->have_model || offset < 0 || count < 0 || offset + count > h->K + 2) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  return comm_allreduce_red(h, offset, count);
}

int tsem_comm_allreduce_host(tsem_comm* c, void* data, int64_t count, int dtype) {
  if (!c || !c->active() || (!data && count) || count < 0 || dtype < 0 || dtype > 3) return TSEM_ERR_ARG;
  if (count == 0) return TSEM_OK;
  if (hipSetDevice(c->device) != hipSuccess) { g_comm_err = "hipSetDevice failed"; return TSEM_ERR_HIP; }
  const size_t bytes = (size_t)count * 8;
  if (c->stage_bytes < bytes) {
    if (c->d_stage) (void)hipFree(c->d_stage);
    c->d_stage = nullptr; c->stage_bytes = 0;
    if (hipMalloc(&c->d_stage, bytes) != hipSuccess) { g_comm_err = "hipMalloc failed (all-reduce staging)"; return TSEM_ERR_NOMEM; }
    c->stage_bytes = bytes;
  }
  // Host vectors travel on the null stream.  The same communicator also serves an engine's own stream (tsem_em_chunk);
  // RCCL wants one stream per communicator at a time, so everything the device still has queued is drained first
  // (these are set-up and report sums: a device synchronisation costs nothing here).
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(c->d_stage, data, bytes, hipMemcpyHostToDevice);
  int rc = TSEM_OK;
  if (e == hipSuccess) rc = comm_allreduce_dev(c, c->d_stage, (size_t)count, dtype, nullptr, g_comm_err);
  if (rc) return rc;
  if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
  if (e == hipSuccess) e = hipMemcpy(data, c->d_stage, bytes, hipMemcpyDeviceToHost);
  if (e != hipSuccess) { g_comm_err = std::string("all-reduce staging: ") + hipGetErrorString(e); return TSEM_ERR_HIP; }
  return TSEM_OK;
}

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

static int rowpass_args(tsem_ctx* h, int which, RowPassArgs& A) {
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
static int rowpass_grid(tsem_ctx* h) { return (int)std::min<int64_t>(8192, std::max<int64_t>(1, (h->N + 15) / 16)); }

static int export_z_with(tsem_ctx* h, RowPassArgs& A, double* z) {
  double* d_z = nullptr;
  TSEM_ALLOC(d_z, h->nnz);
  A.zout = d_z;
  if (h->N) k_rowpass<RP_EXPORT_Z><<<rowpass_grid(h), 256, (size_t)A.lut_len * 8, h->stream>>>(A);
  TSEM_HIP(hipGetLastError());
  if (h->nnz) TSEM_HIP(hipMemcpyAsync(z, d_z, sizeof(double) * h->nnz, hipMemcpyDeviceToHost, h->stream));
  TSEM_HIP(hipStreamSynchronize(h->stream));
  (void)hipFree(d_z);
  return TSEM_OK;
}

int tsem_export_z(tsem_ctx* h, int which, double* z) {
  if (!h || !h->d_indptr || !z) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
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
  TSEM_ALLOC(d_nb, h->N);
  A.nbest = d_nb;
  if (h->N) k_rowpass<RP_BEST><<<rowpass_grid(h), 256, (size_t)A.lut_len * 8, h->stream>>>(A);
  TSEM_HIP(hipGetLastError());
  if (h->N) TSEM_HIP(hipMemcpyAsync(nbest, d_nb, sizeof(int32_t) * h->N, hipMemcpyDeviceToHost, h->stream));
  TSEM_HIP(hipStreamSynchronize(h->stream));
  (void)hipFree(d_nb);
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
  TSEM_ALLOC(d_nb, h->N); TSEM_ALLOC(d_rows, h->N); TSEM_ALLOC(d_n, 1);
  A.nbest = d_nb;
  k_rowpass<RP_BEST><<<rowpass_grid(h), 256, (size_t)A.lut_len * 8, h->stream>>>(A);
  TSEM_HIP(hipGetLastError());
  size_t tb = 0;
  TiedRow pred{d_nb};
  rocprim::counting_iterator<int32_t> first(0);
  TSEM_HIP(rocprim::select(nullptr, tb, first, d_rows, d_n, (size_t)h->N, pred, h->stream));
  void* tmp = nullptr;
  TSEM_HIP(hipMalloc(&tmp, tb ? tb : 1));
  TSEM_HIP(rocprim::select(tmp, tb, first, d_rows, d_n, (size_t)h->N, pred, h->stream));
  unsigned long long n = 0;
  TSEM_HIP(hipMemcpyAsync(&n, d_n, 8, hipMemcpyDeviceToHost, h->stream));
  TSEM_HIP(hipStreamSynchronize(h->stream));
  (void)hipFree(tmp);
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
  (void)hipFree(d_nb); (void)hipFree(d_rows); (void)hipFree(d_n);
  if (d_cnt) (void)hipFree(d_cnt);
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
  (void)hipFree(lo);
  return TSEM_OK;
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
    uint32_t has_zero = 0;
    TSEM_HIP(hipMemcpyAsync(&has_zero, h->d_ucount + h->K, 4, hipMemcpyDeviceToHost, h->stream));
    TSEM_HIP(hipStreamSynchronize(h->stream));
    if (method == TSEM_RA_ALL && which == TSEM_Z_INITIAL && !has_zero && h->d_colcount) {
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
  double *d_cs = nullptr, *d_mask = nullptr;
  int32_t* d_picks = nullptr;
  TSEM_ALLOC(d_cs, h->K);
  TSEM_HIP(hipMemsetAsync(d_cs, 0, sizeof(double) * h->K, h->stream));
  if (mask) TSEM_ALLOC(d_mask, h->nnz);
  if (method == TSEM_RA_CHOOSE && picks) {
    TSEM_ALLOC(d_picks, h->N);
    if (h->N) TSEM_HIP(hipMemcpyAsync(d_picks, picks, sizeof(int32_t) * h->N, hipMemcpyHostToDevice, h->stream));
  }
  A.colsums = d_cs; A.zout = d_mask; A.picks = d_picks;
  double* d_lo = nullptr;
  if (int rc = rowpass_lo_begin(h, A, h->K, &d_lo)) return rc;
  if (h->N && h->d_colmap && h->d_col_of_pc && h->P > 0) {
    // hot slots of every part in LDS.  `all` emits one value per stored entry, so it wants as many slots as fit: one
    // 1024-thread workgroup per CU with ~150 KB of accumulators.  The other modes emit at most a few values per ROW
    // and the pass is bound by the latency of its dependent loads (row pointers -> entries), not by atomics: two
    // workgroups per CU (32 waves) with half the slots each (option "rowpass_wgs").
    const int wgs = (method == TSEM_RA_ALL || h->opt_rowpass_wgs < 2) ? 1 : 2;
    A.colmap = h->d_colmap; A.col_of_pc = h->d_col_of_pc; A.P = h->P; A.Kp = h->Kp;
    A.Hs = std::max(0, std::min(h->Kp, (int)((TS_LDS_MAX / wgs - 8192 / wgs - 1024 - A.lut_len * 8) / 8 / h->P)));
    TSEM_HIP(hipFuncSetAttribute((const void*)k_rowpass<RP_REASSIGN>, hipFuncAttributeMaxDynamicSharedMemorySize, TS_LDS_MAX - 1024));
    void (*kern)(RowPassArgs) = k_rowpass<RP_REASSIGN>;
    switch (method) {
      case TSEM_RA_EXCLUDE: kern = k_rowpass<RP_REASSIGN, TSEM_RA_EXCLUDE>; break;
      case TSEM_RA_CHOOSE:  kern = k_rowpass<RP_REASSIGN, TSEM_RA_CHOOSE>; break;
      case TSEM_RA_AVERAGE: kern = k_rowpass<RP_REASSIGN, TSEM_RA_AVERAGE>; break;
      case TSEM_RA_CONF:    kern = k_rowpass<RP_REASSIGN, TSEM_RA_CONF>; break;
      case TSEM_RA_UNIQUE:  kern = k_rowpass<RP_REASSIGN, TSEM_RA_UNIQUE>; break;
      case TSEM_RA_ALL:     kern = k_rowpass<RP_REASSIGN, TSEM_RA_ALL>; break;
    }
    TSEM_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, TS_LDS_MAX - 1024));
    kern<<<h->n_cu * wgs, 1024, (size_t)(A.P * A.Hs + A.lut_len) * 8, h->stream>>>(A);
  } else if (h->N) {
    k_rowpass<RP_REASSIGN><<<rowpass_grid(h), 256, (size_t)A.lut_len * 8, h->stream>>>(A);
  }
  TSEM_HIP(hipGetLastError());
  if (int rc = rowpass_lo_end(h, d_cs, d_lo, h->K)) return rc;
  TSEM_HIP(hipMemcpyAsync(colsums, d_cs, sizeof(double) * h->K, hipMemcpyDeviceToHost, h->stream));
  if (mask && h->nnz) TSEM_HIP(hipMemcpyAsync(mask, d_mask, sizeof(double) * h->nnz, hipMemcpyDeviceToHost, h->stream));
  TSEM_HIP(hipStreamSynchronize(h->stream));
  (void)hipFree(d_cs);
  if (d_mask) (void)hipFree(d_mask);
  if (d_picks) (void)hipFree(d_picks);
  return TSEM_OK;
}

// One pass for the column sums output_report takes from one z (model.py:432-457): conf | exclude | average, and the rows
// with several best hits (the only rows `choose` treats differently from `exclude`) compacted in row order.
int tsem_report_colsums(tsem_ctx* h, int which, double thresh, double* out3K, int64_t* n_ties) {
  if (!h || !h->d_indptr || !out3K) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  RowPassArgs A;
  if (int rc = rowpass_args(h, which, A)) return rc;
  dfree(h->d_tie_rows); dfree(h->d_tie_cnt); h->n_ties = 0;
  if (n_ties) *n_ties = 0;
  const int K = h->K;
  double* d_cs = nullptr;
  TSEM_ALLOC(d_cs, 3 * (int64_t)K);
  TSEM_HIP(hipMemsetAsync(d_cs, 0, sizeof(double) * 3 * K, h->stream));
  if (h->N) {
    // per-row best-hit counts, the compacted tie list and the scan's scratch stay allocated between reports (two N-sized
    // vectors: allocating and freeing them cost more than the pass's kernels)
    if (!h->d_rep_nb) { TSEM_ALLOC(h->d_rep_nb, h->N); TSEM_ALLOC(h->d_rep_rows, h->N); TSEM_ALLOC(h->d_rep_n, 1); }
    int32_t *const d_nb = h->d_rep_nb, *const d_rows = h->d_rep_rows;
    unsigned long long* const d_n = h->d_rep_n;
    A.thresh = thresh; A.colsums = d_cs; A.nbest = d_nb;
    void (*kern)(RowPassArgs) = k_rowpass<RP_REPORT>;
    if (h->opt_report_kernel != 0 && which != TSEM_Z_USER && h->d_rid16 && h->d_col_of_id && A.lut_len > 0) {
      // the streaming report kernel (k_report_rows): lanes per row x entries per lane = the smallest capacity that
      // fewer than 0.5 % of the rows exceed (row-length histogram of tsem_rowstats); the rest goes to k_report_slow
      const int IDN = h->Kpad;
      ReportArgs R;
      R.N = h->N; R.nnz = h->nnz; R.K = K; R.IDN = IDN; R.indptr = h->d_indptr; R.rid = h->d_rid16; R.raw = h->d_raw;
      R.lut = h->d_lut; R.lut_len = A.lut_len; R.cnat2 = nullptr; R.thresh = thresh; R.nbest = d_nb;
      const bool init = A.pi == nullptr;
      double *d_g = nullptr, *d_c2 = nullptr;
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
      // LDS: one workgroup of rr_nt() threads per CU (or two, option rowpass_wgs).  The final z wants pi*theta of as many
      // ids as fit (8 B each) next to a few thousand accumulator slots (16 B each); the initial z has no pi*theta.
      const int wgs = h->opt_rowpass_wgs >= 2 && h->opt_report_wgs2 ? 2 : 1;
      const int lds_avail = TS_LDS_MAX / wgs - 2048 - R.lut_len * 8;
      const int slot_bytes = exact ? 24 : 16;                // conf (f64) + two counters (+ the low pieces)
      R.Hs = std::min(IDN, init ? lds_avail / slot_bytes : std::min(3072, lds_avail / slot_bytes / 4));
      R.HC = init ? 0 : std::max(0, std::min(IDN, (lds_avail - R.Hs * slot_bytes) / 8));
      int cap = 256;                                       // G x E
      if (h->opt_report_lanes > 0) {
        cap = (int)h->opt_report_lanes;
      } else if (h->have_rowstats) {
        for (int q = 0; q < 6; ++q)
          if ((double)h->len_gt[q] <= 0.005 * (double)h->N) { cap = 8 << q; break; }
      }
      void (*rk)(ReportArgs) = nullptr;
#define RK(G_, E_) (init ? k_report_rows<G_, E_, true> : k_report_rows<G_, E_, false>)
      if (cap <= 8) rk = RK(1, 8); else if (cap <= 16) rk = RK(1, 16); else if (cap <= 32) rk = RK(2, 16);
      else if (cap <= 64) rk = RK(4, 16); else if (cap <= 128) rk = RK(8, 16); else rk = RK(16, 16);
#undef RK
      TSEM_HIP(hipFuncSetAttribute((const void*)rk, hipFuncAttributeMaxDynamicSharedMemorySize, TS_LDS_MAX - 1024));
      R.dbg = (int)h->opt_report_dbg;
      R.defer_rows = d_rows; R.defer_n = d_n;               // (d_rows is the tie list later: the slow kernel is done with it by then)
      TSEM_HIP(hipMemsetAsync(d_n, 0, sizeof(unsigned long long), h->stream));
      rk<<<h->n_cu * wgs, rr_nt(init), (size_t)R.lut_len * 8 + (size_t)R.HC * 8 + (size_t)R.Hs * slot_bytes, h->stream>>>(R);
      TSEM_HIP(hipGetLastError());
      if (init) k_report_slow<true><<<h->n_cu * 2, 256, 0, h->stream>>>(R);
      else k_report_slow<false><<<h->n_cu * 2, 256, 0, h->stream>>>(R);
      TSEM_HIP(hipGetLastError());
      k_report_finish<<<cdiv64(IDN, 256), 256, 0, h->stream>>>(IDN, K, h->d_col_of_id, R.g_conf, R.g_n1, R.g_n2, R.g_avgt, R.g_conf_lo, R.g_avgt_lo, d_cs);
      TSEM_HIP(hipGetLastError());
      TSEM_HIP(hipStreamSynchronize(h->stream));
      (void)hipFree(d_g);
      if (d_c2) (void)hipFree(d_c2);
    } else if (h->d_colmap && h->d_col_of_pc && h->P > 0) {
      const int wgs = h->opt_rowpass_wgs < 2 ? 1 : 2;
      A.colmap = h->d_colmap; A.col_of_pc = h->d_col_of_pc; A.P = h->P; A.Kp = h->Kp;
      A.Hs = std::max(0, std::min(h->Kp, (int)((TS_LDS_MAX / wgs - 8192 / wgs - 1024 - A.lut_len * 8) / 8 / h->P / 3)));
      TSEM_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, TS_LDS_MAX - 1024));
      double* d_lo = nullptr;
      if (int rc = rowpass_lo_begin(h, A, 3 * (int64_t)K, &d_lo)) return rc;
      kern<<<h->n_cu * wgs, 1024, (size_t)(3 * A.P * A.Hs + A.lut_len) * 8, h->stream>>>(A);
      TSEM_HIP(hipGetLastError());
      if (int rc = rowpass_lo_end(h, d_cs, d_lo, 3 * (int64_t)K)) return rc;
    } else {
      double* d_lo = nullptr;
      if (int rc = rowpass_lo_begin(h, A, 3 * (int64_t)K, &d_lo)) return rc;
      kern<<<rowpass_grid(h), 256, (size_t)A.lut_len * 8, h->stream>>>(A);
      TSEM_HIP(hipGetLastError());
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
  (void)hipFree(d_cs);
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
  RowPassArgs A;
  if (int rc = rowpass_args(h, which, A)) return rc;
  if (rows)
    for (int64_t i = 0; i < n; ++i)
      if (rows[i] < 0 || rows[i] >= h->N) TSEM_FAIL(TSEM_ERR_ARG, "tsem_reassign_rows: row out of range");
  double* d_cs = nullptr;
  int32_t *d_rows = nullptr, *d_picks = nullptr;
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
  A.method = method; A.thresh = thresh; A.colsums = d_cs; A.picks = d_picks;
  A.rowlist = rows ? d_rows : h->d_tie_rows; A.nlist = n;
  const int grid = (int)std::min<int64_t>(8192, std::max<int64_t>(1, (n + 15) / 16));
  double* d_lo = nullptr;
  if (int rc = rowpass_lo_begin(h, A, h->K, &d_lo)) return rc;
  k_rowpass<RP_REASSIGN><<<grid, 256, (size_t)A.lut_len * 8, h->stream>>>(A);
  TSEM_HIP(hipGetLastError());
  if (int rc = rowpass_lo_end(h, d_cs, d_lo, h->K)) return rc;
  TSEM_HIP(hipMemcpyAsync(colsums, d_cs, sizeof(double) * h->K, hipMemcpyDeviceToHost, h->stream));
  TSEM_HIP(hipStreamSynchronize(h->stream));
  (void)hipFree(d_cs);
  if (d_rows) (void)hipFree(d_rows);
  if (d_picks) (void)hipFree(d_picks);
  return TSEM_OK;
}

int tsem_reassign_groups(tsem_ctx* h, int method, double thresh, int which, const int32_t* picks,
                         const int32_t* group_of_row, int32_t n_groups, double* out) {
  if (!h || !h->d_indptr || !group_of_row || !out || n_groups < 0) return TSEM_ERR_ARG;
  if (method < TSEM_RA_EXCLUDE || method > TSEM_RA_ALL) TSEM_FAIL(TSEM_ERR_ARG, "bad reassign method");
  if (int rc = ensure_device(h)) return rc;
  for (int64_t i = 0; i < h->N; ++i)
    if (group_of_row[i] >= n_groups) TSEM_FAIL(TSEM_ERR_ARG, "group_of_row entry out of range");
  RowPassArgs A;
  if (int rc = rowpass_args(h, which, A)) return rc;
  A.method = method; A.thresh = thresh;
  const int64_t n_out = (int64_t)n_groups * h->K;
  double* d_out = nullptr;
  int32_t *d_picks = nullptr, *d_grp = nullptr;
  TSEM_ALLOC(d_out, n_out);
  TSEM_ALLOC(d_grp, h->N);
  TSEM_HIP(hipMemsetAsync(d_out, 0, sizeof(double) * std::max<int64_t>(1, n_out), h->stream));
  if (h->N) TSEM_HIP(hipMemcpyAsync(d_grp, group_of_row, sizeof(int32_t) * h->N, hipMemcpyHostToDevice, h->stream));
  if (method == TSEM_RA_CHOOSE && picks) {
    TSEM_ALLOC(d_picks, h->N);
    if (h->N) TSEM_HIP(hipMemcpyAsync(d_picks, picks, sizeof(int32_t) * h->N, hipMemcpyHostToDevice, h->stream));
  }
  A.colsums = d_out; A.picks = d_picks; A.group = d_grp;
  double* d_lo = nullptr;
  if (int rc = rowpass_lo_begin(h, A, n_out, &d_lo)) return rc;
  if (h->N && n_out) k_rowpass<RP_REASSIGN><<<rowpass_grid(h), 256, (size_t)A.lut_len * 8, h->stream>>>(A);
  TSEM_HIP(hipGetLastError());
  if (int rc = rowpass_lo_end(h, d_out, d_lo, n_out)) return rc;
  if (n_out) TSEM_HIP(hipMemcpyAsync(out, d_out, sizeof(double) * n_out, hipMemcpyDeviceToHost, h->stream));
  TSEM_HIP(hipStreamSynchronize(h->stream));
  (void)hipFree(d_out); (void)hipFree(d_grp);
  if (d_picks) (void)hipFree(d_picks);
  return TSEM_OK;
}

int tsem_mstep(tsem_ctx* h, const double* z, double* pi_hat, double* theta_hat) {
  if (!h || !h->have_model || !z || !pi_hat || !theta_hat) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  RowPassArgs A;
  if (int rc = rowpass_args(h, TSEM_Z_CUR, A)) return rc;
  double *d_z = nullptr, *d_cs = nullptr;
  TSEM_ALLOC(d_z, h->nnz);
  TSEM_ALLOC(d_cs, h->K);
  if (h->nnz) TSEM_HIP(hipMemcpyAsync(d_z, z, sizeof(double) * h->nnz, hipMemcpyHostToDevice, h->stream));
  TSEM_HIP(hipMemsetAsync(d_cs, 0, sizeof(double) * h->K, h->stream));
  A.colsums = d_cs;
  if (h->N) k_mstep_rows<<<rowpass_grid(h), 256, 0, h->stream>>>(A, d_z);
  if (comm_on(h)) {                                         // row-sharded: thetasum over all ranks (model.py:731)
    if (int rc = comm_allreduce_dev(h->comm, d_cs, (size_t)h->K, 0, h->stream, h->err)) { (void)hipFree(d_z); (void)hipFree(d_cs); return rc; }
  }
  const double tpw = h->theta_prior * h->w_max, ppw = h->pi_prior * h->w_max;
  k_hats<<<cdiv64(h->K, 256), 256, 0, h->stream>>>(h->K, d_cs, h->d_pisum0, tpw, h->W_amb + tpw * h->K, ppw,
                                                  h->W_tot + ppw * h->K, h->d_tmp_pi, h->d_tmp_theta);
  TSEM_HIP(hipGetLastError());
  TSEM_HIP(hipMemcpyAsync(pi_hat, h->d_tmp_pi, sizeof(double) * h->K, hipMemcpyDeviceToHost, h->stream));
  TSEM_HIP(hipMemcpyAsync(theta_hat, h->d_tmp_theta, sizeof(double) * h->K, hipMemcpyDeviceToHost, h->stream));
  TSEM_HIP(hipStreamSynchronize(h->stream));
  (void)hipFree(d_z); (void)hipFree(d_cs);
  return TSEM_OK;
}

int tsem_calc_lnl(tsem_ctx* h, const double* z, const double* pi, const double* theta, double* lnl) {
  if (!h || !h->have_model || !z || !pi || !theta || !lnl) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  RowPassArgs A;
  if (int rc = rowpass_args(h, TSEM_Z_CUR, A)) return rc;
  double* d_z = nullptr;
  TSEM_ALLOC(d_z, h->nnz);
  if (h->nnz) TSEM_HIP(hipMemcpyAsync(d_z, z, sizeof(double) * h->nnz, hipMemcpyHostToDevice, h->stream));
  TSEM_HIP(hipMemcpyAsync(h->d_tmp_pi, pi, sizeof(double) * h->K, hipMemcpyHostToDevice, h->stream));
  TSEM_HIP(hipMemcpyAsync(h->d_tmp_theta, theta, sizeof(double) * h->K, hipMemcpyHostToDevice, h->stream));
  A.pi = h->d_tmp_pi; A.theta = h->d_tmp_theta;
  int grid = std::min(4096, rowpass_grid(h));
  if (h->N) k_lnl_rows<<<grid, 256, 0, h->stream>>>(A, d_z, h->d_lnl_part);
  k_sum_parts<<<1, 256, 0, h->stream>>>(h->d_lnl_part, h->N ? grid : 0, h->d_lnl_part, 0, h->d_lnl_part + 8000);
  TSEM_HIP(hipGetLastError());
  if (comm_on(h)) {
    if (int rc = comm_allreduce_dev(h->comm, h->d_lnl_part + 8000, 1, 0, h->stream, h->err)) { (void)hipFree(d_z); return rc; }
  }
  TSEM_HIP(hipMemcpyAsync(lnl, h->d_lnl_part + 8000, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  TSEM_HIP(hipStreamSynchronize(h->stream));
  (void)hipFree(d_z);
  return TSEM_OK;
}

// ---------------------------------------------------------------------------
// numpy's LEGACY random stream for `choose` (sparse_plus.py:140-154)
// ---------------------------------------------------------------------------
// choose_random draws one np.random.choice per row with several best hits, on numpy's global legacy RandomState — the
// stream `telescope assign` seeds (telescope_assign.py:429-431).  One draw below a bound c is, in numpy's C
// (legacy-distributions / _bounded_integers, masked rejection): mask = the smallest 2^b - 1 >= c - 1, then 32-bit
// Mersenne-Twister outputs until (output & mask) <= c - 1.  np.random.randint(0, counts) on an array does exactly that
// per element (46 ms for the 5.6e6 tied rows of the 50M-row benchmark: it was the largest item of the whole report);
// this is the same loop in C on the caller's MT19937 state (np.random.get_state() -> here -> np.random.set_state()),
// bit for bit the same picks and the same state afterwards (tests/test_host_logic.py).  Host code: the stream is
// sequential by definition.
int tsem_legacy_randint(uint32_t* key624, int32_t* pos, const int32_t* counts, int64_t n, int32_t* out) {
  if (!key624 || !pos || (!counts && n) || (!out && n) || n < 0 || *pos < 0 || *pos > 624) return TSEM_ERR_ARG;
  constexpr int NN = 624, MM = 397;
  constexpr uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MATRIX_A = 0x9908b0dfu;
  uint32_t* mt = key624;
  int p = *pos;
  // the 624 outputs of the current state, tempered in one vectorisable sweep (the draw loop then only masks and compares:
  // 2.0 -> ~1 ns per draw; the state array itself stays untempered, as numpy keeps it)
  uint32_t buf[NN];
  auto temper_all = [&]() {
    for (int kk = 0; kk < NN; ++kk) {
      uint32_t y = mt[kk];
      y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
      buf[kk] = y;
    }
  };
  auto refill = [&]() {
    int kk = 0;
    for (; kk < NN - MM; ++kk) { const uint32_t y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER); mt[kk] = mt[kk + MM] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u); }
    for (; kk < NN - 1; ++kk) { const uint32_t y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER); mt[kk] = mt[kk + (MM - NN)] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u); }
    const uint32_t y = (mt[NN - 1] & UPPER) | (mt[0] & LOWER);
    mt[NN - 1] = mt[MM - 1] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u);
    p = 0;
    temper_all();
  };
  if (p < NN) temper_all();
  for (int64_t i = 0; i < n; ++i) {
    const int32_t c = counts[i];
    if (c <= 0) return TSEM_ERR_ARG;                       // (numpy raises "low >= high")
    const uint32_t rng = (uint32_t)c - 1u;
    if (rng == 0) { out[i] = 0; continue; }                // no random number is consumed
    const uint32_t mask = 0xFFFFFFFFu >> __builtin_clz(rng);   // the smallest 2^b - 1 >= rng
    uint32_t v;
    do {
      if (p == NN) refill();
      v = buf[p++] & mask;
    } while (v > rng);
    out[i] = (int32_t)v;
  }
  *pos = p;
  return TSEM_OK;
}

// ---------------------------------------------------------------------------
// csr_matrix_plus primitives (stateless)
// ---------------------------------------------------------------------------
static int csr_prim(int device, int64_t n_rows, int32_t n_cols, const int64_t* indptr, const double* data,
                    double* out_d, int8_t* out_b) {
  if (hipSetDevice(device) != hipSuccess) { g_create_err = "hipSetDevice failed (no usable HIP device)"; return TSEM_ERR_HIP; }
  if (n_rows < 0 || !indptr) return TSEM_ERR_ARG;
  int64_t nnz = indptr[n_rows];
  int64_t* d_ip = nullptr; double *d_in = nullptr, *d_od = nullptr; int8_t* d_ob = nullptr;
  bool ok = hipMalloc((void**)&d_ip, sizeof(int64_t) * (n_rows + 1)) == hipSuccess &&
            hipMalloc((void**)&d_in, sizeof(double) * std::max<int64_t>(1, nnz)) == hipSuccess;
  if (ok && out_d) ok = hipMalloc((void**)&d_od, sizeof(double) * std::max<int64_t>(1, nnz)) == hipSuccess;
  if (ok && out_b) ok = hipMalloc((void**)&d_ob, std::max<int64_t>(1, nnz)) == hipSuccess;
  int rc = TSEM_OK;
  if (!ok) { g_create_err = "hipMalloc failed"; rc = TSEM_ERR_NOMEM; }
  if (ok) {
    (void)hipMemcpy(d_ip, indptr, sizeof(int64_t) * (n_rows + 1), hipMemcpyHostToDevice);
    if (nnz) (void)hipMemcpy(d_in, data, sizeof(double) * nnz, hipMemcpyHostToDevice);
    int grid = (int)std::min<int64_t>(8192, std::max<int64_t>(1, (n_rows + 15) / 16));
    if (n_rows) {
      if (out_d) k_norm_rows<<<grid, 256>>>(n_rows, d_ip, d_in, d_od);
      if (out_b) k_binmax_rows<<<grid, 256>>>(n_rows, n_cols, d_ip, d_in, d_ob);
    }
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { g_create_err = hipGetErrorString(e); rc = TSEM_ERR_HIP; }
    if (rc == TSEM_OK && nnz) {
      if (out_d) (void)hipMemcpy(out_d, d_od, sizeof(double) * nnz, hipMemcpyDeviceToHost);
      if (out_b) (void)hipMemcpy(out_b, d_ob, nnz, hipMemcpyDeviceToHost);
    }
  }
  if (d_ip) (void)hipFree(d_ip);
  if (d_in) (void)hipFree(d_in);
  if (d_od) (void)hipFree(d_od);
  if (d_ob) (void)hipFree(d_ob);
  return rc;
}

// mode 0: out = data * (1 / sum(data))   norm()   sparse_plus.py:47-48
// mode 1: out = data * (1 / max(data))   scale()  sparse_plus.py:94-95   (max over the matrix incl. implicit zeros)
// mode 2: out = data * recip0(row max)   scale(1) sparse_plus.py:96-97
int tsem_csr_scale(int device, int mode, int64_t n_rows, int32_t n_cols, const int64_t* indptr, const double* data,
                   double* out) {
  if (hipSetDevice(device) != hipSuccess) { g_create_err = "hipSetDevice failed (no usable HIP device)"; return TSEM_ERR_HIP; }
  if (n_rows < 0 || !indptr || mode < 0 || mode > 2) return TSEM_ERR_ARG;
  const int64_t nnz = indptr[n_rows];
  int64_t* d_ip = nullptr; double *d_in = nullptr, *d_out = nullptr, *d_part = nullptr;
  const int G = 512;
  bool ok = hipMalloc((void**)&d_ip, sizeof(int64_t) * (n_rows + 1)) == hipSuccess &&
            hipMalloc((void**)&d_in, sizeof(double) * std::max<int64_t>(1, nnz)) == hipSuccess &&
            hipMalloc((void**)&d_out, sizeof(double) * std::max<int64_t>(1, nnz)) == hipSuccess &&
            hipMalloc((void**)&d_part, sizeof(double) * G) == hipSuccess;
  int rc = TSEM_OK;
  if (!ok) { g_create_err = "hipMalloc failed"; rc = TSEM_ERR_NOMEM; }
  if (ok) {
    (void)hipMemcpy(d_ip, indptr, sizeof(int64_t) * (n_rows + 1), hipMemcpyHostToDevice);
    if (nnz) (void)hipMemcpy(d_in, data, sizeof(double) * nnz, hipMemcpyHostToDevice);
    if (mode == 2) {
      int grid = (int)std::min<int64_t>(8192, std::max<int64_t>(1, (n_rows + 15) / 16));
      if (n_rows) k_scale_rows<<<grid, 256>>>(n_rows, n_cols, d_ip, d_in, d_out);
    } else if (nnz) {
      k_reduce_all<<<G, 256>>>(d_in, nnz, mode == 1, d_part);
      std::vector<double> part(G);
      (void)hipMemcpy(part.data(), d_part, sizeof(double) * G, hipMemcpyDeviceToHost);
      double r = mode == 1 ? -INFINITY : 0.0;
      for (int i = 0; i < G; ++i) r = mode == 1 ? std::max(r, part[i]) : r + part[i];
      if (mode == 1 && nnz < n_rows * (int64_t)n_cols) r = std::max(r, 0.0);
      k_scale_all<<<cdiv64(nnz, 256), 256>>>(d_in, nnz, 1.0 / r, d_out);
    }
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { g_create_err = hipGetErrorString(e); rc = TSEM_ERR_HIP; }
    if (rc == TSEM_OK && nnz) (void)hipMemcpy(out, d_out, sizeof(double) * nnz, hipMemcpyDeviceToHost);
  }
  if (d_ip) (void)hipFree(d_ip);
  if (d_in) (void)hipFree(d_in);
  if (d_out) (void)hipFree(d_out);
  if (d_part) (void)hipFree(d_part);
  return rc;
}

int tsem_csr_norm_rows(int device, int64_t n_rows, const int64_t* indptr, const double* data, double* out) {
  return csr_prim(device, n_rows, 0, indptr, data, out, nullptr);
}
int tsem_csr_binmax_rows(int device, int64_t n_rows, int32_t n_cols, const int64_t* indptr, const double* data,
                         int8_t* out) {
  return csr_prim(device, n_rows, n_cols, indptr, data, nullptr, out);
}

// ---------------------------------------------------------------------------
// instrumentation
// ---------------------------------------------------------------------------
int tsem_kernel_stats(tsem_ctx* h, int reset, double* em_ms, int64_t* em_launches, int64_t* algo_bytes) {
  if (!h) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  TSEM_HIP(hipStreamSynchronize(h->stream));
  for (size_t i = 0; i + 1 < h->ev_used; i += 2) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, h->ev[i], h->ev[i + 1]) == hipSuccess) h->em_ms_acc += ms;
  }
  h->ev_used = 0;
  if (em_ms) *em_ms = h->em_ms_acc;
  if (em_launches) *em_launches = h->em_timed;             // launches that were timed (option "kernel_timing")
  // one EM pass must read every stored entry of the ambiguous rows once:
  // 4 B packed local row/col + the value AS STORED (8 B fp64 Q, or a 2 B score code) per entry,
  // + 2 B row weight code per row
  if (algo_bytes) *algo_bytes = h->nnz_amb * (h->fmt_code ? 6 : 12) + h->N_amb * 2;
  if (reset) { h->em_ms_acc = 0; h->em_launches = 0; h->em_timed = 0; }
  return TSEM_OK;
}

int tsem_debug_fused_prof(tsem_ctx* h, uint64_t* out /* 64*16 */) {
  if (!h || !h->d_prof || !out) return TSEM_ERR_ARG;
  TSEM_HIP(hipStreamSynchronize(h->stream));
  TSEM_HIP(hipMemcpy(out, h->d_prof, 64 * 16 * 8, hipMemcpyDeviceToHost));
  return TSEM_OK;
}

__global__ void k_log1p_probe(int n, const double* x, double* y) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = ts_log1p_pos(x[i]);
}
/* the device log1p of the lnl passes on caller-supplied x >= 0 (accuracy test hook) */
int tsem_debug_log1p(int device, int32_t n, const double* x, double* y) {
  if (!x || !y || n < 0) return TSEM_ERR_ARG;
  if (hipSetDevice(device) != hipSuccess) return TSEM_ERR_HIP;
  double *dx = nullptr, *dy = nullptr;
  if (hipMalloc((void**)&dx, 8 * (size_t)std::max(1, n)) != hipSuccess || hipMalloc((void**)&dy, 8 * (size_t)std::max(1, n)) != hipSuccess)
    return TSEM_ERR_NOMEM;
  (void)hipMemcpy(dx, x, 8 * (size_t)n, hipMemcpyHostToDevice);
  if (n) k_log1p_probe<<<(n + 255) / 256, 256>>>(n, dx, dy);
  hipError_t e = hipMemcpy(y, dy, 8 * (size_t)n, hipMemcpyDeviceToHost);
  (void)hipFree(dx); (void)hipFree(dy);
  return e == hipSuccess ? TSEM_OK : TSEM_ERR_HIP;
}

__global__ void k_log1p_tab_probe(int n, const double* x, double* y) {
  __shared__ double2 tab[FZ_LOGTAB];
  if (threadIdx.x < FZ_LOGTAB) {
    const double ci = 1.0 + (double)threadIdx.x * (1.0 / FZ_LOGTAB);
    tab[threadIdx.x] = make_double2(1.0 / ci, ts_log1p_pos((double)threadIdx.x * (1.0 / FZ_LOGTAB)));
  }
  __syncthreads();
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = fz_log1p_tab(x[i], tab);
}
/* the table-driven log1p of the FUSED lnl pass (fz_log1p_tab) on caller-supplied x >= 0 (accuracy test hook) */
int tsem_debug_log1p_tab(int device, int32_t n, const double* x, double* y) {
  if (!x || !y || n < 0) return TSEM_ERR_ARG;
  if (hipSetDevice(device) != hipSuccess) return TSEM_ERR_HIP;
  double *dx = nullptr, *dy = nullptr;
  if (hipMalloc((void**)&dx, 8 * (size_t)std::max(1, n)) != hipSuccess || hipMalloc((void**)&dy, 8 * (size_t)std::max(1, n)) != hipSuccess)
    return TSEM_ERR_NOMEM;
  (void)hipMemcpy(dx, x, 8 * (size_t)n, hipMemcpyHostToDevice);
  if (n) k_log1p_tab_probe<<<(n + 255) / 256, 256>>>(n, dx, dy);
  hipError_t e = hipMemcpy(y, dy, 8 * (size_t)n, hipMemcpyDeviceToHost);
  (void)hipFree(dx); (void)hipFree(dy);
  return e == hipSuccess ? TSEM_OK : TSEM_ERR_HIP;
}

/* debug: the packed (local row << 16 | local column) words of sub-block (block, part); returns their number */
int64_t tsem_debug_subblock(tsem_ctx* h, int64_t block, int32_t part, uint32_t* out, int64_t cap) {
  if (!h || !h->d_prc || !h->d_sb_off || block < 0 || block >= h->nb || part < 0 || part >= h->P) return TSEM_ERR_ARG;
  int64_t o[2];
  if (hipMemcpy(o, h->d_sb_off + block * h->P + part, 16, hipMemcpyDeviceToHost) != hipSuccess) return TSEM_ERR_HIP;
  const int64_t n = std::min(cap, o[1] - o[0]);
  if (n > 0 && hipMemcpy(out, h->d_prc + o[0], sizeof(uint32_t) * n, hipMemcpyDeviceToHost) != hipSuccess) return TSEM_ERR_HIP;
  return n;
}

int tsem_layout_info(tsem_ctx* h, int64_t* info) {
  if (!h || !info) return TSEM_ERR_ARG;
  info[0] = h->P; info[1] = h->Kp; info[2] = h->R; info[3] = h->nb;
  info[4] = h->N_amb; info[5] = h->N_uni; info[6] = h->nnz_amb; info[7] = h->nnz_pad;
  info[8] = h->n_twin_cols; info[9] = h->G1; info[10] = h->G2; info[11] = h->use_fused ? 1 : 0;
  info[12] = h->last_slow_path; info[13] = h->max_subblock; info[14] = h->fmt_code ? 2 : 8; info[15] = h->n_hot_cols;
  info[16] = h->use_fused ? (int64_t)fz_lds_bytes(h, fz_fmt(h) != 0) : 0;   // dynamic LDS per workgroup of the fused kernel
  info[17] = h->sorted_layout ? 1 : 0; info[18] = h->geo; info[19] = h->n_fallbacks;
  info[20] = h->n_bin_repeats; info[21] = h->opt_reproducible ? (h->len_gt[5] ? 2 : 1) : 0;     // 2: some row has more than 256 entries, see telescope_em.h
  info[22] = h->exact_single ? 1 : 0;                      // reproducible: both pieces in one pass
  info[23] = 0;
  return TSEM_OK;
}

}  // extern "C"
