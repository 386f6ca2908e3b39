// libtelescope_em.so, set-up unit: row statistics (Y, w, W_tot, W_amb, pisum0: model.py:679-699), column signatures (exact twins),
// the column partition and the blocked layout of the ambiguous rows (tsem_build_layout), model and parameter set-up.
#include "tsem_internal.h"
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

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
    // the row half of the hash: a full 64-bit mix once per lane and row; the entry half (round 5) is a 32-bit multiply-xorshift of the
    // score — ~6 instead of ~20 instructions per entry: the kernel is bound by its arithmetic, and all the signature has to do is tell
    // columns with EQUAL counts apart (a filter: k_update still ties two columns only while their sums agree to 1e-12)
    const uint32_t hrow = (uint32_t)(ts_mix64(0x7715ull ^ ((uint64_t)(row_offset + i) * TS_GOLDEN)) >> 32);
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
          const uint32_t r = (j & 1) ? w >> 16 : w & 0xFFFFu;
          uint32_t hv = hrow ^ (r * 0x9E3779B1u);
          hv ^= hv >> 15; hv *= 0x85EBCA77u; hv ^= hv >> 13;
          atomicAdd(&hh[c], ((unsigned long long)hv << 32) | 1ull);
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
  // (round 5) the table holds what the loop needs of a column — part << 24 | popularity id — instead of the column-map word: no
  // multiply per entry; and a step counts its (at most 16) entries per part in 8-bit fields of ONE word, widened once per step
  auto pack = [&](uint32_t cmw) -> uint32_t { const uint32_t p = cmw >> CM_PS; return (p << 24) | ((cmw & CM_SM) * (uint32_t)P + p); };
  if (LM) {
    for (int t = threadIdx.x; t < K; t += blockDim.x) pc_cm[t] = pack(colmap_g[t]);
    __syncthreads();
  }
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
      for (int j = 0; j < E; ++j) { const uint32_t c = k0 + j < len ? ix[j / 4][j & 3] : 0u; cm[j] = LM ? pc_cm[c] : pack(colmap_g[c]); }
      uint32_t idv[E];
      unsigned long long c8 = 0;                           // 8 x 8-bit counters of this step
#pragma unroll
      for (int j = 0; j < E; ++j) {
        idv[j] = cm[j] & 0xFFFFFFu;
        if (k0 + j < len) c8 += 1ull << (8 * (cm[j] >> 24));
      }
      {
        const uint32_t a4 = (uint32_t)c8, b4 = (uint32_t)(c8 >> 32);
        lo += (unsigned long long)(a4 & 0xFFu) | ((unsigned long long)((a4 >> 8) & 0xFFu) << 16) | ((unsigned long long)((a4 >> 16) & 0xFFu) << 32) | ((unsigned long long)(a4 >> 24) << 48);
        hi += (unsigned long long)(b4 & 0xFFu) | ((unsigned long long)((b4 >> 8) & 0xFFu) << 16) | ((unsigned long long)((b4 >> 16) & 0xFFu) << 32) | ((unsigned long long)(b4 >> 24) << 48);
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
// sub-block counts -> padded sizes (in place), with the largest count and the sum of the counts on the way
__global__ __launch_bounds__(256) void k_sb_pad(int64_t n, int64_t* __restrict__ cnt, unsigned long long* __restrict__ st /* [0] max, [1] sum */) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const long long c = i < n ? cnt[i] : 0;
  if (i < n) cnt[i] = (c + (TS_STRANDS * 4 - 1)) / (TS_STRANDS * 4) * (TS_STRANDS * 4);
  long long m = c, sm = c;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { m = max(m, __shfl_xor(m, o, 64)); sm += __shfl_xor(sm, o, 64); }
  if ((threadIdx.x & 63) == 0 && sm) { atomicMax(&st[0], (unsigned long long)m); atomicAdd(&st[1], (unsigned long long)sm); }
}
// entry offsets -> quad offsets of the fused kernel (n offsets + one trailing zero)
__global__ void k_sb_q32(int64_t n, const int64_t* __restrict__ off, uint32_t* __restrict__ q32) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= n) q32[i] = i < n ? (uint32_t)(off[i] >> 2) : 0u;
}
// layout statistic: ambiguous rows whose entries all fall into ONE column part (such a row's normaliser needs no exchange)
__global__ __launch_bounds__(256) void k_single_part_rows(int64_t na, const unsigned long long* __restrict__ pc, unsigned long long* __restrict__ out) {
  unsigned n = 0;
  for (int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; a < na; a += (int64_t)gridDim.x * blockDim.x) {
    const unsigned long long lo = pc[2 * a], hi = pc[2 * a + 1];
    int parts = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) parts += (((q < 4 ? lo : hi) >> (16 * (q & 3))) & 0xFFFF) != 0;
    n += parts == 1;
  }
  const int t = sg_sum_i<64>((int)n);
  if ((threadIdx.x & 63) == 0 && t) atomicAdd(out, (unsigned long long)t);
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
    for (int64_t k = s + lane; k < e; k += RS_SUB) atomicAdd(&cnt[colmap[indices[k]] >> CM_PS], 1u);
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
      uint32_t p = cm >> CM_PS;
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
      prc[pos] = ((uint32_t)lr << 16) | ((cm & CM_SM) + (t & ((1u << ((cm >> CM_LS) & 7u)) - 1u)));   // hot column: deal over its slots
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
      for (int k = lane; k < rlen[lr]; k += RS_SUB) atomicAdd(&cnt[lr * P + (colmap[indices[s0 + k]] >> CM_PS)], 1u);
    }
  }
  __syncthreads();
  {                                                        // exclusive scan down the rows: one WAVE per part, 64 rows per step (round 5: one
    const int wave = threadIdx.x >> 6, ln = threadIdx.x & 63, nw = blockDim.x >> 6;   // thread per part walked the R rows serially)
    for (int q = wave; q < P; q += nw) {
      uint32_t run = 0, last = 0;
      for (int base = 0; base < R; base += 64) {
        const int lr = base + ln;
        const uint32_t c = lr < R ? cnt[lr * P + q] : 0u;
        uint32_t incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(incl, d, 64); if (ln >= d) incl += t; }
        if (lr < R) cnt[lr * P + q] = run + incl - c;
        const unsigned long long m = __ballot(c != 0u);
        if (m) last = (uint32_t)(base + 63 - __clzll((long long)m));
        run += __shfl(incl, 63, 64);
      }
      if (ln == 0) { total[q] = run; lastrow[q] = last; }
    }
  }
  __syncthreads();
  // A row's entries keep their CSR order inside each part: the position of an entry = the row's cursor for its part
  // (read-only after the scan) + the number of earlier entries of the row in that part, counted with wave ballots over
  // the 16 lanes that walk the row — no LDS atomic per entry (2e9 of them with a return value were half of this
  // kernel's time), and the layout is the same from run to run.
  // (later in round 5) ... counted with a prefix sum over the row's 16 lanes instead of one ballot per part: every lane contributes a
  // one in the 8-bit field of its part (parts 0-3 in one word, 4-7 in a second), four DPP row shifts give the inclusive scan, lane 15's
  // value is the step's total per part; the row's running counts sit in 16-bit fields of two 64-bit words (a row has fewer than
  // 65 536 entries).  ~20 VALU instructions per step where the P ballots took ~10 each (the kernel's static code shrank by a
  // quarter); measured 8.1 -> 7.9 ms at 2e9 entries: the 3.4e9 VALU wave-instructions of profiles/r05_setup_pmc.txt were not
  // what it waits for either — its 2- and 4-byte scattered stores (7.5e8 L2 write requests) are.  Same positions as before.
  auto scan16 = [](uint32_t x) -> uint32_t {               // inclusive scan over a DPP row (= the 16 lanes of a row group)
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false);
    return x;
  };
  auto widen = [](uint32_t x) -> unsigned long long {      // 4 x 8-bit fields -> 4 x 16-bit fields
    const uint32_t lo = (x & 0xFFu) | ((x & 0xFF00u) << 8), hi = ((x >> 16) & 0xFFu) | ((x >> 8) & 0xFF0000u);
    return ((unsigned long long)hi << 32) | lo;
  };
  // one step of 16 entries: column-map word cm of this lane's entry (anything for lanes past the row's end), its score
  auto place = [&](int lr, unsigned long long (&run)[2], bool valid, uint32_t cm, uint32_t code) {
    const uint32_t p = cm >> CM_PS, sh8 = (p & 3u) * 8u;
    const bool upper = p >= 4u;
    const uint32_t w0 = scan16(valid && !upper ? 1u << sh8 : 0u);
    uint32_t inc = w0, t = (uint32_t)(run[0] >> ((p & 3u) * 16u)) & 0xFFFFu;
    run[0] += widen((uint32_t)__builtin_amdgcn_update_dpp(0, (int)w0, 0x15F, 0xF, 0xF, false));   // (row_newbcast:15: the row's total)
    if (P > 4) {
      const uint32_t w1 = scan16(valid && upper ? 1u << sh8 : 0u);
      if (upper) { inc = w1; t = (uint32_t)(run[1] >> ((p & 3u) * 16u)) & 0xFFFFu; }
      run[1] += widen((uint32_t)__builtin_amdgcn_update_dpp(0, (int)w1, 0x15F, 0xF, 0xF, false));
    }
    if (valid) {
      t += ((inc >> sh8) & 0xFFu) - 1u;                    // earlier entries of this step in the same part (inclusive count - itself)
      t += cnt[lr * P + p];
      const int64_t pos = sbase[p] + t;
      if (pcode) pcode[pos] = (uint16_t)code;
      else pval[pos] = lut[code];
      prc[pos] = ((uint32_t)lr << 16) | ((cm & CM_SM) + (t & ((1u << ((cm >> CM_LS) & 7u)) - 1u)));   // hot column: deal over its slots
    }
  };
  auto cm_of_id = [&](uint32_t id) -> uint32_t {           // the ids k_row_partcounts wrote: a coalesced 2-byte read instead of a gather
    const uint32_t slot = P == 1 ? id : __umulhi(id, magicP);   // id / P, exact for 16-bit ids and 2 <= P <= 8 (ceil(2^32 / 1) does not fit 32 bits)
    const uint32_t lg = (int)id < nsplit ? (uint32_t)lgtab[id] : 0u;
    return ((id - slot * (uint32_t)P) << CM_PS) | (lg << CM_LS) | slot;
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
      unsigned long long run[2] = {0ull, 0ull};
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
      unsigned long long run[2] = {0ull, 0ull};
      for (int k0 = 0; k0 < len; k0 += RS_SUB) {           // (all 16 lanes stay in the loop: the ballots need them)
        const bool valid = k0 + lane < len;
        place(lr, run, valid, valid ? colmap[indices[s0 + k0 + lane]] : 0u, valid ? (uint32_t)raw[s0 + k0 + lane] : 0u);
      }
    }
  }
  for (int p = 0; p < P; ++p) {                            // padding: value 0 (written here: the buffers of this path are not zero-filled), row = last row
    const int64_t base = sbase[p], end = sb_off[b * P + p + 1];
    for (int64_t pos = base + total[p] + threadIdx.x; pos < end; pos += blockDim.x) {
      prc[pos] = lastrow[p] << 16;
      if (pcode) pcode[pos] = (uint16_t)0; else pval[pos] = 0.0;
    }
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
// Round 5: IN PLACE (a thread reads its whole window into registers before it writes it back, and windows are disjoint): no second
// copy of the layout — 6 B per entry less at the peak of a build, two GB-sized hipMalloc / hipFree pairs less.
// ... and (later in round 5) with COOPERATIVE loads and stores: a thread used to read and write its own window with 16-byte accesses 256 B
// (128 B) apart from its neighbour's — 64 cache lines per wave instruction, and the texture-address unit takes a line per clock: ~30 of
// the 52 us a window took per thread went there (TCP_TOTAL_CACHE_ACCESSES / TCP_TCC_WRITE_REQ 1.0e9, profiles/r05_setup_pmc.txt).  Now
// a WAVE owns 64 consecutive windows; it loads them with contiguous 16-byte pieces (8 lines per instruction), passes them through an LDS
// tile (windows x 32 words, row stride 36 words: both the piece-wise and the row-wise b128 accesses are conflict-free per 16 lanes) so
// that lane l ends up with window l in registers, and stores them back the same way.  The walk itself is unchanged: same layout bit for bit.
constexpr int DC_S = 36;                                   // LDS row stride in words (32 + 4)
__global__ __launch_bounds__(DC_NT) void k_sb_deconflict(int64_t n_win, const uint32_t* prc_in, const uint16_t* code_in,
                                                          uint32_t* prc_out, uint16_t* code_out) {
  __shared__ __attribute__((aligned(16))) uint32_t dc_tile[DC_NT / 64][64 * DC_S];
  uint32_t* const tile = dc_tile[threadIdx.x >> 6];
  const int ln = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * DC_NT + threadIdx.x) >> 6, nwave = ((int64_t)gridDim.x * DC_NT) >> 6;
  // a chunk = 32 words of each of the wave's 64 windows: 128 B per window, `wstride` words from one window's chunk to the next;
  // piece t = q * 64 + lane of a chunk is the 16 bytes at word 4 * (t % 8) of window t / 8
  // (addresses: a wave-uniform base + one 32-bit lane offset + a constant per piece — 64-bit addresses per piece, hoisted out of the
  //  window loop, were 96 registers)
  const uint32_t lo64 = (uint32_t)(ln >> 3) * 64u + 4u * (uint32_t)(ln & 7), lo32 = (uint32_t)(ln >> 3) * 32u + 4u * (uint32_t)(ln & 7);
  auto chunk_in = [&](const uint32_t* src, uint32_t wstride, int64_t wb, uint32_t (&out)[32]) {
    const uint32_t* const bp = src + wb * (int64_t)wstride;  // wave-uniform
    const uint32_t lo = wstride == 64u ? lo64 : lo32;
    const int64_t left = n_win - wb;                          // windows of this wave that exist (>= 1)
    uint4 v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q)
      v[q] = q * 8 + (ln >> 3) < left ? *reinterpret_cast<const uint4*>(bp + (lo + (uint32_t)q * 8u * wstride)) : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int t = q * 64 + ln;
      *reinterpret_cast<uint4*>(tile + (t >> 3) * DC_S + 4 * (t & 7)) = v[q];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint4 r = *reinterpret_cast<const uint4*>(tile + ln * DC_S + 4 * j);
      out[4 * j] = r.x; out[4 * j + 1] = r.y; out[4 * j + 2] = r.z; out[4 * j + 3] = r.w;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  auto chunk_out = [&](uint32_t* dst, uint32_t wstride, int64_t wb, const uint32_t (&in)[32]) {
    uint32_t* const bp = dst + wb * (int64_t)wstride;
    const uint32_t lo = wstride == 64u ? lo64 : lo32;
    const int64_t left = n_win - wb;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      *reinterpret_cast<uint4*>(tile + ln * DC_S + 4 * j) = make_uint4(in[4 * j], in[4 * j + 1], in[4 * j + 2], in[4 * j + 3]);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int t = q * 64 + ln;
      const uint4 r = *reinterpret_cast<const uint4*>(tile + (t >> 3) * DC_S + 4 * (t & 7));
      if (q * 8 + (ln >> 3) < left) *reinterpret_cast<uint4*>(bp + (lo + (uint32_t)q * 8u * wstride)) = r;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  for (int64_t wb = wave * 64; wb < n_win; wb += nwave * 64) {
    uint32_t P[64], Cd[64];
    {
      uint32_t h0[32], h1[32], cw[32];
      chunk_in(prc_in, 64, wb, h0);                        // (one chunk after the other: all 24 loads in flight at once would cost 96 registers)
      __builtin_amdgcn_sched_barrier(0);
      chunk_in(prc_in + 32, 64, wb, h1);
      __builtin_amdgcn_sched_barrier(0);
      chunk_in(reinterpret_cast<const uint32_t*>(code_in), 32, wb, cw);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 32; ++i) { P[i] = h0[i]; P[32 + i] = h1[i]; Cd[2 * i] = cw[i] & 0xFFFFu; Cd[2 * i + 1] = cw[i] >> 16; }
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
    {
      uint32_t h0[32], h1[32], cw[32];
#pragma unroll
      for (int i = 0; i < 32; ++i) { h0[i] = P[i]; h1[i] = P[32 + i]; cw[i] = Cd[2 * i] | (Cd[2 * i + 1] << 16); }
      chunk_out(prc_out, 64, wb, h0);
      __builtin_amdgcn_sched_barrier(0);
      chunk_out(prc_out + 32, 64, wb, h1);
      __builtin_amdgcn_sched_barrier(0);
      chunk_out(reinterpret_cast<uint32_t*>(code_out), 32, wb, cw);
    }
  }
}

__global__ void k_make_ctab(int K, const double* __restrict__ pi, const double* __restrict__ theta,
                            const uint32_t* __restrict__ colmap, int Kp, double* __restrict__ ctab) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= K) return;
  const uint32_t cm = colmap[j];
  const int pc = (int)(cm >> CM_PS) * Kp + (int)(cm & CM_SM), copies = 1 << ((cm >> CM_LS) & 7u);
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


// popularity ids: id = slot * P + part of the column's first slot in the blocked layout (popular columns come first in
// every part, so small ids are popular columns); cnat2 by id
__global__ __launch_bounds__(256) void k_rid16_rows(int64_t N, const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
    const uint32_t* __restrict__ colmap, int P, int only_short, uint16_t* __restrict__ rid) {
  const int sub = threadIdx.x / RS_SUB, lane = threadIdx.x % RS_SUB, subs = blockDim.x / RS_SUB;
  for (int64_t i = (int64_t)blockIdx.x * subs + sub; i < N; i += (int64_t)gridDim.x * subs) {
    const int64_t s = indptr[i], e = indptr[i + 1];
    if (only_short && e - s > 1) continue;                 // (the ambiguous rows were written by k_row_partcounts)
    for (int64_t k = s + lane; k < e; k += RS_SUB) { const uint32_t cm = colmap[indices[k]]; rid[k] = (uint16_t)((cm & CM_SM) * P + (cm >> CM_PS)); }
  }
}

// the CSR column ids back from the popularity ids (option "drop_csr_indices"): eight entries per thread, 16-byte loads of the ids
__global__ __launch_bounds__(256) void k_indices_from_rid(int64_t nnz, const uint16_t* __restrict__ rid, const int32_t* __restrict__ col_of_id,
                                                          int32_t* __restrict__ indices) {
  for (int64_t k0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; k0 < nnz; k0 += (int64_t)gridDim.x * blockDim.x * 8) {
    if (k0 + 8 <= nnz) {
      const cs_u32x4_a2 w = *reinterpret_cast<const cs_u32x4_a2*>(rid + k0);
      int32_t c[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) c[j] = col_of_id[(j & 1) ? w[j / 2] >> 16 : w[j / 2] & 0xFFFFu];
      *reinterpret_cast<cs_u32x4_a4*>(indices + k0) = cs_u32x4_a4{(unsigned)c[0], (unsigned)c[1], (unsigned)c[2], (unsigned)c[3]};
      *reinterpret_cast<cs_u32x4_a4*>(indices + k0 + 4) = cs_u32x4_a4{(unsigned)c[4], (unsigned)c[5], (unsigned)c[6], (unsigned)c[7]};
    } else {
      for (int64_t k = k0; k < nnz; ++k) indices[k] = col_of_id[rid[k]];
    }
  }
}

extern "C" {

// HIP loads a translation unit's code object at the first launch (or attribute query) of one of its kernels: 5.8 ms for this unit in
// the first engine of a process, which used to sit at the head of tsem_rowstats.  The matrix loaders call this right after their last
// kernel goes out (k_check_csr / k_gen_rows), so the load happens while the device is busy with work that has to be done anyway.
void tsem_setup_preload(void) {
  hipFuncAttributes a;
  (void)hipFuncGetAttributes(&a, (const void*)k_class_flags);
}

int tsem_ensure_indices(tsem_ctx* h) {
  if (h->d_indices || !h->d_indptr || h->nnz == 0) return TSEM_OK;
  if (!h->d_rid16 || !h->d_col_of_id) TSEM_FAIL(TSEM_ERR_ARG, "the CSR column ids were dropped and there are no popularity ids to rebuild them from");
  TSEM_ALLOC(h->d_indices, h->nnz + TS_ENTRY_PAD);
  TSEM_HIP(hipMemsetAsync(h->d_indices + h->nnz, 0, sizeof(int32_t) * TS_ENTRY_PAD, h->stream));
  k_indices_from_rid<<<(unsigned)std::min<int64_t>((int64_t)h->n_cu * 32, (h->nnz / 8 + 255) / 256 + 1), 256, 0, h->stream>>>(
      h->nnz, h->d_rid16, h->d_col_of_id, h->d_indices);
  TSEM_HIP(hipGetLastError());
  return TSEM_OK;
}

// option "reproducible": the slots' bounds as a run finds them: 2^E > the largest fragment weight >= every contribution w * z
// (refined column by column, see k_bin_check).  Called when parameters are set from outside, so that a run's bits depend on its
// starting point only, not on what the context computed before.
int tsem_bin_reset(tsem_ctx* h) {
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



// option "use_likelihood": can a part keep three tables (pi*theta of the current and of the previous parameters, the accumulators)
// in LDS with at most 8 parts?  Score codes only (fp64 entries leave the log1p no registers), not together with the exact sums.
static bool lnl3_possible(const tsem_ctx* h) {
  const int K = h->K;
  return !h->opt_reproducible && h->em_kernel != TSEM_EMK_TWOPASS && fz_wants_codes(h) && h->opt_precision == 0 &&
         (h->opt_P > 0 ? (K + (int)h->opt_P - 1) / (int)h->opt_P + 64 <= TS_MAX_KP_LNL : (K + TS_MAX_KP_LNL - 64 - 1) / (TS_MAX_KP_LNL - 64) <= FZ_MAX_P);
}

int tsem_choose_geometry(tsem_ctx* h) {
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
    // option "use_likelihood": three tables per part too (pi*theta of the current and of the previous parameters, the accumulators) —
    // score codes only (fp64 entries leave the log1p no registers), not together with the exact sums (four tables)
    h->lnl3 = h->opt_lnl_fused && lnl3_possible(h);
    const int max_kp = h->exact_single ? TS_MAX_KP3 - 64 : (h->lnl3 ? TS_MAX_KP_LNL - 64 : TS_MAX_KP);
    int P = h->opt_P > 0 ? (int)h->opt_P : (K + max_kp - 1) / max_kp;
    if (P < 1) P = 1;
    // SPLIT layout (round 4): more columns than 8 parts of 7680 hold.  A part's pi*theta table and its accumulators then live in LDS
    // one at a time — a row-sum pass and a scatter pass per iteration, every entry read twice — with parts of up to 15 360 columns:
    // K <= 122 880 stays on the fused kernel (the two-pass kernels beyond; they took 7x the time per entry at K = 100k).  Teams of
    // 5-8 only (the instantiations that exist); option "split" = 1 forces it on a smaller matrix (tests), 0 forbids it.
    constexpr int SPLIT_MAX_KP = 2 * TS_MAX_KP;
    h->split = false;
    if (h->em_kernel != TSEM_EMK_TWOPASS && !h->opt_reproducible && h->opt_precision == 0 && na > 0 && h->opt_split != 0) {
      const int p2 = std::max(5, (K + SPLIT_MAX_KP - 1) / SPLIT_MAX_KP);       // (a part that needs every slot has no spare ones for hot columns)
      if (h->opt_P > 0) {
        const int kp = (K + P - 1) / P;
        h->split = P >= 5 && P <= FZ_MAX_P && ((kp > TS_MAX_KP && kp <= SPLIT_MAX_KP) || h->opt_split == 1);
      } else if ((P > FZ_MAX_P || h->opt_split == 1) && p2 <= FZ_MAX_P) {
        P = std::max(p2, std::min(P, FZ_MAX_P));
        h->split = true;
      }
      if (h->split) { h->lnl3 = false; h->exact_single = false; }
    }
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
    // more than 64 column parts (K > 491 520): no blocked layout — the reference takes any number of loci (model.py:643), so the
    // EM pass and the log-likelihood fall back to plain CSR row passes (global gathers of pi*theta, fp64 atomics on the column
    // sums: k_em_rows / k_lnl_rows_amb).  The column map is the identity cut into virtual parts, which is all k_update needs.
    h->em_rows = P > 64 && h->opt_P <= 0;
    if (P > 64 && !h->em_rows) TSEM_FAIL(TSEM_ERR_ARG, "more than 64 column parts (K > 491520) is not supported");
    if (h->em_rows) { h->split = false; h->lnl3 = false; h->exact_single = false; }
    if (h->em_rows && h->opt_reproducible) TSEM_FAIL(TSEM_ERR_ARG, "reproducible mode needs the fused kernel (K <= 61440)");
    int Kp = h->em_rows ? TS_MAX_KP : (K + P - 1) / P;
    if (Kp > (h->split ? SPLIT_MAX_KP : TS_MAX_KP)) TSEM_FAIL(TSEM_ERR_ARG, "parts option leaves more than 7680 columns per part");
    // spare accumulator slots per part for very popular columns (build_layout splits them)
    h->hot_extra = (h->opt_hot_split && !h->em_rows) ? std::min(64, (h->split ? SPLIT_MAX_KP : (h->exact_single ? TS_MAX_KP3 : (h->lnl3 ? TS_MAX_KP_LNL : TS_MAX_KP))) - Kp) : 0;
    Kp += h->hot_extra;
    h->P = P; h->Kp = Kp; h->Kpad = P * Kp;
    h->use_fused = (h->em_kernel != TSEM_EMK_TWOPASS) && P <= FZ_MAX_P && !h->em_rows;   // AUTO: fused when the layout allows it
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
      if (P <= 4 && h->geo == 2 && !h->lnl3 && 1.07 * fz_cap(2) * P / std::max(2.0, mean_len) > 1.4 * fz_rmax(2)) h->geo = 3;   // (MODE 4 has no geometry 3: registers)
      if (h->opt_geo >= 0 && P <= 4) h->geo = (h->opt_geo == 2 || (h->opt_geo == 3 && !h->lnl3)) ? (int)h->opt_geo : 0;
      if (h->opt_geo >= 0 && P > 4) h->geo = h->opt_geo == 2 ? 2 : 1;
      double r = 1.07 * fz_cap(h->geo) * P / std::max(2.0, mean_len);
      const int lut_bytes = (h->lut_len > 0 && h->lut_len <= 2048) ? h->lut_len * 8 : 0;   // the score table shares LDS with the rings
      int rmax = std::min(fz_rmax(h->geo), (TS_LDS_MAX - 2560 - (h->split ? Kp + 2 : ((h->exact_single || h->lnl3) ? 3 : 2) * Kp) * 8 - lut_bytes - std::max(h->opt_reproducible ? Kp * 2 + 16 : 0, FZ_LOGTAB * 16 + 16)) / ((fz_yr(h->geo) + (h->lnl3 ? 4 : 2)) * 8));
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
  if (int rc = tsem_ensure_indices(h)) return rc;
  const int64_t N = h->N;
  const int K = h->K;
  PhaseTimer pt(h->stream);
  DevTmp t_wpart, t_fa, t_fu, t_pis, t_lg;                   // freed on every return path
  TSEM_ALLOC(h->d_row_code, N); TSEM_ALLOC(h->d_row_cls, N);   // (kept: 3 B per row; tsem_export_rowinfo)
  uint16_t* const d_code = h->d_row_code; uint8_t* const d_cls = h->d_row_cls;
  const int grid = (int)std::min<int64_t>(4096, std::max<int64_t>(1, (N + 15) / 16));
  TSEM_TMP(t_wpart, sizeof(double) * 2 * grid);
  double* const d_wpart = t_wpart.as<double>();
  TSEM_ALLOC(h->d_pisum0, K);
  TSEM_ALLOC(h->d_ucount, K + 1);
  TSEM_HIP(hipMemsetAsync(h->d_ucount, 0, sizeof(uint32_t) * (K + 1), h->stream));
  TSEM_TMP(t_pis, sizeof(double) * PIS_LEVELS * K);
  double* const d_pis_lv = t_pis.as<double>();
  TSEM_HIP(hipMemsetAsync(d_pis_lv, 0, sizeof(double) * PIS_LEVELS * K, h->stream));
  int pis_e2 = 0;
  (void)std::frexp(h->lut_host[h->lut_len - 1] > 0 ? h->lut_host[h->lut_len - 1] : 1.0, &pis_e2);   // Q < 2^e2 (the table is increasing)
  TSEM_HIP(hipMemsetAsync(h->d_maxcode, 0, 4, h->stream));
  TSEM_HIP(hipMemsetAsync(d_wpart, 0, sizeof(double) * 2 * grid, h->stream));
  TSEM_TMP(t_lg, 64);
  unsigned long long* const d_lg = t_lg.as<unsigned long long>();
  TSEM_HIP(hipMemsetAsync(d_lg, 0, 64, h->stream));
  pt.lap("rowstats: allocations");
  if (N) {
    const double mean_len = (double)h->nnz / (double)N;    // lanes per row x 16 entries >= ~1.5 mean row lengths
    const int G = mean_len * 1.5 <= 16 ? 1 : mean_len * 1.5 <= 32 ? 2 : mean_len * 1.5 <= 64 ? 4 : mean_len * 1.5 <= 128 ? 8 : 16;
    auto rk = G == 1 ? k_rowstats<1> : G == 2 ? k_rowstats<2> : G == 4 ? k_rowstats<4> : G == 8 ? k_rowstats<8> : k_rowstats<16>;
    rk<<<grid, 256, 0, h->stream>>>(N, h->d_indptr, h->d_indices, h->d_raw, h->d_lut, d_code, d_cls,
                                    d_wpart, h->d_maxcode, d_pis_lv, pis_e2 + 1023, h->d_ucount, K, d_lg);
  }
  k_pisum_finish<<<cdiv64(K, 256), 256, 0, h->stream>>>(K, d_pis_lv, h->d_pisum0);
  TSEM_HIP(hipGetLastError());
  pt.lap("rowstats: k_rowstats kernel");
  TSEM_HIP(hipMemcpyAsync(h->len_gt, d_lg, 6 * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
  std::vector<double> wpart(2 * grid);
  uint32_t maxcode = 0;
  TSEM_HIP(hipMemcpyAsync(wpart.data(), d_wpart, sizeof(double) * 2 * grid, hipMemcpyDeviceToHost, h->stream));
  TSEM_HIP(hipMemcpyAsync(&maxcode, h->d_maxcode, 4, hipMemcpyDeviceToHost, h->stream));
  uint32_t has_zero = 0;                                     // some stored score is 0 (fixed from here on: kept on the host)
  TSEM_HIP(hipMemcpyAsync(&has_zero, h->d_ucount + K, 4, hipMemcpyDeviceToHost, h->stream));
  TSEM_HIP(hipStreamSynchronize(h->stream));
  h->has_zero_score = has_zero != 0;
  double wt = 0, wa = 0;
  for (int i = 0; i < grid; ++i) { wt += wpart[2 * i]; wa += wpart[2 * i + 1]; }
  if (stats3) { stats3[0] = wt; stats3[1] = wa; stats3[2] = (N && h->nnz) ? h->lut_host[maxcode] : 0.0; }
  if (pisum0) TSEM_HIP(hipMemcpy(pisum0, h->d_pisum0, sizeof(double) * K, hipMemcpyDeviceToHost));
  // (Round 5: the ~7 ms this section takes in the FIRST engine of a process — 0.08 ms in the second — are a one-off of the runtime's
  //  first sizeable host copy.  Routing the read-backs through a pinned buffer, or storing them into device-mapped host memory from a
  //  kernel, only moved the 7 ms to the next hipMemcpy, and with pinned memory registered the GB-sized hipMallocs of the layout took
  //  60-340 ms each in two of three runs: reverted.  gpurun_out logs r5_setup_trace3-5.)

  pt.lap("rowstats: k_rowstats + sums");
  // column signatures (popularity + twin detection)
  if (col_count && col_hash) {
    DevTmp t_hash;
    TSEM_TMP(t_hash, sizeof(unsigned long long) * K);
    unsigned long long* const d_hash = t_hash.as<unsigned long long>();
    TSEM_ALLOC(h->d_colcount, K);                          // LOCAL stored entries per column: reassign('all', initial) of this rank
    unsigned long long* const d_cnt = h->d_colcount;
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
  }
  pt.lap("rowstats: column signatures");
  // compact ambiguous and unique rows
  TSEM_TMP(t_fa, 4 * (size_t)(N + 1)); TSEM_TMP(t_fu, 4 * (size_t)(N + 1));
  int32_t* const d_fa = t_fa.as<int32_t>(); int32_t* const d_fu = t_fu.as<int32_t>();
  int32_t na = 0, nu = 0;
  TSEM_HIP(hipMemsetAsync(d_fa + N, 0, 4, h->stream));
  TSEM_HIP(hipMemsetAsync(d_fu + N, 0, 4, h->stream));
  if (N) {
    k_class_flags<<<cdiv64(N, 256), 256, 0, h->stream>>>(N, d_cls, d_fa, d_fu);
    size_t tb = 0;
    TSEM_HIP(rocprim::exclusive_scan(nullptr, tb, d_fa, d_fa, (int32_t)0, (size_t)(N + 1), rocprim::plus<int32_t>(), h->stream));
    DevTmp t_scan;
    TSEM_TMP(t_scan, tb);
    void* const tmp = t_scan.p;
    TSEM_HIP(rocprim::exclusive_scan(tmp, tb, d_fa, d_fa, (int32_t)0, (size_t)(N + 1), rocprim::plus<int32_t>(), h->stream));
    TSEM_HIP(rocprim::exclusive_scan(tmp, tb, d_fu, d_fu, (int32_t)0, (size_t)(N + 1), rocprim::plus<int32_t>(), h->stream));
    TSEM_HIP(hipMemcpyAsync(&na, d_fa + N, 4, hipMemcpyDeviceToHost, h->stream));
    TSEM_HIP(hipMemcpyAsync(&nu, d_fu + N, 4, hipMemcpyDeviceToHost, h->stream));
    TSEM_HIP(hipStreamSynchronize(h->stream));
  }
  pt.lap("rowstats: class flags + scans");
  h->N_amb = na; h->N_uni = nu;
  if (int rc = tsem_choose_geometry(h)) return rc;
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
  pt.lap("rowstats: compact rows");
  h->have_rowstats = true;
  return TSEM_OK;
}

/* Y (model.py:679: 1 where the row has several stored entries) and the row weights w_i = max_j Q_ij (model.py:690) of the
 * local rows, as tsem_rowstats left them on the device — the per-row inputs of every EM pass. */
int tsem_export_rowinfo(tsem_ctx* h, uint8_t* Y, double* weights) {
  if (!h || !h->have_rowstats || !h->d_row_code || !h->d_row_cls) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  const int64_t N = h->N;
  std::vector<uint8_t> cls((size_t)N);
  std::vector<uint16_t> code((size_t)N);
  if (N) {
    TSEM_HIP(hipMemcpy(cls.data(), h->d_row_cls, (size_t)N, hipMemcpyDeviceToHost));
    TSEM_HIP(hipMemcpy(code.data(), h->d_row_code, 2 * (size_t)N, hipMemcpyDeviceToHost));
  }
  for (int64_t i = 0; i < N; ++i) {
    if (Y) Y[i] = cls[i] == 2 ? 1 : 0;
    if (weights) weights[i] = cls[i] ? h->lut_host[code[i]] : 0.0;
  }
  return TSEM_OK;
}

// stable LSD radix sort of (key, idx) pairs by 64-bit key (11-bit digits; passes whose digit is constant are skipped): the two K-sized
// host sorts of a set-up (columns by popularity, column signatures for the twin search) took 1.3 ms each with std::stable_sort /
// std::sort at K = 30k — a fixed cost that is a quarter of the set-up of a 2M-row matrix
static void radix_sort_pairs(std::vector<uint64_t>& key, std::vector<int>& idx) {
  const size_t n = key.size();
  std::vector<uint64_t> k2(n);
  std::vector<int> i2(n);
  uint32_t hist[2048];
  for (int pass = 0; pass < 6; ++pass) {
    const int sh = 11 * pass;
    for (uint32_t& x : hist) x = 0u;
    for (size_t i = 0; i < n; ++i) hist[(key[i] >> sh) & 0x7FFu] += 1u;
    if (n && hist[(key[0] >> sh) & 0x7FFu] == n) continue;
    uint32_t run = 0;
    for (uint32_t& x : hist) { const uint32_t c = x; x = run; run += c; }
    for (size_t i = 0; i < n; ++i) { const uint32_t q = hist[(key[i] >> sh) & 0x7FFu]++; k2[q] = key[i]; i2[q] = idx[i]; }
    key.swap(k2); idx.swap(i2);
  }
}

// ---------------------------------------------------------------------------
// layout: column partition by popularity, blocked COO of ambiguous rows
// ---------------------------------------------------------------------------
int tsem_build_layout(tsem_ctx* h) {
  PhaseTimer pt(h->stream);
  const int K = h->K;
  const int64_t na = h->N_amb;
  if (int rc = tsem_ensure_indices(h)) return rc;          // (a rebuild after option "drop_csr_indices": from the ids of the layout about to go)
  TSEM_HIP(hipStreamSynchronize(h->stream));
  tsem_free_layout(h);
  h->nnz_amb = 0;
  h->n_single_part = 0;
  if (h->em_rows) {
    // identity column map in virtual parts of Kp columns: pc == column (k_update / k_make_ctab / k_colreduce index conventions hold)
    std::vector<uint32_t> cm((size_t)K);
    std::vector<int32_t> cpc((size_t)h->Kpad, -1);
    for (int j = 0; j < K; ++j) { cm[j] = ((uint32_t)(j / h->Kp) << CM_PS) | (uint32_t)(j % h->Kp); cpc[j] = j; }
    TSEM_ALLOC(h->d_colmap, K);
    TSEM_ALLOC(h->d_col_of_pc, h->Kpad);
    TSEM_HIP(hipMemcpy(h->d_colmap, cm.data(), sizeof(uint32_t) * K, hipMemcpyHostToDevice));
    TSEM_HIP(hipMemcpy(h->d_col_of_pc, cpc.data(), sizeof(int32_t) * h->Kpad, hipMemcpyHostToDevice));
    h->nb = 0; h->N_amb_pad = 1; h->nnz_pad = 0; h->max_subblock = 0; h->n_hot_cols = 0;
    h->nnz_amb = h->nnz - h->N_uni;
    h->fmt_code = h->fmt_wcode = false; h->sorted_layout = false;
    h->G1 = h->G2 = 1;
    return TSEM_OK;
  }
  // 1. column popularity: global entry counts handed in by set_model
  const std::vector<uint64_t>& counts = h->col_count;
  // 2. parts: deal columns by popularity so every part carries ~equal nnz
  const int P = h->P, Kp = h->Kp;
  std::vector<int> order(K);
  std::iota(order.begin(), order.end(), 0);
  {                                                        // most popular first, equal counts by column index (a stable sort by count, descending)
    std::vector<uint64_t> key((size_t)K);
    for (int j = 0; j < K; ++j) key[j] = ~counts[j];
    radix_sort_pairs(key, order);
  }
  // colmap[j] = part << CM_PS | log2(copies) << CM_LS | first slot (tsem_device.h).  A column that holds a large share of its
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
    colmap[j] = ((uint32_t)p << CM_PS) | ((uint32_t)lg << CM_LS) | (uint32_t)cursor[p];
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
    // the padding holds id 0: lanes past the LAST row's end index the pi*theta tables with what they find there (tsem_host.hip)
    TSEM_HIP(hipMemsetAsync(h->d_rid16 + h->nnz, 0, sizeof(uint16_t) * TS_ENTRY_PAD, h->stream));
  }
  pt.lap("layout: maps to the device, rid16 alloc");
  // 3. row blocks.  Two-pass layout: R rows each.  Fused layout: as many consecutive rows as the
  //    register tile takes (no part may exceed FZ_CAP entries, at most R rows) — rows per block vary,
  //    every block still owns R row SLOTS (holes at the end), so all kernels keep b*R+lr indexing.
  const int R = h->R;
  int64_t nb = 0;
  int64_t* d_bs = nullptr;                                 // first compact row of every block, [nb + 1]
  unsigned long long* d_pc = nullptr;                      // per-row part counts (fused layout only)
  TSEM_SCOPED(d_bs); TSEM_SCOPED(d_pc);                    // (temporaries of this function: freed on every return path)
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
    {
      DevTmp t_sp;
      TSEM_TMP(t_sp, 8);
      TSEM_HIP(hipMemsetAsync(t_sp.p, 0, 8, h->stream));
      k_single_part_rows<<<(unsigned)std::min<int64_t>(4096, (na + 255) / 256), 256, 0, h->stream>>>(na, d_pc, t_sp.as<unsigned long long>());
      unsigned long long nsp = 0;
      TSEM_HIP(hipMemcpyAsync(&nsp, t_sp.p, 8, hipMemcpyDeviceToHost, h->stream));
      TSEM_HIP(hipStreamSynchronize(h->stream));
      h->n_single_part = (int64_t)nsp;
    }
    const int cap = fz_cap(h->geo) - TS_STRANDS * 4;          // sub-blocks are padded to TS_STRANDS*4 entries
    // chunk length: >= 64 blocks' worth of rows (the forced break at a chunk end costs ~0.8 % more blocks; round 2 used 256 blocks' worth,
    // 484 sequential waves for 47M rows: 2 x 2.5 ms; four times as many waves walk a quarter each)
    const int64_t L = std::max<int64_t>((int64_t)R * 64, (na + 16383) / 16384);
    const int64_t nch = (na + L - 1) / L;
    int64_t *d_cnt = nullptr, *d_off = nullptr;
    int* d_flag = nullptr;
    TSEM_SCOPED(d_cnt); TSEM_SCOPED(d_off); TSEM_SCOPED(d_flag);
    TSEM_ALLOC(d_cnt, nch + 1); TSEM_ALLOC(d_off, nch + 1); TSEM_ALLOC(d_flag, 1);
    TSEM_HIP(hipMemsetAsync(d_flag, 0, sizeof(int), h->stream));
    TSEM_HIP(hipMemsetAsync(d_cnt, 0, sizeof(int64_t) * (nch + 1), h->stream));
    k_block_greedy<<<(unsigned)nch, 64, 0, h->stream>>>(na, P, R, cap, L, 0, d_pc, d_cnt, nullptr, nullptr, d_flag);
    TSEM_HIP(hipGetLastError());
    {
      size_t tb = 0;
      TSEM_HIP(rocprim::exclusive_scan(nullptr, tb, d_cnt, d_off, (int64_t)0, (size_t)(nch + 1), rocprim::plus<int64_t>(), h->stream));
      void* tmp = nullptr;
      TSEM_SCOPED(tmp);
      TSEM_HIP(hipMalloc(&tmp, tb ? tb : 1));
      TSEM_HIP(rocprim::exclusive_scan(tmp, tb, d_cnt, d_off, (int64_t)0, (size_t)(nch + 1), rocprim::plus<int64_t>(), h->stream));
      int flag = 0;
      TSEM_HIP(hipMemcpyAsync(&nb, d_off + nch, sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
      TSEM_HIP(hipMemcpyAsync(&flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, h->stream));
      TSEM_HIP(hipStreamSynchronize(h->stream));
      if (flag) { h->use_fused = false; nb = 0; }          // one row overflows the register tile
    }
    if (h->use_fused) {
      TSEM_ALLOC(d_bs, nb + 1);
      k_block_greedy<<<(unsigned)nch, 64, 0, h->stream>>>(na, P, R, cap, L, 1, d_pc, d_cnt, d_off, d_bs, d_flag);
      TSEM_HIP(hipGetLastError());
      TSEM_HIP(hipMemcpyAsync(d_bs + nb, &na, sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
      TSEM_HIP(hipStreamSynchronize(h->stream));
    }
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
  // 4. sub-block sizes -> offsets, on the device (round 5: the counts used to travel to the host and the offsets back, 12 MB of
  //    pageable copies and a host loop: 2.3 ms at 146k blocks): pad every count to TS_STRANDS*4 entries, exclusive scan, 32-bit quad
  //    offsets for the fused kernel; three scalars come back (largest sub-block, stored entries, padded total)
  int64_t off = 0;
  TSEM_ALLOC(h->d_sb_off, nb * P + 1);
  if (nb) {
    int64_t* d_cnt = nullptr;
    unsigned long long* d_st = nullptr;
    TSEM_SCOPED(d_cnt); TSEM_SCOPED(d_st);
    TSEM_ALLOC(d_cnt, nb * P + 1); TSEM_ALLOC(d_st, 2);
    TSEM_HIP(hipMemsetAsync(d_st, 0, 16, h->stream));
    TSEM_HIP(hipMemsetAsync(d_cnt + nb * P, 0, 8, h->stream));
    if (d_pc) k_sb_count_pc<<<(unsigned)nb, 64, 0, h->stream>>>(nb, P, d_bs, d_pc, d_cnt);
    else k_sb_count<<<(unsigned)nb, 256, 0, h->stream>>>(na, R, P, h->d_slot_row, h->d_indptr, h->d_indices, h->d_colmap, d_cnt);
    k_sb_pad<<<cdiv64(nb * P, 256), 256, 0, h->stream>>>(nb * P, d_cnt, d_st);
    TSEM_HIP(hipGetLastError());
    size_t tb = 0;
    TSEM_HIP(rocprim::exclusive_scan(nullptr, tb, d_cnt, h->d_sb_off, (int64_t)0, (size_t)(nb * P + 1), rocprim::plus<int64_t>(), h->stream));
    void* tmp = nullptr;
    TSEM_SCOPED(tmp);
    TSEM_HIP(hipMalloc(&tmp, tb ? tb : 1));
    TSEM_HIP(rocprim::exclusive_scan(tmp, tb, d_cnt, h->d_sb_off, (int64_t)0, (size_t)(nb * P + 1), rocprim::plus<int64_t>(), h->stream));
    unsigned long long st[2] = {0, 0};
    TSEM_HIP(hipMemcpyAsync(st, d_st, 16, hipMemcpyDeviceToHost, h->stream));
    TSEM_HIP(hipMemcpyAsync(&off, h->d_sb_off + nb * P, 8, hipMemcpyDeviceToHost, h->stream));
    TSEM_HIP(hipStreamSynchronize(h->stream));
    h->nnz_amb += (int64_t)st[1];
    if (h->use_fused) {
      h->max_subblock = (int64_t)st[0];
      if ((int64_t)st[0] > fz_cap(h->geo)) h->use_fused = false;
    }
  } else {
    TSEM_HIP(hipMemsetAsync(h->d_sb_off, 0, 8, h->stream));
    if (h->use_fused) h->max_subblock = 0;
  }
  h->nnz_pad = off;
  if (h->use_fused && (off >> 2) < 0xFFFFFFFFll) {
    TSEM_ALLOC(h->d_sb_q32, nb * P + 2);
    k_sb_q32<<<cdiv64(nb * P + 2, 256), 256, 0, h->stream>>>(nb * P + 1, h->d_sb_off, h->d_sb_q32);
    TSEM_HIP(hipGetLastError());
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
  // (zero-filled for the strand-transposed fill only, whose padding is whatever it does not write; the row-order fill writes
  //  every position, padding included — 12-24 GB of memsets, ~3 ms at 2e9 entries, went away with that)
  const bool will_sort = h->use_fused && R * P <= FILL_MAX_RP && (h->opt_sorted >= 0 ? h->opt_sorted != 0 : true) && nb > 0;
  TSEM_ALLOC(h->d_prc, off);
  if (!will_sort) TSEM_HIP(hipMemsetAsync(h->d_prc, 0, sizeof(uint32_t) * std::max<int64_t>(1, off), h->stream));
  if (h->fmt_code) {
    TSEM_ALLOC(h->d_pcode, off);
    if (!will_sort) TSEM_HIP(hipMemsetAsync(h->d_pcode, 0, sizeof(uint16_t) * std::max<int64_t>(1, off), h->stream));   // code 0 -> Q = 0
  } else {
    TSEM_ALLOC(h->d_pval, off);
    if (!will_sort) TSEM_HIP(hipMemsetAsync(h->d_pval, 0, sizeof(double) * std::max<int64_t>(1, off), h->stream));
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
  // (the small table of the fill lives until the one synchronisation behind the kernels; nothing GB-sized is allocated next to a
  //  running kernel: that took 60-300 ms in round 5's first attempt at overlapping this section)
  uint8_t* d_lgtab = nullptr;
  TSEM_SCOPED(d_lgtab);
  if (nb && h->sorted_layout) {
    // the popularity ids stand in for the column-map gather when every row's ids are written (they are: k_row_partcounts +
    // k_rid16_rows above) and the split columns' ids fit the small table
    const uint16_t* rid_fill = nullptr;
    int nsplit = 0;
    if (h->d_rid16 && P <= 8) {
      std::vector<uint8_t> lgt;
      for (int j = 0; j < K; ++j) {
        const uint32_t cm = colmap[j], lg = (cm >> CM_LS) & 7u;
        if (lg) { const uint32_t id = (cm & CM_SM) * P + (cm >> CM_PS); if (id >= lgt.size()) lgt.resize(id + 1, 0); lgt[id] = (uint8_t)lg; }
      }
      nsplit = (int)lgt.size();
      if (nsplit <= 4096) {
        TSEM_ALLOC(d_lgtab, std::max(1, nsplit));
        if (nsplit) TSEM_HIP(hipMemcpyAsync(d_lgtab, lgt.data(), nsplit, hipMemcpyHostToDevice, h->stream));
        TSEM_HIP(hipStreamSynchronize(h->stream));         // (lgt is a local)
        rid_fill = h->d_rid16;
      }
    }
    // (reproducible mode keeps the row order: a row's entries in a sub-block then form ONE run, which ends in at most two LDS
    // atomics on its row sum — two additions commute, three need not)
    const bool deconflict = h->fmt_code && h->opt_deconflict != 0 && !h->opt_reproducible && off >= 64;
    const uint32_t magicP = (uint32_t)((0x100000000ull + (uint64_t)P - 1) / (uint64_t)P);
    k_sb_fill_sorted<<<(unsigned)nb, 256, fill_lds_bytes(R, P), h->stream>>>(na, R, P, h->d_slot_row, h->d_indptr, h->d_indices, h->d_raw, h->d_lut,
                                                         h->d_colmap, h->d_sb_off, h->d_pval, h->d_pcode, h->d_prc,
                                                         d_pc ? d_bs : nullptr, d_pc, rid_fill, magicP, nsplit, d_lgtab);
    TSEM_HIP(hipGetLastError());
    if (deconflict) {
      const int64_t n_win = off / 64;
      k_sb_deconflict<<<(unsigned)std::min<int64_t>(n_win / DC_NT + 1, (int64_t)h->n_cu * 32), DC_NT, 0, h->stream>>>(
          n_win, h->d_prc, h->d_pcode, h->d_prc, h->d_pcode);
      TSEM_HIP(hipGetLastError());
    }
    // While the device fills the layout (15 ms at 2e9 entries) the host loads the code object of the fused kernel's unit — 3 ms in a
    // fresh process, at the first hipFuncSetAttribute / launch of one of its kernels — instead of doing so afterwards.
    if (h->use_fused) {
      fz_fn f0 = P <= FZ_MAX_P ? fz_kernel(P, h->split ? 5 : 0, fz_fmt(h), h->geo) : nullptr;
      if (f0) (void)hipFuncSetAttribute((const void*)f0, hipFuncAttributeMaxDynamicSharedMemorySize, TS_LDS_MAX - 1024);
    }
    tsem_report_preload();                                 // (round 6: the report unit's too — ~10 ms that used to sit in the first report)
    TSEM_HIP(hipStreamSynchronize(h->stream));
    // option "drop_csr_indices": the fill was the last reader of the CSR column ids (the report pass and this layout carry 2-byte
    // popularity ids; col = col_of_id[id]): 10 instead of 14 B per stored entry stay resident.
    if (rid_fill && h->d_col_of_id && (h->opt_drop_indices == 1 || (h->opt_drop_indices < 0 && h->nnz >= 4000000000ll))) dfree(h->d_indices);
  } else if (nb) {
    k_sb_fill<<<(unsigned)nb, 256, 0, h->stream>>>(na, R, P, h->d_slot_row, h->d_indptr, h->d_indices, h->d_raw, h->d_lut,
                                                  h->d_colmap, h->d_sb_off, h->d_pval, h->d_pcode, h->d_prc);
    TSEM_HIP(hipGetLastError());
  }
  TSEM_HIP(hipStreamSynchronize(h->stream));
  pt.lap("layout: fill + conflict-aware order");
  if (!h->use_fused) TSEM_ALLOC(h->d_ypart, (int64_t)P * h->N_amb_pad);   // partial row sums of the two-pass kernels
  // launch geometry
  const size_t lds1 = (size_t)(Kp + R) * 8, lds2 = (size_t)(2 * Kp + R) * 8;
  if (lds2 > (size_t)TS_LDS_MAX - 1024 && !h->use_fused) TSEM_FAIL(TSEM_ERR_ARG, "LDS budget exceeded (reduce block_rows)");   // (the two-pass kernels' tables; the split layout has its own budget)
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
      TSEM_HIP(hipMemsetAsync(h->d_xflags, 0, sizeof(uint32_t) * FZ_SYNC_WORDS, h->stream));
      if (!h->fmt_code && !h->fmt_wcode) {                 // fp64 row weights; otherwise the kernel reads d_amb_wcode
        TSEM_ALLOC(h->d_amb_w, h->N_amb_pad);
        k_row_weights<<<cdiv64(h->N_amb_pad, 256), 256, 0, h->stream>>>(h->N_amb_pad, h->d_amb_wcode, h->d_lut, h->d_amb_w);
      }
      if (h->split) {                                      // row-sum pass, scatter pass, log-likelihood over a column half
        for (int mode : {5, 7, 8}) {
          fz_fn f = fz_kernel(P, mode, fz_fmt(h), h->geo);
          if (!f) TSEM_FAIL(TSEM_ERR_ARG, "split layout: no fused kernel for this team size / geometry");
          TSEM_HIP(hipFuncSetAttribute((const void*)f, hipFuncAttributeMaxDynamicSharedMemorySize, TS_LDS_MAX - 1024));
        }
        TSEM_ALLOC(h->d_rinv, h->N_amb_pad);               // the row factors pass A hands to pass B
        TSEM_ALLOC(h->d_ypart, (int64_t)P * h->N_amb_pad); // the members' partial row sums (row-sum pass -> k_row_factors)
        TSEM_HIP(hipMemsetAsync(h->d_rinv, 0, sizeof(double) * h->N_amb_pad, h->stream));
      } else
      for (int mode = 0; mode < 2; ++mode)
        TSEM_HIP(hipFuncSetAttribute((const void*)fz_kernel(P, mode, fz_fmt(h), h->geo),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, TS_LDS_MAX - 1024));
      if (h->lnl3 && !(h->fmt_code && fz_kernel(P, 4, fz_fmt(h), h->geo))) h->lnl3 = false;   // (not score codes after all: the per-iteration lnl pass stays)
      h->lag_valid = false;
      if (h->lnl3) {
        TSEM_HIP(hipFuncSetAttribute((const void*)fz_kernel(P, 4, fz_fmt(h), h->geo), hipFuncAttributeMaxDynamicSharedMemorySize, TS_LDS_MAX - 1024));
        TSEM_ALLOC(h->d_rinv, h->N_amb_pad);
        TSEM_HIP(hipMemsetAsync(h->d_rinv, 0, sizeof(double) * h->N_amb_pad, h->stream));
      }
      if (h->opt_reproducible) {
        if (h->exact_single && !fz_kernel(P, 3, fz_fmt(h), h->geo)) h->exact_single = false;   // (not score codes after all: fz_lds_bytes then counts two tables again)
        if (h->exact_single) TSEM_ALLOC(h->d_fpartial2, (int64_t)h->fz_teams * h->Kpad);
        fz_fn f2 = fz_kernel(P, h->exact_single ? 3 : 2, fz_fmt(h), h->geo);
        if (!f2) TSEM_FAIL(TSEM_ERR_ARG, "reproducible mode needs the fused kernel with a score table of at most 2048 entries");
        TSEM_HIP(hipFuncSetAttribute((const void*)f2, hipFuncAttributeMaxDynamicSharedMemorySize, TS_LDS_MAX - 1024));
        TSEM_ALLOC(h->d_ebias, h->Kpad); TSEM_ALLOC(h->d_ovf, h->Kpad); TSEM_ALLOC(h->d_red_hi, K + 2); TSEM_ALLOC(h->d_binflag, 4);
        TSEM_ALLOC(h->d_ehist, 2 * (size_t)K + 2);
        if (int rc = tsem_bin_reset(h)) return rc;
      }
    }
  }
  if (int rc = tsem_twopass_attributes(h)) return rc;
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
    // order by (count, hash, column): a stable radix sort of the column indices by count, then the (short) runs of equal counts by
    // (hash, column)
    std::vector<int> ord(K);
    std::iota(ord.begin(), ord.end(), 0);
    {
      std::vector<uint64_t> key(col_count, col_count + K);
      radix_sort_pairs(key, ord);
      for (int i = 0; i < K;) {
        int j = i;
        while (j + 1 < K && key[j + 1] == key[i]) ++j;
        if (j > i)
          std::sort(ord.begin() + i, ord.begin() + j + 1, [&](int a, int b) { return col_hash[a] != col_hash[b] ? col_hash[a] < col_hash[b] : a < b; });
        i = j + 1;
      }
    }
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
  if (int rc = tsem_build_layout(h)) return rc;
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

int tsem_make_ctabs(tsem_ctx* h) {
  k_make_ctab<<<cdiv64(h->K, 256), 256, 0, h->stream>>>(h->K, h->d_pi, h->d_theta, h->d_colmap, h->Kp, h->d_ctab);
  k_make_ctab<<<cdiv64(h->K, 256), 256, 0, h->stream>>>(h->K, h->d_pi_prev, h->d_theta_prev, h->d_colmap, h->Kp, h->d_ctab_prev);
  TSEM_HIP(hipGetLastError());
  return TSEM_OK;
}

// em(use_likelihood=True) on a model that was laid out without option "use_likelihood": rebuild the blocked layout with three
// tables per part, so that the EM pass can sum the previous iteration's log-likelihood (fused kernel MODE 4).  A no-op when the layout
// has it already, or cannot have it (two-pass kernels, fp64 entries, option "reproducible", more than 8 x 5056 columns): the
// per-iteration lnl pass then stays.
int tsem_prepare_likelihood(tsem_ctx* h) {
  if (!h || !h->have_model) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  if (h->lnl3 || h->lnl3_declined || !h->use_fused || h->nb == 0 || !lnl3_possible(h)) return TSEM_OK;
  h->opt_lnl_fused = 1;
  TSEM_HIP(hipStreamSynchronize(h->stream));
  if (int rc = tsem_choose_geometry(h)) return rc;
  if (!h->lnl3) {                                          // the geometry declined it (e.g. the split layout): nothing to rebuild, now or later
    h->lnl3_declined = true;
    h->opt_lnl_fused = 0;
    return TSEM_OK;
  }
  if (int rc = tsem_build_layout(h)) return rc;
  TSEM_ALLOC(h->d_ctab, h->Kpad); TSEM_ALLOC(h->d_ctab_prev, h->Kpad);
  TSEM_HIP(hipMemsetAsync(h->d_ctab, 0, sizeof(double) * h->Kpad, h->stream));
  TSEM_HIP(hipMemsetAsync(h->d_ctab_prev, 0, sizeof(double) * h->Kpad, h->stream));
  if (int rc = tsem_make_ctabs(h)) return rc;
  TSEM_HIP(hipStreamSynchronize(h->stream));
  return TSEM_OK;
}

int tsem_set_params(tsem_ctx* h, const double* pi, const double* theta) {
  if (!h || !h->have_model || !pi || !theta) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  h->lag_valid = false;
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
  return tsem_bin_reset(h);
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


}  // extern "C"
