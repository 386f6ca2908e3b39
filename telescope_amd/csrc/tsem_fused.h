// Fused single-pass EM kernel (E-step row sums + M-step column scatter in ONE
// read of the matrix) for the column-partitioned blocked-COO layout.
//
// Why it looks like this (measurements: profiles/r01_primitives_ubench.log,
// profiles/r01_ring_ubench.log, profiles/r01_fused_timeline.txt; profiles/HISTORY.md 4.3):
//   * per-entry gathers of pi*theta and scatter-adds of the column sums only keep
//     up with the HBM stream when they hit LDS (global fp64 atomics: 22 G/s, LDS:
//     >500 G/s); K*16 B does not fit one CU's 160 KB, so columns are split in P
//     parts and a TEAM of P workgroups (one per CU, each owning one part's tables)
//     walks the same row blocks;
//   * a row's normaliser needs all P partial sums -> the team exchanges R partial
//     sums per block.  Teams are formed inside the launch from the hardware XCC_ID so
//     that all members share one XCD and the exchange is served by that XCD's L2
//     (plain stores stay in L2; agent-scope/sc1 loads bypass the reader's L1).
//     Values travel as 8-byte tagged granules (mantissa LSB = epoch parity): no
//     flag, no drain, no fence;
//   * a CU's vector-memory path returns data IN ORDER, so any exchange load waits
//     for every streaming load issued before it (measured 12 k cycles).  Instead of
//     fighting that, the exchange RIDES the stream: dedicated exchange waves publish
//     block k's partial sums one step after they were accumulated and load the partners'
//     two steps later, right after the barrier -- ahead of that step's prefetch burst,
//     behind the previous one -- and combine them in the same step;
//   * each member keeps its sub-block's numerators in REGISTERS between row-sum
//     phase (step k) and scatter phase (step k+4): a ring of 6 register sets, prefetch
//     distance 2 steps, so every stored entry is read from HBM exactly once and the
//     memory pipe never drains (data-path ceiling of this schedule: 6.2 TB/s,
//     tools/ubench/ring.hip);
//   * entries are 4 B (local row, local column) + either the fp64 Q value (FMT 0) or
//     its 2-byte score code, looked up in an LDS copy of the score table (FMT 1): the
//     same fp64 number at half the HBM bytes;
//   * a sub-block is stored in row order, so the row sums are reduced in registers and
//     across lanes (fz_row_sums) and only the end of each run of equal rows issues an
//     LDS atomic: ds_add_f64 costs 3x a gather (profiles/r01_lds_ubench.log).
// No assumption is made about dispatch order or block->XCD placement: teams,
// their size and the block round-robin all derive from tickets taken at run
// time; every wait is bounded and reports through an error word.
#pragma once

constexpr int FZ_NT = 1024;            // threads per workgroup
// Geometry (GEO): the exchange waves keep (P-1) partner values per row pair in registers, so larger
// teams use more exchange waves with fewer row pairs per lane; short rows need more row slots per
// block to fill the register tile.
//   0 : 2 exchange waves x 2 row pairs per lane (R <= 512), 14 data waves   teams of 1-4
//   1 : 3 exchange waves x 1 row pair  per lane (R <= 384), 13 data waves   teams of 5-8
//   2 : 3 exchange waves x 2 row pairs per lane (R <= 768), 13 data waves   teams of 1-4, short rows; teams of 5-8 whose rows cannot
//       fill the tiles with 384 row slots (round 3: (P - 1) x 2 partner values per exchange lane, 118-126 VGPRs, no spill)
//   3 : 3 exchange waves x 3 row pairs per lane (R <= 1152), 13 data waves  teams of 1-4, rows too short to fill the tile with
//       768 of them (round 3).  The row slots are what limits such a block: 48 B of LDS each (y ring 4 deep + s ring 2 deep)
//       next to 121 KB of tables.  Here the exchange wave reads a block's partial sums ONCE, at the publish, keeps its own
//       copy in registers until the combine and zeroes the LDS slot at once: the y ring is 2 deep, a row slot costs 32 B,
//       and 1024-1152 rows fit where 768 did.
__host__ __device__ constexpr int fz_nxw(int g) { return g == 0 ? 2 : 3; }          // exchange waves
__host__ __device__ constexpr int fz_rp(int g) { return g == 1 ? 1 : (g == 3 ? 3 : 2); }   // row pairs per lane
__host__ __device__ constexpr int fz_yr(int g) { return g == 3 ? 2 : 4; }           // depth of the y ring
__host__ __device__ constexpr int fz_dt(int g) { return FZ_NT - 64 * fz_nxw(g); }   // data threads
__host__ __device__ constexpr int fz_cap(int g) { return fz_dt(g) * 4; }            // entries per register tile
__host__ __device__ constexpr int fz_rmax(int g) { return 2 * 64 * fz_rp(g) * fz_nxw(g); }
constexpr int FZ_MAX_P = 8;
#ifndef FZ_GAP_STEPS
#define FZ_GAP_STEPS 1
#endif
constexpr int FZ_GAP = FZ_GAP_STEPS;   // steps between a block's publish and the partner loads
constexpr int FZ_NS = 5 + FZ_GAP;      // register sets: block k lives in set k % FZ_NS
constexpr int FZ_DL = 2;               // prefetch distance (steps)
constexpr int FZ_LAG = 3 + FZ_GAP;     // scatter lag (steps) = FZ_NS - FZ_DL
// (y ring, fz_yr(GEO) deep: row sums of block k live from step k to its combine at k+3 — or, geometry 3, to its publish at k+1)
constexpr int FZ_XS = 8;               // exchange slots per team (ring)
constexpr unsigned FZ_SPIN_LIMIT = 2000000u;
#ifndef FZ_SKIP_IDLE_WAVES
#define FZ_SKIP_IDLE_WAVES 1
#endif
constexpr int FZ_PROF_SLOTS = 16;
#ifndef FZ_LAG_CERR
#define FZ_LAG_CERR 0
#endif

// sync words (uint32): [0..7] per-XCD tickets, [8] registered WGs, [9] error
constexpr int FZ_SYNC_WORDS = 16;

struct FusedArgs {
  int P, Kp, R;
  int64_t nb, N_amb_pad;
  const int64_t* sb_off;
  const uint32_t* sb_q32;   // sb_off / 4 as 32-bit quad indices (fused kernel only)
  // FMT 0: fp64 entries, fp64 row weights (score table too large for LDS)   FMT 1: 2-byte score codes
  // FMT 2: fp64 entries, row weights as 2-byte codes (6 B less exchange traffic per row and member)
  const double* pval;       // FMT 0/2: Q values (fp64), 8 B per entry
  const uint16_t* pcode;    // FMT 1: raw score codes, 2 B per entry; Q = lut[code] bit for bit (sparse_plus.py:89-91)
  const double* lut;        // FMT 1: the score table, copied to LDS [lut_len]
  int lut_len;
  const uint16_t* wcode;    // FMT 1: [N_amb_pad] row weight as a code, w_i = lut[max code of the row]
  const uint32_t* prc;
  const double* ctab;
  const double* ctab2;  // lnl mode: pi*theta of the CURRENT params (ctab then holds the previous ones); MODE 4: of the PREVIOUS params
  double* lnl_out;      // lnl mode: one partial sum per workgroup [grid]
  int sorted;           // 1: sub-blocks in row order (row sums reduced in registers), 0: strand-transposed (one atomic per entry)
  int lnl_mode;         // 0: EM pass (scatter w*z); 1: sum z(prev) * log1p(Q * c_cur)  (model.py:744-760)
  const double* wrow;   // [N_amb_pad] fragment weight w_i = max_j Q_ij (0 in the padding)
  double* partial;      // [team slots][P*Kp], zero-filled before launch
  double* xchg;         // [team][FZ_XS][P][R] tagged granules, zero-filled before launch
  uint32_t* sync;       // zero-filled before launch
  const uint32_t* ctl;  // device-side loop control (tsem_em_chunk): ctl[0] != 0 -> the run has stopped, return at once
  // lnl pass, two forms enqueued back to back (tsem_em.hip launch_fused): `sel` counts the stored entries of columns whose log(pi*theta)
  // may put them into the exact branch of the log-table form; the launch with sel_want = 0 (log tables) runs while that count is
  // <= sel_thr, the one with sel_want = 1 (a logarithm per entry) when it is above — the other returns at once.  null: no selection
  const unsigned long long* sel;
  unsigned long long sel_thr;
  int sel_want;
  // MODE 2 (option "reproducible"): the column scatter adds pre-rounded pieces of w*z, so that every LDS accumulator sums EXACTLY
  // and the order in which the hardware serves the atomics stops mattering (profiles/HISTORY.md 5.1)
  const uint16_t* ebias;  // [P*Kp] biased exponent eb of the slot's bound 2^E (every contribution of the slot is < 2^E)
  int bin;                // 1: the high piece (multiples of 2^(E-30)), 2: the low piece (the remainder in multiples of 2^(E-60))
  double* partial2;       // MODE 3 (both pieces in ONE pass, a second accumulator table in LDS): the low pieces' team partials
  uint8_t* ovf;           // [P*Kp] set to 1 when a contribution reached its slot's bound (the host raises E and repeats the pass)
  // MODE 4 (`--use_likelihood`, model.py:783-789): the EM pass of iteration t+1 also sums the log-likelihood of iteration t,
  //   lnl_t = sum z_t * log1p(Q * c_t),  z_t = (Q * c_{t-1}) * recip0(rowsum_t):
  // Q * c_t are this pass's numerators; c_{t-1} is a third table in LDS (ctab2); recip0(rowsum_t) was stored by pass t (rinv, one
  // double per row slot, written by member 0 of the team at the combine) and is staged through LDS a block ahead by the exchange wave.
  double* rinv;           // [N_amb_pad] read (rows of the blocks to come) and rewritten (rows just combined) by every MODE 4 pass
  // SPLIT layout (round 4: parts of up to 15 424 columns, i.e. K up to 8 x 15 360 on the fused path; a part's pi*theta table AND its
  // accumulators no longer fit the LDS together, so an iteration is two lighter passes with ONE table each):
  //   MODE 5  row sums: c = pi*theta in LDS, phase 1 only; NO exchange — every member writes its partial row sums to
  //           ypart[member][row slot] (8 B per row and member) and runs free; k_row_factors then forms rinv[slot] = w_i *
  //           recip0(sum over the members, in member order: the fused combine's bits).  (A first version kept the team exchange and
  //           let member 0 store the factors: with nothing but phase 1 to hide it behind, the exchange round trip — 3400 clk at 768 row
  //           slots — WAS the step: 1.50 ms for the pass at 10M x 100k x 40 where the scatter pass takes 0.49.)
  //   MODE 7  scatter: the accumulators in LDS, no exchange (the exchange wave stages rinv through the s ring), acc[j] += Q_ij * s_i —
  //           the column sum is pi_j theta_j * sum_i Q_ij s_i, the common factor is applied by k_colreduce
  //   MODE 8  log-likelihood over ONE HALF of the part's columns (both tables of that half in LDS): sum z log1p(Q c_cur) with
  //           z = Q c_prev * rinv; entries of the other half are skipped; two launches (koff = 0, Kh)
  int Kh, koff;           // MODE 8: columns per half, first local column of this launch's half
  double* ypart;          // MODE 5: [P][N_amb_pad] partial row sums
  int lag;                // 0: rinv holds nothing yet (first pass of a run): the lnl partials of this launch are zero
  // log tables (round 5; MODE 1 and MODE 4 with the score table in LDS): log1p(Q c) = log Q + log c + 1 / (Q c) wherever Q c >= 2^27
  const double* lctab;    // [P*Kp] log(pi * theta) of the parameters whose log1p the pass takes (MODE 1: takes ctab2's place in LDS; MODE 4: a fourth table)
  const double* lqtab;    // [lq_n] log Q — FMT 1: per score code; FMT 2: index = (high word of Q >> lq_shift) - lq_base
  int lq_n, lq_shift, lq_base;   // lq_n == 0 (and lq_lin == 0): no log tables, every term goes through fz_log1p_tab
  // FMT 1 with the reference's score table Q = expm1((code * (1 / max)) * scale) (model.py:653): log Q = t - e^-t + ... IS t = (code * lq_a) * lq_b
  // to the last bits once t >= 38 — no table, no gather: two multiplications.  The host checks it against log(lut[code]) code by code
  // (ensure_log_tables); codes below lq_c0 (t < 38: absent from alignment data, whose Q starts at e^46) take the exact branch.
  int lq_lin, lq_c0;
  double lq_a, lq_b;
  int dbg;              // bit0: skip partner loads (timing experiments only; wrong results)
                        // bit5 / bit6: behave like a hand-off time-out in the EM / lnl pass (tests of the recovery path)
  unsigned long long* prof;   // optional per-step timestamps of team 0 / member 0
  int prof_blocks;
};

__device__ __forceinline__ uint32_t fz_ld_u32(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // sc1: bypasses L1
}

// Every vector-memory access of the steady-state loop is an UNCONDITIONAL raw buffer access: lanes
// (or whole steps) with nothing to do use an out-of-range offset / an empty resource, for which the
// hardware returns zeros (loads) or drops the write (stores).  With no branch around a load the
// compiler can count: a use of block i's registers waits with `s_waitcnt vmcnt(n)`, n = the loads
// issued since — with exec-masked or skipped loads it has to assume vmcnt(0), which drains the
// memory pipe twice per step (measured: 6.09 -> 5.04 ms per pass, profiles/HISTORY.md 4.1).
typedef unsigned int fz_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int fz_u32x2 __attribute__((ext_vector_type(2)));
constexpr int FZ_RSRC_FLAGS = 0x00027000;
#ifndef FZ_STREAM_POLICY
#define FZ_STREAM_POLICY 2
#endif
constexpr int FZ_STREAM = FZ_STREAM_POLICY;              // cache policy of the entry loads: nt (read once) keeps the
                                                         // exchange ring and the tables in L2 (fp64 entries: -3.6 %)
constexpr unsigned FZ_OOB = 0x7FFFFF00u;                 // beyond num_records of every resource used here
__device__ __forceinline__ __amdgpu_buffer_rsrc_t fz_rsrc(const void* base, uint64_t byte_off, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(base) + byte_off), 0, (int)bytes,
                                           FZ_RSRC_FLAGS);
}
__device__ __forceinline__ double2 fz_as_double2(fz_u32x4 t) {
  return make_double2(__hiloint2double((int)t.y, (int)t.x), __hiloint2double((int)t.w, (int)t.z));
}

// Cross-lane moves on the VALU (DPP): gfx9 keeps the wavefront shifts and row broadcasts.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ int fz_dpp_i(int old, int src) {
  int r = __builtin_amdgcn_update_dpp(old, src, CTRL, ROWMASK, 0xF, false);   // no source lane -> `old`
  asm volatile("" : "+v"(r));                                                   // (see fz_shr1_d)
  return r;
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double fz_dpp_d(double old, double src) {
  const int lo = fz_dpp_i<CTRL, ROWMASK>(__double2loint(old), __double2loint(src));
  const int hi = fz_dpp_i<CTRL, ROWMASK>(__double2hiint(old), __double2hiint(src));
  return __hiloint2double(hi, lo);
}
constexpr int FZ_DPP_WAVE_SHL1 = 0x130, FZ_DPP_WAVE_SHR1 = 0x138;
__device__ __forceinline__ double fz_shr1_d(double v) {     // value of lane l-1 (0 for lane 0)
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), FZ_DPP_WAVE_SHR1, 0xF, 0xF, true);
  int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), FZ_DPP_WAVE_SHR1, 0xF, 0xF, true);
  int lo2 = lo;
  // pin the moves where they are written: sunk into a divergent branch they would run with a partial
  // EXEC mask, and a DPP move reads 0 from a lane that is switched off
  asm volatile("" : "+v"(lo2), "+v"(hi));
  return __hiloint2double(hi, lo2);
}
// Row sums of a wave whose lanes hold four consecutive entries each, stored in row order: add the
// products m0..m3 (rows r0..r3) into yb[] with ONE LDS atomic per run of equal rows instead of one
// per entry (ds_add_f64 costs 24 clk per 64 lanes, a gather 8: profiles/r01_lds_ubench.log).  Runs
// inside a lane are summed in registers; a run that continues into the next lanes is handed on with
// wavefront shifts on the VALU (as many rounds as the longest chain of lanes lying inside one row:
// usually 1-3).  Every lane of the wave must call this (idle lanes with idle = true).  The result
// is right for any entry order; the row order only makes the runs long.
__device__ __forceinline__ void fz_row_sums(double* yb, bool idle, uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3,
                                            double m0, double m1, double m2, double m3) {
  const bool e01 = r1 == r0, e12 = r2 == r1, e23 = r3 == r2;
  const double s1 = e01 ? m0 + m1 : m1;                     // running sums of the runs inside the lane
  const double s2 = e12 ? s1 + m2 : m2;
  const double s3 = e23 ? s2 + m3 : m3;
  const int a = idle ? 0xFFFF : (int)r0, b = idle ? 0xFFFD : (int)r3;
  const int prev_b = fz_dpp_i<FZ_DPP_WAVE_SHR1, 0xF>(0xFFFE, b);
  const int next_a = fz_dpp_i<FZ_DPP_WAVE_SHL1, 0xF>(0xFFFC, a);
  const bool joins = prev_b == a;                            // my first run continues the left neighbour's last one
  const bool open = e01 & e12 & e23 & joins;                 // the whole lane lies inside that run: pass the sum on
  double I = idle ? 0.0 : s3;                                // sum of the run that ends with my last entry
  for (unsigned long long chain = __builtin_amdgcn_ballot_w64(open); chain; chain &= chain << 1) {
    const double up = fz_shr1_d(I);
    if (open) I = s3 + up;
  }
  const double left = fz_shr1_d(I);                          // (not inside the ?: — every lane must execute the move)
  const double C = joins ? left : 0.0;                       // what the lanes to the left bring for my first run
  if (!idle) {
    // runs that end before my last entry: the first of them takes the carry
    if (!e01) lds_add(&yb[r0], m0 + C);
    const double C1 = e01 ? C : 0.0;
    if (!e12) lds_add(&yb[r1], s1 + C1);
    if (!e23) lds_add(&yb[r2], s2 + (e12 ? C1 : 0.0));
    if (next_a != b) lds_add(&yb[r3], I);                    // my last run ends here (else the right neighbour has it)
  }
}

// log1p(x), x >= 0 finite, for the fused lnl pass (2e9 evaluations per pass: ts_log1p_pos was 3.7 of its 7.3 ms): 1 + x = 2^k m,
// m = c_i (1 + r) with c_i = 1 + i / 64 from the top six mantissa bits, (1 / c_i, log c_i) from a 64-entry table in LDS (built
// at kernel start with the accurate routine), log1p(r) by its series to r^9 (r < 2^-6), the rounding of 1 + x put back like
// fdlibm does.  Entry 0 is (1, 0), so small x keep their relative accuracy; elsewhere the error is ~1e-16 ABSOLUTE per
// evaluation (1 / c_i is rounded), which is what a sum of z * log1p needs.  Max relative error measured against log1p: see
// test_fast_log1p_of_the_fused_lnl_pass.
// CERR = false (the carrying pass, MODE 4, whose step is bound by its arithmetic): without the correction for the rounding of 1 + x —
// an error of at most 2^-53 ABSOLUTE per evaluation, the size of the table's own (1 / c_i is rounded); relative accuracy for tiny x
// is lost, which a SUM of z * log1p (terms of 1 ... 100 dominate) does not see.
constexpr int FZ_LOGTAB = 64;
template <bool CERR = true>
__device__ __forceinline__ double fz_log1p_tab(double x, const double2* __restrict__ tab) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double u = 1.0 + x;
  const int hu = __double2hiint(u);
  const int k = (hu >> 20) - 1023;
  const double cerr = CERR ? (k > 0 ? 1.0 - (u - x) : x - (u - 1.0)) * __builtin_amdgcn_rcp(u) : 0.0;   // (1 + x) - u, relative to u
  const double2 t = tab[(hu >> 14) & (FZ_LOGTAB - 1)];
  const double m = __hiloint2double((hu & 0x000FFFFF) | 0x3FF00000, __double2loint(u));    // [1, 2)
  const double r = fma(m, t.x, -1.0);
  double p = fma(r, 1.0 / 9.0, -1.0 / 8.0);
  p = fma(r, p, 1.0 / 7.0); p = fma(r, p, -1.0 / 6.0); p = fma(r, p, 1.0 / 5.0); p = fma(r, p, -0.25);
  p = fma(r, p, 1.0 / 3.0); p = fma(r, p, -0.5); p = fma(r, p, 1.0);
  const double dk = (double)k;
  return fma(dk, ln2_hi, t.y + fma(r, p, fma(dk, ln2_lo, cerr)));
}

// log1p(x) for x = Q c given L = log Q + log c (both from tables: a score has a few hundred distinct values, a part a few thousand
// columns; 2e9 logarithms per pass become 2e9 additions).  Three ranges of L:
//   L >= 18.715 (x >= 2^27): log1p(x) = L + 1/x - 1/(2 x^2) + ..., the third term is below 2.8e-17.  1/x = exp(-L) needs a RELATIVE
//     accuracy of 1e-7 to be right to 1e-16 absolute: fp32 (v_exp_f32 of an fp32 product; 18.7 <= L: no overflow, flushes to 0 beyond 87);
//   L < -40 (x < 4.3e-18): log1p(x) = x - ... is below 4.3e-18 — absolute accuracy is what a SUM of z * log1p with terms of 20-100
//     needs: 0 (this is also where a column with pi * theta == 0 lands: log 0 = -inf);
//   between (a column that is dying under a zero prior passes through here for a few dozen iterations): the table-driven log1p of
//     the exact x, which the caller produces on demand (`exact_x()`: the numerator it holds, or Q times a pi*theta fetched from
//     global memory).  A wave-uniform branch: no lane of a wave takes it in the bench workload, few waves do on converged real data
//     (Q >= e^46 for alignment scores: x < 2^27 needs pi * theta < 1e-12).  (exp(L) instead of the fetch: the inlined fp64 exp,
//     four times per lane, spilled 116 VGPRs in the fp64-entry kernel.)
constexpr double FZ_L_FAST = 18.715, FZ_L_ZERO = -40.0;
template <bool CERR, class F>
__device__ __forceinline__ double fz_log1p_of_log(double L, F&& exact_x, const double2* __restrict__ tab, bool force_exact = false, bool live = true) {
  double v = L + (double)__expf(-(float)L);                // (L = -inf: NaN, replaced below)
  // ONE compare on the fast path; everything else hides behind the wave-uniform branch.  Lanes whose z is 0 (`live` false: the padding
  // of a sub-block's last quads, columns whose previous pi*theta is 0) never take it — with the arithmetic log Q the padding's L is an
  // ordinary number, and one wave per step walking into the exact branch (a global load) cost 1 ms of a 4.4 ms pass.
  const bool slow = ((L < FZ_L_FAST) | force_exact) & live;
  if (__builtin_amdgcn_ballot_w64(slow) != 0ull) {
    const bool mid = slow & ((L >= FZ_L_ZERO) | force_exact);
    if (slow) v = 0.0;
    if (__builtin_amdgcn_ballot_w64(mid) != 0ull) {
      if (mid) v = fz_log1p_tab<CERR>(exact_x(), tab);
    }
  }
  return live ? v : 0.0;                                   // always finite: z * v is safe to add
}
// FMT 2 (fp64 entries, score table in LDS): log Q from Q's top bits — the host proved the index unique over the score table
__device__ __forceinline__ double fz_logq_of(double q, const double* __restrict__ lqS, int lq_n, int lq_shift, int lq_base) {
  const unsigned idx = (unsigned)((__double2hiint(q) >> lq_shift) - lq_base);
  return lqS[min(idx, (unsigned)(lq_n - 1))];              // (Q = 0, the padding: some finite entry; its z is 0)
}

struct FzRegs {            // 4 entries per thread: 12 VGPRs (FMT 1: 6 until phase 1 turns the codes into numerators)
  uint4 rc;
  double2 v0, v1;
  uint2 cd;
};

struct FzX {                 // context handed to the exchange wave
  const double* lut;
  double* y; double* s; double* rp; uint32_t* offs; unsigned long long* xbase; uint32_t* err;
  int R, team, T, lane, xw;
  int64_t nblk, nsteps;
};

// Exchange wave of member PP of a P-member team (see k_em_fused for the schedule).
template <int P, int PP, int MODE, int FMT, int GEO>
__device__ __forceinline__ void fz_xchg(const FusedArgs& A, const FzX& X) {
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  constexpr int p = PP;
  constexpr int FZ_RP = fz_rp(GEO);
  constexpr int FZ_YR = fz_yr(GEO);
  constexpr bool OWNREG = GEO == 3;                            // own partial sums travel in registers from publish to combine
  constexpr bool LNL1 = MODE == 1 || MODE == 9;                // the dedicated log-likelihood pass (9: with log tables, see fz_log1p_of_log)
  constexpr bool SPA = MODE == 5;                              // split layout: row-sum pass (the partial sums go to A.ypart, no exchange)
  constexpr bool SPS = MODE == 7 || MODE == 8;                 // split layout: no exchange, the s ring is staged from A.rinv
  constexpr bool NOX = SPA || SPS;
  constexpr int NPART = P > 1 ? P - 1 : 1;
  double* const y = X.y; double* const s = X.s; uint32_t* const offs = X.offs; uint32_t* const err = X.err;
  unsigned long long* const xbase = X.xbase;
  const int R = X.R, team = X.team, T = X.T, lane = X.lane;
  const int64_t nblk = X.nblk, nsteps = X.nsteps;
  // rows served by this exchange wave: wave 0 fills its 16-byte x 64-lane instructions completely
  const int rlo = X.xw * (128 * FZ_RP), rhi = min(R, rlo + 128 * FZ_RP);
  __amdgpu_buffer_rsrc_t xrsrc = fz_rsrc(xbase, 0, (unsigned)(FZ_XS * P * R * 8));
  __amdgpu_buffer_rsrc_t qrsrc = fz_rsrc(A.sb_q32, 0, (unsigned)((A.nb * P + 2) * 4));
  const bool offw = X.xw == 0;                                  // wave that also ferries the sub-block offsets
  const bool nopart = (A.dbg & 1) != 0;                         // timing experiment: partner loads go out of range
  auto slot_of = [&](int64_t k, int q) -> unsigned long long* {
    return xbase + ((int64_t)(k & (FZ_XS - 1)) * P + q) * R;
  };
  auto tag_of = [&](int64_t k) -> unsigned long long {
    return (unsigned long long)((((k / FZ_XS) & 1) ^ 1));
  };
  if (!(A.dbg & 16)) __builtin_amdgcn_s_setprio(3);   // few instructions, all on the critical path of the step
  struct Gen { u64x2 pv[NPART][FZ_RP]; double2 w[FZ_RP]; uint32_t wc[FZ_RP]; uint32_t off; double2 rpv[FZ_RP]; };
  Gen g1;
  g1.off = 0;
  struct Own { u64x2 v[FZ_RP]; };
  Own own_q[2 + FZ_GAP];                                  // (geometry 3) own_q[d]: read d + 1 steps ago
#pragma unroll
  for (int d = 0; d < 2 + FZ_GAP; ++d)
#pragma unroll
    for (int j = 0; j < FZ_RP; ++j) own_q[d].v[j] = (u64x2){0ull, 0ull};
  // issue the partner / weight loads of block k and the offset fetch of block ko (never branched around)
  auto issue = [&](Gen& g, int64_t k, int64_t ko) {
    {
      const bool ok = offw && lane < 2 && ko >= 5 && ko < nblk;
      g.off = __builtin_amdgcn_raw_buffer_load_b32(qrsrc, ok ? (unsigned)(((team + ko * T) * P + p + lane) * 4) : FZ_OOB, 0, 0);
    }
    const bool kv = k >= 0 && k < nblk;
    const uint64_t blk = kv ? (uint64_t)(team + k * T) : 0;
    if (SPS) {                                                  // the row factors pass A left for block k
      __amdgpu_buffer_rsrc_t rr = fz_rsrc(A.rinv, blk * R * 8, kv ? (unsigned)R * 8 : 0);
#pragma unroll
      for (int j = 0; j < FZ_RP; ++j)
        g.rpv[j] = fz_as_double2(__builtin_amdgcn_raw_buffer_load_b128(rr, (unsigned)(rlo + 2 * (lane + 64 * j)) * 8, 0, 0));
    }
    if (!LNL1 && !NOX) {                                        // row weights (the lnl pass uses w = 1)
      if (FMT != 0) {
        __amdgpu_buffer_rsrc_t wr = fz_rsrc(A.wcode, blk * R * 2, kv ? (unsigned)R * 2 : 0);
#pragma unroll
        for (int j = 0; j < FZ_RP; ++j)
          g.wc[j] = __builtin_amdgcn_raw_buffer_load_b32(wr, (unsigned)(rlo + 2 * (lane + 64 * j)) * 2, 0, 0);   // 2 codes
      } else {
        __amdgpu_buffer_rsrc_t wr = fz_rsrc(A.wrow, blk * R * 8, kv ? (unsigned)R * 8 : 0);
#pragma unroll
        for (int j = 0; j < FZ_RP; ++j)
          g.w[j] = fz_as_double2(__builtin_amdgcn_raw_buffer_load_b128(wr, (unsigned)(rlo + 2 * (lane + 64 * j)) * 8, 0, 0));
      }
    }
    if (MODE == 4) {                                            // 1 / rowsum of the PREVIOUS pass for the rows of block i+1 (phase 1 of the next step)
      const int64_t kn = k + 3 + FZ_GAP;
      const bool nv = kn > 0 && kn < nblk && A.lag != 0;        // (block 0: loaded by the prologue)
      __amdgpu_buffer_rsrc_t rr = fz_rsrc(A.rinv, (nv ? (uint64_t)(team + kn * T) : 0) * R * 8, nv ? (unsigned)R * 8 : 0);
#pragma unroll
      for (int j = 0; j < FZ_RP; ++j)
        g.rpv[j] = fz_as_double2(__builtin_amdgcn_raw_buffer_load_b128(rr, (unsigned)(rlo + 2 * (lane + 64 * j)) * 8, 0, 0));
    }
    if (P > 1 && !NOX) {
      // Validity lives in the DESCRIPTOR (an empty resource returns zeros), never in a per-lane select of the
      // offset: the compiler turned `ok ? offset : out-of-range` into two exec-masked loads with one destination
      // and put `s_waitcnt vmcnt(0)` between them — the exchange wave then sat behind the data waves' whole
      // burst in the middle of its issue sequence, every step (round-2 ISA review; r02 timelines "x:issued").
      // Lanes past the last row pair re-read row pair R-2: one more hit on a line the wave loads anyway.
      __amdgpu_buffer_rsrc_t xr = fz_rsrc(xbase, 0, (kv && !nopart) ? (unsigned)(FZ_XS * P * R * 8) : 0u);
#pragma unroll
      for (int j = 0; j < FZ_RP; ++j) {
        const int r = min(rlo + 2 * (lane + 64 * j), R - 2);
#pragma unroll
        for (int q = 0; q < P; ++q) {
          if (q == p) continue;
          // one 16-byte agent-scope (sc1, L1-bypassing) load per row pair and partner
          const unsigned boff = (unsigned)((((k & (FZ_XS - 1)) * P + q) * R + r) * 8);
          fz_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(xr, boff, 0, 16);
          g.pv[q < p ? q : q - 1][j].x = ((unsigned long long)t.y << 32) | t.x;
          g.pv[q < p ? q : q - 1][j].y = ((unsigned long long)t.w << 32) | t.z;
        }
      }
    }
  };
  // combine block k from generation g: s = w * recip0(sum of the P partials), fixed order
  auto combine = [&](Gen& g, const Own& mine, int64_t k, int64_t ko) {
    if (offw && lane < 2 && ko >= 5 && ko < nblk) offs[(ko & 7) * 2 + lane] = g.off;   // blocks 0..4: prologue
    if (k < 0 || k >= nblk) return;
    if (SPS) {                                            // nothing to combine: hand the stored row factors to the data waves
#pragma unroll
      for (int j = 0; j < FZ_RP; ++j) {
        const int r = rlo + 2 * (lane + 64 * j);
        if (r < rhi) *reinterpret_cast<double2*>(&s[(k & 1) * R + r]) = g.rpv[j];
      }
      return;
    }
    if (SPA) return;
#ifdef FZ_EXPERIMENT
    if (A.dbg & 2048) return;                             // timing experiment: no combine at all (s stays 0, members free-run)
#endif
    // (Round 3 tried carrying this member's OWN sums in registers from the publish to the combine, zeroing y at the
    // publish: two LDS operations and one LDS wait less per step — and 2.3x the tag misses, code16 3.33 -> 3.37 ms, fp64
    // unchanged, with one or two steps of gap alike (profiles/r03_exchange_bounds.txt): what the combine costs is
    // WAITING for the slowest member of the team, not its LDS traffic.)
    const unsigned long long tag = tag_of(k);
#pragma unroll
    for (int j = 0; j < FZ_RP; ++j) {
      const int r = rlo + 2 * (lane + 64 * j);
      if (r < rhi) {
        u64x2 own;
        if (OWNREG) own = mine.v[j]; else own = *reinterpret_cast<const u64x2*>(&y[(k & (FZ_YR - 1)) * R + r]);
        if (P > 1 && !nopart) {
          unsigned spins = 0;
          for (;;) {                                    // normally true at once: published 2 steps ago
            bool ok = true;
#pragma unroll
            for (int i = 0; i < NPART; ++i) ok &= ((g.pv[i][j].x & 1ull) == tag) & ((g.pv[i][j].y & 1ull) == tag);
            if (ok) break;
            if (spins == 0) atomicAdd(err + 1, 1u);     // statistics: granules that were not there yet
#pragma unroll
            for (int q = 0; q < P; ++q) {               // slow path: agent-scope (sc1) reloads
              if (q == p) continue;
              const unsigned long long* src = slot_of(k, q) + r;
              g.pv[q < p ? q : q - 1][j].x = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              g.pv[q < p ? q : q - 1][j].y = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (++spins > FZ_SPIN_LIMIT) { atomicOr(err, 2u); break; }
            if ((spins & 255u) == 0 && fz_ld_u32(err)) break;
          }
        }
        double ys0 = 0.0, ys1 = 0.0;
#pragma unroll
        for (int q = 0; q < P; ++q) {                   // fixed order: every member computes the same bits
          u64x2 v = (q == p || nopart) ? own : g.pv[q < p ? q : (q > 0 ? q - 1 : 0)][j];
          if (P > 1) { v.x &= ~1ull; v.y &= ~1ull; }
          ys0 += __longlong_as_double((long long)v.x);
          ys1 += __longlong_as_double((long long)v.y);
        }
        double2 w = make_double2(1.0, 1.0);
        if (!LNL1) w = FMT != 0 ? make_double2(X.lut[g.wc[j] & 0xFFFFu], X.lut[g.wc[j] >> 16]) : g.w[j];
        // z = n * recip0(rowsum) (sparse_plus.py:52), weighted by w_i (model.py:730).  (The exchange wave computes 4-6 of these IEEE
        // divisions per lane and step, on the critical path of a short-row step.  v_rcp_f64 + two fma-corrected Newton steps
        // instead: -3 % at 10 entries per row, -1.5 % at 40 with score codes when written without any special-case handling;
        // made safe for zero / denormal / huge sums — a wave-uniform choice between the sequences, or frexp / ldexp around the
        // short one — the gain is within the repeat spread, and it is not a correctly rounded quotient: not taken.
        // profiles/r03_exchange_bounds.txt section 7.)
        const double ri0 = recip0(ys0), ri1 = recip0(ys1);
        *reinterpret_cast<double2*>(&s[(k & 1) * R + r]) = make_double2(ri0 * w.x, ri1 * w.y);
        if (MODE == 4 && PP == 0) {                       // what the NEXT pass needs of this one's E-step: one member stores it
          __amdgpu_buffer_rsrc_t ro = fz_rsrc(A.rinv, (uint64_t)(team + k * T) * R * 8, (unsigned)R * 8);
          fz_u32x4 sv;
          sv.x = (unsigned)__double2loint(ri0); sv.y = (unsigned)__double2hiint(ri0); sv.z = (unsigned)__double2loint(ri1); sv.w = (unsigned)__double2hiint(ri1);
          __builtin_amdgcn_raw_buffer_store_b128(sv, ro, (unsigned)r * 8, 0, 0);
        }
        if (!OWNREG) *reinterpret_cast<double2*>(&y[(k & (FZ_YR - 1)) * R + r]) = make_double2(0.0, 0.0);
      }
    }
  };
  int64_t i = 0;
  if (A.dbg & 8) {                                      // timing experiment: exchange waves only keep the barriers
    for (; i < nsteps; ++i) {
      if (offw && lane < 2) { const int64_t ko = i + FZ_DL + 2; if (ko >= 5 && ko < nblk) offs[(ko & 7) * 2 + lane] = A.sb_q32[(team + ko * T) * P + p + lane]; }
      __syncthreads();
    }
    return;
  }
  auto xstep = [&]() {
    const bool pr = A.prof && team == 0 && p == 0 && lane == 0 && offw && (int)i < A.prof_blocks;
    if (pr) A.prof[i * FZ_PROF_SLOTS + 6] = clock64();
    // ONE generation of partner values, loaded and combined in the SAME step.  The loads go out right after the
    // publish (two steps after the partners published theirs, ahead of this step's burst of the data waves, which
    // comes at its end); they come back through the in-order memory pipe behind the previous burst, and the wave
    // has nothing else to do meanwhile (loads before the publish: no difference).  Round 1 and the first round-2
    // version kept TWO generations in flight across the barrier, alternating between two structs: the compiler could not prove the one combined last step
    // complete on every path, put `s_waitcnt vmcnt(0..2)` in front of every second publish (whose temporaries
    // landed on that generation's registers), and the wave sat 1000-2400 clk behind the data waves' burst before
    // it published — every second step 500-1000 clk longer (timelines: "x:published" 60 vs 1000-2400 clk).  A
    // single struct combined at the top of the NEXT step instead is bistable: one late combine (a partner that
    // runs behind) delays the next issue behind the burst, whose values then come late, and so on for the rest of
    // the pass.  Same-step is robust, needs half the registers, and the two steps between publish and load make
    // a miss rarer (profiles/r02_ab_exchange.txt).
    // Tried once more at the end of round 2 with UNTRACKED loads (inline-asm buffer loads the compiler keeps no
    // score-board entry for, one explicit `s_waitcnt vmcnt(n)`, two generations, loads a step ahead of their combine):
    // the exchange wave is then done long before the data waves, and the pass is no faster (codes 3.55 = 3.55 ms, fp64
    // entries 4.21 against 4.08 ms on the same box; three times the tag misses with one step between publish and
    // load) — the step is not waiting for the exchange.  (Lesson kept: asm VMEM needs `s_nop 4` in front — the hazard
    // recognizer does not see it, and a descriptor SGPR restored by v_readlane was read too early.)
    // publish y(i-1): tagged granules, 16-byte stores (a tear between halves is harmless).  PLAIN
    // stores: the line stays in the XCD's L2, where the partners' sc1 loads find it (a write-through
    // sc1 store drops it from L2: measured 14 M tag misses per pass vs 0.14 M)
    if (SPS) {
      // split layout, scatter / lnl pass: nothing to publish, the members of a team run free
    } else if (SPA) {
      // split layout, row-sum pass: y(i-1) goes to HBM as it is (k_row_factors sums the members), the slot is zeroed at once; only
      // the lane that owns a row pair touches it (a clamped lane would read the slot after its owner zeroed it)
      const int64_t kp = i - 1;
      const bool pv = kp >= 0 && kp < nblk;
      __amdgpu_buffer_rsrc_t ys = fz_rsrc(A.ypart, ((uint64_t)p * (uint64_t)A.N_amb_pad + (pv ? (uint64_t)(team + kp * T) : 0ull) * (uint64_t)R) * 8,
                                          pv ? (unsigned)R * 8 : 0u);
#pragma unroll
      for (int j = 0; j < FZ_RP; ++j) {
        const int r0 = rlo + 2 * (lane + 64 * j);
        const int r = min(r0, R - 2);
        const fz_u32x4 gq = *reinterpret_cast<const fz_u32x4*>(&y[(kp & (FZ_YR - 1)) * R + r]);
        if (pv && r0 < rhi) *reinterpret_cast<double2*>(&y[(kp & (FZ_YR - 1)) * R + r]) = make_double2(0.0, 0.0);
        __builtin_amdgcn_raw_buffer_store_b128(gq, ys, r0 < rhi ? (unsigned)r * 8 : FZ_OOB, 0, 0);
      }
    } else
    if (OWNREG) {
      // geometry 3: read y(i-1) once — publish it, keep it for the combine three steps on, zero the slot now (it is
      // complete since the barrier that ended step i-1 and nobody else reads it; only the lanes that own a row pair zero
      // it: a clamped lane would race with the owner's read)
      const int64_t kp = i - 1;
      const bool pv = kp >= 0 && kp < nblk;
      const unsigned long long tag = tag_of(kp);
      __amdgpu_buffer_rsrc_t xs = fz_rsrc(xbase, 0, (pv && P > 1) ? (unsigned)(FZ_XS * P * R * 8) : 0u);
#pragma unroll
      for (int d = 1 + FZ_GAP; d > 0; --d) own_q[d] = own_q[d - 1];
#pragma unroll
      for (int j = 0; j < FZ_RP; ++j) {
        const int r0 = rlo + 2 * (lane + 64 * j);
        const int r = min(r0, R - 2);
        fz_u32x4 gq = *reinterpret_cast<const fz_u32x4*>(&y[(kp & (FZ_YR - 1)) * R + r]);
        own_q[0].v[j] = (u64x2){((unsigned long long)gq.y << 32) | gq.x, ((unsigned long long)gq.w << 32) | gq.z};
        if (pv && r0 < rhi) *reinterpret_cast<double2*>(&y[(kp & (FZ_YR - 1)) * R + r]) = make_double2(0.0, 0.0);
        gq.x = (gq.x & ~1u) | (unsigned)tag;
        gq.z = (gq.z & ~1u) | (unsigned)tag;
        // only the OWNER of a row pair stores it: a clamped lane (of this or another exchange wave) may read the slot after the
        // owner zeroed it and would publish zeros over the owner's values (out-of-range offset: the store is dropped)
        const unsigned boff = r0 < rhi ? (unsigned)((((kp & (FZ_XS - 1)) * P + p) * R + r) * 8) : FZ_OOB;
        if (P > 1) __builtin_amdgcn_raw_buffer_store_b128(gq, xs, boff, 0, 0);
      }
    } else
    if (P > 1) {
      const int64_t kp = i - 1;
      const bool pv = kp >= 0 && kp < nblk;
      const unsigned long long tag = tag_of(kp);
#ifdef FZ_EXPERIMENT
      const bool xnp = (A.dbg & 1024) != 0;               // timing experiment: publish stores dropped
#else
      constexpr bool xnp = false;
#endif
      __amdgpu_buffer_rsrc_t xs = fz_rsrc(xbase, 0, (pv && !xnp) ? (unsigned)(FZ_XS * P * R * 8) : 0u);   // (an empty resource drops the store)
#pragma unroll
      for (int j = 0; j < FZ_RP; ++j) {
        const int r = min(rlo + 2 * (lane + 64 * j), R - 2);   // lanes past the last row pair store row pair R-2 again (same bytes)
        fz_u32x4 gq = *reinterpret_cast<const fz_u32x4*>(&y[(kp & (FZ_YR - 1)) * R + r]);
        gq.x = (gq.x & ~1u) | (unsigned)tag;
        gq.z = (gq.z & ~1u) | (unsigned)tag;
        const unsigned boff = (unsigned)((((kp & (FZ_XS - 1)) * P + p) * R + r) * 8);
        __builtin_amdgcn_raw_buffer_store_b128(gq, xs, boff, 0, 0);
      }
    }
    if (pr) A.prof[i * FZ_PROF_SLOTS + 10] = clock64();   // (waits for the LDS reads of the publish: lgkmcnt)
    issue(g1, i - 2 - FZ_GAP, i + FZ_DL + 2);
    if (pr) A.prof[i * FZ_PROF_SLOTS + 7] = clock64();
    combine(g1, own_q[1 + FZ_GAP], i - 2 - FZ_GAP, i + FZ_DL + 2);   // (geometry 3: block i-2-GAP was read 1+GAP publishes before this one's)
    if (MODE == 4) {                                      // stage the row factors of block i+1 for the data waves' next step
#pragma unroll
      for (int j = 0; j < FZ_RP; ++j) {
        const int r = rlo + 2 * (lane + 64 * j);
        if (r < rhi) *reinterpret_cast<double2*>(&X.rp[((i + 1) & 1) * R + r]) = g1.rpv[j];
      }
    }
    if (pr) A.prof[i * FZ_PROF_SLOTS + 5] = clock64();
    if (A.prof && team == 0 && p == 0 && lane == 0 && X.xw == 1 && (int)i < A.prof_blocks) A.prof[i * FZ_PROF_SLOTS + 9] = clock64();
    __syncthreads();
    ++i;
  };
  while (i < nsteps) xstep();
}

template <int PT, int MODE, int FMT, int GEO>
__global__ __launch_bounds__(FZ_NT) void k_em_fused(FusedArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int P = PT;
  constexpr int FZ_DT = fz_dt(GEO);
  constexpr int FZ_YR = fz_yr(GEO);
  const int Kp = A.Kp, R = A.R;
  constexpr bool EXACT = MODE == 2 || MODE == 3;   // option "reproducible"
  constexpr bool LAG = MODE == 4;                  // EM pass + the log-likelihood of the previous iteration
  constexpr bool LNL1 = MODE == 1 || MODE == 9;    // the dedicated log-likelihood pass; MODE 9: log tables in LDS (score table in LDS: FMT 1, 2)
  constexpr bool SPA = MODE == 5, SPB = MODE == 7, SPL = MODE == 8;   // split layout (see FusedArgs): ONE table of Kp entries, or two of Kh
  const int KT = SPL ? A.Kh : Kp;                  // entries per LDS table
  double* c = reinterpret_cast<double*>(smem);
  double* acc = (SPA || SPB) ? c : c + KT;         // (MODE 5 has no accumulators, MODE 7 no pi*theta table)
  double* const acc2 = acc + Kp;                   // MODE 3: the low pieces' accumulators [Kp];  MODE 4: pi*theta of the previous parameters
  double* const cprev = acc2;
  double* y = acc + ((MODE == 3 || LAG) ? 2 : 1) * KT;   // y[FZ_YR][R]  partial row sums (ring)
  double* s = y + FZ_YR * R;                       // s[2][R]      w_i / rowsum_i   (ring)
  double* const rpS = s + 2 * R;                   // MODE 4: rp[2][R]  1 / rowsum_i of the previous pass (ring)
  int* ibox = reinterpret_cast<int*>(s + (LAG ? 4 : 2) * R);   // [0]=ticket [1..8]=xcd counts
  const int tid = threadIdx.x;
  uint32_t* const sync = A.sync;
  uint32_t* const err = sync + 9;
  if (A.ctl && fz_ld_u32(A.ctl) != 0u) return;            // stopped by an earlier update kernel of this chunk
  if (LNL1 && A.sel && ((*A.sel > A.sel_thr) != (A.sel_want != 0))) return;   // the other form of the lnl pass runs (see FusedArgs)
  // start-up timeline (tools/startup_prof.py): 100 MHz wall clock of every workgroup's thread 0 at entry / tickets counted / LDS
  // zeroed / tables loaded / loop start / loop end / exit, behind the per-step slots of team 0
  unsigned long long* const sprof = (A.prof && tid == 0) ? A.prof + 64 * FZ_PROF_SLOTS + (size_t)blockIdx.x * 8 : nullptr;
  if (sprof) sprof[0] = wall_clock64();
  if (A.dbg & ((!LNL1 && MODE != 8) ? 32 : 64)) {     // test hook: what a watchdog time-out leaves behind
    if (blockIdx.x == 0 && tid == 0) atomicOr(err, 2u);
    return;
  }

  // ---- team formation from the hardware XCC id --------------------------------
  // Round 4: the start-up took 17.6 us of a 560 us shard pass (tools/startup_prof.py: every workgroup enters within 1.2 us, but
  // had the tickets counted only 13 us later — two dependent device-scope atomics, a release fence (an L2 write-back per
  // workgroup) and a polling loop of two uncached loads per turn — and loaded its tables after that).  Now: ONE atomic gives the
  // ticket, hence the column part; the part's tables are loaded and the LDS is zeroed WHILE the other workgroups take theirs; the
  // count every workgroup waits for is the sum of the eight ticket counters themselves (no second counter, no fence: nothing but
  // the counters is communicated).
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 7u;
  if (tid == 0) ibox[0] = (int)__hip_atomic_fetch_add(&sync[xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int t = tid; t < KT; t += FZ_NT) acc[t] = 0.0;   // (lnl mode: overwritten with ctab2 below)
  if (MODE == 3) for (int t = tid; t < Kp; t += FZ_NT) acc2[t] = 0.0;
  for (int t = tid; t < FZ_YR * R; t += FZ_NT) y[t] = 0.0;
  for (int t = tid; t < (LAG ? 4 : 2) * R; t += FZ_NT) s[t] = 0.0;
  uint32_t* offs = reinterpret_cast<uint32_t*>(ibox + 16);   // [8][2] sub-block quad ranges (ring), LDS
  double* dum = reinterpret_cast<double*>(ibox + 32);        // [64] one slot per lane: where idle lanes send their (zero) atomics
  double* lutS = dum + 64;                                   // FMT 1: score table [lut_len]
  if (tid < 64) dum[tid] = 0.0;
  if (FMT != 0)
    for (int t = tid; t < A.lut_len; t += FZ_NT) lutS[t] = A.lut[t];
  uint16_t* const eS = reinterpret_cast<uint16_t*>(lutS + A.lut_len);   // MODE 2: [Kp] the slots' exponent bounds
  // MODE 1 / 4 with log tables: log Q behind the score table [lq_n]
  constexpr bool LT = MODE == 9;
  double* const lqS = lutS + A.lut_len;
  if (LT && !(FMT == 1 && A.lq_lin))
    for (int t = tid; t < A.lq_n; t += FZ_NT) lqS[t] = A.lqtab[t];
  // MODE 1: [FZ_LOGTAB] (1 / c_i, log c_i) for fz_log1p_tab, in the same place as MODE 2's table (the two modes never share a launch)
  double2* const logtab = reinterpret_cast<double2*>((reinterpret_cast<uintptr_t>(lutS + A.lut_len + ((LT && !(FMT == 1 && A.lq_lin)) ? A.lq_n : 0)) + 15) & ~(uintptr_t)15);
  if ((LNL1 || LAG || SPL) && tid < FZ_LOGTAB) {
    const double ci = 1.0 + (double)tid * (1.0 / FZ_LOGTAB);
    logtab[tid] = make_double2(1.0 / ci, ts_log1p_pos((double)tid * (1.0 / FZ_LOGTAB)));
  }
  __syncthreads();
  if (sprof) sprof[2] = wall_clock64();
  const int ticket = __builtin_amdgcn_readfirstlane(ibox[0]);   // LDS broadcasts: tell the compiler they are uniform
  const int u = ticket / P, p = ticket % P;
  if (SPL) {                                               // one half of the part's columns: previous and current pi*theta
    for (int t = tid; t < KT; t += FZ_NT) {
      const bool in = A.koff + t < Kp;
      c[t] = in ? A.ctab[p * Kp + A.koff + t] : 0.0;
      acc[t] = in ? A.ctab2[p * Kp + A.koff + t] : 0.0;
    }
  } else if (!SPB) {
    for (int t = tid; t < Kp; t += FZ_NT) c[t] = A.ctab[p * Kp + t];
  }
  if (LNL1) {                                              // the table the log1p is taken of: pi*theta of the current parameters, or its logarithm
    const double* const src = LT ? A.lctab : A.ctab2;
    for (int t = tid; t < Kp; t += FZ_NT) acc[t] = src[p * Kp + t];
  }
  if (LAG)
    for (int t = tid; t < Kp; t += FZ_NT) cprev[t] = A.ctab2[p * Kp + t];
  if (EXACT)
    for (int t = tid; t < Kp; t += FZ_NT) eS[t] = A.ebias[p * Kp + t];
  if (tid == 0) {
    unsigned spins = 0;
    for (;;) {                                             // every workgroup has taken its ticket
      unsigned tot = 0;
      int cnt[8];
#pragma unroll
      for (int x = 0; x < 8; ++x) { cnt[x] = (int)fz_ld_u32(&sync[x]); tot += (unsigned)cnt[x]; }
      if (tot >= gridDim.x) {
#pragma unroll
        for (int x = 0; x < 8; ++x) ibox[1 + x] = cnt[x];
        break;
      }
      __builtin_amdgcn_s_sleep(2);
      if (++spins > FZ_SPIN_LIMIT) { atomicOr(err, 1u); break; }
      if ((spins & 15u) == 0 && fz_ld_u32(err)) break;
    }
    if (sprof) sprof[1] = wall_clock64();
  }
  __syncthreads();
  if (sprof) sprof[3] = wall_clock64();
  int T = 0, tbase = 0;
  for (int x = 0; x < 8; ++x) {
    int teams = __builtin_amdgcn_readfirstlane(ibox[1 + x]) / P;
    if (x < (int)xcc) tbase += teams;
    T += teams;
  }
  const bool valid = (u + 1) * P <= __builtin_amdgcn_readfirstlane(ibox[1 + xcc]) && fz_ld_u32(err) == 0;
  if (!valid || T == 0) return;                          // leftover workgroup of an incomplete team
  const int team = tbase + u;                            // 0..T-1
  unsigned long long* const xbase = reinterpret_cast<unsigned long long*>(A.xchg) + (int64_t)team * FZ_XS * P * R;

  // Block k of this team is row block team + k*T.  Schedule of one block:
  //   step k-2 : prefetch burst          load(k)            -> register set k % 6
  //   step k   : row sums                P1(k)              -> y[k & 3]
  //   step k+1 : publish                 granules of y(k)   (exchange waves)
  //   step k+3 : partner loads + combine s[k & 1] = w / sum_q y_q ; y[k & 3] = 0   (exchange waves, same step)
  //   step k+4 : scatter                 P2(k), set k % 6 is then refilled with block k+6
  // one barrier per step.
  const int64_t nblk = (A.nb > team) ? (A.nb - team + T - 1) / T : 0;
  // Role split: the last 2-3 waves are exchange waves (they hold no matrix entries, so they can afford
  // the registers for the partner values); the others are data waves.  A CU's vector-memory pipe
  // returns in order, so every exchange access is issued right after the barrier: ahead of the data
  // waves' burst of this step (which they issue at its end), behind the burst of the previous one.
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  auto slot_of = [&](int64_t k, int q) -> unsigned long long* {
    return xbase + ((int64_t)(k & (FZ_XS - 1)) * P + q) * R;
  };
  auto tag_of = [&](int64_t k) -> unsigned long long {   // slot k % XS is rewritten every XS blocks; first tag is 1
    return (unsigned long long)((((k / FZ_XS) & 1) ^ 1));
  };
  const int64_t nsteps = nblk + FZ_LAG + 1;               // last scatter is block nblk-1 at step nblk+3
  // prologue: offsets of blocks 0..4 straight into LDS
  if (tid < 10) {
    const int64_t k = tid >> 1;
    if (k < nblk) offs[tid] = A.sb_q32[(team + k * T) * P + p + (tid & 1)];
  }
  if (LAG && A.lag && nblk > 0)
    for (int t = tid; t < R; t += FZ_NT) rpS[t] = A.rinv[(int64_t)team * R + t];
  __syncthreads();
  if (sprof) { sprof[4] = wall_clock64(); sprof[7] = ((unsigned long long)xcc << 48) | ((unsigned long long)team << 32) | ((unsigned long long)p << 16) | (unsigned long long)nblk; }

  double lsum = 0.0;                                      // lnl mode: this thread's share of the sum
#ifdef FZ_EXPERIMENT
  const unsigned long long fz_t0 = clock64();
#endif
  if (tid >= FZ_DT) {
    // ============================ exchange wave ===============================
    // dispatched on the member index so every register array is statically indexed
    FzX X;
    X.lut = lutS; X.y = y; X.s = s; X.rp = rpS; X.offs = offs; X.xbase = xbase; X.err = err; X.R = R; X.team = team; X.T = T;
    X.nblk = nblk; X.nsteps = nsteps; X.lane = (tid - FZ_DT) & 63; X.xw = (tid - FZ_DT) >> 6;
    switch (p) {
      case 0: fz_xchg<P, 0, MODE, FMT, GEO>(A, X); break;
      case 1: if (P > 1) fz_xchg<P, (P > 1 ? 1 : 0), MODE, FMT, GEO>(A, X); break;
      case 2: if (P > 2) fz_xchg<P, (P > 2 ? 2 : 0), MODE, FMT, GEO>(A, X); break;
      case 3: if (P > 3) fz_xchg<P, (P > 3 ? 3 : 0), MODE, FMT, GEO>(A, X); break;
      case 4: if (P > 4) fz_xchg<P, (P > 4 ? 4 : 0), MODE, FMT, GEO>(A, X); break;
      case 5: if (P > 5) fz_xchg<P, (P > 5 ? 5 : 0), MODE, FMT, GEO>(A, X); break;
      case 6: if (P > 6) fz_xchg<P, (P > 6 ? 6 : 0), MODE, FMT, GEO>(A, X); break;
      case 7: if (P > 7) fz_xchg<P, (P > 7 ? 7 : 0), MODE, FMT, GEO>(A, X); break;
      default: break;
    }
  } else {
    // ============================== data waves ================================
    // Loads of block k: resources rebased to the sub-block, so thread t reads quad t at a constant
    // offset and every lane past the sub-block's end (or a whole step past the last block) reads zeros.
    auto load_blk = [&](FzRegs& rr, uint32_t oq0v, uint32_t oq1v, int64_t k) {
      const uint32_t oq0 = __builtin_amdgcn_readfirstlane(oq0v);
      const uint32_t nq = (k >= 0 && k < nblk) ? __builtin_amdgcn_readfirstlane(oq1v) - oq0 : 0u;
      fz_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(fz_rsrc(A.prc, (uint64_t)oq0 * 16, nq * 16), (unsigned)tid * 16, 0, FZ_STREAM);
      rr.rc = make_uint4(t.x, t.y, t.z, t.w);
      if (FMT == 1) {
        fz_u32x2 cd = __builtin_amdgcn_raw_buffer_load_b64(fz_rsrc(A.pcode, (uint64_t)oq0 * 8, nq * 8), (unsigned)tid * 8, 0, FZ_STREAM);
        rr.cd = make_uint2(cd.x, cd.y);
      } else {
        __amdgpu_buffer_rsrc_t vr = fz_rsrc(A.pval, (uint64_t)oq0 * 32, nq * 32);
        rr.v0 = fz_as_double2(__builtin_amdgcn_raw_buffer_load_b128(vr, (unsigned)tid * 32, 0, FZ_STREAM));
        rr.v1 = fz_as_double2(__builtin_amdgcn_raw_buffer_load_b128(vr, (unsigned)tid * 32 + 16, 0, FZ_STREAM));
      }
    };
    // phase 1: numerators n = Q * (pi*theta) kept in the registers, partial row sums into y(k).
    // A lane whose quad is all zeros (past the end of the sub-block, or a step without a block) marks
    // itself idle in rc.x and skips both phases: zeros added to y[0] / acc[0] by every idle lane
    // would serialise on one LDS address.
    constexpr bool lnl = LNL1 || SPL;
    auto phase1 = [&](FzRegs& rr, int64_t k) {
      const bool idle = FMT == 1 ? (rr.cd.x | rr.cd.y) == 0u
                                 : (rr.v0.x == 0.0) & (rr.v0.y == 0.0) & (rr.v1.x == 0.0) & (rr.v1.y == 0.0);
      if (SPL) { if (idle) rr.rc.x = 0xFFFFFFFFu; return; }    // (the row factors come from pass A: nothing to sum)
      if (__builtin_amdgcn_ballot_w64(!idle) == 0) { rr.rc.x = 0xFFFFFFFFu; return; }   // whole wave idle
      double* yb = y + (k & (FZ_YR - 1)) * R;
      double m0 = 0.0, m1 = 0.0, m2 = 0.0, m3 = 0.0;
      if (!idle) {
        double2 q0 = rr.v0, q1 = rr.v1;
        if (FMT == 1) {                                   // Q from the score table: the same fp64 the fp64 layout stores
          q0 = make_double2(lutS[rr.cd.x & 0xFFFFu], lutS[rr.cd.x >> 16]);
          q1 = make_double2(lutS[rr.cd.y & 0xFFFFu], lutS[rr.cd.y >> 16]);
        }
        m0 = q0.x * c[rr.rc.x & 0xFFFF]; m1 = q0.y * c[rr.rc.y & 0xFFFF];
        m2 = q1.x * c[rr.rc.z & 0xFFFF]; m3 = q1.y * c[rr.rc.w & 0xFFFF];
        // EM: the set keeps the numerators for phase 2.  lnl: phase 2 needs Q itself (twice) — the fp64
        // layout has it in the set already, the code layout looks it up again (2 registers per set
        // instead of 8: with four log1p expansions in flight the lnl kernel would spill otherwise)
        if (!lnl) { rr.v0 = make_double2(m0, m1); rr.v1 = make_double2(m2, m3); }
      }
      if (A.sorted) {
        fz_row_sums(yb, idle, rr.rc.x >> 16, rr.rc.y >> 16, rr.rc.z >> 16, rr.rc.w >> 16, m0, m1, m2, m3);
      } else if (!idle) {                                  // strand-transposed order: neighbouring entries never share a row
        lds_add(&yb[rr.rc.x >> 16], m0); lds_add(&yb[rr.rc.y >> 16], m1);
        lds_add(&yb[rr.rc.z >> 16], m2); lds_add(&yb[rr.rc.w >> 16], m3);
      }
      if (idle) rr.rc.x = 0xFFFFFFFFu;
    };
    // phase 2: scatter w * z into the part's column accumulators
    auto phase2 = [&](FzRegs& rr, int64_t k) {
      if (rr.rc.x == 0xFFFFFFFFu) return;
      const double* sb = s + (k & 1) * R;
      if (lnl) {                                          // z = (Q c_prev) * recip0(rowsum);  acc[] holds c_cur
        auto term = [&](double q, uint32_t rc) {
          if (SPL) {                                      // only the entries whose column lies in this launch's half
            const uint32_t j = (rc & 0xFFFFu) - (uint32_t)A.koff;
            if (j < (uint32_t)KT) {
              const double z = (q * c[j]) * sb[rc >> 16];
              if (z != 0.0) lsum += z * fz_log1p_tab(q * acc[j], logtab);
            }
            return;
          }
          const double z = (q * c[rc & 0xFFFF]) * sb[rc >> 16];
          if (z != 0.0) lsum += z * fz_log1p_tab(q * acc[rc & 0xFFFF], logtab);
        };
        if constexpr (LT) {
          // log tables: acc[] holds log(pi*theta) of the current parameters, lqS[] log Q: an addition instead of a logarithm per
          // entry.  The gathers of several entries first — one LDS round trip (each term ends in a wave-uniform branch, across which
          // the compiler moves no load: term by term it was four round trips) — then their arithmetic: all four entries with score
          // codes, two and two with fp64 entries (whose six register sets of 12 leave no room for four: 56 VGPRs spilled).
          const double* const cg = A.ctab2 + p * Kp;     // pi*theta of the current parameters in global memory (the rare range only)
          auto pair = [&](double qa, double qb, double la, double lb, uint32_t rca, uint32_t rcb) {
            const uint32_t ja = rca & 0xFFFF, jb = rcb & 0xFFFF;
            const double za = (qa * c[ja]) * sb[rca >> 16], zb = (qb * c[jb]) * sb[rcb >> 16];
            const double La = la + acc[ja], Lb = lb + acc[jb];
            __builtin_amdgcn_sched_barrier(0);
            lsum = fma(za, fz_log1p_of_log<true>(La, [&]() { return qa * cg[ja]; }, logtab, false, za != 0.0), lsum);   // (always finite; padding has z = 0)
            lsum = fma(zb, fz_log1p_of_log<true>(Lb, [&]() { return qb * cg[jb]; }, logtab, false, zb != 0.0), lsum);
          };
          if (FMT == 1) {
            const uint32_t k0 = rr.cd.x & 0xFFFFu, k1 = rr.cd.x >> 16, k2 = rr.cd.y & 0xFFFFu, k3 = rr.cd.y >> 16;
            const uint32_t j0 = rr.rc.x & 0xFFFF, j1 = rr.rc.y & 0xFFFF, j2 = rr.rc.z & 0xFFFF, j3 = rr.rc.w & 0xFFFF;
            const double q0 = lutS[k0], q1 = lutS[k1], q2 = lutS[k2], q3 = lutS[k3];
            const double z0 = (q0 * c[j0]) * sb[rr.rc.x >> 16], z1 = (q1 * c[j1]) * sb[rr.rc.y >> 16];
            const double z2 = (q2 * c[j2]) * sb[rr.rc.z >> 16], z3 = (q3 * c[j3]) * sb[rr.rc.w >> 16];
            const bool lin = A.lq_lin != 0;                // (a kernel argument: wave-uniform)
            double l0, l1, l2, l3;
            bool f0 = false, f1 = false, f2 = false, f3 = false;
            if (lin) {                                     // log Q = (code * (1 / max)) * scale: no gather
              l0 = ((double)k0 * A.lq_a) * A.lq_b; l1 = ((double)k1 * A.lq_a) * A.lq_b;
              l2 = ((double)k2 * A.lq_a) * A.lq_b; l3 = ((double)k3 * A.lq_a) * A.lq_b;
              f0 = (int)k0 < A.lq_c0; f1 = (int)k1 < A.lq_c0; f2 = (int)k2 < A.lq_c0; f3 = (int)k3 < A.lq_c0;   // (the padding's code 0 has z = 0: not live)
            } else { l0 = lqS[k0]; l1 = lqS[k1]; l2 = lqS[k2]; l3 = lqS[k3]; }
            const double L0 = l0 + acc[j0], L1 = l1 + acc[j1], L2 = l2 + acc[j2], L3 = l3 + acc[j3];
            __builtin_amdgcn_sched_barrier(0);
            lsum = fma(z0, fz_log1p_of_log<true>(L0, [&]() { return q0 * cg[j0]; }, logtab, f0, z0 != 0.0), lsum);
            lsum = fma(z1, fz_log1p_of_log<true>(L1, [&]() { return q1 * cg[j1]; }, logtab, f1, z1 != 0.0), lsum);
            lsum = fma(z2, fz_log1p_of_log<true>(L2, [&]() { return q2 * cg[j2]; }, logtab, f2, z2 != 0.0), lsum);
            lsum = fma(z3, fz_log1p_of_log<true>(L3, [&]() { return q3 * cg[j3]; }, logtab, f3, z3 != 0.0), lsum);
          } else {
            pair(rr.v0.x, rr.v0.y, fz_logq_of(rr.v0.x, lqS, A.lq_n, A.lq_shift, A.lq_base), fz_logq_of(rr.v0.y, lqS, A.lq_n, A.lq_shift, A.lq_base),
                 rr.rc.x, rr.rc.y);
            pair(rr.v1.x, rr.v1.y, fz_logq_of(rr.v1.x, lqS, A.lq_n, A.lq_shift, A.lq_base), fz_logq_of(rr.v1.y, lqS, A.lq_n, A.lq_shift, A.lq_base),
                 rr.rc.z, rr.rc.w);
          }
          return;
        }
        if (FMT == 1) {
          term(lutS[rr.cd.x & 0xFFFFu], rr.rc.x); term(lutS[rr.cd.x >> 16], rr.rc.y);
          term(lutS[rr.cd.y & 0xFFFFu], rr.rc.z); term(lutS[rr.cd.y >> 16], rr.rc.w);
        } else {
          term(rr.v0.x, rr.rc.x); term(rr.v0.y, rr.rc.y); term(rr.v1.x, rr.rc.z); term(rr.v1.y, rr.rc.w);
        }
        return;
      }
      // all four gathers first: an LDS atomic may alias a later LDS read as far as the compiler knows,
      // so `add(acc, v * s[..])` four times in a row serialises gather -> wait -> atomic -> gather ...
      const double s0 = sb[rr.rc.x >> 16], s1 = sb[rr.rc.y >> 16], s2 = sb[rr.rc.z >> 16], s3 = sb[rr.rc.w >> 16];
      lds_add(&acc[rr.rc.x & 0xFFFF], rr.v0.x * s0);
      lds_add(&acc[rr.rc.y & 0xFFFF], rr.v0.y * s1);
      lds_add(&acc[rr.rc.z & 0xFFFF], rr.v1.x * s2);
      lds_add(&acc[rr.rc.w & 0xFFFF], rr.v1.y * s3);
    };
    FzRegs r0, r1, r2, r3, r4, r5;
    {
      FzRegs z;                                           // sets that hold no block yet are idle
      z.rc = make_uint4(0xFFFFFFFFu, 0, 0, 0); z.v0 = z.v1 = make_double2(0.0, 0.0); z.cd = make_uint2(0, 0);
      r0 = r1 = r2 = r3 = r4 = r5 = z;
    }
#if FZ_GAP_STEPS == 2
    FzRegs r6 = r0;
#endif
    int64_t i = 0;
    // step i (lnl pass): `rs` is the set of block i-LAG (scattered, then refilled with block i+2); `rp` the set of block i
    auto step_lnl = [&](FzRegs& rs, FzRegs& rp) {
      const uint32_t oq0 = offs[((i + FZ_DL) & 7) * 2], oq1 = offs[((i + FZ_DL) & 7) * 2 + 1];
      phase2(rs, i - FZ_LAG);
      load_blk(rs, oq0, oq1, i + FZ_DL);
      phase1(rp, i);                                      // waits for burst(i), issued two steps ago
      __syncthreads();
      ++i;
    };
    // step i (EM pass).  The LDS is the busiest unit of the step (SQ_LDS_IDX_ACTIVE 70-82 % of the kernel time,
    // profiles/r02_lds_counters.txt) and its queue is deep, so every LDS round trip a wave WAITS for costs
    // hundreds of cycles.  Hence: (1) all gathers of the step — row factors s of block i-LAG, score table and
    // pi*theta entries of block i, the offsets of the next burst — are issued back to back at the top: ONE
    // round trip per step instead of three; (2) the row-sum atomics of block i go first and the four column
    // scatters of block i-LAG LAST, unconditionally (idle lanes send 0.0 to a private dummy slot), and the wave
    // enters the barrier as soon as all but those four have completed (`s_waitcnt lgkmcnt(4)`; LDS operations of
    // a wave complete in order): nobody reads the column accumulators before the kernel ends, so the scatters
    // drain behind the barrier while the next step's gathers queue up, and the LDS never runs dry at a barrier.
    uint32_t oqa = offs[FZ_DL * 2], oqb = offs[FZ_DL * 2 + 1];            // offsets of burst(DL), used at step 0
    const int lane_id = tid & 63;
    auto step_em = [&](FzRegs& rs, FzRegs& rp) {
      const bool pr = A.prof && team == 0 && p == 0 && tid == 0 && (int)i < A.prof_blocks;
      if (pr) A.prof[i * FZ_PROF_SLOTS + 0] = clock64();
      // (Round 4: the exchange wave's publish is late — 900 instead of 200 clk — in the steps where its two LDS reads of y(i-1) queue
      // behind the data waves' gather burst.  `s_sleep` of 64 / 192 clk here, to let them in first: 18 per row 1.70 = 1.70 / 1.78 ms
      // (codes), 2.20 / 2.24 / 2.25 (fp64 entries); 40 per row 3.35 / 3.37 / 3.51 — nothing, then slower.  profiles/r04_ab_delay.txt)
      const int64_t k2 = i - FZ_LAG;
      const uint32_t on0 = offs[((i + FZ_DL + 1) & 7) * 2], on1 = offs[((i + FZ_DL + 1) & 7) * 2 + 1];   // burst of the NEXT step
      const bool idle2 = rs.rc.x == 0xFFFFFFFFu;
      // (padding has code 0 / value 0 -> numerator 0: no branch needed around the products)
      const bool idle = FMT == 1 ? (rp.cd.x | rp.cd.y) == 0u
                                 : (rp.v0.x == 0.0) & (rp.v0.y == 0.0) & (rp.v1.x == 0.0) & (rp.v1.y == 0.0);
      // A WAVE whose lanes hold nothing in either set — the tail waves of a sub-block that short rows cannot fill: at
      // 10 entries per row 768 row slots fill 58 % of the register tile — skips the step's LDS work altogether.  The
      // step has a floor of LDS instruction ISSUE (~22 wave-instructions per data wave, ~8 clk each even with every lane
      // masked, profiles/HISTORY.md 9.2); idle waves used to pay it in full.  (Wave-uniform branch around LDS operations only: the
      // streaming loads below stay unconditional.)
      const bool wave_idle = FZ_SKIP_IDLE_WAVES && GEO >= 2 && !LAG &&   // (only the short-row geometry leaves whole waves idle; elsewhere the branch costs 1.5 %; MODE 4: it costs registers the log1p needs)
                             (__builtin_amdgcn_ballot_w64(!idle) | __builtin_amdgcn_ballot_w64(!idle2)) == 0ull;
      if (wave_idle) {
        rp.rc.x = 0xFFFFFFFFu;
        if (pr) { A.prof[i * FZ_PROF_SLOTS + 1] = clock64(); A.prof[i * FZ_PROF_SLOTS + 3] = clock64(); }
      } else {
      // ---- gathers ----
      const uint32_t a0 = idle2 ? 0u : rs.rc.x, a1 = rs.rc.y, a2 = rs.rc.z, a3 = rs.rc.w;
      const double* sb = s + (k2 & 1) * R;
      // (Row order: gathering only the outer two row factors of a lane when no lane of the wave spans three rows
      // — two LDS reads less per lane — was tried twice: with the select at the top (a second LDS round trip) 4.20 ->
      // 4.30 ms, with the select deferred to phase 2 3.57 -> 3.73 ms (codes) / 4.09 -> 4.19 ms: the ballot, the
      // compares and the selects cost more issue slots than the two broadcast reads they save.)
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
      if (!SPA) { s0 = sb[a0 >> 16]; s1 = sb[a1 >> 16]; s2 = sb[a2 >> 16]; s3 = sb[a3 >> 16]; }
      double2 q0 = rp.v0, q1 = rp.v1;
#ifdef FZ_EXPERIMENT   // upper bounds (WRONG RESULTS): what conflict-free LDS accesses would buy, per kind of access
      const uint32_t xm_l = (A.dbg & 128) ? 3u : 0xFFFFu, xm_c = (A.dbg & 256) ? 31u : 0xFFFFu;
#else
      constexpr uint32_t xm_l = 0xFFFFu, xm_c = 0xFFFFu;
#endif
      if (FMT == 1) {                                     // Q from the score table: the same fp64 the fp64 layout stores
#ifdef FZ_X_NOLUT   // upper bound (WRONG RESULTS): the step without its four score-table gathers (a conversion on the VALU instead)
        q0 = make_double2((double)(rp.cd.x & 0xFFFFu), (double)(rp.cd.x >> 16));
        q1 = make_double2((double)(rp.cd.y & 0xFFFFu), (double)(rp.cd.y >> 16));
#else
        q0 = make_double2(lutS[rp.cd.x & xm_l], lutS[(rp.cd.x >> 16) & xm_l]);
        q1 = make_double2(lutS[rp.cd.y & xm_l], lutS[(rp.cd.y >> 16) & xm_l]);
#endif
      }
      double c0 = 1.0, c1 = 1.0, c2 = 1.0, c3 = 1.0;           // (MODE 7 scatters Q * s: the column's pi*theta is applied by k_colreduce)
      if (!SPB) { c0 = c[rp.rc.x & xm_c]; c1 = c[rp.rc.y & xm_c]; c2 = c[rp.rc.z & xm_c]; c3 = c[rp.rc.w & xm_c]; }
      // ---- phase 1 of block i: numerators stay in the set, partial row sums into y(i) ----
      const double m0 = q0.x * c0, m1 = q0.y * c1, m2 = q1.x * c2, m3 = q1.y * c3;
      rp.v0 = make_double2(m0, m1); rp.v1 = make_double2(m2, m3);
      double* yb = y + (i & (FZ_YR - 1)) * R;
      if (pr) A.prof[i * FZ_PROF_SLOTS + 1] = clock64();
      if (SPB) {
        // no row sums in the scatter pass
      } else if (A.sorted) {
        fz_row_sums(yb, idle, rp.rc.x >> 16, rp.rc.y >> 16, rp.rc.z >> 16, rp.rc.w >> 16, m0, m1, m2, m3);
      } else {                                            // strand-transposed order: neighbouring entries never share a row
        double* d = dum + lane_id;
        lds_add(idle ? d : &yb[rp.rc.x >> 16], m0); lds_add(idle ? d : &yb[rp.rc.y >> 16], m1);
        lds_add(idle ? d : &yb[rp.rc.z >> 16], m2); lds_add(idle ? d : &yb[rp.rc.w >> 16], m3);
      }
      if (idle) rp.rc.x = 0xFFFFFFFFu;
      if (pr) A.prof[i * FZ_PROF_SLOTS + 3] = clock64();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      // ---- phase 2 of block i-LAG: w*z into the part's column accumulators; ALWAYS four atomics, issued last ----
      if (!SPA) {
        const uint32_t dj = (uint32_t)(dum - acc) + (uint32_t)lane_id;   // the lane's dummy slot as an index into acc[]
#ifdef FZ_EXPERIMENT
        const bool xa = (A.dbg & 512) != 0;                // every lane its own slot: no conflicts, no shared addresses
        const uint32_t j0 = (idle2 | xa) ? dj : (a0 & 0xFFFFu), j1 = (idle2 | xa) ? dj : (a1 & 0xFFFFu);
        const uint32_t j2 = (idle2 | xa) ? dj : (a2 & 0xFFFFu), j3 = (idle2 | xa) ? dj : (a3 & 0xFFFFu);
#else
        const uint32_t j0 = idle2 ? dj : (a0 & 0xFFFFu), j1 = idle2 ? dj : (a1 & 0xFFFFu);
        const uint32_t j2 = idle2 ? dj : (a2 & 0xFFFFu), j3 = idle2 ? dj : (a3 & 0xFFFFu);
#endif
        if (EXACT) {
          // Exact accumulation (Demmel-Nguyen style pre-rounding on a per-slot grid): v = hi + lo + rest with hi a multiple of
          // 2^(E-30) and lo a multiple of 2^(E-60); the sums of the hi pieces (and of the lo pieces) of up to 2^23 contributions
          // below 2^E are exact in fp64, hence the same whatever order the LDS serves the atomics in.  MODE 2: one piece per
          // pass (A.bin), one accumulator table.  MODE 3: both pieces in this pass, two tables (when three tables fit the LDS).
          auto pieces = [&](double v, uint32_t j, bool live, double& hi, double& lo) {
            const uint32_t eb = live ? (uint32_t)eS[j] : 1023u;
            if (live && (uint32_t)(__double2hiint(v) >> 20) >= eb) A.ovf[p * Kp + j] = 1;    // v >= 2^E: the bound was too low
            const double m1 = __hiloint2double((int)(((eb + 22u) << 20) | 0x80000u), 0);    // 1.5 * 2^(E+22): ulp = 2^(E-30)
            hi = (v + m1) - m1;
            const double m2 = __hiloint2double((int)(((eb - 8u) << 20) | 0x80000u), 0);     // 1.5 * 2^(E-8):  ulp = 2^(E-60)
            lo = ((v - hi) + m2) - m2;
          };
          double h0, h1, h2, h3, l0, l1, l2, l3;
          pieces(rs.v0.x * s0, j0, !idle2, h0, l0); pieces(rs.v0.y * s1, j1, !idle2, h1, l1);
          pieces(rs.v1.x * s2, j2, !idle2, h2, l2); pieces(rs.v1.y * s3, j3, !idle2, h3, l3);
          if (MODE == 3) {
            // (idle lanes: both zeros go to the lane's dummy slot of acc[] — acc2 + dj would point past the dummy slots)
            double* const b2 = idle2 ? acc : acc2;
            lds_add(&acc[j0], h0); lds_add(&acc[j1], h1); lds_add(&acc[j2], h2); lds_add(&acc[j3], h3);
            lds_add(&b2[j0], l0); lds_add(&b2[j1], l1); lds_add(&b2[j2], l2); lds_add(&b2[j3], l3);
          } else {
            const bool first = A.bin == 1;
            lds_add(&acc[j0], first ? h0 : l0); lds_add(&acc[j1], first ? h1 : l1);
            lds_add(&acc[j2], first ? h2 : l2); lds_add(&acc[j3], first ? h3 : l3);
          }
        } else {
        lds_add(&acc[j0], rs.v0.x * s0);
        lds_add(&acc[j1], rs.v0.y * s1);
        lds_add(&acc[j2], rs.v1.x * s2);
        lds_add(&acc[j3], rs.v1.y * s3);
        }
      }
      if (LAG) {
        // log-likelihood of the PREVIOUS iteration (model.py:744-760): z = (Q c_prev) * recip0(rowsum_prev), times log1p(Q c) — the
        // numerators this pass has just formed.  Padding has Q = 0 -> z = 0 (no `z != 0` branch: log1p of a finite product is
        // finite).  AFTER the scatter of block i-LAG, whose register set and row factors are dead by now, and one entry after the
        // other: the kernel sits at 128 VGPRs (1024 threads), and a spill in this loop is a scratch access in the in-order memory
        // pipe behind the stream; score codes are looked up again instead of keeping Q alive across the row sums.  `idle` lanes
        // have marked rc.x by now: their Q is 0, and lane-private garbage in rc.x >> 16 must not index LDS -> masked.
        const double* rb = rpS + (i & 1) * R;
        const uint32_t e0 = idle ? 0u : rp.rc.x, e1 = rp.rc.y, e2 = rp.rc.z, e3 = rp.rc.w;
        // all gathers of the four entries first (one LDS round trip), then the four evaluations, each PINNED where it is written:
        // left alone, the compiler sinks the arithmetic below the barrier (nothing but `lsum` at the very end needs it), keeps the
        // twelve gathered values alive into the next step and spills the streaming loads' destinations
        auto term = [&](double q, double cp, double rf, double m) {
          lsum = fma((q * cp) * rf, fz_log1p_tab<FZ_LAG_CERR != 0>(m, logtab), lsum);
          asm volatile("" : "+v"(lsum));
          __builtin_amdgcn_sched_barrier(0);
        };
        if (FMT == 1) {
          const double g0 = lutS[rp.cd.x & 0xFFFFu], g1 = lutS[rp.cd.x >> 16], g2 = lutS[rp.cd.y & 0xFFFFu], g3 = lutS[rp.cd.y >> 16];
          const double p0 = cprev[e0 & 0xFFFF], p1 = cprev[e1 & 0xFFFF], p2 = cprev[e2 & 0xFFFF], p3 = cprev[e3 & 0xFFFF];
          const double f0 = rb[e0 >> 16], f1 = rb[e1 >> 16], f2 = rb[e2 >> 16], f3 = rb[e3 >> 16];
          term(g0, p0, f0, m0); term(g1, p1, f1, m1); term(g2, p2, f2, m2); term(g3, p3, f3, m3);
        } else {                                          // fp64 entries keep Q alive (8 VGPRs): two entries per round trip
          {
            const double p0 = cprev[e0 & 0xFFFF], p1 = cprev[e1 & 0xFFFF], f0 = rb[e0 >> 16], f1 = rb[e1 >> 16];
            term(q0.x, p0, f0, m0); term(q0.y, p1, f1, m1);
          }
          {
            const double p2 = cprev[e2 & 0xFFFF], p3 = cprev[e3 & 0xFFFF], f2 = rb[e2 >> 16], f3 = rb[e3 >> 16];
            term(q1.x, p2, f2, m2); term(q1.y, p3, f3, m3);
          }
        }
      }
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      // refill the set at the END of the step.  (Issuing the burst in the middle — products and slots of phase 2 in
      // temporaries — so that it drains before the next step's exchange loads was tried: fp64 entries 4.4 -> 5.2 ms.)
      load_blk(rs, oqa, oqb, i + FZ_DL);
      if (pr) A.prof[i * FZ_PROF_SLOTS + 2] = clock64();
      if (A.prof && team == 0 && p == 0 && tid == FZ_DT - 64 && (int)i < A.prof_blocks) A.prof[i * FZ_PROF_SLOTS + 8] = clock64();
      __builtin_amdgcn_s_waitcnt(MODE == 3 ? 0xC87F : (SPA ? 0xC07F : 0xC47F));   // lgkmcnt(4): everything but the four (MODE 3: eight; MODE 5: no) scatters above has completed
      oqa = __builtin_amdgcn_readfirstlane(on0); oqb = __builtin_amdgcn_readfirstlane(on1);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (pr) A.prof[i * FZ_PROF_SLOTS + 4] = clock64();
      ++i;
    };
    auto step = [&](FzRegs& rs, FzRegs& rp) {
      if (LNL1 || SPL) step_lnl(rs, rp); else step_em(rs, rp);
    };
    load_blk(r0, offs[0], offs[1], 0);
    load_blk(r1, offs[2], offs[3], 1);
#if FZ_GAP_STEPS == 1
    static_assert(FZ_NS == 6 && FZ_LAG == 4 && FZ_DL == 2, "ring unrolling below assumes 6 sets");
    while (i < nsteps) {                                  // block k lives in set k % 6; (i-4) % 6 == (i+2) % 6
      step(r2, r0); if (i >= nsteps) break;               // i % 6 == 0
      step(r3, r1); if (i >= nsteps) break;
      step(r4, r2); if (i >= nsteps) break;
      step(r5, r3); if (i >= nsteps) break;
      step(r0, r4); if (i >= nsteps) break;
      step(r1, r5);
    }
#else
    static_assert(FZ_NS == 7 && FZ_LAG == 5 && FZ_DL == 2, "ring unrolling below assumes 7 sets");
    while (i < nsteps) {                                  // block k lives in set k % 7; (i-5) % 7 == (i+2) % 7
      step(r2, r0); if (i >= nsteps) break;               // i % 7 == 0
      step(r3, r1); if (i >= nsteps) break;
      step(r4, r2); if (i >= nsteps) break;
      step(r5, r3); if (i >= nsteps) break;
      step(r6, r4); if (i >= nsteps) break;
      step(r0, r5); if (i >= nsteps) break;
      step(r1, r6);
    }
#endif
  }
  __syncthreads();
  if (sprof) sprof[5] = wall_clock64();
  if (LNL1 || LAG || SPL) {                               // one partial per workgroup, summed by k_sum_parts / k_colreduce
    for (int o = 32; o > 0; o >>= 1) lsum += __shfl_down(lsum, o, 64);
    double* wsum = y;                                     // the y ring is idle now
    if ((tid & 63) == 0) wsum[tid >> 6] = lsum;
    __syncthreads();
    if (tid == 0) {
      double t = 0.0;
      for (int w = 0; w < FZ_NT / 64; ++w) t += wsum[w];
      A.lnl_out[team * P + p] = t;
    }
    if (LNL1 || SPL) return;
  }
  if (SPA) return;                                        // the row factors are in A.rinv; no column sums from this pass
#ifdef FZ_EXPERIMENT
  if ((A.dbg & 4096) && A.prof && tid == 0) {             // per-member loop time (cycles) and blocks: prof[(team*P+p)*2 ..]
    A.prof[(team * P + p) * 2] = clock64() - fz_t0;
    A.prof[(team * P + p) * 2 + 1] = (unsigned long long)nblk;
  }
#endif
  double* out = A.partial + (int64_t)team * (P * Kp) + p * Kp;
  for (int t = tid; t < Kp; t += FZ_NT) out[t] = acc[t];
  if (MODE == 3) {
    double* out2 = A.partial2 + (int64_t)team * (P * Kp) + p * Kp;
    for (int t = tid; t < Kp; t += FZ_NT) out2[t] = acc2[t];
  }
  if (sprof) sprof[6] = wall_clock64();
}
