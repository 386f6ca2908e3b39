// The report pass over the FINAL z without padded lanes and without fp64 (round 6; included by tsem_report.hip after ReportArgs).
//
// k_report_rows gives a row G lanes of E entries with G a power of two chosen for the whole matrix: at 40 entries per row and a
// capacity of 64, 38 % of the lane-entries are padding, and the kernel is bound by instruction issue (37 VALU per lane-entry,
// profiles/r05_report_pmc.txt) — its time follows the capacity, not the entries.  Two changes, one kernel (k_report_pack32):
//
// 1. PACKING.  A row takes ceil(len / E) lanes (E = 16 entries per lane, 8 where the rows are short), rows follow each other without
//    gaps inside a wave, and a wave's work is a CHUNK of whole rows that fills 64 lanes (k_report_chunks packs them greedily once per
//    matrix: ~9 % padding in the rows' last lanes + ~4 % at the chunks' ends).  Lane -> row: the chunk's descriptor carries a 64-bit
//    mask of the lanes that start a row; a lane's row is the number of such lanes at or below it (v_bcnt), its place inside the row
//    the distance to the nearest one (v_ffbh).  Descriptors are fetched three iterations ahead, row pointers two, entries one.  A row's
//    (sum, maximum, runner-up, id of the maximum) is a segmented scan over its lanes on the VALU (rp_scan_step); the row's LAST lane
//    holds the totals and decides.
//    (History, all in profiles/r06_report_*: packed rows with the fp64 arithmetic of k_report_rows 4.86 against 5.67 ms — bound by the
//    fp64 per-entry work, the L2 gathers of the cold pi*theta and the winners' atomics; the fp32 filter with LDS atomics per row for
//    sum / max / count 3.16 ms — LDS-bound; the scan 2.4; 16 entries per lane, integer max, LDS mask table 2.2; counted waits and the
//    next chunks' loads at the top of the iteration 2.0-2.15 ms.)
//
// 2. AN fp32 FILTER WITH AN EXACT FALL-BACK.  What the pass emits per row are INTEGERS — which entry is the best hit, whether its z
//    reaches conf_prob (conf's value z / z is 1 to 2^-53) — and for all but a few thousand rows of 5e7 they are decided by a wide
//    margin.  So the numerators are formed in fp32, p = fl32(Q 2^-sQ) * fl32(pi theta 2^60) (scales that keep 149 binades of Q and
//    186 of pi*theta normal; only ratios matter), with the entry's position in its lane in the three or four lowest mantissa bits
//    (the maximum then IS the arg-max).  Every p is within 2^-19.6 of the true numerator, a row sum of <= 16 + 64 additions within 2^-17.3.
//    A row is DECIDED when   its largest p is >= 2^-40 (nothing that underflowed can matter),
//                            no other p lies within 2^-16 of it (the true gap is then > 2^-17: one best hit, no near-tie), and
//                            M / S is farther than 2^-15 from conf_prob (the true z_max lies within 2^-17 of it);
//    everything else — exact ties, near-ties, rows on the threshold, rows whose products all vanish, rows of more than 64 lanes —
//    goes to k_report_slow, which does the exact fp64 arithmetic and forms near-tied row sums in the reference's order (near_band).
//    With 4-byte tables pi*theta of ALL ids fits LDS up to ~34 000 slots (K <= 30 720): no gather leaves the CU; beyond, the tail
//    comes from L2 (COLD).  Unique rows (one entry: pi instead of pi*theta, model.py:706-714) need no arithmetic: their entry is the
//    best hit with z = 1 unless pi or Q vanish — the table's sign bit says whether pi is safely above 0.
//    Winners are not counted here: the row's last lane stores a record (id | pass | decided), 4 B per row, into the best-hit array, and
//    k_report_hist counts the records per id with the whole LDS for its counters — no global atomic, no counter competing with the
//    tables for LDS; the deferred rows' words are rewritten by k_report_slow with their best-hit counts afterwards.
//
// Integer outputs are exact by construction (a decided row's outputs do not depend on rounding; the others are computed exactly);
// `conf` sums 1.0 per passing row where the reference sums fl(z fl(1 / z)) in {1, 1 - 2^-53}.
// Serves: the final z (cnat2 != null), conf_prob > 0.51, no groups, not `reproducible`, lut[0] == 0, lut_len <= 2048; everything
// else keeps k_report_rows.
#pragma once

struct RpChunk { int32_t r0, nr; unsigned long long heads; };   // rows r0 .. r0 + (nr & 127) - 1, the longest of them (nr >> 8) lanes; bit l of
                                                                // heads: lane l starts a row (or the unused tail)
constexpr int RC_TILE = 8192;             // rows per workgroup of k_report_chunks
constexpr int RC_WIN = 1024;              // lanes per window: one thread packs one window greedily

// lanes of a row with E entries per lane; rows beyond 64 lanes take one (they are left to k_report_slow)
__host__ __device__ inline int rp_lanes(int64_t len, int E) { return len > 64 * (int64_t)E || len <= E ? 1 : (int)((len + E - 1) / E); }

// Pack the rows of the matrix into chunks.  One workgroup per tile of RC_TILE rows: lanes per row -> LDS, exclusive prefix, then
// thread t packs the rows whose first lane falls into window t, t + 256, .. of the tile (greedy: a chunk is closed when the next
// row does not fit).  The order of the chunks in `out` is arbitrary (slots are reserved with one atomic per window).  Launched twice:
// with out == null to count the chunks, then to write them into a table of exactly that size.
__global__ __launch_bounds__(256) void k_report_chunks(int64_t N, const int64_t* __restrict__ indptr, int E, RpChunk* __restrict__ out,
                                                       unsigned long long* __restrict__ n_out, int64_t cap) {
  __shared__ uint32_t pre[RC_TILE + RC_TILE / 32 + 2];      // element i at i + i / 32: a thread's 32 consecutive rows hit 32 banks
  __shared__ uint32_t wtot[4];
  auto at = [](int i) { return i + (i >> 5); };
  const int64_t base = (int64_t)blockIdx.x * RC_TILE;
  const int nrows = (int)min<int64_t>(RC_TILE, N - base);
  for (int i = threadIdx.x; i < RC_TILE; i += 256)
    pre[at(i)] = i < nrows ? (uint32_t)rp_lanes(indptr[base + i + 1] - indptr[base + i], E) : 0u;
  __syncthreads();
  uint32_t mine = 0;
  for (int k = 0; k < 32; ++k) mine += pre[at(threadIdx.x * 32 + k)];
  uint32_t inc = mine;
  for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o, 64); if ((int)(threadIdx.x & 63) >= o) inc += t; }
  if ((threadIdx.x & 63) == 63) wtot[threadIdx.x >> 6] = inc;
  __syncthreads();
  uint32_t run = inc - mine;
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) run += wtot[w];
  const uint32_t total = wtot[0] + wtot[1] + wtot[2] + wtot[3];
  for (int k = 0; k < 32; ++k) { const int i = at(threadIdx.x * 32 + k); const uint32_t g = pre[i]; pre[i] = run; run += g; }
  if (threadIdx.x == 255) pre[at(RC_TILE)] = total;
  __syncthreads();
  auto first_row_at = [&](uint32_t lane0) -> int {            // first row of the tile whose first lane is >= lane0
    int lo = 0, hi = nrows;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (pre[at(mid)] < lane0) lo = mid + 1; else hi = mid; }
    return lo;
  };
  const uint32_t nwin = (total + RC_WIN - 1) / RC_WIN;
  for (uint32_t w = threadIdx.x; w < nwin; w += 256) {
    const int ra = first_row_at(w * RC_WIN), rb = w + 1 < nwin ? first_row_at((w + 1) * RC_WIN) : nrows;
    if (ra >= rb) continue;
    int cnt = 1, used = 0;
    for (int r = ra; r < rb; ++r) {
      const int g = (int)(pre[at(r + 1)] - pre[at(r)]);
      if (used + g > 64) { ++cnt; used = 0; }
      used += g;
    }
    const int64_t slot0 = (int64_t)atomicAdd(n_out, (unsigned long long)cnt);
    if (!out || slot0 + cnt > cap) continue;                 // (out == null: the counting launch; the table then holds exactly that many)
    int64_t slot = slot0;
    RpChunk c; c.r0 = (int32_t)(base + ra); c.nr = 0; c.heads = 0ull; used = 0;
    int gmax = 0;
    for (int r = ra; r < rb; ++r) {
      const int g = (int)(pre[at(r + 1)] - pre[at(r)]);
      if (used + g > 64) {
        if (used < 64) c.heads |= 1ull << used;
        c.nr |= gmax << 8;
        out[slot++] = c;
        c.r0 = (int32_t)(base + r); c.nr = 0; c.heads = 0ull; used = 0; gmax = 0;
      }
      c.heads |= 1ull << used; used += g; ++c.nr; gmax = max(gmax, g);
    }
    if (used < 64) c.heads |= 1ull << used;
    c.nr |= gmax << 8;
    out[slot] = c;
  }
}

// ---- tables of the fp32 filter -----------------------------------------------------------------------------------------------
constexpr float RP_FLOOR = 0x1p-40f;                       // decided rows have their largest numerator above this (scaled units)
constexpr float RP_NEAR = 0x1p-16f, RP_TMARGIN = 0x1p-15f;
// t32[id] = fl32(pi theta 2^60) with the sign bit set when pi < 2^-1000 (a unique row on that column is left to the exact path);
// t32[IDN] = 0 (what the LDS-resident ids read from the global table with COLD).  l32[code] = fl32(lut[code] 2^-sq).
__global__ void k_rp32_tables(int IDN, const double* __restrict__ cnat2, int lut_len, const double* __restrict__ lut, int sq,
                              uint32_t* __restrict__ t32, float* __restrict__ l32) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < IDN) {
    const float c = (float)ldexp(cnat2[i], 60);
    t32[i] = (__float_as_uint(c) & 0x7FFFFFFFu) | (cnat2[IDN + i] >= 0x1p-1000 ? 0u : 0x80000000u);
  }
  if (i == IDN) t32[i] = 0u;
  if (i < lut_len) l32[i] = (float)ldexp(lut[i], -sq);
}

// winners per id from the per-row records: bit 17 = the row has a decided winner, bit 16 = its z reaches conf_prob, bits 0-15 its id.
// LDS counters for the ids [id0, id0 + W); flushed into the doubles k_report_finish reads.
__global__ __launch_bounds__(1024) void k_report_hist(int64_t N, const uint32_t* __restrict__ win, int id0, int W,
                                                      double* __restrict__ g_n1, double* __restrict__ g_conf) {
  extern __shared__ unsigned long long hist_lds[];          // [W]: winners | winners that pass << 32 (one LDS atomic per row)
  for (int t = threadIdx.x; t < W; t += blockDim.x) hist_lds[t] = 0ull;
  __syncthreads();
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  auto count = [&](uint32_t w) {
    const uint32_t id = (w & 0xFFFFu) - (uint32_t)id0;
    if ((w & 0x20000u) && id < (uint32_t)W) atomicAdd(&hist_lds[id], (w & 0x10000u) ? 0x100000001ull : 1ull);
  };
  const int64_t n4 = N / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(win) + i);
    count(v.x); count(v.y); count(v.z); count(v.w);
  }
  if (blockIdx.x == 0)
    for (int64_t i = n4 * 4 + threadIdx.x; i < N; i += blockDim.x) count(win[i]);
  __syncthreads();
  for (int t = threadIdx.x; t < W; t += blockDim.x) {
    const unsigned long long c = hist_lds[t];
    if ((uint32_t)c) unsafeAtomicAdd(&g_n1[id0 + t], (double)(uint32_t)c);
    if (c >> 32) unsafeAtomicAdd(&g_conf[id0 + t], (double)(uint32_t)(c >> 32));
  }
}

struct Rp32Args {
  int64_t N; int32_t IDN, HC, lut_len, lut_rep;             // lut_rep: log2 of the copies of the score table in LDS (one per bank class)
  const int64_t* indptr; const uint16_t* rid; const uint16_t* raw;
  const uint32_t* t32; const float* l32;                    // k_rp32_tables
  float thresh;
  uint32_t* win;                                            // [N] winner records (the host passes the best-hit count array: a deferred row's
                                                            // word is rewritten by k_report_slow with its count, after k_report_hist has run)
  int32_t* defer_rows; unsigned long long* defer_n;
  const RpChunk* chunks; int64_t nchunks;
  int dbg;                                                  // timing experiments (wrong results): 16 no scan, 32 no stores, no decisions
};

// A row's (sum, largest, second largest, id of the largest) over its lanes: a segmented inclusive scan on the VALU — row_shr 1 / 2 / 4 / 8
// inside the 16-lane DPP rows, then lane 15 -> the next DPP row and lane 31 -> the upper half, each step applied where the source
// lane still belongs to the same matrix row (pos = the lane's distance from its row's first lane).  The row's LAST lane ends up
// with the totals.  No LDS: a first version's LDS atomics (one add, one max, one count per lane into a slot per row, read back by
// every lane) cost 20 LDS-array cycles each, and with 8 cycles per table gather the LDS was busy for 2.2 of that kernel's 3.2 ms
// (profiles/r06_report_pack32_lds.txt).  The numerators are >= 0, so max / min / compare work on their bit patterns as integers
// (no canonicalising v_max_f32 x, x in front of every float maximum).
__device__ __forceinline__ void rp_store_u32(uint32_t* p, uint32_t v) {
  asm volatile("global_store_dword %0, %1, off" : : "v"(p), "v"(v) : "memory");
}
struct RpTuple { float s; uint32_t m1, m2, w; };
template <int CTRL, int ROWMASK>
__device__ __forceinline__ void rp_scan_step(RpTuple& v, bool take) {
  const float ts = __uint_as_float((uint32_t)fz_dpp_i<CTRL, ROWMASK>(0, (int)__float_as_uint(v.s)));
  const uint32_t t1 = (uint32_t)fz_dpp_i<CTRL, ROWMASK>(0, (int)v.m1);
  const uint32_t t2 = (uint32_t)fz_dpp_i<CTRL, ROWMASK>(0, (int)v.m2);
  const uint32_t tw = (uint32_t)fz_dpp_i<CTRL, ROWMASK>(0, (int)v.w);
  if (take) {
    v.s += ts;
    v.m2 = max(max(min(v.m1, t1), v.m2), t2);               // the runner-up of the union
    v.w = t1 > v.m1 ? tw : v.w;
    v.m1 = max(v.m1, t1);
  }
}

template <int E, bool COLD>
__global__ __launch_bounds__(1024) void k_report_pack32(Rp32Args A) {
  static_assert(E == 8 || E == 16, "entries per lane");
  extern __shared__ uint32_t rp32_lds[];
  // [E + 1][E / 2] words: the lane's code words ANDed with row nl keep its first nl codes | [HC] table of the ids < HC, [1] 0 |
  // [lut_len << lut_rep] score table: copy b of entry k at word (k << lut_rep) + b, i.e. in bank class b — a lane reads the copy of
  // its own lane number: no bank conflict among the 64 gathers of an instruction
  constexpr int W = E / 2, MASKW = (E + 1) * W;
  const int nw = blockDim.x >> 6;
  uint32_t* const tS = rp32_lds + MASKW;
  float* const lS = reinterpret_cast<float*>(tS + A.HC + 1);
  for (int t = threadIdx.x; t < MASKW; t += blockDim.x) {
    const int lim = t / W - 2 * (t % W);
    rp32_lds[t] = lim >= 2 ? 0xFFFFFFFFu : (lim == 1 ? 0x0000FFFFu : 0u);
  }
  for (int t = threadIdx.x; t < A.HC; t += blockDim.x) tS[t] = A.t32[t];
  if (threadIdx.x == 0) tS[A.HC] = 0u;
  for (int t = threadIdx.x; t < (A.lut_len << A.lut_rep); t += blockDim.x) lS[t] = A.l32[t >> A.lut_rep];
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned long long le = (2ull << lane) - 1ull;       // lanes 0 .. lane
  const int64_t cstride = (int64_t)gridDim.x * nw;
  const int64_t nit = (A.nchunks + cstride - 1) / cstride;
  const uint32_t tab0 = (uint32_t)MASKW * 4u;                // byte offset of the id table
  const uint32_t hc4 = (uint32_t)A.HC * 4u, idn4 = (uint32_t)A.IDN * 4u;
  const uint32_t lutoff = tab0 + (uint32_t)(A.HC + 1) * 4u + (uint32_t)(lane & ((1 << A.lut_rep) - 1)) * 4u;   // this lane's copy
  const int lsh = 2 + A.lut_rep;
  const char* const lds0 = reinterpret_cast<const char*>(rp32_lds);
  const char* const tg0 = reinterpret_cast<const char*>(A.t32);

  struct Dsc { int4 w; };                                     // r0, nr, heads lo, heads hi (as loaded)
  struct IpRaw { rr_i64x2_a8 se; int rel, pos, row, gmax; bool valid; };
  struct Ip { int64_t s; int len, pos, row, gmax; bool valid, toolong; };
  struct Ent { rr_u32x4_a2 id[E / 8], cd[E / 8]; };
  auto load_dsc = [&](int64_t it) -> Dsc {
    const int64_t c = it * cstride + (int64_t)blockIdx.x * nw + wv;
    Dsc d;
    d.w = *reinterpret_cast<const int4*>(A.chunks + (c < A.nchunks ? c : 0));
    if (c >= A.nchunks) d.w.y = 0;                            // no rows: every lane idles
    return d;
  };
  auto load_ip = [&](const Dsc& d) -> IpRaw {
    const unsigned long long hm = (((unsigned long long)(uint32_t)d.w.w << 32) | (uint32_t)d.w.z) & le;
    IpRaw p;
    p.rel = __popcll(hm) - 1;
    p.pos = lane - (63 - __clzll((long long)hm));
    p.valid = p.rel < (d.w.y & 127);
    p.gmax = d.w.y >> 8;
    p.row = d.w.x + (p.valid ? p.rel : 0);
    p.se = *reinterpret_cast<const rr_i64x2_a8*>(A.indptr + p.row);
    return p;
  };
  auto derive = [&](const IpRaw& r) -> Ip {
    Ip p; p.s = r.se.x; p.pos = r.pos; p.row = r.row; p.valid = r.valid; p.gmax = r.gmax;
    const int64_t len = r.se.y - r.se.x;
    p.toolong = r.valid && len > 64 * E;
    p.len = r.valid && !p.toolong ? (int)len : 0;
    return p;
  };
  auto load_ent = [&](const Ip& p) -> Ent {
    // lanes past the row's end read what follows it (the arrays carry TS_ENTRY_PAD entries of padding: never out of bounds); the
    // lanes behind a chunk's last row (and every lane of a wave without a chunk) read the chunk's first entries — their distance
    // from the unused tail's head times E would reach up to 1008 entries past them (round 6: a memory fault once in 1128 soak cases)
    const int64_t k = p.s + (p.valid && !p.toolong ? E * p.pos : 0);
    Ent t;
#pragma unroll
    for (int q = 0; q < E / 8; ++q) {
      t.id[q] = *reinterpret_cast<const rr_u32x4_a2*>(A.rid + k + 8 * q);
      t.cd[q] = *reinterpret_cast<const rr_u32x4_a2*>(A.raw + k + 8 * q);
    }
    return t;
  };
  struct Prep { float q[E]; uint32_t c[E]; };                // scaled Q, table word (sign bit: see k_rp32_tables)
  auto prep = [&](const Ip& p, const Ent& t) -> Prep {
    const int nl = min(max(p.len - E * p.pos, 0), E);        // this lane's entries of the row
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 mk[E / 8];                                          // codes past the row's end -> 0 (lut[0] = 0: they add nothing)
#pragma unroll
    for (int q = 0; q < E / 8; ++q) mk[q] = *reinterpret_cast<const u32x4*>(lds0 + (uint32_t)nl * (W * 4u) + 16u * q);
    Prep r;
#pragma unroll
    for (int j = 0; j < E; ++j) {
      const uint32_t wc = t.cd[j / 8][(j / 2) & 3] & mk[j / 8][(j / 2) & 3], wi = t.id[j / 8][(j / 2) & 3];
      const uint32_t code = (j & 1) ? wc >> 16 : wc & 0xFFFFu;
      const uint32_t id4 = (j & 1) ? (wi >> 16) << 2 : (wi & 0xFFFFu) << 2;
      r.q[j] = *reinterpret_cast<const float*>(lds0 + ((code << lsh) + lutoff));
      if (COLD) {
        const bool hot = id4 < hc4;
        const uint32_t cl = *reinterpret_cast<const uint32_t*>(lds0 + tab0 + (hot ? id4 : hc4));   // cold: the 0 behind the table
        const uint32_t cg = *reinterpret_cast<const uint32_t*>(tg0 + (hot ? idn4 : id4));          // hot: the 0 behind the global table
        r.c[j] = cl | cg;
      } else {
        r.c[j] = *reinterpret_cast<const uint32_t*>(lds0 + tab0 + id4);
      }
    }
    return r;
  };
  struct Out { int row, nb; uint32_t w; bool act; };
  auto finish = [&](const Ip& p, const Ent& t, const Prep& r) -> Out {
    RpTuple v; v.s = 0.f; v.m1 = 0u; v.m2 = 0u;
#pragma unroll
    for (int j = 0; j < E; ++j) {
      // the numerator with the entry's position in its low mantissa bits (E = 16: four bits, 2^-20 relative)
      const uint32_t pb = (__float_as_uint(r.q[j] * fabsf(__uint_as_float(r.c[j]))) & ~(uint32_t)(E - 1)) | (uint32_t)j;
      v.s += __uint_as_float(pb);
      v.m2 = max(v.m2, min(v.m1, pb));                         // the second largest so far (m1 >= m2)
      v.m1 = max(v.m1, pb);
    }
    // the lane's largest entry: its position sits in the low bits
    const uint32_t jw = v.m1 & (uint32_t)(E - 1);
    uint32_t wsel = t.id[0][0];
#pragma unroll
    for (int q = 1; q < W; ++q) wsel = (jw >> 1) == (uint32_t)q ? t.id[q / 4][q & 3] : wsel;
    v.w = (jw & 1u) ? wsel >> 16 : wsel & 0xFFFFu;
    const uint32_t q0 = __float_as_uint(r.q[0]), c0 = r.c[0], id0 = t.id[0][0] & 0xFFFFu;   // (what a unique row needs)
    const int gm = __builtin_amdgcn_readfirstlane(p.gmax);     // the longest row of the chunk, in lanes: steps beyond it are skipped
    if (gm > 1 && !(A.dbg & 16)) {
      const int pos = p.pos, l15 = lane & 15, l31 = lane & 31;
      rp_scan_step<0x111, 0xF>(v, pos >= 1);
      if (gm > 2) rp_scan_step<0x112, 0xF>(v, pos >= 2);
      if (gm > 4) rp_scan_step<0x114, 0xF>(v, pos >= 4);
      if (gm > 8) rp_scan_step<0x118, 0xF>(v, pos >= 8);
      rp_scan_step<0x142, 0xA>(v, (lane & 16) != 0 && pos > l15);      // lane 15 of the DPP row below
      rp_scan_step<0x143, 0xC>(v, (lane & 32) != 0 && pos > l31);      // lane 31
    }
    const int g = rp_lanes(p.len, E);
    Out o; o.act = p.valid && p.pos == g - 1; o.row = p.row; o.nb = -1; o.w = 0u;   // the row's last lane holds its totals: it decides
    {
      const float m1 = __uint_as_float(v.m1), m2 = __uint_as_float(v.m2);
      const float ts = A.thresh * v.s;
      const bool pass = m1 > ts * (1.0f + RP_TMARGIN), fail = m1 < ts * (1.0f - RP_TMARGIN);
      if (p.len > 1) {
        if (m1 >= RP_FLOOR && m2 < m1 * (1.0f - RP_NEAR) && (pass || fail)) { o.nb = 1; o.w = v.w | (pass ? 0x30000u : 0x20000u); }
      } else if (p.len == 1) {                                  // a unique row: its entry is the best hit, z = 1, unless pi or Q vanish
        if (!(c0 & 0x80000000u) && __uint_as_float(q0) > 0.f) { o.nb = 1; o.w = id0 | 0x30000u; }
      } else if (!p.toolong) {
        o.nb = 0;                                               // an empty row
      }
    }
    return o;
  };
  // The stores of a chunk go out one iteration LATER, in front of the next loads, and as inline assembly.  gfx9 counts loads and
  // stores in one counter, in issue order.  Seen by the compiler's wait-count pass, a store that may or may not be pending makes every
  // wait a wait for everything; unseen but issued last in the iteration, it sits behind the row-pointer / descriptor loads the next
  // iteration waits for with vmcnt(0), which then waits for the store's acknowledgement as well (2.2 against 1.8 ms).  Issued here it
  // has a whole iteration to complete before anything younger is awaited.  The best-hit count of a decided row (1; 0 for an empty row)
  // is not stored: the host takes the tied rows from the deferred list (k_ties_of_deferred).
  auto emit = [&](const Out& o) {
    if (!o.act || (A.dbg & 32)) return;
    rp_store_u32(&A.win[o.row], o.w);
    if (o.nb < 0) rp_store_u32(reinterpret_cast<uint32_t*>(&A.defer_rows[atomicAdd(A.defer_n, 1ull)]), (uint32_t)o.row);
  };
  if (nit > 0) {
    // Prologue.  The loads must be outstanding in the loop's own order when it is entered — entries, then row pointers, then the
    // descriptor: the wait-count pass merges the state at the loop header over both ways in, and with another order here it waits
    // for everything (vmcnt(0)) at the top of EVERY iteration.  So the row pointers of chunk 1 and the descriptor of chunk 2 are
    // loaded twice: once to get going, once more behind the entries of chunk 0.
    Ip ip0;
    Dsc d1;
    { const Dsc d0 = load_dsc(0); d1 = load_dsc(1); const IpRaw r0 = load_ip(d0); ip0 = derive(r0); }
    __builtin_amdgcn_sched_barrier(0);
    Ent e0 = load_ent(ip0);
    IpRaw r1 = load_ip(d1);
    Dsc d2 = load_dsc(2);
    __builtin_amdgcn_sched_barrier(0);
    Out o; o.act = false; o.row = 0; o.nb = 0; o.w = 0u;
    for (int64_t it = 0; it < nit; ++it) {
      if constexpr (COLD) {
        // the L2 gathers of this chunk first: loads return in order, and behind the next chunks' loads they would wait for those
        const Prep q = prep(ip0, e0);
        __builtin_amdgcn_sched_barrier(0);
        emit(o);                                               // the previous chunk's stores
        const Ip ip1 = derive(r1);
        const Ent e1 = load_ent(ip1);
        const IpRaw r2 = load_ip(d2);
        const Dsc d3 = load_dsc(it + 3);
        __builtin_amdgcn_sched_barrier(0);
        o = finish(ip0, e0, q);
        ip0 = ip1; e0 = e1; r1 = r2; d2 = d3;
      } else {
        // nothing of this chunk leaves the CU: the next chunks' loads go out first and have the whole iteration to arrive in
        emit(o);                                               // the previous chunk's stores
        const Ip ip1 = derive(r1);
        const Ent e1 = load_ent(ip1);
        const IpRaw r2 = load_ip(d2);
        const Dsc d3 = load_dsc(it + 3);
        __builtin_amdgcn_sched_barrier(0);
        const Prep q = prep(ip0, e0);
        o = finish(ip0, e0, q);
        __builtin_amdgcn_sched_barrier(0);
        ip0 = ip1; e0 = e1; r1 = r2; d2 = d3;
      }
    }
    emit(o);
  }
}
