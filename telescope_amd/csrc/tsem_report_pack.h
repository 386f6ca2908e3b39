// The report pass over the FINAL z without padded lanes (round 6; included by tsem_report.hip after ReportArgs / ReportEmit).
//
// k_report_rows gives a row G lanes of E entries with G a power of two chosen for the whole matrix: at 40 entries per row and a
// capacity of 64, 38 % of the lane-entries are padding, and the kernel is bound by instruction issue (37 VALU per lane-entry,
// profiles/r05_report_pmc.txt) — its time follows the capacity, not the entries.  Here a row takes ceil(len / 8) lanes, rows
// follow each other without gaps inside a wave, and a wave's work is a CHUNK of whole rows that fills 64 lanes (k_report_chunks
// packs them greedily once per matrix: ~9 % padding in the rows' last lanes + ~4 % at the chunks' ends).  What changes with it:
//   * lane -> row: the chunk's descriptor carries a 64-bit mask of the lanes that start a row; a lane's row is the number of
//     such lanes at or below it (v_bcnt), its place inside the row the distance to the nearest one (v_ffbh) — no search, no
//     per-row prefix sums in the kernel.  Descriptors are fetched three iterations ahead, the rows' pointers two, the entries one;
//   * row sum and row maximum: one ds_add_f64 and one ds_max_u64 (numerators are >= 0: their bit patterns order like integers)
//     per LANE into the row's slot of a wave-private LDS array, read back by every lane of the row — LDS instructions of a wave
//     execute in order, so no barrier; the count of entries inside the near-tie band below the maximum goes the same way;
//   * no division: a row whose largest numerator M is alone inside the band has ONE best hit, the lane that holds it counts it;
//     z_max >= conf_prob is decided as M against conf_prob * rowsum with the band's margin (the reference's z_max = fl(M fl(1 / S))
//     lies within 2^-51 of M / S); conf's value z / z is 1.0 (fl(z fl(1 / z)) is 1 or 1 - 2^-53: a float output, 1e-16 relative);
//     everything inside the margins — several numerators in the band that are not all equal, z_max within the band of conf_prob —
//     goes to k_report_slow, which forms the row sum in the reference's order (near_band); exact ties are emitted here;
//   * pi*theta: ids below HC from LDS, the rest from L2, both fetched unconditionally and ADDED (the LDS table ends with a 0.0
//     that the cold ids read, the global one that the hot ids read): an add instead of three selects;
//   * winners: 32-bit LDS counters for the Hs most popular ids (n1 | conf), one packed 64-bit global atomic for the others.
// Same arithmetic per entry as k_report_rows (lut[code] * c[id], products rounded before they are added: this unit is compiled
// with -ffp-contract=off), so the integer outputs are the same bits; the float outputs differ by summation order only.
// Serves: the final z (cnat2 != null), conf_prob > 0.51, no groups, not `reproducible`, lut[0] == 0; everything else keeps
// k_report_rows.  Rows longer than 512 entries go to k_report_slow.
#pragma once

struct RpChunk { int32_t r0, nr; unsigned long long heads; };   // rows r0 .. r0 + nr - 1; bit l of heads: lane l starts a row (or the unused tail)
constexpr int RP_E = 8;                   // entries per lane
constexpr int RP_MAXLEN = 64 * RP_E;      // longest row handled here
constexpr int RC_TILE = 8192;             // rows per workgroup of k_report_chunks
constexpr int RC_WIN = 1024;              // lanes per window: one thread packs one window greedily

__host__ __device__ inline int rp_lanes(int64_t len) { return len > RP_MAXLEN || len <= RP_E ? 1 : (int)((len + RP_E - 1) / RP_E); }

// upper bound of the number of chunks (two consecutive chunks of a window hold more than 64 lanes together; every window and
// every tile ends with a partial one)
static inline int64_t rp_chunk_cap(int64_t N, int64_t nnz) {
  const int64_t lanes = nnz / RP_E + N;
  return lanes / 32 + lanes / RC_WIN + 2 * (N / RC_TILE + 1) + 64;
}

// Pack the rows of the matrix into chunks.  One workgroup per tile of RC_TILE rows: lanes per row -> LDS, exclusive prefix, then
// thread t packs the rows whose first lane falls into window t, t + 256, .. of the tile (greedy: a chunk is closed when the next
// row does not fit).  The order of the chunks in `out` is arbitrary (slots are reserved with one atomic per window).
__global__ __launch_bounds__(256) void k_report_chunks(int64_t N, const int64_t* __restrict__ indptr, RpChunk* __restrict__ out,
                                                       unsigned long long* __restrict__ n_out, int64_t cap) {
  __shared__ uint32_t pre[RC_TILE + RC_TILE / 32 + 2];      // element i at i + i / 32: a thread's 32 consecutive rows hit 32 banks
  __shared__ uint32_t wtot[4];
  auto at = [](int i) { return i + (i >> 5); };
  const int64_t base = (int64_t)blockIdx.x * RC_TILE;
  const int nrows = (int)min<int64_t>(RC_TILE, N - base);
  for (int i = threadIdx.x; i < RC_TILE; i += 256)
    pre[at(i)] = i < nrows ? (uint32_t)rp_lanes(indptr[base + i + 1] - indptr[base + i]) : 0u;
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
    if (slot0 + cnt > cap) continue;                         // (cannot happen: rp_chunk_cap; the host checks the count)
    int64_t slot = slot0;
    RpChunk c; c.r0 = (int32_t)(base + ra); c.nr = 0; c.heads = 0ull; used = 0;
    for (int r = ra; r < rb; ++r) {
      const int g = (int)(pre[at(r + 1)] - pre[at(r)]);
      if (used + g > 64) {
        if (used < 64) c.heads |= 1ull << used;
        out[slot++] = c;
        c.r0 = (int32_t)(base + r); c.nr = 0; c.heads = 0ull; used = 0;
      }
      c.heads |= 1ull << used; used += g; ++c.nr;
    }
    if (used < 64) c.heads |= 1ull << used;
    out[slot] = c;
  }
}

// by id: n1 | conf << 32 of the packed kernel's winners -> the doubles k_report_finish reads
__global__ void k_report_unpack(int IDN, const unsigned long long* __restrict__ g_pack, double* __restrict__ g_n1, double* __restrict__ g_conf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= IDN) return;
  const unsigned long long p = g_pack[i];
  if (p) { g_n1[i] += (double)(uint32_t)p; g_conf[i] += (double)(uint32_t)(p >> 32); }
}

struct RpSlot { double sum; unsigned long long max; };     // a row's slot of the wave's LDS array

__global__ __launch_bounds__(1024) void k_report_pack(ReportArgs A, const RpChunk* __restrict__ chunks, int64_t nchunks,
                                                      unsigned long long* __restrict__ g_pack) {
  extern __shared__ double rr_lds[];
  // [HC] pi*theta of the most popular ids, [1] 0.0 | [lut_len] score table | per wave [64] slots | [Hs] n1, [Hs] conf | per wave [64] counts
  const int nw = blockDim.x >> 6;
  double* const cH = rr_lds;
  double* const lutS = cH + A.HC + 1;
  RpSlot* const slots_all = reinterpret_cast<RpSlot*>(lutS + A.lut_len);
  uint32_t* const hot1 = reinterpret_cast<uint32_t*>(slots_all + nw * 64);
  uint32_t* const hotc = hot1 + A.Hs;
  uint32_t* const cnts_all = hotc + A.Hs;
  for (int t = threadIdx.x; t < A.HC; t += blockDim.x) cH[t] = A.cnat2[t];
  if (threadIdx.x == 0) cH[A.HC] = 0.0;
  for (int t = threadIdx.x; t < A.lut_len; t += blockDim.x) lutS[t] = A.lut[t];
  for (int t = threadIdx.x; t < nw * 64; t += blockDim.x) { slots_all[t].sum = 0.0; slots_all[t].max = 0ull; cnts_all[t] = 0u; }
  for (int t = threadIdx.x; t < A.Hs; t += blockDim.x) { hot1[t] = 0u; hotc[t] = 0u; }
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  RpSlot* const slots = slots_all + wv * 64;
  uint32_t* const cnts = cnts_all + wv * 64;
  const unsigned long long le = (2ull << lane) - 1ull;       // lanes 0 .. lane
  const int64_t cstride = (int64_t)gridDim.x * nw;
  const int64_t nit = (nchunks + cstride - 1) / cstride;
  const uint32_t lutoff = (uint32_t)((A.HC + 1) * 8);        // byte offset of the score table in LDS
  const uint32_t hc8 = (uint32_t)A.HC * 8u, idn8 = (uint32_t)A.IDN * 8u;
  const char* const lds0 = reinterpret_cast<const char*>(rr_lds);
  const char* const cg0 = reinterpret_cast<const char*>(A.cnat2);   // [IDN] pi*theta | [IDN] pi | [1] 0.0, by id

  struct Dsc { int4 w; };                                     // r0, nr, heads lo, heads hi (as loaded)
  struct IpRaw { rr_i64x2_a8 se; int rel, pos, row; bool valid; };
  struct Ip { int64_t s; int len, rel, pos, row; bool valid, toolong; };
  struct Ent { rr_u32x4_a2 id, cd; };
  auto load_dsc = [&](int64_t it) -> Dsc {
    const int64_t c = it * cstride + (int64_t)blockIdx.x * nw + wv;
    Dsc d;
    d.w = *reinterpret_cast<const int4*>(chunks + (c < nchunks ? c : 0));
    if (c >= nchunks) d.w.y = 0;                              // no rows: every lane idles
    return d;
  };
  auto load_ip = [&](const Dsc& d) -> IpRaw {
    const unsigned long long hm = (((unsigned long long)(uint32_t)d.w.w << 32) | (uint32_t)d.w.z) & le;
    IpRaw p;
    p.rel = __popcll(hm) - 1;
    p.pos = lane - (63 - __clzll((long long)hm));
    p.valid = p.rel < d.w.y;
    p.row = d.w.x + (p.valid ? p.rel : 0);
    p.se = *reinterpret_cast<const rr_i64x2_a8*>(A.indptr + p.row);
    return p;
  };
  auto derive = [&](const IpRaw& r) -> Ip {
    Ip p; p.s = r.se.x; p.rel = r.rel; p.pos = r.pos; p.row = r.row; p.valid = r.valid;
    const int64_t len = r.se.y - r.se.x;
    p.toolong = r.valid && len > RP_MAXLEN;
    p.len = r.valid && !p.toolong ? (int)len : 0;
    return p;
  };
  auto load_ent = [&](const Ip& p) -> Ent {
    // lanes past the row's end read what follows it (the arrays carry TS_ENTRY_PAD entries of padding: never out of bounds)
    const int64_t k = p.s + (p.toolong ? 0 : RP_E * p.pos);
    Ent t;
    t.id = *reinterpret_cast<const rr_u32x4_a2*>(A.rid + k);
    t.cd = *reinterpret_cast<const rr_u32x4_a2*>(A.raw + k);
    return t;
  };
  struct Prep { double n[RP_E]; uint32_t cdm[RP_E / 2]; };
  auto prep = [&](const Ip& p, const Ent& t) -> Prep {
    const int nl = min(max(p.len - RP_E * p.pos, 0), RP_E);   // this lane's entries of the row
    const bool amb = p.len > 1;                                // ambiguous rows: pi*theta, unique rows: pi (model.py:706-714)
    const uint32_t lim8 = amb ? hc8 : 0u, goff8 = amb ? 0u : idn8;
    Prep q;
#pragma unroll
    for (int w = 0; w < RP_E / 2; ++w) {
      const int lim = nl - 2 * w;                              // codes past the row's end -> 0 (lut[0] = 0: they add nothing)
      q.cdm[w] = t.cd[w] & (lim >= 2 ? 0xFFFFFFFFu : (lim == 1 ? 0x0000FFFFu : 0u));
    }
#pragma unroll
    for (int j = 0; j < RP_E; ++j) {
      const uint32_t wc = q.cdm[j / 2], wi = t.id[j / 2];
      const uint32_t code8 = (j & 1) ? (wc >> 16) << 3 : (wc & 0xFFFFu) << 3;
      const uint32_t id8 = (j & 1) ? (wi >> 16) << 3 : (wi & 0xFFFFu) << 3;
      const bool hot = id8 < lim8;
      const uint32_t la = hot ? id8 : hc8;                     // cold: the 0.0 behind the table
      const uint32_t ga = hot ? 2u * idn8 : id8 + goff8;       // hot: the 0.0 behind the global table
      const double x = *reinterpret_cast<const double*>(lds0 + lutoff + code8);
      const double cl = *reinterpret_cast<const double*>(lds0 + la);
      const double cg = *reinterpret_cast<const double*>(cg0 + ga);
      q.n[j] = x * (cl + cg);
    }
    return q;
  };
  auto push = [&](int32_t code) { A.defer_rows[atomicAdd(A.defer_n, 1ull)] = code; };
  auto finish = [&](const Ip& p, const Ent& t, const Prep& q) {
    const double* n = q.n;
    double s = 0.0, m = 0.0; uint32_t wid8 = 0u;
#pragma unroll
    for (int j = 0; j < RP_E; ++j) {
      const uint32_t wi = t.id[j / 2];
      s += n[j];
      const bool gt = n[j] > m;
      m = fmax(m, n[j]);
      wid8 = gt ? ((j & 1) ? wi >> 16 : wi & 0xFFFFu) : wid8;
    }
    RpSlot* const sl = slots + p.rel;
    uint32_t* const cn = cnts + p.rel;
    if (p.valid) {
      __hip_atomic_fetch_add(&sl->sum, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_max(&sl->max, (unsigned long long)__double_as_longlong(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    const double S = __hip_atomic_load(&sl->sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const double M = __longlong_as_double((long long)__hip_atomic_load(&sl->max, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    const double lo = M * (1.0 - TS_NEAR_BAND);
    int nnl = 0;
#pragma unroll
    for (int j = 0; j < RP_E; ++j) nnl += n[j] >= lo ? 1 : 0;
    const bool live = p.valid && M > 0.0;                      // (M == 0: the row's pattern is empty, model.py:720; every n >= lo = 0 then)
    if (live) __hip_atomic_fetch_add(cn, (uint32_t)nnl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    uint32_t nn = __hip_atomic_load(cn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const bool head = p.valid && p.pos == 0;
    // z_max >= conf_prob without the division: M against conf_prob * S, with the band's margin on either side
    const double ts = A.thresh * S, tm = ts * (2.0 * TS_NEAR_BAND);
    const bool pass = M > ts + tm, near_t = !pass && M >= ts - tm;
    bool near = live && near_t;
    int nb = live ? 1 : 0;
    if (__builtin_amdgcn_ballot_w64(live && nn > 1u) != 0ull) {   // several numerators inside the band: exact ties, or a near-tie
      int nbl = 0;
#pragma unroll
      for (int j = 0; j < RP_E; ++j) nbl += n[j] == M ? 1 : 0;
      const bool tied = live && nn > 1u;
      if (tied) __hip_atomic_fetch_add(cn, (uint32_t)nbl << 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const uint32_t both = __hip_atomic_load(cn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (tied) {
        nb = (int)(both >> 16);
        if (nb != (int)(both & 0xFFFFu)) near = true;           // not all of them equal
        else if (!near) {                                       // nb >= 2 best hits: z_max <= 1/2 < conf_prob
          const double share = 1.0 * recip0((double)nb);
          const ReportEmit<0> EM{A, nullptr, nullptr, nullptr, 0, nullptr};
#pragma unroll
          for (int j = 0; j < RP_E; ++j) {
            const uint32_t wi = t.id[j / 2];
            if (n[j] == M) EM.tie((j & 1) ? wi >> 16 : wi & 0xFFFFu, nb, nb == 2 ? 0.5 : share, 0);
          }
        }
      }
      nn = both & 0xFFFFu;
    }
    if (head) {                                                 // reset the row's slot for the wave's next chunk; the row's outputs
      __hip_atomic_store(&sl->sum, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_store(&sl->max, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_store(cn, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (p.toolong) push((int32_t)p.row);
      else if (near) push(~(int32_t)p.row);
      else A.nbest[p.row] = nb;
    }
    if (live && !near && nn == 1u && m == M) {                  // this lane holds the row's only best hit
      if ((int)wid8 < A.Hs) {
        atomicAdd(&hot1[wid8], 1u);
        if (pass) atomicAdd(&hotc[wid8], 1u);
      } else {
        atomicAdd(&g_pack[wid8], pass ? 0x100000001ull : 1ull);
      }
    }
  };
  if (nit > 0) {
    Dsc d2 = load_dsc(2);
    IpRaw r1, r0;
    { const Dsc d0 = load_dsc(0), d1 = load_dsc(1); r0 = load_ip(d0); r1 = load_ip(d1); }
    Ip ip0 = derive(r0);
    Ent e0 = load_ent(ip0);
    for (int64_t it = 0; it < nit; ++it) {
      const Prep q = prep(ip0, e0);                            // this chunk's gathers first ...
      __builtin_amdgcn_sched_barrier(0);
      const Ip ip1 = derive(r1);                               // ... then the loads of the next ones: they have this chunk's
      const Ent e1 = load_ent(ip1);                            //     arithmetic to arrive in
      const IpRaw r2 = load_ip(d2);
      const Dsc d3 = load_dsc(it + 3);
      __builtin_amdgcn_sched_barrier(0);
      finish(ip0, e0, q);
      ip0 = ip1; e0 = e1; r1 = r2; d2 = d3;
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < A.Hs; t += blockDim.x) {
    const unsigned long long c = (unsigned long long)hot1[t] | ((unsigned long long)hotc[t] << 32);
    if (c) atomicAdd(&g_pack[t], c);
  }
}
