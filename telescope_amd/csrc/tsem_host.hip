// libtelescope_em.so — MI355X (gfx950 / CDNA4) engine for Telescope's EM reassignment path.  C ABI in include/telescope_em.h.
// This unit: the handle (create / destroy / options / stream), loading the score matrix (tsem_load_scores: host arrays borrowed
// for the call; tsem_generate: the synthetic matrix on the device, bit-exact twin of telescope_amd/synthetic.py), the score table
// (Q = expm1((r / max) * 100), model.py:653), instrumentation and debug entry points.
//
// Data layout in HBM:
//   * canonical CSR of uint16 raw scores (indptr int64, indices int32, raw u16)
//     + the Q lookup table lut[r] = expm1(r/max*100) (model.py:653);
//   * for the EM hot loop, the AMBIGUOUS rows (Y_i = 1, model.py:679) re-laid
//     as a column-partitioned blocked COO ("PCOO"): columns are dealt by
//     popularity into P parts of Kp <= 7680 columns so that one part's
//     pi*theta table AND its fp64 column accumulators fit in LDS; rows are cut
//     in blocks of R; sub-block (b,p) is a contiguous run of
//     {fp64 Q value | 2-byte score code, u32 (local row << 16 | local col)} = 12 | 6 B per entry.
//   Global fp64 atomics reach only ~22 G/s on MI355X (2 G/s on hot columns)
//   while LDS gathers / ds_add_f64 keep up with the 5.8 TB/s HBM stream
//   (profiles/r01_primitives_ubench.log), hence every per-entry gather and
//   scatter of the hot loop goes through LDS.
#include "tsem_internal.h"
#include <rocprim/device/device_scan.hpp>

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
  int32_t fam = 0;
  if (dist == 2) {
    const double v = (double)(ts_hash3(seed ^ TS_SALT_FAM, row, 0) >> 11) * (1.0 / 9007199254740992.0);
    fam = (int32_t)floor(__dmul_rn((double)((K - 1) / TS_FAMILY), __dmul_rn(__dmul_rn(v, v), v)));
  }
  for (int k = 0; k < len; ++k) {
    if (k == 0 && has0) { cols[0] = 0; continue; }
    for (int attempt = 0;; ++attempt) {
      uint64_t h = ts_hash3(seed, row, (uint64_t)(k + 256 * attempt));
      double u = (double)(h >> 11) * (1.0 / 9007199254740992.0);
      int32_t cand;
      if (dist == 2) {                                     // 'family': every column of the row inside the row's family of TS_FAMILY loci
        cand = 1 + TS_FAMILY * fam + (int32_t)floor(__dmul_rn((double)TS_FAMILY, u));
      } else {
        if (dist == 1) u = __dmul_rn(__dmul_rn(u, u), u);
        cand = 1 + (int32_t)floor(__dmul_rn((double)(K - 1), u));
      }
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

std::string g_create_err;

extern "C" {

void tsem_free_layout(tsem_ctx* h) {
  dfree(h->d_ebias); dfree(h->d_ovf); dfree(h->d_red_hi); dfree(h->d_binflag); dfree(h->d_ehist);
  dfree(h->d_colmap); dfree(h->d_col_of_pc); dfree(h->d_rid16); dfree(h->d_col_of_id); dfree(h->d_sb_off); dfree(h->d_pval); dfree(h->d_pcode); dfree(h->d_prc);
  dfree(h->d_ypart); dfree(h->d_partial); dfree(h->d_xchg); dfree(h->d_xflags); dfree(h->d_fz_aux); h->fz_clean = false; dfree(h->d_fpartial); dfree(h->d_fpartial2); dfree(h->d_amb_w); dfree(h->d_sb_q32); dfree(h->d_rinv); h->lag_valid = false;
  dfree(h->d_lctab); dfree(h->d_lqtab); dfree(h->d_lq_mid); h->lq_n = 0; h->lq_tried = false;
  h->fused_launched = false;
}
void tsem_free_matrix(tsem_ctx* h) {
  dfree(h->d_indptr); dfree(h->d_indices); dfree(h->d_raw); dfree(h->d_lut);
  dfree(h->d_amb_row); dfree(h->d_amb_wcode); dfree(h->d_amb_wcode_c); dfree(h->d_slot_row); dfree(h->d_uni_col); dfree(h->d_uni_code);
  dfree(h->d_pisum0); dfree(h->d_twin_rep); dfree(h->d_ucount); dfree(h->d_colcount); dfree(h->d_row_code); dfree(h->d_row_cls);
  tsem_free_layout(h);
  dfree(h->d_pi); dfree(h->d_theta); dfree(h->d_pi_prev); dfree(h->d_theta_prev);
  dfree(h->d_ctab); dfree(h->d_ctab_prev); dfree(h->d_red_own); dfree(h->d_tmp_pi); dfree(h->d_tmp_theta);
  dfree(h->d_c32); dfree(h->d_cs32); dfree(h->d_lut32); dfree(h->d_cnat);
  dfree(h->d_ctl); dfree(h->d_ctld); dfree(h->d_lnls); dfree(h->d_pi_first); dfree(h->d_theta_first); dfree(h->d_user_z);
  dfree(h->d_tie_rows); dfree(h->d_tie_cnt); h->n_ties = 0;
  if (h->d_rep_chunks) { (void)hipFree(h->d_rep_chunks); h->d_rep_chunks = nullptr; h->n_rep_chunks = 0; }
  dfree(h->d_rep_nb); dfree(h->d_rep_rows); dfree(h->d_rep_n); dfree(h->d_exact_n); dfree(h->d_flag_bits); h->flag_words = 0;
  dfree(h->d_group); h->n_groups = 0;
  if (h->d_gtile) { (void)hipFree(h->d_gtile); h->d_gtile = nullptr; h->gtile_bytes = 0; }
  if (h->d_rep_tmp) { (void)hipFree(h->d_rep_tmp); h->d_rep_tmp = nullptr; h->rep_tmp_bytes = 0; }
  h->first_pending = false;
  h->d_red = nullptr;
  h->have_rowstats = h->have_model = false;
  h->N = h->nnz = 0; h->K = 0;
  h->max_code = -1; h->min_code = -1;
  h->lnl3_declined = false;
}



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
  tsem_free_matrix(h);
  dfree(h->d_diffs); dfree(h->d_lnl_part); dfree(h->d_maxcode); dfree(h->d_xerr);
  for (auto& ev : h->ev) (void)hipEventDestroy(ev);
  for (auto& ev : h->pev) (void)hipEventDestroy(ev);
  for (auto& ev : h->ev_rep) if (ev) (void)hipEventDestroy(ev);
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
  else if (k == "split") h->opt_split = v;
  else if (k == "value_format") h->opt_format = v;
  else if (k == "hot_split") h->opt_hot_split = v;
  else if (k == "geometry") h->opt_geo = v;
  else if (k == "sorted_fill") h->opt_sorted = v;
  else if (k == "deconflict") h->opt_deconflict = v;
  else if (k == "reproducible") h->opt_reproducible = v;
  else if (k == "em_precision") h->opt_precision = v;
  else if (k == "kernel_timing") h->opt_timing = v;
  else if (k == "drop_csr_indices") h->opt_drop_indices = v;   // -1 auto (>= 4e9 entries), 0 never, 1 always: see tsem_common.h
  else if (k == "phase_timing") h->opt_phase = v;          // HIP events between the phases of every chunked iteration (tsem_phase_times); a diagnostic: ~5 events per iteration
  else if (k == "report_shortcuts") h->opt_shortcuts = v;
  else if (k == "rowpass_wgs") h->opt_rowpass_wgs = v;
  else if (k == "report_kernel") h->opt_report_kernel = v;   // 0: the generic row pass (k_rowpass<RP_REPORT>) instead of k_report_rows
  else if (k == "report_wgs2") h->opt_report_wgs2 = v;
  else if (k == "report_dbg") h->opt_report_dbg = v;
  else if (k == "report_lanes") h->opt_report_lanes = v;     // capacity (lanes per row x entries per lane) of k_report_rows: 8 .. 256 (0 = from the row lengths)
  else if (k == "issue_early") h->opt_issue = v;       // (kept for old scripts; the exchange has one order now)
  else if (k == "group_tile_bytes") h->opt_group_tile = v;   // per-group sums: bytes of output (groups x K doubles) computed per pass over the matrix
  else if (k == "use_likelihood") h->opt_lnl_fused = v;      // before the matrix is laid out (tsem_set_model), or followed by tsem_prepare_likelihood
  else if (k == "fused_prof") {
    if (v && !h->d_prof) { if (hipMalloc((void**)&h->d_prof, TS_PROF_WORDS * 8) != hipSuccess) return TSEM_ERR_NOMEM; }
    if (h->d_prof) (void)hipMemset(h->d_prof, 0, TS_PROF_WORDS * 8);
    h->prof_steps = v == 2 ? 0 : 64;                       // 2: the start-up stamps only (the per-step stamps slow team 0 down)
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
  h->lut0_zero = lut[0] == 0.0;
  h->lut_increasing = lut_len >= 2 && lut[1] > 0.0;
  for (int i = 2; i < lut_len && h->lut_increasing; ++i) h->lut_increasing = lut[i] > lut[i - 1];
  dfree(h->d_lqtab); h->lq_n = 0; h->lq_tried = false;       // (log Q follows the score table)
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
  tsem_free_matrix(h);
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
  // ... and what they find there must be a valid column / score code / (tsem_setup.hip) popularity id: the report kernels mask the
  // VALUES of entries past a row's end, but index tables with them first (k_report_pack32<E, true> reads its global pi*theta table at
  // the id: a stale 65535 would be 256 KB into whatever follows a 200 KB table)
  TSEM_HIP(hipMemsetAsync(h->d_indices + nnz, 0, sizeof(int32_t) * TS_ENTRY_PAD, h->stream));
  TSEM_HIP(hipMemsetAsync(h->d_raw + nnz, 0, sizeof(uint16_t) * TS_ENTRY_PAD, h->stream));
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
  tsem_setup_preload();                                      // (the set-up unit's code object loads while the check runs)
  uint32_t badw[4] = {0, 0, 0, 0};
  TSEM_HIP(hipMemcpyAsync(badw, d_bad, sizeof(badw), hipMemcpyDeviceToHost, h->stream));
  TSEM_HIP(hipStreamSynchronize(h->stream));
  uint32_t bad = badw[0] | ((badw[2] | badw[3]) && !(badw[0] & 1u) ? 8u : 0u);
  (void)hipFree(d_bad);
  if (bad) {
    tsem_free_matrix(h);
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
  if (dist < 0 || dist > 2 || (dist == 2 && n_cols <= TS_FAMILY)) TSEM_FAIL(TSEM_ERR_ARG, "dist: 0 uniform, 1 zipf, 2 family (more than 256 columns)");
  tsem_free_matrix(h);
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
  TSEM_HIP(hipMemsetAsync(h->d_indices + nnz, 0, sizeof(int32_t) * TS_ENTRY_PAD, h->stream));   // (see tsem_load_scores)
  TSEM_HIP(hipMemsetAsync(h->d_raw + nnz, 0, sizeof(uint16_t) * TS_ENTRY_PAD, h->stream));
  if (n) {
    k_gen_rows<<<cdiv64(n, 128), 128, 0, h->stream>>>(row_begin, n, n_cols, seed, dist, h->d_indptr, h->d_indices, h->d_raw);
    TSEM_HIP(hipGetLastError());
    tsem_setup_preload();                                    // (the set-up unit's code object loads while the generator runs)
    // ... and so does the runtime's path for sizeable copies to pageable host memory: a matrix that comes from the host pays that
    // one-off (7 ms, once per process) inside its upload (tools/time_setup_host.py: the row statistics' read-backs then take 0.09 ms);
    // the generated matrix has no upload, and the one-off used to land in tsem_rowstats' read-backs instead
    {
      std::vector<unsigned char> warm((size_t)1 << 18);
      (void)hipMemcpyAsync(warm.data(), h->d_indptr, std::min<size_t>(warm.size(), sizeof(int64_t) * (size_t)(n + 1)), hipMemcpyDeviceToHost, h->stream);
      (void)hipStreamSynchronize(h->stream);
    }
    TSEM_HIP(hipStreamSynchronize(h->stream));
  }
  (void)hipFree(d_cdf);
  return TSEM_OK;
}

// out[0]: the largest score; out[1] (preset to 0xFFFF): the smallest one above 0 — the range of log Q for the lnl pass's choice of
// form (tsem_em.hip k_log_tab), found on the way
__global__ void k_max_u16(const uint16_t* __restrict__ v, int64_t n, uint32_t* __restrict__ out) {
  uint32_t m = 0, lo = 0xFFFFu;
  auto low = [](uint32_t a, uint32_t x) -> uint32_t { return x != 0u && x < a ? x : a; };
  // eight scores per 16-byte load (hipMalloc alignment); the tail one by one
  const uint4* v4 = reinterpret_cast<const uint4*>(v);
  const int64_t n8 = n / 8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    const uint4 w = v4[i];
    const uint32_t a = max(max(w.x & 0xFFFFu, w.x >> 16), max(w.y & 0xFFFFu, w.y >> 16));
    const uint32_t b = max(max(w.z & 0xFFFFu, w.z >> 16), max(w.w & 0xFFFFu, w.w >> 16));
    m = max(m, max(a, b));
    lo = low(low(low(low(lo, w.x & 0xFFFFu), w.x >> 16), w.y & 0xFFFFu), w.y >> 16);
    lo = low(low(low(low(lo, w.z & 0xFFFFu), w.z >> 16), w.w & 0xFFFFu), w.w >> 16);
  }
  for (int64_t i = n8 * 8 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    m = max(m, (uint32_t)v[i]);
    lo = low(lo, (uint32_t)v[i]);
  }
  m = (uint32_t)sg_max_i<64>((int)m);
  lo = 0xFFFFu - (uint32_t)sg_max_i<64>((int)(0xFFFFu - lo));
  if ((threadIdx.x & 63) == 0 && m) { atomicMax(out, m); atomicMin(out + 1, lo); }
}

int tsem_max_score(tsem_ctx* h, int32_t* max_score) {
  if (!h || !h->d_indptr || !max_score) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  if (h->max_code < 0) {                                     // (the matrix does not change between load / generate calls)
    uint32_t mm[2] = {0u, 0xFFFFu};
    TSEM_HIP(hipMemcpyAsync(h->d_maxcode, mm, 8, hipMemcpyHostToDevice, h->stream));
    if (h->nnz) k_max_u16<<<1024, 256, 0, h->stream>>>(h->d_raw, h->nnz, h->d_maxcode);
    TSEM_HIP(hipMemcpyAsync(mm, h->d_maxcode, 8, hipMemcpyDeviceToHost, h->stream));
    TSEM_HIP(hipStreamSynchronize(h->stream));
    h->max_code = (int32_t)mm[0];
    h->min_code = mm[0] ? (int32_t)mm[1] : -1;             // (smallest stored score above 0; -1: none)
  }
  *max_score = h->max_code;
  return TSEM_OK;
}

// Q for every raw score with the C library's expm1: what a host WITHOUT numpy installs (tests/c_host/run_bundled.c).  Not a device
// function and no handle: plain host arithmetic in the reference's operation order, (r * (1 / max)) * scale (model.py:653 via
// sparse_plus.py:89-91).
int tsem_score_lut(int32_t max_score, double scale_factor, double* lut) {
  if (max_score < 0 || max_score > 65535 || !lut) return TSEM_ERR_ARG;
  if (max_score == 0) { lut[0] = 0.0; return TSEM_OK; }
  const double inv = 1.0 / (double)max_score;
  for (int32_t r = 0; r <= max_score; ++r) lut[r] = std::expm1(((double)r * inv) * scale_factor);
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
  if (indices && h->nnz) { if (int rc = tsem_ensure_indices(h)) return rc; }
  if (indices && h->nnz) TSEM_HIP(hipMemcpy(indices, h->d_indices, sizeof(int32_t) * h->nnz, hipMemcpyDeviceToHost));
  if (raw && h->nnz) TSEM_HIP(hipMemcpy(raw, h->d_raw, sizeof(uint16_t) * h->nnz, hipMemcpyDeviceToHost));
  return TSEM_OK;
}

// Column parts (the tables of one part must fit LDS), rows per block and the fused kernel's geometry, from
// the row statistics and the options; called again when a handle falls back to the two-pass kernels.
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

int tsem_report_stats(tsem_ctx* h, double* kernel_ms, int64_t* algo_bytes, int64_t* deferred_rows, int32_t* kernel) {
  if (!h) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  float ms = 0.f;
  if (h->rep_timed && h->ev_rep[0] && h->ev_rep[1]) {
    TSEM_HIP(hipStreamSynchronize(h->stream));
    if (hipEventElapsedTime(&ms, h->ev_rep[0], h->ev_rep[1]) != hipSuccess) ms = 0.f;
  }
  if (kernel_ms) *kernel_ms = ms;
  if (algo_bytes) *algo_bytes = h->nnz * 4 + h->N * 12;
  if (deferred_rows) *deferred_rows = h->rep_deferred;
  if (kernel) *kernel = h->rep_kernel;
  return TSEM_OK;
}

/* free / total bytes of the device's memory (hipMemGetInfo), and — with a handle — what the handle's matrix keeps resident, by kind:
 * resident[0] CSR row pointers + scores, [1] CSR column ids (0 after option "drop_csr_indices"), [2] popularity ids,
 * [3] blocked layout (entries + offsets), [4] per-row arrays */
int tsem_device_memory(tsem_ctx* h, int device, int64_t* free_bytes, int64_t* total_bytes, int64_t* resident5) {
  if (hipSetDevice(h ? h->device : device) != hipSuccess) return TSEM_ERR_HIP;
  size_t f = 0, t = 0;
  if (hipMemGetInfo(&f, &t) != hipSuccess) return TSEM_ERR_HIP;
  if (free_bytes) *free_bytes = (int64_t)f;
  if (total_bytes) *total_bytes = (int64_t)t;
  if (resident5 && h) {
    const int64_t nz = h->nnz + TS_ENTRY_PAD;
    resident5[0] = h->d_indptr ? 8 * (h->N + 1) + (h->d_raw ? 2 * nz : 0) : 0;
    resident5[1] = h->d_indices ? 4 * nz : 0;
    resident5[2] = h->d_rid16 ? 2 * nz : 0;
    resident5[3] = (h->d_prc ? 4 * h->nnz_pad : 0) + (h->d_pcode ? 2 * h->nnz_pad : 0) + (h->d_pval ? 8 * h->nnz_pad : 0) +
                   (h->d_sb_off ? 12 * (h->nb * h->P + 2) : 0);
    resident5[4] = (h->d_row_code ? 3 * h->N : 0) + (h->d_amb_row ? 6 * h->N_amb : 0) + (h->d_slot_row ? 6 * h->N_amb_pad : 0) +
                   (h->d_uni_col ? 6 * h->N_uni : 0) + (h->d_amb_w ? 8 * h->N_amb_pad : 0) + (h->d_rinv ? 8 * h->N_amb_pad : 0) +
                   (h->d_ypart ? 8 * (int64_t)h->P * h->N_amb_pad : 0);
  }
  return TSEM_OK;
}

int tsem_phase_times(tsem_ctx* h, int reset, double* ms6, int64_t* n_iter) {
  if (!h) return TSEM_ERR_ARG;
  if (ms6) for (int i = 0; i < 6; ++i) ms6[i] = h->phase_ms[i];
  if (n_iter) *n_iter = h->phase_n;
  if (reset) { for (double& v : h->phase_ms) v = 0.0; h->phase_n = 0; }
  return TSEM_OK;
}

int tsem_debug_fused_prof(tsem_ctx* h, uint64_t* out /* 64*16 */) {
  if (!h || !h->d_prof || !out) return TSEM_ERR_ARG;
  TSEM_HIP(hipStreamSynchronize(h->stream));
  TSEM_HIP(hipMemcpy(out, h->d_prof, 64 * 16 * 8, hipMemcpyDeviceToHost));
  return TSEM_OK;
}

int tsem_debug_fused_startup(tsem_ctx* h, uint64_t* out /* 512*8 */) {
  if (!h || !h->d_prof || !out) return TSEM_ERR_ARG;
  TSEM_HIP(hipStreamSynchronize(h->stream));
  TSEM_HIP(hipMemcpy(out, h->d_prof + 64 * 16, 512 * 8 * 8, hipMemcpyDeviceToHost));
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

__global__ void k_log1p_of_log_probe(int n, const double* lq, const double* lc, const double* q, const double* c, double* y) {
  __shared__ double2 tab[FZ_LOGTAB];
  if (threadIdx.x < FZ_LOGTAB) {
    const double ci = 1.0 + (double)threadIdx.x * (1.0 / FZ_LOGTAB);
    tab[threadIdx.x] = make_double2(1.0 / ci, ts_log1p_pos((double)threadIdx.x * (1.0 / FZ_LOGTAB)));
  }
  __syncthreads();
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = fz_log1p_of_log<true>(lq[i] + lc[i], [&]() { return q[i] * c[i]; }, tab);
}
/* log1p(Q c) as the log-table lnl passes form it (fz_log1p_of_log: log Q + log c + exp(-L) | 0 | the table-driven log1p of the exact
 * product) on caller-supplied lq = log Q, lc = log c, q, c (accuracy test hook) */
int tsem_debug_log1p_of_log(int device, int32_t n, const double* lq, const double* lc, const double* q, const double* c, double* y) {
  if (!lq || !lc || !q || !c || !y || n < 0) return TSEM_ERR_ARG;
  if (hipSetDevice(device) != hipSuccess) return TSEM_ERR_HIP;
  double* d = nullptr;
  const size_t m = (size_t)std::max(1, n);
  if (hipMalloc((void**)&d, 8 * 5 * m) != hipSuccess) return TSEM_ERR_NOMEM;
  (void)hipMemcpy(d, lq, 8 * (size_t)n, hipMemcpyHostToDevice); (void)hipMemcpy(d + m, lc, 8 * (size_t)n, hipMemcpyHostToDevice);
  (void)hipMemcpy(d + 2 * m, q, 8 * (size_t)n, hipMemcpyHostToDevice); (void)hipMemcpy(d + 3 * m, c, 8 * (size_t)n, hipMemcpyHostToDevice);
  if (n) k_log1p_of_log_probe<<<(n + 255) / 256, 256>>>(n, d, d + m, d + 2 * m, d + 3 * m, d + 4 * m);
  hipError_t e = hipMemcpy(y, d + 4 * m, 8 * (size_t)n, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  return e == hipSuccess ? TSEM_OK : TSEM_ERR_HIP;
}

// What a pure streaming read reaches on this GPU: grid-stride, eight 16-byte non-temporal loads in flight per thread (the
// best plain variant of tools/ubench/stream.hip), sum into a register, no store.  The "measured-stream peak" beside the
// nominal 8 TB/s in bench.py's roofline block.
typedef unsigned int sp_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(1024) void k_stream_probe(const sp_u32x4* __restrict__ p, int64_t n16, uint32_t* __restrict__ out) {
  sp_u32x4 acc = {0, 0, 0, 0};
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 7 * stride < n16; i += 8 * stride) {
    sp_u32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(p + i + u * stride);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  for (; i < n16; i += stride) acc += __builtin_nontemporal_load(p + i);
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = acc.x;      // (never true for the memset pattern: keeps the loads alive)
}
/* best of `reps` timed launches over a scratch buffer of `bytes` (allocated and freed here); *gbs = bytes / time */
int tsem_debug_stream_read(int device, int64_t bytes, int32_t reps, double* gbs) {
  if (!gbs || bytes < (1 << 20) || reps < 1) return TSEM_ERR_ARG;
  if (hipSetDevice(device) != hipSuccess) return TSEM_ERR_HIP;
  DevTmp buf, out;
  if (hipMalloc(&buf.p, (size_t)bytes) != hipSuccess) { buf.p = nullptr; return TSEM_ERR_NOMEM; }
  if (hipMalloc(&out.p, 64) != hipSuccess) { out.p = nullptr; return TSEM_ERR_NOMEM; }
  if (hipMemset(buf.p, 1, (size_t)bytes) != hipSuccess) return TSEM_ERR_HIP;
  hipEvent_t a, b;
  if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return TSEM_ERR_HIP;
  const int64_t n16 = bytes / 16;
  float best = 1e30f;
  for (int r = 0; r <= reps; ++r) {                          // launch 0 warms up
    (void)hipEventRecord(a, nullptr);
    k_stream_probe<<<4096, 1024>>>(buf.as<sp_u32x4>(), n16, out.as<uint32_t>());
    (void)hipEventRecord(b, nullptr);
    if (hipEventSynchronize(b) != hipSuccess) { (void)hipEventDestroy(a); (void)hipEventDestroy(b); return TSEM_ERR_HIP; }
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    if (r && ms < best) best = ms;
  }
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  *gbs = (double)(n16 * 16) / ((double)best * 1e-3) / 1e9;
  return TSEM_OK;
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
  info[20] = h->n_bin_repeats; info[21] = h->opt_reproducible ? (h->bin_inexact ? 3 : (h->len_gt[5] ? 2 : 1)) : 0;     // 2: some row has more than 256 entries, see telescope_em.h
  info[22] = h->exact_single ? 1 : 0;                      // reproducible: both pieces in one pass
  info[26] = h->em_rows ? 1 : 0;                           // K beyond 64 column parts: plain CSR row passes (no blocked layout)
  info[25] = h->n_single_part;                             // ambiguous rows whose entries all lie in ONE column part (they would need no exchange)
  info[24] = h->split ? 1 : 0;                             // split layout: two light passes per iteration (K > 61 440)
  // the lnl pass's choice of form (tsem_em.hip k_log_tab): stored entries in columns that may reach the exact branch of the log form,
  // as counted before the LAST lnl pass (-1: no choice armed), and the count above which the per-entry logarithm ran instead
  info[29] = -1; info[30] = (int64_t)(1e-3 * (double)h->nnz);
  if (h->d_lq_mid && h->lq_choice[0] > 0 && hipSetDevice(h->device) == hipSuccess) {
    unsigned long long m = 0;
    if (hipMemcpyAsync(&m, h->d_lq_mid, 8, hipMemcpyDeviceToHost, h->stream) == hipSuccess && hipStreamSynchronize(h->stream) == hipSuccess) info[29] = (int64_t)m;
  }
  info[28] = h->lq_lin;                                    // ... and log Q is (code / max) * scale itself: no table (code entries, the reference's score table)
  info[27] = h->lq_n;                                      // entries of the log Q table of the lnl passes (0: not built yet / does not fit / does not apply)
  info[23] = h->lnl3 ? 1 : 0;                              // the layout lets the EM pass carry the previous iteration's log-likelihood (option "use_likelihood")
  // rows whose sum the report / row passes redid in the reference's order of additions since the matrix was loaded (near-ties of two z
  // values or of a z value and conf_prob: tsem_report.hip, tsem_npsum.h)
  info[31] = 0;
  if (h->d_exact_n && hipSetDevice(h->device) == hipSuccess) {
    unsigned long long m = 0;
    if (hipMemcpyAsync(&m, h->d_exact_n, 8, hipMemcpyDeviceToHost, h->stream) == hipSuccess && hipStreamSynchronize(h->stream) == hipSuccess) info[31] = (int64_t)m;
  }
  return TSEM_OK;
}


}  // extern "C"
