// Shared declarations for libtelescope_em.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cmath>
#include <string>
#include <vector>

#include <rccl/rccl.h>   // types and enum values only: the library is resolved at run time (nccl_api, tsem.hip), never linked

#include "telescope_em.h"

#define TSEM_HIP(call)                                                           \
  do {                                                                           \
    hipError_t e_ = (call);                                                      \
    if (e_ != hipSuccess) {                                                      \
      h->err = std::string(#call) + ": " + hipGetErrorString(e_) + " (" +        \
               __FILE__ + ":" + std::to_string(__LINE__) + ")";                  \
      return TSEM_ERR_HIP;                                                       \
    }                                                                            \
  } while (0)

#define TSEM_FAIL(code, msg)                                                     \
  do {                                                                           \
    h->err = (msg);                                                              \
    return (code);                                                               \
  } while (0)

// ---- counter-based hash shared with telescope_amd/synthetic.py --------------
#define TS_GOLDEN 0x9E3779B97F4A7C15ull
#define TS_M1 0xBF58476D1CE4E5B9ull
#define TS_M2 0x94D049BB133111EBull
#define TS_SALT_LEN 0xA5A5A5A5A5A5A5A5ull
#define TS_SALT_UNIQ 0x5BD1E9955BD1E995ull
#define TS_SALT_COL0 0xC2B2AE3D27D4EB4Full
#define TS_SALT_SCORE 0x165667B19E3779F9ull
#define TS_SALT_FAM 0x27D4EB2F165667C5ull
#define TS_FAMILY 256                 /* loci per family of the synthetic distribution 2 ('family', telescope_amd/synthetic.py) */

__host__ __device__ inline uint64_t ts_mix64(uint64_t z) {
  z += TS_GOLDEN;
  z = (z ^ (z >> 30)) * TS_M1;
  z = (z ^ (z >> 27)) * TS_M2;
  return z ^ (z >> 31);
}
__host__ __device__ inline uint64_t ts_hash3(uint64_t seed, uint64_t row, uint64_t k) {
  return ts_mix64(ts_mix64(seed ^ (row * TS_GOLDEN)) ^ (k * TS_M1));
}

// LDS budget for the per-part tables (c and acc, 8 B each per column).
constexpr int TS_LDS_TABLE_BYTES = 120 * 1024;
constexpr int TS_MAX_KP = TS_LDS_TABLE_BYTES / 16;   // 7680 columns per part
constexpr int TS_MAX_KP_LNL = 5120;                  // ... when a part keeps pi*theta of TWO parameter sets and the accumulators (option "use_likelihood": 24 B per column)
constexpr int TS_MAX_KP3 = 4800;                     // ... when a part keeps THREE tables in LDS (option "reproducible", one pass: 26 B per column)
constexpr int TS_LDS_MAX = 160 * 1024;
constexpr int TS_ENTRY_PAD = 320;   // entries of padding behind indices[] / raw[] (k_report_rows reads 16-entry lanes past a row's end)
constexpr int TS_STRANDS = 16;   // strand-transposed entry order inside a sub-block

struct tsem_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  int n_cu = 256;

  // ---- CSR of raw scores (rows owned by this rank) ----
  int64_t N = 0, nnz = 0;
  int32_t K = 0;
  int64_t* d_indptr = nullptr;
  int32_t* d_indices = nullptr;
  uint16_t* d_raw = nullptr;
  int64_t opt_drop_indices = -1;    // option "drop_csr_indices": free the CSR column ids (4 B per entry) once the blocked layout and the 2-byte popularity ids
                                    // exist — col = col_of_id[rid16], rebuilt on demand (tsem_ensure_indices) for the generic row passes, z export, a
                                    // layout rebuild.  -1 auto: from 4e9 stored entries on; 0 never; 1 always
  double* d_lut = nullptr;
  int lut_len = 0;
  std::vector<double> lut_host;
  bool lut0_zero = false;           // lut[0] == 0: a masked score code adds nothing (k_report_pack)
  bool lut_increasing = false;      // lut[1] > 0 and strictly increasing from there: a larger score code is a larger Q (k_report_init_codes)

  // ---- row classes ----
  int64_t N_amb = 0, N_uni = 0, nnz_amb = 0;
  int32_t* d_amb_row = nullptr;     // [N_amb]  compact ambiguous row -> CSR row
  uint16_t* d_amb_wcode = nullptr;  // [N_amb_pad] per row SLOT: max raw code of the row (w = lut[code]), 0 = hole
  uint16_t* d_amb_wcode_c = nullptr;  // [N_amb]   same, per compact ambiguous row
  int32_t* d_slot_row = nullptr;    // [N_amb_pad] row slot -> CSR row, -1 = hole
  int32_t* d_uni_col = nullptr;     // [N_uni]
  uint16_t* d_uni_code = nullptr;   // [N_uni]
  uint32_t* d_maxcode = nullptr;    // [1]
  uint16_t* d_row_code = nullptr;   // [N] per CSR row: its largest raw score (w_i = lut[code], model.py:690); tsem_rowstats, kept for tsem_export_rowinfo
  uint8_t* d_row_cls = nullptr;     // [N] 0 empty row, 1 unique (Y_i = 0), 2 ambiguous (Y_i = 1, model.py:679)
  int32_t max_code = -1;            // largest raw score of the resident matrix (-1: not taken yet); tsem_max_score
  int32_t min_code = -1;            // smallest stored score above 0 (taken by the same pass; -1: none / not taken yet)
  bool have_rowstats = false;
  bool bin_inexact = false;         // option "reproducible": a pass gave up moving a column's grid after 40 repeats (its sums are not exact)

  // ---- model scalars (GLOBAL after set_model) ----
  double W_tot = 0, W_amb = 0, w_max = 0, pi_prior = 0, theta_prior = 0;
  double* d_pisum0 = nullptr;  // [K]
  uint32_t* d_ucount = nullptr;            // [K+1] unique rows with a positive score per column (local); [K]: some stored score is 0
  bool has_zero_score = false;             // ... read back once by tsem_rowstats
  unsigned long long* d_colcount = nullptr;   // [K] stored entries per column (local rows)
  bool em_cur = false, em_prev = false;    // current / previous pi, theta come from tsem_set_model or the M-step, not from tsem_set_params
  int64_t opt_split = -1;                  // split layout: -1 when K needs it, 1 forced (tests), 0 never
  int64_t opt_issue = -1;                  // fused kernel, exchange wave: partner loads before the combine (1), after it (0), -1 auto
  int64_t opt_rowpass_wgs = 2;             // workgroups per CU of the reassign row pass (modes other than `all`)
  int64_t opt_report_kernel = 1;           // tsem_report_colsums runs k_report_rows (0: the generic k_rowpass<RP_REPORT>)
  int64_t opt_report_dbg = 0;
  int64_t opt_report_wgs2 = 0;             // 1: two of its workgroups per CU with half the LDS tables each (experiments)
  int64_t opt_report_lanes = 0;            // its capacity per row, lanes x entries per lane (0 = auto)
  unsigned long long len_gt[6] = {0, 0, 0, 0, 0, 0};   // rows with more than 8, 16, 32, 64, 128, 256 entries (tsem_rowstats)
  int64_t opt_shortcuts = 1;               // tsem_reassign answers `all`(initial) and `unique` from the setup counts
  int32_t* d_twin_rep = nullptr;  // [K] representative column of each exact-twin class
  std::vector<uint64_t> col_count;  // global entries per column
  std::vector<int32_t> twin_rep_host;
  int64_t row_offset = 0;      // global index of this rank's first row
  int n_twin_cols = 0;
  bool have_model = false;

  // ---- column partition + blocked COO layout of ambiguous rows ----
  int P = 1, Kp = 0, Kpad = 0, R = 2048;
  int64_t nb = 0, N_amb_pad = 0, nnz_pad = 0;
  uint32_t* d_colmap = nullptr;     // [K]    col -> (part<<16 | lcol)
  int32_t* d_col_of_pc = nullptr;   // [Kpad] part*Kp+lcol -> col or -1
  uint16_t* d_rid16 = nullptr;      // [nnz + TS_ENTRY_PAD] popularity id (slot * P + part) of every stored entry's column: the report pass
  int32_t* d_col_of_id = nullptr;   // [Kpad] id -> col or -1
  int64_t* d_sb_off = nullptr;      // [nb*P+1] entry offsets (multiples of 4)
  double* d_pval = nullptr;         // [nnz_pad]  Q values (fp64 entry format)
  uint16_t* d_pcode = nullptr;      // [nnz_pad]  raw score codes (code16 entry format: Q = lut[code])
  int64_t opt_sorted = -1;          // -1 auto, 0: strand-transposed sub-blocks, 1: row-ordered sub-blocks (fused layout)
  int64_t opt_timing = 1;           // HIP events around every n-th EM pass (tsem_kernel_stats); 0 = none
  int64_t opt_precision = 0;        // 1: the EM pass in fp32 arithmetic (diagnostic for the config-3 tolerance sweep)
  float *d_c32 = nullptr, *d_cs32 = nullptr, *d_lut32 = nullptr;
  int64_t opt_reproducible = 0;     // 1: order-independent (exact, binned) column sums in the fused EM pass: pi / theta / lnl bit-identical from run to run,
                                    //    two passes per iteration (profiles/HISTORY.md 5.1)
  uint16_t* d_ebias = nullptr;      // [Kpad] per slot: biased exponent of the bound 2^E of its contributions
  uint8_t* d_ovf = nullptr;         // [Kpad] a contribution reached its slot's bound in the last pass
  double* d_red_hi = nullptr;       // [K+2] column sums of the high pieces
  int64_t opt_lnl_fused = 0;        // option "use_likelihood" = 1: lay the matrix out so that the EM pass can sum the previous iteration's log-likelihood
                                    //    as well (fused kernel MODE 4: three tables per part in LDS, tsem_fused.h); tsem_em_chunk then needs no lnl pass per iteration
  bool lnl3 = false;                // the current layout allows it
  bool lnl3_declined = false;       // tsem_prepare_likelihood asked once and the geometry said no (reset with the matrix)
  bool em_rows = false;             // K beyond 64 x 7680 columns: no blocked layout at all, the EM pass and the log-likelihood are plain CSR row passes
                                    // with global gathers and fp64 atomics (any K; slow: a completeness path, tsem_em.hip k_em_rows)
  int64_t n_single_part = 0;        // ambiguous rows with all their entries in one column part (layout statistic, tsem_layout_info[25])
  bool split = false;               // SPLIT layout (K > 8 x 7680 on the fused path): parts of up to 15 424 columns, one LDS table per pass — a row-sum pass and a
                                    // scatter pass per iteration (tsem_fused.h MODE 5 / 7), the log-likelihood over column halves (MODE 8)
  double* d_rinv = nullptr;         // [N_amb_pad] recip0(row sum) of the last MODE 4 pass (what the next one needs of its E-step)
  // log tables of the log-likelihood passes (round 5): log1p(Q c) = log Q + log c + 1 / (Q c) for Q c >= 2^27 — one add and a table
  // look-up instead of a logarithm per stored entry (tsem_fused.h, fz_lnl_term)
  double* d_lctab = nullptr;        // [Kpad] log(pi * theta), permuted like d_ctab; rebuilt before every pass that reads it
  double* d_lqtab = nullptr;        // [lq_n] log Q: per score code (code entries), or indexed by the top bits of Q (fp64 entries)
  int lq_n = 0, lq_shift = 0, lq_base = 0;   // fp64 entries: index = (high word of Q >> lq_shift) - lq_base
  bool lq_tab_fits = false;         // the log Q table fits the LDS the layout leaves
  int lq_lin = 0, lq_c0 = 0;        // code entries with the reference's own score table: log Q = (code * lq_a) * lq_b from lq_c0 on (no table at all)
  double lq_a = 0, lq_b = 0;
  bool lq_tried = false;            // the tables were attempted for this layout (lq_n == 0 afterwards: they do not fit / do not apply)
  double lq_lo = 0, lq_hi = 0;      // log Q of the smallest / largest stored score (> 0): which columns may reach the exact branch of the log form
  unsigned long long* d_lq_mid = nullptr;   // [1] stored entries of such columns, counted by k_log_tab before every lnl pass (the device picks the form)
  int64_t lq_choice[2] = {0, 0};    // lnl passes enqueued with the selection armed | (diagnostic) reserved
  bool lag_agreed = false;          // row-sharded runs: EVERY rank can run MODE 4 (decided once per run, dropped for good after a time-out anywhere)
  bool lag_valid = false;           // the iteration committed last still owes its lnl, and d_rinv / d_ctab_prev are what the next MODE 4 pass needs for it
  bool exact_single = false;        // reproducible: both pieces in ONE pass (three tables per part fit the LDS with <= 8 parts)
  double* d_fpartial2 = nullptr;    // [fz_teams][Kpad] the low pieces' team partials of that pass
  int16_t* d_ehist = nullptr;       // [2K] per column: exponent (+4) of its last sum, and by how many bits it fell in the last iteration
  uint32_t* d_binflag = nullptr;    // [1] some column wants the pass repeated with another exponent
  int64_t n_bin_repeats = 0;        // passes repeated because an exponent had to move
  int64_t opt_deconflict = -1;      // conflict-aware entry order inside the rows of the row-ordered code layout (k_sb_deconflict): -1 auto = on
                                    //    (round 3: 4 ms of setup at 2e9 entries for -6 % per EM pass; round 2's version cost 14 ms and was opt-in)
  int64_t opt_geo = -1;             // -1 auto; 0 / 2 force the geometry of teams of 1-4 (experiments)
  double run_len_est = 0.0;         // mean entries per ambiguous row and column part (set by tsem_rowstats)
  bool sorted_layout = false;       // sub-blocks stored in row order (k_sb_fill_sorted)
  int geo = 0;                      // fused kernel geometry (tsem_fused.h): exchange waves x row pairs per lane
  bool fmt_wcode = false;           // fp64 entries, but the score table sits in LDS for the row weights (kernel FMT 2)
  bool fmt_code = false;            // entry format of the blocked layout: false = fp64 values, true = 2-byte codes
  int64_t opt_hot_split = 1;        // 1: very popular columns get several accumulator slots (see build_layout)
  int hot_extra = 0;                // spare slots per part reserved for them
  int n_hot_cols = 0;               // columns that were split
  int64_t opt_format = 0;           // 0 auto (codes when the fused kernel runs and the table fits LDS), 1 fp64, 2 codes
  uint32_t* d_prc = nullptr;        // [nnz_pad]  lrow<<16 | lcol
  double* d_ypart = nullptr;        // [P][N_amb_pad] partial row sums
  int G1 = 1, G2 = 1, T1 = 512, T2 = 1024;
  double* d_partial = nullptr;      // [G2][Kpad]
  double* d_lnl_part = nullptr;     // [4096]
  int em_kernel = TSEM_EMK_AUTO;
  int64_t opt_R = 0, opt_P = 0, opt_chunk = 0, opt_dbg = 0;
  bool use_fused = false;
  int64_t max_subblock = 0;
  int64_t last_slow_path = 0;       // exchange granules that missed their tag in the last checked pass
  int fz_grid = 0, fz_teams = 0;
  double* d_fpartial = nullptr;     // [fz_teams][Kpad]
  double* d_amb_w = nullptr;        // [N_amb_pad] fragment weights
  uint32_t* d_sb_q32 = nullptr;     // [nb*P+2] sub-block offsets / 4
  bool fused_launched = false;
  unsigned long long* d_prof = nullptr;
  int prof_steps = 64;

  // ---- parameters ----
  double *d_pi = nullptr, *d_theta = nullptr, *d_pi_prev = nullptr, *d_theta_prev = nullptr;
  double *d_ctab = nullptr, *d_ctab_prev = nullptr;  // [Kpad] permuted pi*theta
  double* d_user_z = nullptr;       // [nnz] caller-assigned z (TSEM_Z_USER), NaN = not in z's pattern
  int32_t* d_tie_rows = nullptr;    // rows with several best hits, in row order, and their counts: left by the last
  int32_t* d_tie_cnt = nullptr;     // tsem_report_colsums for tsem_report_ties / tsem_reassign_rows
  int64_t n_ties = 0;
  int32_t* d_group = nullptr;       // [N] row -> group of the per-group sums (tsem_set_groups), -1 = none
  int32_t n_groups = 0;
  void* d_gtile = nullptr; size_t gtile_bytes = 0;   // one tile of per-group sums (by column | by id), kept between calls
  int64_t opt_group_tile = 0;       // bytes of per-group output computed per pass over the matrix (0 = 1 GB)
  int32_t *d_rep_nb = nullptr, *d_rep_rows = nullptr;   // [N] scratch of tsem_report_colsums, kept between calls
  unsigned long long* d_rep_n = nullptr;
  struct RpChunk* d_rep_chunks = nullptr; int64_t n_rep_chunks = 0; int rep_chunk_E = 0;   // k_report_pack's packing of the rows into wave-sized chunks (tsem_report_pack.h)
  uint32_t* d_flag_bits = nullptr; int64_t flag_words = 0;   // near-tie bitmap of the generic row passes (+ their counter), kept between calls
  unsigned long long* d_exact_n = nullptr;   // [1] rows whose sum a report / row pass redid in the reference's order of additions (near-ties, tsem_npsum.h)
                                             //     since the matrix was loaded; tsem_layout_info[31]
  void* d_rep_tmp = nullptr; size_t rep_tmp_bytes = 0;
  double* d_red = nullptr;          // reduce buffer (K+2), internal or bound
  double* d_red_own = nullptr;
  int64_t red_count = 0;
  double* d_diffs = nullptr;        // [TS_DIFF_RING]
  double *d_tmp_pi = nullptr, *d_tmp_theta = nullptr;
  double* d_cnat = nullptr;         // [K] pi*theta in column order for the CSR row passes

  // ---- fused-kernel exchange state ----
  double* d_xchg = nullptr;
  uint32_t* d_xflags = nullptr;
  uint32_t* d_fz_aux = nullptr;     // [0] k_update's block counter, [2..3] error word / miss counter saved by its cleanup
  bool fz_clean = false;            // sync words and exchange ring are zero (k_update cleaned them after the last pass)
  uint32_t* d_xerr = nullptr;

  // ---- device-side loop control (tsem_em_chunk) ----
  uint32_t* d_ctl = nullptr;        // [0] stop: 0 run, 1 converged, 2 a rank's EM pass timed out, 3 ... its lnl pass  [1] iterations committed
  double* d_ctld = nullptr;         // [0] lnl of the previous iteration  [1..2] lnl reduce slots (value, error flag)
  double* d_lnls = nullptr;         // [TS_DIFF_RING]
  double *d_pi_first = nullptr, *d_theta_first = nullptr;   // params after the first iteration of the run (model.py:776-778)
  bool first_pending = false;       // the next committed update saves them
  double lnl_prev_seed = INFINITY;  // lnl the first iteration of the next run is compared with (model.py:683,786)
  int64_t n_fallbacks = 0;          // time-outs answered by switching to the two-pass kernels

  // ---- communicator (row-sharded runs) ----
  tsem_comm* comm = nullptr;

  // ---- instrumentation ----
  int64_t opt_phase = 0;            // option "phase_timing": HIP events between the phases of every iteration of tsem_em_chunk (tsem_phase_times)
  std::vector<hipEvent_t> pev;      // TS_PHASE_MARKS events per iteration of the current chunk
  std::vector<uint8_t> pev_set;     // ... which of them were recorded
  int pev_iter = -1;                // iteration of the chunk being enqueued (-1: outside tsem_em_chunk, nothing is marked)
  double phase_ms[6] = {0, 0, 0, 0, 0, 0};   // pass | column reduce | all-reduce | update | gap to the next iteration | first mark to last mark
  int64_t phase_n = 0;              // iterations summed into phase_ms
  std::vector<hipEvent_t> ev;       // pairs
  hipEvent_t ev_rep[2] = {nullptr, nullptr};   // around the dominant kernel of the last tsem_report_colsums (tsem_report_stats)
  bool rep_timed = false; int rep_kernel = 0; int64_t rep_deferred = 0;
  size_t ev_used = 0;
  double em_ms_acc = 0;
  int64_t em_launches = 0, em_timed = 0;
};

constexpr int TS_DIFF_RING = 65536;
constexpr int TS_PHASE_MARKS = 5;   // before the pass | after it | after the column reduce | after the all-reduce | after the update
constexpr int TS_PROF_WORDS = 64 * 16 + 512 * 8;   // option "fused_prof": per-step slots of team 0 + start-up stamps of up to 512 workgroups

struct tsem_local_group;            // in-process transport: several handles on ONE device (tsem_comm_create_local)

struct tsem_comm {
  ncclComm_t nccl = nullptr;        // RCCL transport (one process per GPU)
  tsem_local_group* local = nullptr;   // in-process transport (tests, several engines sharing a GPU)
  int device = 0, rank = 0, world = 1;
  uint64_t epoch = 0;               // collectives issued so far (local transport: slot parity)
  void* d_stage = nullptr;          // staging buffer of tsem_comm_allreduce_host
  size_t stage_bytes = 0;
  bool active() const { return nccl != nullptr || local != nullptr; }
};
