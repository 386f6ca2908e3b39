/*
 * telescope_em.h — C ABI of libtelescope_em.so, the MI355X (gfx950) engine for
 * Telescope's EM reassignment path.
 *
 * The reference has no FFI layer: its boundary is the Python class
 * `TelescopeLikelihood` (telescope/utils/model.py:631-865) on top of
 * `csr_matrix_plus` (telescope/utils/sparse_plus.py:24-174) and scipy.sparse.
 * Each entry point below names the reference code it replaces.  The Python
 * host mirror (telescope_amd/likelihood.py) binds these with ctypes; the
 * binding a reference maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C, opaque handle, `int` status (0 = TSEM_OK, <0 = error; text via
 *     tsem_last_error).  No exceptions or abort() cross the boundary.
 *   - one handle == one GPU == one host thread (not thread-safe).  Multi-GPU is
 *     one process per GPU; the only per-iteration exchange is a sum all-reduce
 *     of the reduce buffer (K+2 doubles: the per-locus column sums and an error
 *     flag).  With a communicator attached (tsem_comm_*) the library issues it
 *     itself — ncclAllReduce (RCCL over xGMI) on the engine's stream between the
 *     EM pass and the parameter update, no host round trip (tsem_em_chunk).  A
 *     host may instead do it between tsem_em_pass() and tsem_em_update().
 *   - host pointers are borrowed for the duration of the call; the library owns
 *     all device memory except a reduce buffer bound with
 *     tsem_bind_reduce_buffer().
 *   - there is no CPU fallback: tsem_create fails if no HIP device is usable.
 */
#ifndef TELESCOPE_EM_H
#define TELESCOPE_EM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSEM_OK            0
#define TSEM_ERR_ARG      -1   /* bad argument / wrong state              */
#define TSEM_ERR_HIP      -2   /* HIP runtime error (text in last_error)  */
#define TSEM_ERR_NOMEM    -3
#define TSEM_ERR_TIMEOUT  -4   /* in-kernel hand-off watchdog fired       */

/* reassign methods, model.py:808-865 */
#define TSEM_RA_EXCLUDE 0
#define TSEM_RA_CHOOSE  1
#define TSEM_RA_AVERAGE 2
#define TSEM_RA_CONF    3
#define TSEM_RA_UNIQUE  4
#define TSEM_RA_ALL     5

/* which parameters define z: */
#define TSEM_Z_PREV     0   /* params before the last M-step == reference self.z (model.py:795) */
#define TSEM_Z_CUR      1   /* current params */
#define TSEM_Z_INITIAL  2   /* Q.norm(1), model.py:837 (initial=True)      */
#define TSEM_Z_USER     4   /* the z installed with tsem_set_user_z, used as is (a caller assigned tl.z, model.py:837) */
#define TSEM_Z_FIRST    3   /* tsem_get_params only: params after the FIRST iteration of the last run
                               == pi_init / theta_init (model.py:776-778) */

/* EM kernel selection (tsem_set_option "em_kernel") */
#define TSEM_EMK_AUTO     0
#define TSEM_EMK_TWOPASS  1   /* phase kernels, entries read twice                 */
#define TSEM_EMK_FUSED    2   /* persistent single-pass kernel with row-sum exchange */

typedef struct tsem_ctx tsem_ctx;

/* ---- lifecycle ---------------------------------------------------------- */
int  tsem_create(tsem_ctx** out, int device);
void tsem_destroy(tsem_ctx* h);
const char* tsem_last_error(const tsem_ctx* h);      /* h may be NULL: last create error */
int  tsem_set_stream(tsem_ctx* h, void* hip_stream); /* launch on this hipStream_t (default: null stream) */
/* options, set before tsem_rowstats (the layout is built from them):
 *   "em_kernel"    TSEM_EMK_*                      "block_rows", "parts"  override the layout geometry
 *   "value_format" 0 auto (2-byte score codes + LDS score table when the fused kernel runs and the table
 *                  has <= 2048 entries), 1 fp64 Q values, 2 codes (error if not possible)
 *   "hot_split"    1 (default): very popular columns get several accumulator slots
 *   "row_offset"   global index of this rank's first row (synthetic generator, column signatures)
 *   "report_shortcuts" 1 (default): tsem_reassign answers `all` (initial) and `unique` from counts taken at
 *                  setup instead of a pass over the matrix (the same numbers; 0 forces the pass)
 *   "report_kernel" which kernels serve tsem_report_colsums / tsem_reassign_rows / tsem_reassign_groups: 1 (default) the streaming
 *                  report kernels, incl. the passes over the INITIAL z that run on the 2-byte score codes alone (no `conf` column
 *                  wanted, a strictly increasing score table, no stored score of 0); 2 the streaming kernels without those
 *                  codes-only passes (timing comparisons); 0 the generic 16-lanes-per-row pass for everything (same results)
 *   "kernel_timing" n: HIP events around every n-th EM pass for tsem_kernel_stats (default 1, 0 = off)
 *   "phase_timing" 1: HIP events at the phase boundaries of every chunked iteration (tsem_phase_times)
 *   "drop_csr_indices" free the CSR column ids (4 of the 14 B per stored entry the default layout keeps resident) once the
 *                  blocked layout and the 2-byte popularity ids exist; they are rebuilt on demand (col = col_of_id[id]) for
 *                  the generic row passes, z export, tsem_export_csr and a layout rebuild.  -1 (default): from 4e9 stored
 *                  entries on; 0 never; 1 always
 *   "deconflict"   conflict-aware entry order inside the rows of the row-ordered code layout (LDS bank conflicts of the
 *                  column scatter 3.2 -> 2.4 lanes per class: -6 % per EM pass for ~4 ms of setup at 2e9 entries);
 *                  default -1 = on, 0 = off
 *   "reproducible" 1: ORDER-INDEPENDENT column sums in the fused EM pass.  Every contribution w*z is cut into a high and a low piece on
 *                  a per-column power-of-two grid; sums of such pieces are exact in fp64, so the unordered LDS atomics add up to
 *                  the same bits whatever their order: pi, theta, lnl and the iteration count are bit-identical from run to run
 *                  (same build, same device type, same number of ranks).  Two passes per iteration, plus a repeated pass when a
 *                  column's grid has to move (tsem_layout_info[20] counts them); the rows keep their entry order ("deconflict"
 *                  is off), so that a row's partial sum is one run ending in at most two atomics.  That holds for rows of up to
 *                  256 entries; with longer rows tsem_layout_info[21] reports 2 instead of 1 and the last bit of such a row's
 *                  sum may depend on timing.  3: a pass gave up moving a column's grid after 40 repeats — its sums are not
 *                  guaranteed exact (never seen; reported instead of hidden).  A hand-off time-out in this mode redoes the pass on the
 *                  fused kernel (the two-pass kernels have no exact sums); repeated time-outs end the run with TSEM_ERR_TIMEOUT on every rank.  Needs the fused kernel and a score table of <= 2048 entries; column sums are
 *                  within (entries of the column) x 2^-41 of exact, typically one fp64 rounding.  The float-valued sums of
 *                  tsem_reassign / _rows / _groups / tsem_report_colsums (conf, average) are accumulated exactly as well
 *                  (two atomics per value).  1: both pieces in ONE pass over THREE tables per part when they fit the LDS (score
 *                  codes, at most 4800 columns per part with at most 8 parts — i.e. K <= 19k with teams of 4, K <= 38k with
 *                  long rows) — ~1.65x the default mode's time per iteration instead of ~2.5x; 2: always the two-pass form
 *                  (tsem_layout_info[22] says which one runs).  Default 0.
 *   "split"        how many loci the layout takes.  K <= 61 440 (8 column parts of 7680): the fused single-pass kernel.  K <= 122 880:
 *                  the SPLIT layout of the fused kernel (-1, default: when K needs it) — parts of up to 15 360 columns, one LDS table
 *                  per pass: a row-sum pass and a scatter pass per iteration (every entry read twice), the log-likelihood over two
 *                  halves of every part's columns; ~1.4-1.5x the fused kernel's time per entry.  1 forces it on a smaller matrix
 *                  (tests; needs "parts" >= 5), 0 forbids it (the two-pass kernels then take over at K > 61 440).  K <= 491 520:
 *                  the two-pass kernels (64 column parts).  Beyond: plain CSR row passes with global fp64 atomics — any K, an order
 *                  of magnitude slower per entry (tsem_layout_info[24] / [26] say which form runs).
 *   "em_precision" 1: the EM pass in fp32 arithmetic (row sums, posteriors and column sums in fp32) — a
 *                  DIAGNOSTIC for the fp32-vs-fp64 tolerance sweep of BASELINE config 3, not a product path
 *   "fused_dbg", "fused_prof", "chunk_blocks"     timing experiments */
int  tsem_set_option(tsem_ctx* h, const char* key, int64_t value);
int  tsem_synchronize(tsem_ctx* h);

/* ---- score matrix (model.py:638-653; sparse_plus.py:89-91) ---------------
 * CSR of uint16 raw scores for the rows this rank owns.  `lut[r]` is
 * Q = expm1((r * (1/max_score)) * 100.) for r = 0..lut_len-1, computed by the
 * host with the reference's numpy expression so Q is bit-identical; max_score
 * is the GLOBAL maximum (all ranks).  The arrays are borrowed for the call, copied to HBM and validated there
 * (non-decreasing row pointers, column ids in [0, n_cols), scores < lut_len): TSEM_ERR_ARG and no matrix otherwise.
 * `lut` may be NULL: the table then follows with tsem_set_lut once the (global) maximum is known — tsem_max_score
 * takes it on the device, so the host never has to scan the entries. */
int  tsem_load_scores(tsem_ctx* h, int64_t n_rows, int32_t n_cols,
                      const int64_t* indptr, const int32_t* indices,
                      const uint16_t* raw, const double* lut, int32_t lut_len);
/* Generate rows [row_begin,row_end) of the synthetic matrix on the device
 * (telescope_amd/synthetic.py is the bit-exact CPU twin).  `len_cdf` is
 * synthetic.poisson_cdf_u32(mean).  dist: 0 uniform, 1 zipf, 2 family (every row inside one family of 256 consecutive loci).  */
int  tsem_generate(tsem_ctx* h, int64_t row_begin, int64_t row_end, int32_t n_cols,
                   const uint32_t* len_cdf, int32_t cdf_len, uint64_t seed,
                   int32_t dist, double uniq_frac);
/* largest raw score of the local rows; (re)install the Q lookup table once the
 * GLOBAL maximum is known (model.py:640,653) */
int  tsem_max_score(tsem_ctx* h, int32_t* max_score);
int  tsem_set_lut(tsem_ctx* h, const double* lut, int32_t lut_len);
/* The table for a host without numpy (the C host of tests/c_host/): lut[r] = expm1((r * (1 / max_score)) * scale_factor), r = 0 ..
 * max_score, with the C library's expm1.  NOTE what this is not: numpy's `expm1` — what the reference evaluates (model.py:653) — is a
 * SIMD routine on AVX-512 hosts and differs from libm's in the last bit of ~10 % of the entries (measured: 23 of 213 for max_score
 * 212, 6493 of 65536).  Every result then agrees with the reference's to ~1e-15 relative instead of bit for bit; a host that needs the
 * reference's bits passes the table the reference's own numpy produced (telescope_amd/likelihood.py score_lut does).  No handle, no
 * device: host arithmetic only. */
int  tsem_score_lut(int32_t max_score, double scale_factor, double* lut /* max_score + 1 */);
int  tsem_dims(tsem_ctx* h, int64_t* n_rows, int32_t* n_cols, int64_t* nnz);
/* copy the CSR back to the host (tests of the generator) */
int  tsem_export_csr(tsem_ctx* h, int64_t* indptr, int32_t* indices, uint16_t* raw);

/* ---- model setup (model.py:679-699) --------------------------------------
 * tsem_rowstats: Y, weights w_i = max_j Q_ij, and the LOCAL sums
 *   stats[0]=sum w, stats[1]=sum w*Y, stats[2]=max w, pisum0[K] = sum of Q over
 *   unique rows per column; col_count[K] = stored entries per column and
 *   col_hash[K] = order-independent 64-bit signature sum_i hash(global row, score)
 *   (wrap-around).  Multi-GPU hosts all-reduce them (sum,sum,max,sum,sum,sum).
 *   Columns with equal (count, hash) are exact twins (same fragments, same
 *   scores): the reference keeps their pi/theta bit-identical because scipy
 *   accumulates every column in row order, and `reassign` ties depend on it, so
 *   the update step gives twins one shared accumulation.
 * tsem_set_model: global stats + priors; builds the column-partitioned EM
 *   layout and resets pi = theta = 1/K (model.py:667,673).  */
int  tsem_rowstats(tsem_ctx* h, double* stats3, double* pisum0,
                   uint64_t* col_count, uint64_t* col_hash);
/* Y[N] (model.py:679) and weights[N] = w_i (model.py:690) of the local rows as tsem_rowstats left them on the
 * device (either pointer may be NULL): the reference's `tl.Y` / `tl._weights`. */
int  tsem_export_rowinfo(tsem_ctx* h, uint8_t* Y, double* weights);
int  tsem_set_model(tsem_ctx* h, const double* stats3, const double* pisum0,
                    const uint64_t* col_count, const uint64_t* col_hash,
                    double pi_prior, double theta_prior);

/* ---- parameters ---------------------------------------------------------- */
int  tsem_set_params(tsem_ctx* h, const double* pi, const double* theta);
int  tsem_get_params(tsem_ctx* h, int which /*TSEM_Z_PREV|TSEM_Z_CUR|TSEM_Z_FIRST*/, double* pi, double* theta);

/* ---- EM iteration (estep model.py:702-722 + mstep 724-742, fused) ---------
 * tsem_em_pass: local fused E+M over this rank's rows with the current
 *   params; leaves red[0..K) = local thetasum, red[K] = 0, red[K+1] = 0 in the
 *   reduce buffer (device).  Asynchronous.
 * tsem_em_update: consumes the (all-reduced) reduce buffer: theta_hat, pi_hat
 *   (model.py:733-740), diff_est = sum|pi_hat - pi| (model.py:781); current
 *   params become "prev".  Synchronous when diff_est != NULL.
 * tsem_lnl_pass: local part of calculate_lnl(z(prev), cur) (model.py:744-760)
 *   into red[K]; tsem_read_reduce fetches the buffer.  */
int  tsem_reduce_buffer(tsem_ctx* h, void** dptr, int64_t* count);
int  tsem_bind_reduce_buffer(tsem_ctx* h, void* dptr, int64_t count); /* count >= K+2 doubles */
int  tsem_em_pass(tsem_ctx* h);
int  tsem_em_update(tsem_ctx* h, double* diff_est);
int  tsem_lnl_pass(tsem_ctx* h);
int  tsem_read_reduce(tsem_ctx* h, double* out, int64_t offset, int64_t count);
/* n fixed iterations (pass [+ all-reduce] + update) with no host round trip; the
 * per-iteration diff_est values are left in a device ring and copied out at the
 * end.  == em() with em_epsilon=0, max_iter=n. */
int  tsem_em_steps(tsem_ctx* h, int32_t n, double* diffs_out /* n or NULL */);
/* The loop body of em() (model.py:771-797) for up to n_max iterations, enqueued
 * back to back: fused E+M pass, column reduce, all-reduce (communicator
 * attached), parameter update; with use_likelihood also the lnl pass, its
 * all-reduce and the |lnl - lnl_prev| test.  Convergence (diff_est < epsilon,
 * or the lnl test) is decided ON THE DEVICE: the update kernel raises a stop
 * flag and every kernel enqueued behind it returns at once, so the host
 * synchronises once per chunk, not once per iteration, and the state after the
 * call is exactly the reference's after its last iteration.  flags bit 0
 * starts a run (lnl_prev = inf or tsem_set_prev_lnl, the next update saves
 * pi_init / theta_init).
 * *n_done = iterations committed, *stopped = 1 when the convergence test fired.
 * use_likelihood on a layout built for it (option "use_likelihood" = 1 before
 * tsem_set_model, or tsem_prepare_likelihood): no lnl pass per iteration — the
 * EM pass of iteration t+1 also sums lnl_t = sum z_t log1p(Q c_t) (model.py:
 * 783-789: its numerators ARE Q c_t; pi*theta of the previous parameters is a
 * third LDS table, the previous row sums are kept per row), and that
 * iteration's update kernel tests |lnl_t - lnl_(t-1)| < epsilon BEFORE it
 * commits: a run that converges in iteration t ends in the state of iteration
 * t.  The lnl of the chunk's LAST iteration is then unknown when the chunk
 * returns (lnls_out[n_done - 1] = NaN): it arrives as *lnl_carry of the next
 * chunk — which may do nothing else: n_done = 0, stopped = 1 — or, with flags
 * bit 1 (last chunk of the run), from the dedicated lnl pass before the call
 * returns.  *lnl_carry is NaN when nothing was owed.
 * A hand-off time-out of the fused kernel on ANY rank (the flag travels in the
 * all-reduce) leaves the parameters untouched on every rank; the failing rank
 * rebuilds its layout for the two-pass kernels and the iteration is redone. */
int  tsem_em_chunk(tsem_ctx* h, int32_t n_max, double epsilon, int32_t use_likelihood, int32_t flags,
                   int32_t* n_done, int32_t* stopped, double* diffs_out /* n_max or NULL */,
                   double* lnls_out /* n_max or NULL */, double* lnl_carry /* 1 or NULL */);
/* em(use_likelihood=True) on a model laid out without option "use_likelihood": rebuild the blocked layout so that the EM
 * pass can carry the log-likelihood (above).  A no-op when the layout has it or cannot have it (two-pass kernels, fp64
 * entries, option "reproducible", K > 8 x 5056): tsem_em_chunk then runs the lnl pass every iteration. */
int  tsem_prepare_likelihood(tsem_ctx* h);
/* Switch this handle to the two-pass kernels (rebuilds the blocked layout, keeps the parameters): what
 * tsem_em_chunk does after a time-out; public for hosts that drive pass / update themselves. */
int  tsem_fallback_twopass(tsem_ctx* h);
/* After tsem_em_update returned TSEM_ERR_TIMEOUT (every rank does, the flag is all-reduced): the rank whose
 * own fused pass raised the error word switches to the two-pass kernels (*switched = 1), the others do
 * nothing; then every rank redoes the pass. */
int  tsem_recover_timeout(tsem_ctx* h, int32_t* switched);
/* calculate_lnl(z(prev params), current params) (model.py:800-801), summed over the ranks of an attached
 * communicator; synchronous.  Redone on the two-pass kernels after a time-out of the fused pass. */
int  tsem_final_lnl(tsem_ctx* h, double* lnl);
/* The log-likelihood the FIRST iteration of the next run (tsem_em_chunk with first != 0) is compared with under
 * use_likelihood: the reference compares with self.lnl as the previous em() left it (model.py:786; inf on a fresh
 * model, model.py:683).  tsem_set_model resets it to inf, tsem_em_run leaves its final value. */
int  tsem_set_prev_lnl(tsem_ctx* h, double lnl);
/* Full EM loop (model.py:762-806) on top of tsem_em_chunk. */
int  tsem_em_run(tsem_ctx* h, double epsilon, int32_t max_iter, int32_t use_likelihood,
                 int32_t* n_iter, int32_t* converged, double* lnl,
                 double* diffs /* max_iter */, double* lnls /* max_iter or NULL */,
                 double* pi_init, double* theta_init /* K each or NULL */);

/* ---- communicator (row-sharded runs) ---------------------------
 * One RCCL communicator per process / GPU, created from an id that rank 0
 * generates and the host ships to the other ranks by any means (the Python
 * host uses its torch.distributed group).  Attached to a handle, it makes
 * tsem_em_chunk / tsem_em_steps / tsem_em_run all-reduce the reduce buffer
 * (sum, fp64, K+2) and the log-likelihood scalar on the handle's stream.
 * tsem_comm_allreduce_host: in-place sum of a host fp64 / uint64 vector over the
 * communicator (setup sums, reassign column sums).  */
#define TSEM_COMM_ID_BYTES 128
typedef struct tsem_comm tsem_comm;
int  tsem_comm_unique_id(void* id128);
int  tsem_comm_create(tsem_comm** out, int device, const void* id128, int rank, int world);
void tsem_comm_destroy(tsem_comm* c);
const char* tsem_comm_last_error(void);
int  tsem_comm_attach(tsem_ctx* h, tsem_comm* c);             /* c == NULL detaches */
int  tsem_comm_allreduce(tsem_ctx* h, int64_t offset, int64_t count);   /* reduce buffer [offset, offset+count), async */
int  tsem_comm_allreduce_host(tsem_comm* c, void* data, int64_t count, int dtype /*0 f64 sum, 1 u64 sum, 2 f64 max, 3 i64 max*/);
/* RCCL is resolved at run time, not linked: the librccl already mapped into the process (a torch process carries
 * one) or else librccl.so.1 from the loader path — ONE copy, chosen deliberately; without RCCL the library still
 * loads and runs single-GPU.  Writes "rccl <version> (<path>)" (or why it is unavailable) into buf. */
int  tsem_comm_library_info(char* buf, int32_t cap);
/* In-process transport: `world` handles on ONE device, one host thread each (tests of the row-sharded protocol on
 * a one-GPU box; hosts that share a GPU between engines).  Same tsem_em_chunk, same reduce buffer, error slot and
 * device-side stop flag; the all-reduce is a copy into per-rank slots, a host rendezvous of the ranks' threads at
 * ENQUEUE time, stream waits on the peers' events and a sum in rank order (bit-identical on every rank). */
typedef struct tsem_local_group tsem_local_group;
int  tsem_comm_local_group(tsem_local_group** out, int device, int world /* <= 8 */);
void tsem_comm_local_group_destroy(tsem_local_group* g);      /* after every communicator of the group */
int  tsem_comm_create_local(tsem_comm** out, tsem_local_group* g, int rank);

/* ---- results -------------------------------------------------------------- */
/* z aligned to the CSR pattern of the loaded scores (-1 where the reference
 * drops the entry from z's pattern, i.e. where Q_ij * pi_j[*theta_j] == 0).  model.py:795 / 837.  */
int  tsem_export_z(tsem_ctx* h, int which, double* z /* nnz */);
/* install (z != NULL) or drop (NULL) a caller-supplied z aligned to the CSR pattern, NaN where z has no
 * entry: `which` = TSEM_Z_USER then makes reassign / best_counts / reassign_groups read it as is, without
 * renormalising — what the reference does with whatever `self.z` holds (model.py:837) */
int  tsem_set_user_z(tsem_ctx* h, const double* z /* nnz or NULL */);
/* estep on explicit params (public estep(pi,theta), model.py:702-722) */
int  tsem_estep(tsem_ctx* h, const double* pi, const double* theta, double* z /* nnz */);
/* public mstep(z) (model.py:724-742) and calculate_lnl(z,pi,theta) (744-760) on a
 * caller-supplied z aligned to the CSR pattern (0 where z has no entry); with a communicator
 * attached both sum over the ranks' row shards before the closed forms / the return */
int  tsem_mstep(tsem_ctx* h, const double* z, double* pi_hat, double* theta_hat);
int  tsem_calc_lnl(tsem_ctx* h, const double* z, const double* pi, const double* theta, double* lnl);
/* rows' best-hit counts for `choose` (sparse_plus.py:117-129): nbest[i] */
int  tsem_best_counts(tsem_ctx* h, int which, int32_t* nbest /* n_rows */);
/* the same, compacted on the device: only the rows with SEVERAL best hits (the ones `choose` draws a random
 * number for, sparse_plus.py:147-149), in row order: rows[i] = local row index, counts[i] = its number of best hits.
 * *n = how many there are; if n > cap nothing is copied and TSEM_ERR_ARG is returned (call again with cap >= n). */
int  tsem_best_ties(tsem_ctx* h, int which, int64_t cap, int32_t* rows, int32_t* counts, int64_t* n);
/* reassign(method, thresh, initial).sum(0) (model.py:808-865,435-457) -> colsums[K]
 * (local rows); optional per-entry mask aligned to the CSR pattern (nnz doubles,
 * NULL to skip).  `picks[i]` (choose only) = ordinal of the chosen best hit. */
int  tsem_reassign(tsem_ctx* h, int method, double thresh, int which,
                   const int32_t* picks, double* colsums, double* mask);
/* The column sums `Telescope.output_report` takes from ONE z (model.py:432-457) in ONE pass over the rows:
 * out[0..K) = reassign('conf', thresh).sum(0), out[K..2K) = 'exclude', out[2K..3K) = 'average' (local rows).  The
 * rows with several best hits — the only rows `choose` treats differently from `exclude` (sparse_plus.py:140-154) —
 * are compacted on the device in row order: *n_ties of them; tsem_report_ties copies their indices and numbers of
 * best hits out (cap >= n_ties), and tsem_reassign_rows(TSEM_RA_CHOOSE, ..., rows = NULL, picks, n_ties, out) adds up
 * the picked entries of exactly those rows, so that  choose = exclude + that.  tsem_reassign_rows with a caller's
 * row list gives the contribution of those rows to any method (picks[i] belongs to rows[i]).
 * thresh < 0: the caller needs no `conf` column (out[0..K) is then not meaningful).  For the INITIAL z the pass then runs on
 * the 2-byte score codes alone — a row's best hits are its largest codes when the score table is strictly increasing and no
 * stored score is 0 — with no score-table look-up and no floating point (k_report_init_codes). */
int  tsem_report_colsums(tsem_ctx* h, int which, double thresh, double* out /* 3*K */, int64_t* n_ties);
int  tsem_report_ties(tsem_ctx* h, int64_t cap, int32_t* rows, int32_t* counts);
int  tsem_reassign_rows(tsem_ctx* h, int method, double thresh, int which, const int32_t* rows,
                        const int32_t* picks, int64_t n, double* colsums);
/* z[ridx, fidx] and reassign(method, thresh)[ridx, fidx] for the entries of a LIST of rows: what Telescope.update_sam reads per
 * alignment (model.py:483,508-511 — `prob = tl.z[ridx, fidx]`, `mat[ridx, fidx] > 0`), without the N x K matrices.  Row rows[i]'s
 * entries (CSR order) go to z_out / mask_out [out_off[i], out_off[i + 1]) (out_off[n] = total; the slices must have the rows'
 * lengths: TSEM_ERR_ARG otherwise).  z_out: -1 where the reference drops the entry from z's pattern.  picks[i] (choose) belongs to
 * list row i.  Either output may be NULL. */
int  tsem_rows_lookup(tsem_ctx* h, int which, int method, double thresh, int64_t n, const int32_t* rows, const int32_t* picks,
                      const int64_t* out_off, double* z_out, double* mask_out);
/* The random picks of `choose` (sparse_plus.py:140-154: np.random.choice per row with several best hits) in numpy's
 * LEGACY stream, on the host: out[i] = the draw np.random.randint(0, counts[i]) would return, taken in order from the
 * MT19937 state (key624, *pos) = np.random.get_state()[1:3]; the state is advanced exactly as numpy advances it, so
 * the caller's stream continues unchanged after np.random.set_state.  counts[i] >= 1 (1 consumes nothing). */
int  tsem_legacy_randint(uint32_t* key624, int32_t* pos, const int32_t* counts, int64_t n, int32_t* out);
/* the same assignment summed per GROUP of rows: scTelescope.output_report's per-barcode count matrix
 * `_assignments[_rows, :].sum(0)` for every barcode (model.py:611-625).  group_of_row[i] in
 * [0, n_groups) or -1 (row in no group); out is row-major [n_groups][K], caller-allocated.
 * tsem_set_groups copies the map to the device once (range-checked there) — the six methods of a report
 * then pass group_of_row = NULL; a non-NULL map is set first.  The device computes the groups in tiles
 * of at most 1 GB of device scratch (option "group_tile_bytes": the tile by column plus, for the streaming
 * kernel, the same tile by id), one pass over the matrix per tile: n_groups x K
 * doubles only have to fit the caller's `out`.  exclude / average / conf (conf_prob > 0.5) stream the
 * 4-byte id + score-code arrays like tsem_report_colsums; the other methods take the generic row pass. */
int  tsem_set_groups(tsem_ctx* h, const int32_t* group_of_row /* N, or NULL to drop */, int32_t n_groups);
int  tsem_reassign_groups(tsem_ctx* h, int method, double thresh, int which, const int32_t* picks,
                          const int32_t* group_of_row /* N, or NULL: the map of tsem_set_groups */, int32_t n_groups, double* out);

/* ---- csr_matrix_plus primitives on arbitrary fp64 CSR (sparse_plus.py) ---- */
int  tsem_csr_norm_rows(int device, int64_t n_rows, const int64_t* indptr,
                        const double* data, double* out);            /* norm(1)  :46-52  */
int  tsem_csr_binmax_rows(int device, int64_t n_rows, int32_t n_cols, const int64_t* indptr,
                          const double* data, int8_t* out);          /* binmax(1):117-129 */
/* mode 0: norm() :47-48   mode 1: scale() :94-95   mode 2: scale(1) :96-97 */
int  tsem_csr_scale(int device, int mode, int64_t n_rows, int32_t n_cols, const int64_t* indptr,
                    const double* data, double* out);

/* ---- instrumentation ------------------------------------------------------ */
/* HIP-event time (ms) and number of TIMED launches of the dominant EM kernel(s) since the
 * last call with reset=1 (option "kernel_timing" = n times every n-th pass, 0 none; default 1);
 * algorithmic bytes one EM pass reads.  */
int  tsem_kernel_stats(tsem_ctx* h, int reset, double* em_ms, int64_t* em_launches,
                       int64_t* algo_bytes_per_pass);
/* The last tsem_report_colsums of the handle: HIP-event time (ms) of its dominant kernel — the pass over the stored entries — when
 * option "kernel_timing" is not 0 (else 0), the algorithmic bytes that pass reads and writes (4 B per stored entry: popularity id +
 * score code; 8 B row pointer + 4 B output per row), the rows it left to the exact row kernel (too long, tied, near-tied, on the
 * threshold), and which kernel it was: 0 none yet, 1 generic row pass, 2 capacity kernel (k_report_rows), 3 score codes only
 * (k_report_init_codes), 4 packed fp32 filter (k_report_pack32).  model.py:432-457 is what the pass computes. */
int  tsem_report_stats(tsem_ctx* h, double* kernel_ms, int64_t* algo_bytes, int64_t* deferred_rows, int32_t* kernel);
/* Option "phase_timing" = 1: tsem_em_chunk records a HIP event at every phase boundary of every iteration it enqueues (a
 * diagnostic: ~5 events per iteration on the engine's stream).  ms6 = summed milliseconds over *n_iter iterations of
 * { EM pass | column reduce | all-reduce of the K+2 sums (0 without a communicator) | update | gap to the next iteration's
 * pass | first mark to last mark }.  What bench.py prints as `phase_us`. */
int  tsem_phase_times(tsem_ctx* h, int reset, double* ms6, int64_t* n_iter);
/* Free / total bytes of the device's memory (hipMemGetInfo; h may be NULL, then `device` is asked) and, with a handle, the bytes its
 * matrix keeps resident: resident5 = { CSR row pointers + scores | CSR column ids (0 after option "drop_csr_indices") | popularity
 * ids | blocked layout | per-row arrays }.  What a capacity plan needs: 14 B per stored entry by default (score codes), 10 after
 * the drop, + ~20 B per row. */
int  tsem_device_memory(tsem_ctx* h, int device, int64_t* free_bytes, int64_t* total_bytes, int64_t* resident5);
/* Facts about the resident layout, 32 values (the Python binding names them: _lib.Engine.layout_info).  Of the later ones: [23] the
 * EM pass carries the previous iteration's lnl; [24] split layout; [26] plain CSR row passes; [27] entries of the lnl pass's log Q
 * table (0: none), [28] log Q is arithmetic (no table); [29] stored entries that k_log_tab counted for the exact branch of the log
 * form before the LAST lnl pass (-1: no choice armed) and [30] the count above which the per-entry logarithm runs instead — reading
 * [29] synchronises the handle's stream; [31] rows whose row sum the report / row passes recomputed in the reference's own order of
 * additions (scipy's `sum(axis=1)` = np.add.reduceat: a0 + pairwise(a1 ..)) since the matrix was loaded — the rows where two z values,
 * or a z value and conf_prob, are closer than the rounding of 1 / rowsum could decide (sparse_plus.py:99-129 compares with `==`). */
int  tsem_layout_info(tsem_ctx* h, int64_t* info32);
/* per-block shader-clock stamps of team 0 / member 0 of the fused kernel (option "fused_prof") */
int  tsem_debug_fused_prof(tsem_ctx* h, uint64_t* out512);
/* the same option's start-up timeline: per workgroup b (up to 512), out[8 b ..] = 100 MHz wall clock at entry / tickets counted /
 * LDS zeroed / tables loaded / loop start / loop end / exit, and (team << 32 | member << 16 | blocks) */
int  tsem_debug_fused_startup(tsem_ctx* h, uint64_t* out4096);
/* the packed local row / local column words of one sub-block of the blocked layout (layout studies) */
int64_t tsem_debug_subblock(tsem_ctx* h, int64_t block, int32_t part, uint32_t* out, int64_t cap);
/* y[i] = the device log1p the lnl passes use (finite x >= 0), for accuracy tests against libm */
int  tsem_debug_log1p(int device, int32_t n, const double* x, double* y);
/* GB/s of a pure streaming read (16-byte non-temporal loads, 8 in flight per thread) over a scratch buffer of `bytes`,
 * best of `reps` launches: the measured-stream peak quoted beside the nominal HBM peak */
int  tsem_debug_stream_read(int device, int64_t bytes, int32_t reps, double* gbs);
/* the same for the table-driven log1p of the fused lnl pass (64-entry table in LDS, ~1e-16 absolute error per evaluation) */
int  tsem_debug_log1p_tab(int device, int32_t n, const double* x, double* y);
/* ... and of the log-table form of the lnl passes (round 5): log1p(q c) from lq = log q and lc = log c — L + exp(-L) for
 * L = lq + lc >= 18.715, 0 below -40, the table-driven log1p of the exact product q * c between */
int  tsem_debug_log1p_of_log(int device, int32_t n, const double* lq, const double* lc, const double* q, const double* c, double* y);

#ifdef __cplusplus
}
#endif
#endif /* TELESCOPE_EM_H */
