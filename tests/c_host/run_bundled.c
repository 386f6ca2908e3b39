/* A C99 host of libtelescope_em.so with no Python in the process: the bundled `telescope test` matrix (1000 fragments x 59 loci)
 * from a flat binary (tools/make_c_host_fixture.py dumps tests/golden/bundled_raw_scores.npz), through
 *     tsem_create -> tsem_load_scores -> tsem_max_score -> tsem_set_lut -> tsem_rowstats -> tsem_set_model -> tsem_em_run
 *     -> tsem_report_colsums,
 * i.e. TelescopeLikelihood.__init__ + em() + reassign('exclude').sum(0) of the reference (model.py:635-700, 762-806, 808-865).
 * Prints the iteration count, the final log-likelihood and the per-locus `exclude` counts; tests/test_gpu_round6.py compiles it
 * with gcc, runs it and compares with tests/golden/case_bundled.npz (16 iterations, lnl 95252.596293).
 *
 *   usage: run_bundled <bundled_flat.bin> [libm]
 *     libm: take the score table from tsem_score_lut (the C library's expm1) instead of the numpy table in the file
 *
 * What this host had to know that the header does not do for it — both documented in telescope_em.h:
 *   - the score table Q = expm1(score / max * 100): the reference's bits come from numpy's expm1, so the fixture carries that
 *     table; tsem_score_lut gives libm's (1 ulp apart in ~10 % of the entries, results agree to ~1e-15);
 *   - nothing else: a single-GPU host hands tsem_rowstats's outputs straight to tsem_set_model (a multi-GPU host all-reduces
 *     them in between). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "telescope_em.h"

#define CK(call) do { int rc_ = (call); if (rc_ != TSEM_OK) { \
  fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, tsem_last_error(h)); return 2; } } while (0)

static void* xread(FILE* f, size_t bytes) {
  void* p = malloc(bytes ? bytes : 1);
  if (!p || fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "short read\n"); exit(3); }
  return p;
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s bundled_flat.bin [libm]\n", argv[0]); return 1; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 1; }
  int64_t hdr[4];                                            /* rows, columns, stored entries, table length */
  if (fread(hdr, 8, 4, f) != 4) { fprintf(stderr, "bad header\n"); return 3; }
  const int64_t n = hdr[0], nnz = hdr[2];
  const int32_t k = (int32_t)hdr[1], lut_len = (int32_t)hdr[3];
  int64_t* indptr = (int64_t*)xread(f, 8 * (size_t)(n + 1));
  int32_t* indices = (int32_t*)xread(f, 4 * (size_t)nnz);
  uint16_t* raw = (uint16_t*)xread(f, 2 * (size_t)nnz);
  if ((2 * nnz) % 8) free(xread(f, 8 - (size_t)((2 * nnz) % 8)));   /* the table starts on an 8-byte boundary */
  double* lut = (double*)xread(f, 8 * (size_t)lut_len);
  fclose(f);

  tsem_ctx* h = NULL;
  if (tsem_create(&h, 0) != TSEM_OK) { fprintf(stderr, "tsem_create: %s\n", tsem_last_error(NULL)); return 2; }
  CK(tsem_load_scores(h, n, k, indptr, indices, raw, NULL, 0));
  int32_t max_score = 0;
  CK(tsem_max_score(h, &max_score));
  if (max_score + 1 != lut_len) { fprintf(stderr, "fixture: table of %d entries for max score %d\n", lut_len, max_score); return 3; }
  if (argc > 2 && strcmp(argv[2], "libm") == 0) CK(tsem_score_lut(max_score, 100.0, lut));
  CK(tsem_set_lut(h, lut, lut_len));

  double stats[3];
  double* pisum0 = (double*)malloc(8 * (size_t)k);
  uint64_t* cnt = (uint64_t*)malloc(8 * (size_t)k);
  uint64_t* hsh = (uint64_t*)malloc(8 * (size_t)k);
  CK(tsem_rowstats(h, stats, pisum0, cnt, hsh));
  CK(tsem_set_model(h, stats, pisum0, cnt, hsh, 0.0, 200000.0));   /* telescope's defaults: pi_prior 0, theta_prior 200000 */

  enum { MAX_ITER = 100 };
  int32_t n_iter = 0, converged = 0;
  double lnl = 0.0, diffs[MAX_ITER];
  CK(tsem_em_run(h, 1e-7, MAX_ITER, 0, &n_iter, &converged, &lnl, diffs, NULL, NULL, NULL));

  double* out = (double*)malloc(8 * 3 * (size_t)k);
  int64_t n_ties = 0;
  CK(tsem_report_colsums(h, TSEM_Z_PREV, 0.9, out, &n_ties));     /* the final z is the E-step before the last M-step (model.py:795) */
  int64_t info[32];
  CK(tsem_layout_info(h, info));

  printf("iterations %d converged %d\n", n_iter, converged);
  printf("lnl %.17g\n", lnl);
  printf("last_diff %.17g\n", diffs[n_iter - 1]);
  printf("ties %lld near_tie_rows %lld\n", (long long)n_ties, (long long)info[31]);
  printf("exclude");
  for (int32_t j = 0; j < k; ++j) printf(" %lld", (long long)(out[k + j] + 0.5));
  printf("\n");
  tsem_destroy(h);
  free(out); free(hsh); free(cnt); free(pisum0); free(lut); free(raw); free(indices); free(indptr);
  return 0;
}
