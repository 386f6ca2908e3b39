/* TEST INFRASTRUCTURE — a second, independent CPU restatement of Telescope's EM loop, in plain C.
 *
 * oracle/telescope_oracle.py issues the reference's own scipy operator sequence (and is the
 * `cpu_baseline` of bench.py because its cost structure is the reference's).  This file restates
 * the same arithmetic the way a CPU programmer would write it — one fused pass over the CSR rows per
 * iteration, OpenMP over rows, thread-private column accumulators — so that
 *   (1) the GPU engine is checked against two independently written oracles, and
 *   (2) bench.py can time a STRONG CPU baseline (all host cores) beside the reference-equivalent one.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product never
 * does (telescope_amd/ has no reference to oracle/).
 *
 * Follows /root/reference/telescope/utils/model.py:
 *   setup   635-700  Q = expm1((raw/max)*100) via the caller's table, Y_i = [row length > 1], w_i = max_j Q_ij,
 *                    W_tot, W_amb, w_max, pisum0 = column sums of Q over the unique rows
 *   estep   702-722  n_ij = Q_ij * pi_j * theta_j (Y_i = 1) | Q_ij * pi_j (Y_i = 0);  z = n * recip0(rowsum)
 *   mstep   724-742  thetasum_j = sum_i z_ij w_i Y_i;  theta = (thetasum + tp) / (W_amb + tp K);
 *                    pi = (pisum0 + thetasum + pp) / (W_tot + pp K),  tp = theta_prior * w_max, pp = pi_prior * w_max
 *   lnl     744-760  sum_ij z_ij log1p(inner_ij), inner built like n but with the NEW pi, theta
 *   loop    762-806  diff = sum |pi_new - pi|; stop at diff < eps (or, use_likelihood, |lnl - previous lnl| < eps) or max_iter;
 *                    lnl of (last z, final pi/theta)
 * Summation order differs from scipy's (row-parallel partial sums), so results agree to rounding, not bitwise.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline double recip0(double v) { double r = 1.0 / v; return isinf(r) ? 0.0 : r; }   /* sparse_plus.py:16-22 */

/* returns the number of iterations run, or -1 on allocation failure.
 * out: pi[K], theta[K], pi_init[K] (after the first iteration), *lnl, *converged, diffs[max_iter] (may be NULL) */
/* (oracle_em_fused3: lnls[max_iter] (may be NULL) = the log-likelihood of every iteration under use_likelihood, model.py:785) */
int oracle_em_fused3(int64_t N, int32_t K, const int64_t* indptr, const int32_t* indices, const uint16_t* raw,
                     const double* lut, double pi_prior, double theta_prior, double epsilon, int32_t max_iter,
                     int32_t use_likelihood, int32_t nthreads, double* pi, double* theta, double* pi_init, double* lnl,
                     int32_t* converged, double* diffs, double* lnls) {
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
  const int T = omp_get_max_threads();
#else
  const int T = 1;
#endif
  double* acc = (double*)calloc((size_t)T * K, sizeof(double));
  double* pisum0 = (double*)calloc((size_t)K, sizeof(double));
  double* c = (double*)malloc(sizeof(double) * K);
  double* pin = (double*)malloc(sizeof(double) * K);
  double* thn = (double*)malloc(sizeof(double) * K);
  double* w = (double*)malloc(sizeof(double) * (N > 0 ? N : 1));
  if (!acc || !pisum0 || !c || !pin || !thn || !w) return -1;
  double W_tot = 0.0, W_amb = 0.0, w_max = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : W_tot, W_amb) reduction(max : w_max)
  for (int64_t i = 0; i < N; ++i) {                                   /* model.py:679-699 */
    double m = 0.0;
    for (int64_t k = indptr[i]; k < indptr[i + 1]; ++k) { double q = lut[raw[k]]; if (q > m) m = q; }
    w[i] = m; W_tot += m; if (m > w_max) w_max = m;
    if (indptr[i + 1] - indptr[i] > 1) W_amb += m;
    else for (int64_t k = indptr[i]; k < indptr[i + 1]; ++k) {
      const double q = lut[raw[k]];
#pragma omp atomic
      pisum0[indices[k]] += q;
    }
  }
  const double tp = theta_prior * w_max, pp = pi_prior * w_max;
  for (int j = 0; j < K; ++j) pi[j] = theta[j] = 1.0 / K;             /* model.py:667,673 */
  int it = 0, conv = 0;
  double total_lnl = INFINITY;
  while (!conv && it < max_iter) {
    for (int j = 0; j < K; ++j) c[j] = pi[j] * theta[j];
    memset(acc, 0, sizeof(double) * (size_t)T * K);
#pragma omp parallel
    {
#ifdef _OPENMP
      double* a = acc + (size_t)omp_get_thread_num() * K;
#else
      double* a = acc;
#endif
#pragma omp for schedule(static)
      for (int64_t i = 0; i < N; ++i) {                               /* E-step row by row, M-step scatter fused */
        const int64_t s = indptr[i], e = indptr[i + 1];
        if (e - s < 2) continue;                                      /* unique rows only feed pisum0 */
        double sum = 0.0;
        for (int64_t k = s; k < e; ++k) sum += lut[raw[k]] * c[indices[k]];
        const double r = recip0(sum) * w[i];
        for (int64_t k = s; k < e; ++k) a[indices[k]] += (lut[raw[k]] * c[indices[k]]) * r;
      }
    }
    double diff = 0.0;
    for (int j = 0; j < K; ++j) {                                     /* model.py:731-740 */
      double ts = 0.0;
      for (int t = 0; t < T; ++t) ts += acc[(size_t)t * K + j];
      thn[j] = (ts + tp) / (W_amb + tp * K);
      pin[j] = (pisum0[j] + ts + pp) / (W_tot + pp * K);
      diff += fabs(pin[j] - pi[j]);
    }
    ++it;
    if (diffs) diffs[it - 1] = diff;
    conv = diff < epsilon;                                            /* model.py:781,792 */
    if (use_likelihood || conv || it >= max_iter) {
      /* lnl: z from the parameters BEFORE this M-step, inner from the ones after it — of every iteration under
       * use_likelihood (model.py:783-789), else of the last one (model.py:795-801) */
      double l = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : l)
      for (int64_t i = 0; i < N; ++i) {
        const int64_t s = indptr[i], e = indptr[i + 1];
        const int amb = (e - s) > 1;
        double sum = 0.0;
        for (int64_t k = s; k < e; ++k) sum += lut[raw[k]] * (amb ? c[indices[k]] : pi[indices[k]]);
        const double r = recip0(sum);
        for (int64_t k = s; k < e; ++k) {
          const double q = lut[raw[k]];
          const double z = (q * (amb ? c[indices[k]] : pi[indices[k]])) * r;
          const double inner = q * (amb ? pin[indices[k]] * thn[indices[k]] : pin[indices[k]]);
          if (z != 0.0) l += z * log1p(inner);
        }
      }
      if (use_likelihood) conv = fabs(l - total_lnl) < epsilon;       /* model.py:786-788; self.lnl starts at inf (model.py:676) */
      total_lnl = l;
      if (lnls) lnls[it - 1] = l;
    }
    memcpy(pi, pin, sizeof(double) * K);
    memcpy(theta, thn, sizeof(double) * K);
    if (it == 1 && pi_init) memcpy(pi_init, pi, sizeof(double) * K);  /* model.py:776-778 */
  }
  if (lnl) *lnl = total_lnl;
  if (converged) *converged = conv;
  free(acc); free(pisum0); free(c); free(pin); free(thn); free(w);
  return it;
}

int oracle_em_fused2(int64_t N, int32_t K, const int64_t* indptr, const int32_t* indices, const uint16_t* raw,
                     const double* lut, double pi_prior, double theta_prior, double epsilon, int32_t max_iter,
                     int32_t use_likelihood, int32_t nthreads, double* pi, double* theta, double* pi_init, double* lnl,
                     int32_t* converged, double* diffs) {
  return oracle_em_fused3(N, K, indptr, indices, raw, lut, pi_prior, theta_prior, epsilon, max_iter, use_likelihood, nthreads, pi,
                          theta, pi_init, lnl, converged, diffs, (double*)0);
}

int oracle_em_fused(int64_t N, int32_t K, const int64_t* indptr, const int32_t* indices, const uint16_t* raw,
                    const double* lut, double pi_prior, double theta_prior, double epsilon, int32_t max_iter,
                    int32_t nthreads, double* pi, double* theta, double* pi_init, double* lnl, int32_t* converged,
                    double* diffs) {
  return oracle_em_fused2(N, K, indptr, indices, raw, lut, pi_prior, theta_prior, epsilon, max_iter, 0, nthreads, pi, theta,
                          pi_init, lnl, converged, diffs);
}

/* reassign('exclude').sum(0) (model.py:837-842, sparse_plus.py:99-129) with z = estep(pi, theta) computed row by
 * row: counts[j] = number of rows whose UNIQUE largest posterior sits in column j (rows whose maximum is shared
 * by several entries are dropped).  Entries whose numerator is exactly 0 are not in z's pattern (model.py:720).
 * Returns 0. */
int oracle_exclude_counts(int64_t N, int32_t K, const int64_t* indptr, const int32_t* indices, const uint16_t* raw,
                          const double* lut, const double* pi, const double* theta, int64_t* counts) {
  (void)K;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < N; ++i) {
    const int64_t s = indptr[i], e = indptr[i + 1];
    const int amb = (e - s) > 1;
    double sum = 0.0;
    for (int64_t k = s; k < e; ++k) sum += lut[raw[k]] * (amb ? pi[indices[k]] * theta[indices[k]] : pi[indices[k]]);
    const double r = recip0(sum);
    double zmax = -1.0; int nbest = 0; int64_t kbest = -1;
    for (int64_t k = s; k < e; ++k) {
      const double n = lut[raw[k]] * (amb ? pi[indices[k]] * theta[indices[k]] : pi[indices[k]]);
      if (n == 0.0) continue;
      const double z = n * r;
      if (z > zmax) { zmax = z; nbest = 1; kbest = k; } else if (z == zmax) ++nbest;
    }
    if (nbest == 1) {
#pragma omp atomic
      counts[indices[kbest]] += 1;
    }
  }
  return 0;
}

/* The column sums Telescope.output_report takes from ONE z (model.py:432-457): reassign('conf', thresh), ('exclude') and
 * ('average') summed over the rows, with z = estep(pi, theta) (entries whose numerator is exactly 0 are not in z's pattern,
 * model.py:720) or, initial != 0, z = Q.norm(1) (model.py:837: the stored pattern, zeros included).
 *   conf     model.py:851-856  v = z where z >= thresh else 0;  v.norm(1) = v * recip0(rowsum v)
 *   exclude  model.py:839-842  binmax(1) (sparse_plus.py:117-129: data == row maximum, the maximum taken over the FULL row of
 *                              K columns, i.e. at least 0 when the pattern does not fill the row), rows with exactly one best hit
 *   average  model.py:847-850  binmax(1).norm(1): 1 / (number of best hits) per best hit
 * Thread-private accumulators added in thread order: the same bits for the same thread count.  Returns 0, -1 without memory. */
int oracle_report_sums(int64_t N, int32_t K, const int64_t* indptr, const int32_t* indices, const uint16_t* raw,
                       const double* lut, const double* pi, const double* theta, int32_t initial, double thresh,
                       int32_t nthreads, double* conf, int64_t* exclude, double* average) {
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
  const int T = omp_get_max_threads();
#else
  const int T = 1;
#endif
  double* ac = (double*)calloc((size_t)T * K, sizeof(double));
  double* aa = (double*)calloc((size_t)T * K, sizeof(double));
  int64_t* ae = (int64_t*)calloc((size_t)T * K, sizeof(int64_t));
  if (!ac || !aa || !ae) { free(ac); free(aa); free(ae); return -1; }
#pragma omp parallel
  {
#ifdef _OPENMP
    const size_t t0 = (size_t)omp_get_thread_num() * K;
#else
    const size_t t0 = 0;
#endif
    double *c_ = ac + t0, *a_ = aa + t0;
    int64_t* e_ = ae + t0;
#pragma omp for schedule(static)
    for (int64_t i = 0; i < N; ++i) {
      const int64_t s = indptr[i], e = indptr[i + 1];
      const int amb = (e - s) > 1;
      double sum = 0.0;
      for (int64_t k = s; k < e; ++k) {
        const int32_t j = indices[k];
        sum += lut[raw[k]] * (initial ? 1.0 : (amb ? pi[j] * theta[j] : pi[j]));
      }
      const double r = recip0(sum);
      /* pass 1: row maximum over the pattern (and the implicit zeros of a row shorter than K), confident mass */
      int64_t npat = 0;
      double zmax = -INFINITY, csum = 0.0;
      for (int64_t k = s; k < e; ++k) {
        const int32_t j = indices[k];
        const double n = lut[raw[k]] * (initial ? 1.0 : (amb ? pi[j] * theta[j] : pi[j]));
        if (!initial && n == 0.0) continue;
        const double z = n * r;
        ++npat;
        if (z > zmax) zmax = z;
        if (z >= thresh) csum += z;
      }
      if (npat < K && zmax < 0.0) zmax = 0.0;
      if (npat == 0) continue;
      const double cr = recip0(csum);
      int nbest = 0;
      for (int64_t k = s; k < e; ++k) {
        const int32_t j = indices[k];
        const double n = lut[raw[k]] * (initial ? 1.0 : (amb ? pi[j] * theta[j] : pi[j]));
        if (!initial && n == 0.0) continue;
        if (n * r == zmax) ++nbest;
      }
      for (int64_t k = s; k < e; ++k) {
        const int32_t j = indices[k];
        const double n = lut[raw[k]] * (initial ? 1.0 : (amb ? pi[j] * theta[j] : pi[j]));
        if (!initial && n == 0.0) continue;
        const double z = n * r;
        if (z >= thresh) c_[j] += z * cr;
        if (z == zmax) { a_[j] += 1.0 / (double)nbest; if (nbest == 1) e_[j] += 1; }
      }
    }
  }
  for (int j = 0; j < K; ++j) {
    double c = 0.0, a = 0.0; int64_t x = 0;
    for (int t = 0; t < T; ++t) { c += ac[(size_t)t * K + j]; a += aa[(size_t)t * K + j]; x += ae[(size_t)t * K + j]; }
    if (conf) conf[j] = c;
    if (average) average[j] = a;
    if (exclude) exclude[j] = x;
  }
  free(ac); free(aa); free(ae);
  return 0;
}
