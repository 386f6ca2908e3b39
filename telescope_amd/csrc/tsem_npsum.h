// The order of additions of the reference's ROW SUM, for the rows whose integer outputs hang on its last bit.
//
// sparse_plus.py:51 normalises z by `self.sum(1)`; scipy's CSR `sum(axis=1)` is `np.add.reduceat(data, indptr[rows])`
// (scipy/sparse/_compressed.py `_minor_reduce`), and numpy's reduceat hands every segment a_0 .. a_{m-1} to the add loop
// as a binary reduce: out = a_0 + pairwise(a_1 .. a_{m-1}), with numpy's published pairwise scheme (numpy/_core/src/umath/
// loops_utils.h.src, `@TYPE@_pairwise_sum`):
//     n < 8     res = 0; res += a_i, left to right
//     n <= 128  eight accumulators r_j = a_j; r_j += a_{i+j} for i = 8, 16, .. while i < n - n % 8;
//               res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)); the last n % 8 terms are added left to right
//     n > 128   n2 = n / 2 rounded down to a multiple of 8;  pairwise(a, n2) + pairwise(a + n2, n - n2)
// The sequence is what scipy holds when it sums: for the model's z the NON-ZERO products of a row in CSR order (`_amb + _uni`,
// model.py:713-714, drops exact zeros), for the initial z every stored entry.
//
// Plain C++ with no HIP dependency (tests/test_host_logic.py compiles it with g++ and checks it against np.add.reduceat);
// on the device one lane walks a flagged row with it — a handful of rows per matrix, see tsem_report.hip.
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define TS_NPSUM_FN __host__ __device__ inline
#define TS_NPSUM_NOUNROLL _Pragma("nounroll")      // (one lane walks a rare row: keep the code and its registers small)
#else
#define TS_NPSUM_FN inline
#define TS_NPSUM_NOUNROLL
#endif

// `next()` yields the sequence's terms in order; exactly `n` of them are taken
template <class Next>
TS_NPSUM_FN double np_pairwise_leaf(Next& next, int64_t n) {
  if (n < 8) {
    double res = 0.0;
    TS_NPSUM_NOUNROLL
    for (int64_t i = 0; i < n; ++i) res += next();
    return res;
  }
  double r[8];
  for (int j = 0; j < 8; ++j) r[j] = next();
  int64_t i = 8;
  TS_NPSUM_NOUNROLL
  for (; i < n - (n % 8); i += 8)
    for (int j = 0; j < 8; ++j) r[j] += next();
  double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  TS_NPSUM_NOUNROLL
  for (; i < n; ++i) res += next();
  return res;
}

template <class Next>
TS_NPSUM_FN double np_pairwise(Next& next, int64_t n) {
  if (n <= 128) return np_pairwise_leaf(next, n);
  // the recursion of the last rule, unrolled onto an explicit stack (the terms are consumed left to right whatever the shape
  // of the tree): at most log2(2^63 / 128) levels
  // (volatile: the arrays must stay in memory — promoted to registers they cost the calling kernel 240 VGPRs for a path one lane
  //  takes on a handful of rows)
  volatile int64_t len[60];
  volatile double left[60];
  volatile signed char stage[60];
  int sp = 0;
  len[0] = n; stage[0] = 0; left[0] = 0.0;
  double ret = 0.0;
  TS_NPSUM_NOUNROLL
  while (sp >= 0) {
    const int64_t m = len[sp];
    if (m <= 128) { ret = np_pairwise_leaf(next, m); --sp; continue; }
    int64_t n2 = m / 2;
    n2 -= n2 % 8;
    if (stage[sp] == 0) { stage[sp] = 1; ++sp; len[sp] = n2; stage[sp] = 0; continue; }
    if (stage[sp] == 1) { left[sp] = ret; stage[sp] = 2; ++sp; len[sp] = m - n2; stage[sp] = 0; continue; }
    ret = left[sp] + ret;
    --sp;
  }
  return ret;
}

// np.add.reduceat over one segment of m terms (m >= 1)
template <class Next>
TS_NPSUM_FN double np_reduceat_sum(Next& next, int64_t m) {
  const double a0 = next();
  if (m <= 1) return a0;
  return a0 + np_pairwise(next, m - 1);
}
