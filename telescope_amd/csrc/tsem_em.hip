// libtelescope_em.so, EM unit: one EM iteration = fused pass (tsem_fused.h; or the two-pass kernels) -> k_colreduce ->
// [all-reduce] -> k_update (model.py:718-742, 781), the log-likelihood passes (model.py:744-760), option "reproducible",
// the fp32 diagnostic pass, and the chunked device-side loop of em() (model.py:762-806).
#include "tsem_internal.h"

// ============================================================================
// EM hot loop — two-pass form (phase 1: partial row sums; phase 2: scatter)
// ============================================================================
// WG = (part p, stripe g).  LDS: ctab[Kp] | y[R]
template <int NT>
__global__ __launch_bounds__(NT) void k_phase1(int P, int Kp, int R, int64_t b0, int64_t nb, int G, int64_t N_amb_pad,
    const int64_t* __restrict__ sb_off, const double* __restrict__ pval, const uint32_t* __restrict__ prc,
    const double* __restrict__ ctab, double* __restrict__ ypart, const uint32_t* __restrict__ ctl) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (ctl && __hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;   // the run has stopped (tsem_em_chunk)
  double* c = reinterpret_cast<double*>(smem);
  double* y = c + Kp;
  const int p = blockIdx.x % P, g = blockIdx.x / P;
  for (int t = threadIdx.x; t < Kp; t += NT) c[t] = ctab[p * Kp + t];
  for (int t = threadIdx.x; t < R; t += NT) y[t] = 0.0;
  __syncthreads();
  for (int64_t b = b0 + g; b < nb; b += G) {
    const int64_t q0 = sb_off[b * P + p] >> 2, q1 = sb_off[b * P + p + 1] >> 2;
    for (int64_t q = q0 + threadIdx.x; q < q1; q += NT) {
      uint4 rc = reinterpret_cast<const uint4*>(prc)[q];
      double2 v0 = reinterpret_cast<const double2*>(pval)[2 * q];
      double2 v1 = reinterpret_cast<const double2*>(pval)[2 * q + 1];
      lds_add(&y[rc.x >> 16], v0.x * c[rc.x & 0xFFFF]);
      lds_add(&y[rc.y >> 16], v0.y * c[rc.y & 0xFFFF]);
      lds_add(&y[rc.z >> 16], v1.x * c[rc.z & 0xFFFF]);
      lds_add(&y[rc.w >> 16], v1.y * c[rc.w & 0xFFFF]);
    }
    __syncthreads();
    double* out = ypart + (int64_t)p * N_amb_pad + b * R;
    for (int t = threadIdx.x; t < R; t += NT) { out[t] = y[t]; y[t] = 0.0; }
    __syncthreads();
  }
}

// LDS: ctab[Kp] | acc[Kp] | s[R].  thetasum partials -> partial[g][p*Kp + l]
template <int NT>
__global__ __launch_bounds__(NT) void k_phase2_em(int P, int Kp, int R, int64_t b0, int64_t nb, int G, int accumulate, int64_t N_amb_pad,
    const int64_t* __restrict__ sb_off, const double* __restrict__ pval, const uint32_t* __restrict__ prc,
    const double* __restrict__ ctab, const double* __restrict__ ypart, const uint16_t* __restrict__ wcode,
    const double* __restrict__ lut, double* __restrict__ partial, const uint32_t* __restrict__ ctl) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (ctl && __hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
  double* c = reinterpret_cast<double*>(smem);
  double* acc = c + Kp;
  double* s = acc + Kp;
  const int p = blockIdx.x % P, g = blockIdx.x / P;
  {
    const double* prev = partial + (int64_t)g * (P * Kp) + p * Kp;
    for (int t = threadIdx.x; t < Kp; t += NT) { c[t] = ctab[p * Kp + t]; acc[t] = accumulate ? prev[t] : 0.0; }
  }
  for (int64_t b = b0 + g; b < nb; b += G) {
    __syncthreads();
    for (int t = threadIdx.x; t < R; t += NT) {
      int64_t a = b * R + t;
      double ys = 0.0;
      for (int pp = 0; pp < P; ++pp) ys += ypart[(int64_t)pp * N_amb_pad + a];
      // z = n * recip0(rowsum) (sparse_plus.py:52), weighted by w_i (model.py:730)
      s[t] = recip0(ys) * lut[wcode[a]];
    }
    __syncthreads();
    const int64_t q0 = sb_off[b * P + p] >> 2, q1 = sb_off[b * P + p + 1] >> 2;
    for (int64_t q = q0 + threadIdx.x; q < q1; q += NT) {
      uint4 rc = reinterpret_cast<const uint4*>(prc)[q];
      double2 v0 = reinterpret_cast<const double2*>(pval)[2 * q];
      double2 v1 = reinterpret_cast<const double2*>(pval)[2 * q + 1];
      lds_add(&acc[rc.x & 0xFFFF], (v0.x * c[rc.x & 0xFFFF]) * s[rc.x >> 16]);
      lds_add(&acc[rc.y & 0xFFFF], (v0.y * c[rc.y & 0xFFFF]) * s[rc.y >> 16]);
      lds_add(&acc[rc.z & 0xFFFF], (v1.x * c[rc.z & 0xFFFF]) * s[rc.z >> 16]);
      lds_add(&acc[rc.w & 0xFFFF], (v1.y * c[rc.w & 0xFFFF]) * s[rc.w >> 16]);
    }
  }
  __syncthreads();
  double* out = partial + (int64_t)g * (P * Kp) + p * Kp;
  for (int t = threadIdx.x; t < Kp; t += NT) out[t] = acc[t];
}

// lnl over ambiguous rows: sum z(prev) * log1p(Q * c_cur)   (model.py:755-758)
// LDS: c_old[Kp] | c_new[Kp] | r[R]
template <int NT>
__global__ __launch_bounds__(NT) void k_phase2_lnl(int P, int Kp, int R, int64_t nb, int G, int64_t N_amb_pad,
    const int64_t* __restrict__ sb_off, const double* __restrict__ pval, const uint32_t* __restrict__ prc,
    const double* __restrict__ ctab_old, const double* __restrict__ ctab_new, const double* __restrict__ ypart,
    double* __restrict__ lnl_part) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double scratch[16];
  double* co = reinterpret_cast<double*>(smem);
  double* cn = co + Kp;
  double* r = cn + Kp;
  const int p = blockIdx.x % P, g = blockIdx.x / P;
  for (int t = threadIdx.x; t < Kp; t += NT) { co[t] = ctab_old[p * Kp + t]; cn[t] = ctab_new[p * Kp + t]; }
  double acc = 0.0;
  for (int64_t b = g; b < nb; b += G) {
    __syncthreads();
    for (int t = threadIdx.x; t < R; t += NT) {
      int64_t a = b * R + t;
      double ys = 0.0;
      for (int pp = 0; pp < P; ++pp) ys += ypart[(int64_t)pp * N_amb_pad + a];
      r[t] = recip0(ys);
    }
    __syncthreads();
    const int64_t e0 = sb_off[b * P + p], e1 = sb_off[b * P + p + 1];
    for (int64_t e = e0 + threadIdx.x; e < e1; e += NT) {
      uint32_t rc = prc[e];
      double v = pval[e];
      double z = (v * co[rc & 0xFFFF]) * r[rc >> 16];
      if (z != 0.0) acc += z * ts_log1p_pos(v * cn[rc & 0xFFFF]);
    }
  }
  double t = block_sum(acc, scratch);
  if (threadIdx.x == 0) lnl_part[blockIdx.x] = t;
}

// lnl over unique rows: z = n * recip0(n), n = Q*pi_prev; inner = Q*pi_cur
__global__ __launch_bounds__(256) void k_lnl_unique(int64_t N_uni, const int32_t* __restrict__ ucol,
    const uint16_t* __restrict__ ucode, const double* __restrict__ lut, const double* __restrict__ pi_old,
    const double* __restrict__ pi_new, double* __restrict__ lnl_part) {
  __shared__ double scratch[16];
  double acc = 0.0;
  for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < N_uni; u += (int64_t)gridDim.x * blockDim.x) {
    int col = ucol[u];
    double q = lut[ucode[u]];
    double n = q * pi_old[col];
    if (n != 0.0) {
      double z = n * recip0(n);
      if (z != 0.0) acc += z * ts_log1p_pos(q * pi_new[col]);
    }
  }
  double t = block_sum(acc, scratch);
  if (threadIdx.x == 0) lnl_part[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void k_sum_parts(const double* __restrict__ a, int na, const double* __restrict__ b,
                                                   int nbb, double* __restrict__ out) {
  __shared__ double scratch[16];
  double acc = 0.0;
  for (int t = threadIdx.x; t < na; t += blockDim.x) acc += a[t];
  for (int t = threadIdx.x; t < nbb; t += blockDim.x) acc += b[t];
  double t = block_sum(acc, scratch);
  if (threadIdx.x == 0) out[0] = t;
}

// red[col] = sum_g partial[g][pc]   (fixed order -> deterministic given partials).  A block handles 64
// slots x 4 interleaved slices of the team axis, so the strided reads of one slot overlap instead of
// forming a chain of G dependent loads.  `sync` (fused kernel): only the teams that formed wrote
// their slice — G = sum over XCDs of floor(tickets / P).
// red[K] = 1 when this rank's fused pass raised its watchdog error word (the flag is summed with the column
// sums by the all-reduce, so every rank's update kernel sees that SOME rank failed), red[K+1] = 0.
__global__ __launch_bounds__(256) void k_colreduce(int Kpad, int G, const double* __restrict__ partial,
                            const int32_t* __restrict__ col_of_pc, const uint32_t* __restrict__ colmap,
                            double* __restrict__ red, int K, const uint32_t* __restrict__ sync, int P,
                            const uint32_t* __restrict__ ctl, unsigned long long* __restrict__ fz_xchg, int64_t fz_xchg_n,
                            const double* __restrict__ lnl_part, const double* __restrict__ lnl_uni, int lnl_nu,
                            const double* __restrict__ cmul = nullptr /* split layout: the partials are sums of Q s, times pi*theta of the slot */,
                            const uint32_t* __restrict__ errlog = nullptr /* split layout: the error word k_keep_err saved of the row-sum pass */) {
  // the fused kernel's exchange ring must be zero at its next launch: cleared here, by the ~950 blocks of the
  // kernel that follows every fused pass (no launch of its own, no fence: the next fused launch is a kernel boundary away)
  if (fz_xchg) {
    typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
    u64x2_t* const x2 = reinterpret_cast<u64x2_t*>(fz_xchg);                  // (hipMalloc alignment; the ring holds an even number of words)
    const u64x2_t zz = {0ull, 0ull};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < fz_xchg_n / 2; i += (int64_t)gridDim.x * blockDim.x)
      x2[i] = zz;
  }
  // 32 slots x 8 interleaved slices of the team axis per block: 8 independent loads per thread in flight (the
  // kernel is latency-bound: 64 teams x 30k slots = 15 MB; 13.6 us with 4 slices of 16 loads, r01 profile)
  constexpr int NS = 8, NC = 32;
  __shared__ double part[NS][NC];
  if (ctl && __hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;   // the run has stopped
  if (sync) {
    int t = 0;
    for (int x = 0; x < 8; ++x) t += (int)(sync[x] / (uint32_t)P);
    G = min(G, t);
  }
  const int pcl = threadIdx.x % NC, slice = threadIdx.x / NC;
  const int pc = blockIdx.x * NC + pcl;
  if (blockIdx.x == 0 && threadIdx.x == 0) { red[K] = ((sync && sync[9]) || (errlog && errlog[0])) ? 1.0 : 0.0; if (!lnl_part) red[K + 1] = 0.0; }
  if (lnl_part && blockIdx.x == 1 % gridDim.x) {
    // fused kernel MODE 4: the log-likelihood of the PREVIOUS iteration rides in slot K+1 (summed over the ranks with the column
    // sums): one partial per workgroup of the teams that formed + the unique rows' partials, fixed order
    __shared__ double lsc[16];
    double a = 0.0;
    for (int t = threadIdx.x; t < G * P; t += blockDim.x) a += lnl_part[t];
    for (int t = threadIdx.x; t < lnl_nu; t += blockDim.x) a += lnl_uni[t];
    const double tot = block_sum(a, lsc);
    if (threadIdx.x == 0) red[K + 1] = tot;
  }
  const int col = pc < Kpad ? col_of_pc[pc] : -1;        // -1: padding, or a secondary slot of a split column
  double s = 0.0;
  if (col >= 0) {
    const int copies = 1 << ((colmap[col] >> CM_LS) & 7u);
    if (copies == 1) {
#pragma unroll 8
      for (int g = slice; g < G; g += NS) s += partial[(int64_t)g * Kpad + pc];
    } else {
      for (int g = slice; g < G; g += NS)
        for (int c = 0; c < copies; ++c) s += partial[(int64_t)g * Kpad + pc + c];
    }
  }
  part[slice][pcl] = s;
  __syncthreads();
  if (slice == 0 && col >= 0) {                          // fixed order -> deterministic given the partials
    const double t = ((part[0][pcl] + part[1][pcl]) + (part[2][pcl] + part[3][pcl])) + ((part[4][pcl] + part[5][pcl]) + (part[6][pcl] + part[7][pcl]));
    // split layout: sum_i (Q_ij c_j) s_i = c_j sum_i Q_ij s_i (a column whose pi*theta is exactly 0 holds nothing, as in the reference)
    red[col] = cmul ? (cmul[pc] == 0.0 ? 0.0 : cmul[pc] * t) : t;
  }
}

// split layout: rinv[slot] = [w_i *] recip0(sum of the members' partial row sums), members in order — the bits of the fused combine
__global__ __launch_bounds__(256) void k_row_factors(int64_t n, int P, const double* __restrict__ ypart, const uint16_t* __restrict__ wcode,
                                                    const double* __restrict__ wrow, const double* __restrict__ lut, int weighted,
                                                    double* __restrict__ rinv, const uint32_t* __restrict__ ctl) {
  if (ctl && __hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;   // the run has stopped
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;                   // two slots per thread (16-byte accesses; n is even)
  if (i >= n) return;
  double t0 = 0.0, t1 = 0.0;
  for (int p = 0; p < P; ++p) {
    const double2 v = *reinterpret_cast<const double2*>(ypart + (int64_t)p * n + i);
    t0 += v.x; t1 += v.y;
  }
  double r0 = recip0(t0), r1 = recip0(t1);
  if (weighted) {
    if (wcode) { r0 *= lut[wcode[i]]; r1 *= lut[wcode[i + 1]]; }
    else { r0 *= wrow[i]; r1 *= wrow[i + 1]; }
  }
  *reinterpret_cast<double2*>(rinv + i) = make_double2(r0, r1);
}

__global__ void k_keep_err(const uint32_t* sync, uint32_t* errlog) { errlog[0] |= sync[9]; errlog[1] = sync[10]; }

// Loop control of tsem_em_chunk, evaluated on the device so that the host need not synchronise every
// iteration: ctl[0] = stop flag (0 run, 1 converged, 2 a rank's EM pass timed out), ctl[1] = iterations
// committed since the host last cleared it.
struct UpdCtl {
  uint32_t* ctl;          // null: legacy stepwise use (always commit unless the error flag is up)
  double eps;             // converged = diff_est < eps (model.py:792) unless use_lnl
  int use_lnl;            // convergence is decided by k_lnl_check instead (model.py:785-789)
  double* pi_first;       // non-null: also store the new parameters here (pi_init / theta_init, model.py:776-778)
  double* theta_first;
  // lagged log-likelihood test (fused kernel MODE 4): red[K+1] is the lnl of the iteration committed LAST; if it ends the run
  // (model.py:785-789) this pass's sums are dropped — the state stays the reference's after its last iteration
  int lag_check;
  double* ctld;           // [0] the lnl before that one
  double* lnl_slot;       // where the value goes (the iteration's slot of the chunk's trace, or the carry slot)
};

// M-step closed forms (model.py:733-740) + per-block partials of diff_est (model.py:781)
__global__ __launch_bounds__(256) void k_update(UpdCtl C, int K, const double* __restrict__ red,
    const double* __restrict__ pisum0, double theta_pw, double theta_den, double pi_pw, double pi_den,
    double* __restrict__ pi, double* __restrict__ theta, double* __restrict__ pi_prev,
    double* __restrict__ theta_prev, const uint32_t* __restrict__ colmap, int Kp,
    double* __restrict__ ctab, double* __restrict__ ctab_prev, const int32_t* __restrict__ twin_rep,
    double* __restrict__ diff_part, double* __restrict__ diff_out, uint32_t* __restrict__ done,
    uint32_t* __restrict__ fz_sync, uint32_t* __restrict__ fz_errlog, unsigned long long* __restrict__ fz_xchg, int64_t fz_xchg_n) {
  __shared__ double scratch[16];
  __shared__ bool last;
  // the fused kernel's sync words and exchange ring must be zero at its next launch: do it here (this
  // kernel runs once per EM pass, after the pass) instead of three memsets in front of every launch
  if (fz_sync) {
    (void)fz_xchg; (void)fz_xchg_n;                        // (the ring itself is cleared by k_colreduce)
    if (blockIdx.x == 0 && threadIdx.x < 16) {            // FZ_SYNC_WORDS
      if (threadIdx.x == 9) atomicOr(&fz_errlog[0], fz_sync[9]);          // keep the error word / miss counter for the host
      if (threadIdx.x == 10) fz_errlog[1] = fz_sync[10];
      fz_sync[threadIdx.x] = 0u;
    }
  }
  if (C.ctl && __hip_atomic_load(C.ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;   // the run has stopped: nothing to commit
  // some rank's fused pass timed out (flag summed by the all-reduce): the column sums are incomplete, so NO
  // rank commits; the last block raises stop = 2 and the host redoes the iteration (tsem_em_chunk)
  const bool failed = red[K] > 0.0;
  // (every block reads the same two numbers; ctld[0] is rewritten by the LAST block only, after all have passed this point)
  const double lag_lnl = (C.lag_check && !failed) ? red[K + 1] : 0.0;
  const bool lag_stop = C.lag_check && !failed && fabs(lag_lnl - C.ctld[0]) < C.eps;
  double d = 0.0;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < K && !failed && !lag_stop; j += gridDim.x * blockDim.x) {
    // exact twin columns share one accumulation (see k_colsig) as long as their
    // sums agree to rounding, i.e. their parameters are still symmetric
    const int jr = twin_rep[j];
    double ts = red[j];
    if (jr != j) {
      double tr = red[jr];
      if (fabs(ts - tr) <= 1e-12 * fmax(fabs(ts), fabs(tr))) ts = tr;
    }
    double th = (ts + theta_pw) / theta_den;
    double ps = pisum0[j] + ts;
    double ph = (ps + pi_pw) / pi_den;
    double po = pi[j], to = theta[j];
    d += fabs(ph - po);
    pi_prev[j] = po; theta_prev[j] = to;
    pi[j] = ph; theta[j] = th;
    if (C.pi_first) { C.pi_first[j] = ph; C.theta_first[j] = th; }
    const uint32_t cm = colmap[j];
    const int pc = (int)(cm >> CM_PS) * Kp + (int)(cm & CM_SM), copies = 1 << ((cm >> CM_LS) & 7u);
    const double cold = ctab[pc], cnew = ph * th;
    for (int c = 0; c < copies; ++c) { ctab_prev[pc + c] = cold; ctab[pc + c] = cnew; }
  }
  double t = block_sum(d, scratch);
  if (threadIdx.x == 0) {
    diff_part[blockIdx.x] = t;
    __threadfence();
    last = atomicAdd(done, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (last) {                                              // fixed order -> deterministic
    __threadfence();
    double v = 0.0;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += blockDim.x) v += __hip_atomic_load(&diff_part[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    double tt = block_sum(v, scratch);
    if (threadIdx.x == 0) {
      *done = 0u;
      if (failed) {
        if (C.ctl) C.ctl[0] = 2u;
        *diff_out = -1.0;                                  // (legacy stepwise hosts: negative = not committed)
      } else if (lag_stop) {                               // the previous iteration was the last one: nothing committed
        *C.lnl_slot = lag_lnl; C.ctld[0] = lag_lnl;
        C.ctl[0] = 1u;
      } else {
        if (C.lag_check) { *C.lnl_slot = lag_lnl; C.ctld[0] = lag_lnl; }
        *diff_out = tt;
        if (C.ctl) {
          C.ctl[1] += 1u;
          if (!C.use_lnl && tt < C.eps) C.ctl[0] = 1u;     // model.py:792
        }
      }
    }
  }
}

// use_likelihood convergence test (model.py:785-789): lred[0] = all-reduced log-likelihood of the iteration just
// committed, lred[1] > 0 when some rank's lnl pass timed out.
__global__ void k_lnl_check(uint32_t* ctl, double* ctld, const double* __restrict__ lred, double eps,
                            double* __restrict__ lnl_out) {
  if (ctl[0]) return;
  if (lred[1] > 0.0) { ctl[0] = 3u; return; }
  const double l = lred[0];
  *lnl_out = l;
  if (fabs(l - ctld[0]) < eps) ctl[0] = 1u;
  ctld[0] = l;
}
// lnl partial + error flag of this rank into the two lnl reduce slots
__global__ void k_lnl_slots(const double* __restrict__ red_lnl, const uint32_t* __restrict__ sync, double* __restrict__ lred,
                            const uint32_t* __restrict__ errlog = nullptr /* split layout: error words of the earlier launches of the pass */) {
  lred[0] = *red_lnl;
  lred[1] = ((sync && sync[9]) || (errlog && errlog[0])) ? 1.0 : 0.0;
}

static_assert(FZ_SYNC_WORDS == 16, "k_update clears 16 sync words");

// ---- reduced-precision EM pass (BASELINE config 3: the fp32 leg of the tolerance sweep) -------------
// A DIAGNOSTIC, not a product path: the E-step products, the row sums, the posteriors and the column sums are all
// fp32.  Q = expm1(100 s / max) reaches 2.7e43 > FLT_MAX, so the score table is scaled by 2^-64
// (exact) before rounding to fp32, and pi*theta by 1 / max_j(pi*theta) (z is invariant under both); products
// that still underflow are lost — which is what the sweep is there to measure.  One 16-lane group per row of the
// canonical CSR, fp32 global atomics for the column sums.
constexpr int F32_SHIFT = 64;
__global__ void k_cmax(int K, const double* __restrict__ pi, const double* __restrict__ theta, unsigned long long* __restrict__ out) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  double v = j < K ? pi[j] * theta[j] : 0.0;
  v = sg_max<64>(v);
  if ((threadIdx.x & 63) == 0) atomicMax(out, (unsigned long long)__double_as_longlong(v));   // non-negative doubles order like integers
}
__global__ void k_make_c32(int K, const double* __restrict__ pi, const double* __restrict__ theta,
                           const unsigned long long* __restrict__ cmax, float* __restrict__ c32) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= K) return;
  const double m = __longlong_as_double((long long)*cmax);
  c32[j] = (float)((pi[j] * theta[j]) / (m > 0.0 ? m : 1.0));
}
__global__ void k_lut32(int n, const double* __restrict__ lut, float* __restrict__ lut32) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) lut32[i] = (float)ldexp(lut[i], -F32_SHIFT);
}
__global__ __launch_bounds__(256) void k_em_rows_f32(int64_t N, const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
    const uint16_t* __restrict__ raw, const float* __restrict__ lut32, const float* __restrict__ c32, float* __restrict__ colsums) {
  const int sub = threadIdx.x / RP_SUB, lane = threadIdx.x % RP_SUB, subs = blockDim.x / RP_SUB;
  for (int64_t row = (int64_t)blockIdx.x * subs + sub; row < N; row += (int64_t)gridDim.x * subs) {
    const int64_t s = indptr[row], e = indptr[row + 1];
    if (e - s < 2) continue;                               // unique rows feed pi through pisum0 only (model.py:699)
    float y = 0.f, w = 0.f;
    for (int64_t k = s + lane; k < e; k += RP_SUB) { const float q = lut32[raw[k]]; y += q * c32[indices[k]]; w = fmaxf(w, q); }
#pragma unroll
    for (int o = RP_SUB / 2; o > 0; o >>= 1) { y += __shfl_xor(y, o, RP_SUB); w = fmaxf(w, __shfl_xor(w, o, RP_SUB)); }
    float r = 1.f / y;
    if (isinf(r)) r = 0.f;                                 // recip0, sparse_plus.py:16-22
    for (int64_t k = s + lane; k < e; k += RP_SUB) {
      const float v = ((lut32[raw[k]] * c32[indices[k]]) * r) * w;
      if (v != 0.f) unsafeAtomicAdd(&colsums[indices[k]], v);
    }
  }
}
// Any K (more than 64 column parts): the EM pass as a plain CSR row pass — 16 lanes per row, pi*theta gathered from global memory,
// w*z added to the column sums with global fp64 atomics.  A completeness path (the reference takes any number of loci), an
// order of magnitude slower per entry than the fused kernel.  red[0..K) must be zero; unique rows feed pi through pisum0 only.
__global__ __launch_bounds__(256) void k_em_rows(int64_t N, const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
    const uint16_t* __restrict__ raw, const double* __restrict__ lut, const double* __restrict__ c, double* __restrict__ red,
    const uint32_t* __restrict__ ctl) {
  if (ctl && __hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;   // the run has stopped
  const int sub = threadIdx.x / RP_SUB, lane = threadIdx.x % RP_SUB, subs = blockDim.x / RP_SUB;
  for (int64_t row = (int64_t)blockIdx.x * subs + sub; row < N; row += (int64_t)gridDim.x * subs) {
    const int64_t s = indptr[row], e = indptr[row + 1];
    if (e - s < 2) continue;
    double y = 0.0, w = 0.0;
    for (int64_t k = s + lane; k < e; k += RP_SUB) { const double q = lut[raw[k]]; y += q * c[indices[k]]; w = fmax(w, q); }
    y = sg_sum<RP_SUB>(y); w = sg_max<RP_SUB>(w);
    const double r = recip0(y) * w;                         // sparse_plus.py:16-22, model.py:730
    for (int64_t k = s + lane; k < e; k += RP_SUB) {
      const double v = (lut[raw[k]] * c[indices[k]]) * r;
      if (v != 0.0) unsafeAtomicAdd(&red[indices[k]], v);
    }
  }
}
// ... and calculate_lnl's ambiguous rows (model.py:744-760): sum z(prev) log1p(Q c_cur), one partial per workgroup
__global__ __launch_bounds__(256) void k_lnl_rows_amb(int64_t N, const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
    const uint16_t* __restrict__ raw, const double* __restrict__ lut, const double* __restrict__ c_prev, const double* __restrict__ c_cur,
    double* __restrict__ part) {
  __shared__ double scratch[16];
  const int sub = threadIdx.x / RP_SUB, lane = threadIdx.x % RP_SUB, subs = blockDim.x / RP_SUB;
  double acc = 0.0;
  for (int64_t row = (int64_t)blockIdx.x * subs + sub; row < N; row += (int64_t)gridDim.x * subs) {
    const int64_t s = indptr[row], e = indptr[row + 1];
    if (e - s < 2) continue;
    double y = 0.0;
    for (int64_t k = s + lane; k < e; k += RP_SUB) y += lut[raw[k]] * c_prev[indices[k]];
    y = sg_sum<RP_SUB>(y);
    const double r = recip0(y);
    for (int64_t k = s + lane; k < e; k += RP_SUB) {
      const double q = lut[raw[k]];
      const double z = (q * c_prev[indices[k]]) * r;
      if (z != 0.0) acc += z * ts_log1p_pos(q * c_cur[indices[k]]);
    }
  }
  const double t = block_sum(acc, scratch);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}
__global__ void k_red_from_f32(int K, const float* __restrict__ colsums, double* __restrict__ red) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < K) red[j] = ldexp((double)colsums[j], F32_SHIFT);
  if (j == 0) { red[K] = 0.0; red[K + 1] = 0.0; }
}

extern "C" {

// ---------------------------------------------------------------------------
// EM pass / update / lnl
// ---------------------------------------------------------------------------
static int launch_phase1(tsem_ctx* h, const double* ctab, int64_t b0 = 0, int64_t b1 = -1, bool em = false) {
  if (h->nb == 0) return TSEM_OK;
  if (b1 < 0) b1 = h->nb;
  const size_t lds1 = (size_t)(h->Kp + h->R) * 8;
  k_phase1<512><<<h->G1 * h->P, 512, lds1, h->stream>>>(h->P, h->Kp, h->R, b0, b1, h->G1, h->N_amb_pad, h->d_sb_off,
                                                        h->d_pval, h->d_prc, ctab, h->d_ypart, em ? h->d_ctl : nullptr);
  TSEM_HIP(hipGetLastError());
  return TSEM_OK;
}

static int begin_timing(tsem_ctx* h, hipEvent_t** pair) {
  *pair = nullptr;
  // option "kernel_timing" = n: HIP events around every n-th EM pass (0 = never, default 1).  An event pair costs
  // the stream several microseconds per iteration (profiles/r02_comm_overhead.txt): the Python host switches it
  // off for em(), bench.py samples every 4th launch of the timed region.
  if (h->opt_timing <= 0 || (h->em_launches % h->opt_timing) != 0) return TSEM_OK;
  if (h->ev_used + 2 > 8192) return TSEM_OK;
  while (h->ev.size() < h->ev_used + 2) {
    hipEvent_t e;
    TSEM_HIP(hipEventCreate(&e));
    h->ev.push_back(e);
  }
  *pair = &h->ev[h->ev_used];
  h->ev_used += 2;
  h->em_timed += 1;
  TSEM_HIP(hipEventRecord((*pair)[0], h->stream));
  return TSEM_OK;
}

// option "phase_timing": one HIP event per phase boundary of every iteration a chunk enqueues (tsem_em_chunk sets pev_iter and reads
// the events back after its synchronisation).  Marks: 0 before the pass, 1 after it, 2 after the column reduce, 3 after the
// all-reduce, 4 after the update.  A diagnostic (bench.py's `phase_us`): each event costs the stream a few microseconds.
static int phase_mark(tsem_ctx* h, int k) {
  if (h->opt_phase <= 0 || h->pev_iter < 0) return TSEM_OK;
  const size_t idx = (size_t)h->pev_iter * TS_PHASE_MARKS + (size_t)k;
  while (h->pev.size() <= idx) {
    hipEvent_t e;
    TSEM_HIP(hipEventCreate(&e));
    h->pev.push_back(e);
  }
  if (h->pev_set.size() <= idx) h->pev_set.resize(idx + 1, 0);
  TSEM_HIP(hipEventRecord(h->pev[idx], h->stream));
  h->pev_set[idx] = 1;
  return TSEM_OK;
}
// ... read back after the chunk's synchronisation: n iterations were enqueued (all of them ran unless the device stopped the chunk:
// kernels behind a stop return at once, their phases then measure launch overhead only — callers time fixed-length chunks)
static void phase_harvest(tsem_ctx* h, int n) {
  if (h->opt_phase <= 0) return;
  auto el = [&](size_t a, size_t b, double* out) -> bool {
    if (a >= h->pev_set.size() || b >= h->pev_set.size() || !h->pev_set[a] || !h->pev_set[b]) return false;
    float ms = 0;
    if (hipEventElapsedTime(&ms, h->pev[a], h->pev[b]) != hipSuccess) return false;
    *out = ms;
    return true;
  };
  for (int i = 0; i < n; ++i) {
    const size_t b = (size_t)i * TS_PHASE_MARKS;
    double pass = 0, cr = 0, ar = 0, up = 0, tot = 0, gap = 0;
    if (!el(b, b + 4, &tot)) continue;
    if (!el(b, b + 1, &pass) || !el(b + 1, b + 2, &cr)) { pass = 0; cr = 0; (void)el(b, b + 2, &pass); }   // (no mark between pass and reduce)
    (void)el(b + 2, b + 3, &ar);
    (void)el(b + 3, b + 4, &up);
    h->phase_ms[0] += pass; h->phase_ms[1] += cr; h->phase_ms[2] += ar; h->phase_ms[3] += up; h->phase_ms[5] += tot;
    if (i + 1 < n && el(b + 4, b + TS_PHASE_MARKS, &gap)) h->phase_ms[4] += gap;
    h->phase_n += 1;
  }
  std::fill(h->pev_set.begin(), h->pev_set.end(), 0);
}

// ---- log tables of the log-likelihood passes (tsem_fused.h, fz_log1p_of_log) ----
// ... and, on the way, how many stored entries sit in columns whose log(pi*theta) may take an entry into the exact branch of the log
// form (FZ_L_ZERO <= log Q + log c < FZ_L_FAST for SOME stored score: lq_lo / lq_hi bound log Q): in that branch a wave fetches pi*theta
// from global memory and waits for it behind its whole prefetch queue.  Columns on their way to pi = 0 pass through that range: at
// K = 50k, ~100 per row, a third of the columns is there after 20 iterations and the pass takes 5.0 ms where the per-entry logarithm
// takes 3.2 (profiles/r05_lnl_evolution.txt).  The two forms of the pass read the count and one of them returns at once.
__global__ void k_log_tab(int n, const double* __restrict__ c, double* __restrict__ lc, const int32_t* __restrict__ col_of_pc,
                          const unsigned long long* __restrict__ colcount, double lq_lo, double lq_hi, unsigned long long* __restrict__ mid) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long m = 0;
  if (t < n) {
    const double l = log(c[t]);                            // log 0 = -inf: the kernel's "zero" range
    lc[t] = l;
    if (mid && colcount) {
      // the share of the column's entries whose log Q puts them between the two limits, for scores spread evenly over [lq_lo, lq_hi]
      // (real scores crowd near the top: an over-estimate, i.e. the per-entry logarithm takes over a little early)
      const int j = col_of_pc[t];                          // (-1: padding, the further slots of a split column)
      if (j >= 0 && l + lq_hi >= FZ_L_ZERO && l + lq_lo < FZ_L_FAST) {
        const double w = lq_hi - lq_lo;                    // (lq_lo = -inf: codes the arithmetic log Q forces into the branch — every entry counts)
        const bool spread = w > 0.0 && w < INFINITY;
        const double below_fast = spread ? fmin(1.0, fmax(0.0, (FZ_L_FAST - l - lq_lo) / w)) : 1.0;
        const double below_zero = spread ? fmin(1.0, fmax(0.0, (FZ_L_ZERO - l - lq_lo) / w)) : 0.0;
        m = (unsigned long long)((double)colcount[j] * (below_fast - below_zero) + 0.5);
      }
    }
  }
  if (mid) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m += __shfl_xor(m, o, 64);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(mid, m);
  }
}
// log Q, built once per layout and score table.  Code entries: one value per code.  fp64 entries: a direct-index table on the top bits
// of Q — the fewest mantissa bits that keep the codes 1 .. max apart — provided it fits the LDS the layout leaves.  lq_n stays 0 when
// the tables do not apply (no score table in LDS) or do not fit: the passes then evaluate the logarithm per entry as before.
static int ensure_log_tables(tsem_ctx* h) {
  if (h->lq_tried) return TSEM_OK;
  h->lq_tried = true;
  h->lq_n = 0;
  const int fmt = fz_fmt(h);
  if (fmt == 0 || !h->use_fused || h->split || h->lut_len <= 1 || (int)h->lut_host.size() != h->lut_len) return TSEM_OK;
  const size_t room = (size_t)(TS_LDS_MAX - 1024) - fz_lds_bytes(h, true);
  std::vector<double> tab;
  h->lq_lin = 0;
  if (fmt == 1) {
    // The reference's table (model.py:653: Q = expm1((code * (1 / max)) * 100.)) has log Q = t to the last bits once t = (code / max) * 100
    // >= 38: then the kernel needs no log Q table at all — which is what lets the layouts whose row slots fill the LDS (short rows
    // at K = 30k) use the log form too.  Checked code by code against log(lut[code]); any other table takes the look-up.
    {
      const int mx = h->lut_len - 1;
      const double a = 1.0 / (double)mx, b = std::nearbyint(std::log1p(h->lut_host[mx]));
      int c0 = -1;
      bool ok = mx >= 1 && b >= 40.0 && b <= 700.0;
      for (int r = 1; r <= mx && ok; ++r) {
        const double t = ((double)r * a) * b;
        if (t < 38.0) continue;
        if (c0 < 0) c0 = r;
        const double lq = std::log(h->lut_host[r]);
        ok = std::fabs(lq - t) <= 4.0 * (std::nextafter(t, INFINITY) - t);
      }
      if (ok && c0 > 0) { h->lq_lin = 1; h->lq_c0 = c0; h->lq_a = a; h->lq_b = b; }
    }
    h->lq_tab_fits = (size_t)h->lut_len * 8 <= room;
    if (!h->lq_lin && !h->lq_tab_fits) return TSEM_OK;
    tab.resize(h->lut_len);
    for (int i = 0; i < h->lut_len; ++i) tab[i] = std::log(h->lut_host[i]);   // (log 0 = -inf; kept for layout_info / when the table is not the reference's)
    h->lq_shift = 0; h->lq_base = 0;
  } else {
    auto hi = [](double q) { int64_t b; std::memcpy(&b, &q, 8); return (int)(b >> 32); };
    bool ok = false;
    for (int bits = 0; bits <= 8 && !ok; ++bits) {
      const int shift = 20 - bits;
      int lo = INT32_MAX, top = INT32_MIN;
      for (int i = 0; i < h->lut_len; ++i) {
        const double q = h->lut_host[i];
        if (!(q > 0.0) || !std::isfinite(q)) continue;
        lo = std::min(lo, hi(q) >> shift); top = std::max(top, hi(q) >> shift);
      }
      if (lo > top) return TSEM_OK;
      const int64_t n = (int64_t)top - lo + 1;
      if (n <= 0 || (size_t)n * 8 > room) break;            // more bits only make it larger
      std::vector<double> t((size_t)n, 0.0);
      std::vector<uint8_t> used((size_t)n, 0);
      bool unique = true;
      for (int i = 0; i < h->lut_len && unique; ++i) {
        const double q = h->lut_host[i];
        if (!(q > 0.0) || !std::isfinite(q)) continue;
        const int k = (hi(q) >> shift) - lo;
        if (used[k] && t[k] != std::log(q)) unique = false;
        used[k] = 1; t[k] = std::log(q);
      }
      if (unique) { tab.swap(t); h->lq_shift = shift; h->lq_base = lo; ok = true; }
    }
    if (!ok) return TSEM_OK;
  }
  TSEM_ALLOC(h->d_lqtab, tab.size());
  TSEM_HIP(hipMemcpyAsync(h->d_lqtab, tab.data(), sizeof(double) * tab.size(), hipMemcpyHostToDevice, h->stream));
  TSEM_HIP(hipStreamSynchronize(h->stream));               // (tab is a local)
  fz_fn f9 = fz_kernel(h->P, 9, fmt, h->geo);
  if (!f9) return TSEM_OK;
  TSEM_HIP(hipFuncSetAttribute((const void*)f9, hipFuncAttributeMaxDynamicSharedMemorySize, TS_LDS_MAX - 1024));
  TSEM_ALLOC(h->d_lctab, h->Kpad);
  // the range of log Q over the stored scores, for the choice between the two forms of the pass (k_log_tab)
  h->lq_lo = 0.0; h->lq_hi = 0.0;
  if (h->d_colcount && h->d_col_of_pc && h->nnz > 0 && h->max_code > 0 && h->max_code < h->lut_len && h->min_code >= 1 && h->min_code <= h->max_code) {
    const int mn = h->min_code;                            // (tsem_max_score found both ends in its one pass over the scores)
    if (h->lut_host[mn] > 0.0 && std::isfinite(h->lut_host[h->max_code])) {
      TSEM_ALLOC(h->d_lq_mid, 1);
      h->lq_lo = std::log(h->lut_host[mn]); h->lq_hi = std::log(h->lut_host[h->max_code]);
      // the arithmetic log Q sends every code below lq_c0 into the exact branch: with such scores stored, take the look-up where its
      // table fits, else let every live column count (-> the per-entry logarithm runs)
      if (h->lq_lin && mn < h->lq_c0) {
        if (h->lq_tab_fits) h->lq_lin = 0; else h->lq_lo = -INFINITY;
      }
    }
  }
  h->lq_n = (int)tab.size();
  return TSEM_OK;
}

// One launch of the persistent fused kernel.  mode 0: EM pass (column sums of w*z into d_fpartial);
// mode 1: log-likelihood of the ambiguous rows (one partial per workgroup into d_lnl_part).
static int launch_fused(tsem_ctx* h, int mode, hipEvent_t* pair, int bin = 0) {
  if (!h->fz_clean) {                                      // (k_update leaves them zero after every EM pass)
    if (h->fused_launched && h->d_fz_aux)                  // keep the error word of a launch nobody cleaned up after
      k_keep_err<<<1, 1, 0, h->stream>>>(h->d_xflags, h->d_fz_aux + 2);
    TSEM_HIP(hipMemsetAsync(h->d_xflags, 0, sizeof(uint32_t) * FZ_SYNC_WORDS, h->stream));
    if (h->P > 1) TSEM_HIP(hipMemsetAsync(h->d_xchg, 0, sizeof(double) * (size_t)h->fz_teams * FZ_XS * h->P * h->R, h->stream));
  }
  h->fz_clean = false;
  int kmode = mode;                                        // the instantiation launched (9 = mode 1 with log tables)
  if (mode == 1 || mode == 8)                              // teams that do not form write nothing
    TSEM_HIP(hipMemsetAsync(h->d_lnl_part + (mode == 8 ? (size_t)bin * h->fz_grid : 0), 0, sizeof(double) * (size_t)h->fz_grid, h->stream));
  FusedArgs A;
  A.P = h->P; A.Kp = h->Kp; A.R = h->R; A.nb = h->nb; A.N_amb_pad = h->N_amb_pad;
  A.sb_off = h->d_sb_off; A.sb_q32 = h->d_sb_q32; A.pval = h->d_pval; A.prc = h->d_prc;
  const bool lnl = mode == 1;                               // modes 2, 3 (exact column sums) and 4 (+ the previous lnl) are EM passes
  A.ctab = lnl ? h->d_ctab_prev : h->d_ctab; A.ctab2 = mode == 4 ? h->d_ctab_prev : h->d_ctab; A.lnl_out = h->d_lnl_part; A.lnl_mode = lnl ? 1 : 0;
  A.rinv = h->d_rinv; A.lag = (mode == 4 && h->lag_valid) ? 1 : 0;
  A.Kh = (h->Kp + 1) / 2; A.koff = 0; A.ypart = h->d_ypart;
  if (mode == 5 && bin == 1) { A.ctab = h->d_ctab_prev; A.lnl_mode = 1; }           // split layout, lnl: unweighted recip0 of the PREVIOUS parameters' row sums
  if (mode == 8) { A.ctab = h->d_ctab_prev; A.ctab2 = h->d_ctab; A.koff = bin * A.Kh; A.lnl_out = h->d_lnl_part + (size_t)bin * h->fz_grid; }
  A.wrow = h->d_amb_w; A.partial = h->d_fpartial; A.xchg = h->d_xchg; A.sorted = h->sorted_layout ? 1 : 0;
  A.sync = h->d_xflags;
  A.prof = (mode == 0 || mode == 4 || mode == 5) ? h->d_prof : nullptr; A.prof_blocks = A.prof ? h->prof_steps : 0; A.dbg = (int)h->opt_dbg;
  A.ctl = h->d_ctl;
  A.ebias = h->d_ebias; A.bin = bin; A.ovf = h->d_ovf; A.partial2 = h->d_fpartial2;

  A.pcode = h->d_pcode; A.lut = h->d_lut; A.lut_len = fz_fmt(h) ? h->lut_len : 0; A.wcode = h->d_amb_wcode;
  size_t ldsf = fz_lds_bytes(h, fz_fmt(h) != 0);
  A.lctab = nullptr; A.lqtab = nullptr; A.lq_n = 0; A.lq_shift = 0; A.lq_base = 0; A.lq_lin = 0; A.lq_c0 = 0; A.lq_a = 0; A.lq_b = 0;
  A.sel = nullptr; A.sel_thr = 0; A.sel_want = 0;
  if (mode == 1 && !(h->opt_dbg & 8192)) {                  // (fused_dbg bit 13: the per-entry logarithm, for A/B timing and tests)
    if (int rc = ensure_log_tables(h)) return rc;
    if (h->lq_n > 0) {
      // fused_dbg bit 15: the log form whatever the parameters look like (tests of its exact branch, A/B timing)
      const bool choose = h->d_lq_mid && !(h->opt_dbg & 32768);
      if (choose) TSEM_HIP(hipMemsetAsync(h->d_lq_mid, 0, sizeof(unsigned long long), h->stream));
      k_log_tab<<<cdiv64(h->Kpad, 256), 256, 0, h->stream>>>(h->Kpad, h->d_ctab, h->d_lctab, h->d_col_of_pc, h->d_colcount, h->lq_lo, h->lq_hi,
                                                              choose ? h->d_lq_mid : nullptr);
      TSEM_HIP(hipGetLastError());
      if (choose) {
        // a wave takes the exact branch when ANY of its 256 entries of a step is in the range: with a share f of the entries there,
        // 1 - (1 - f)^256 of the steps stall — 22 % at f = 1e-3, about where the two forms cost the same
        A.sel = h->d_lq_mid; A.sel_thr = (unsigned long long)(1e-3 * (double)h->nnz); A.sel_want = 0;
        h->lq_choice[0] += 1;
      }
      A.lctab = h->d_lctab; A.lqtab = h->d_lqtab; A.lq_n = h->lq_n; A.lq_shift = h->lq_shift; A.lq_base = h->lq_base;
      A.lq_lin = (fz_fmt(h) == 1 && !((h->opt_dbg & 16384) && h->lq_tab_fits)) ? h->lq_lin : 0;   // (fused_dbg bit 14: the look-up even where the arithmetic form applies — tests)
      A.lq_c0 = h->lq_c0; A.lq_a = h->lq_a; A.lq_b = h->lq_b;
      if (!A.lq_lin) ldsf += (size_t)h->lq_n * 8;
      kmode = 9;
    }
  }
  if ((lnl || mode == 8) && h->fz_grid > 2048) TSEM_FAIL(TSEM_ERR_ARG, "fused lnl: more workgroups than partial slots");
  fz_fn fn = fz_kernel(h->P, kmode, fz_fmt(h), h->geo);
  if (!fn) TSEM_FAIL(TSEM_ERR_ARG, mode >= 2 ? "reproducible mode needs the fused kernel with a score table of at most 2048 entries"
                                               : "fused kernel supports at most 8 column parts");
  if (pair) TSEM_HIP(hipEventRecord(pair[0], h->stream));   // time the kernel, not the memsets
  fn<<<h->fz_grid, FZ_NT, ldsf, h->stream>>>(A);
  TSEM_HIP(hipGetLastError());
  if (kmode == 9 && A.sel) {                                // ... and the per-entry-logarithm form behind it: exactly one of the two runs
    fz_fn f1 = fz_kernel(h->P, 1, fz_fmt(h), h->geo);
    if (!f1) TSEM_FAIL(TSEM_ERR_ARG, "fused lnl: no kernel for this team size / geometry");
    FusedArgs B = A;
    B.lctab = nullptr; B.lqtab = nullptr; B.lq_n = 0; B.lq_lin = 0; B.sel_want = 1;
    f1<<<h->fz_grid, FZ_NT, fz_lds_bytes(h, fz_fmt(h) != 0), h->stream>>>(B);
    TSEM_HIP(hipGetLastError());
  }
  h->fused_launched = true;
  return TSEM_OK;
}

static int launch_row_factors(tsem_ctx* h, int weighted) {
  const bool codes = h->fmt_code || h->fmt_wcode;
  k_row_factors<<<cdiv64(h->N_amb_pad / 2 + 1, 256), 256, 0, h->stream>>>(h->N_amb_pad, h->P, h->d_ypart, codes ? h->d_amb_wcode : nullptr,
                                                                          codes ? nullptr : h->d_amb_w, h->d_lut, weighted, h->d_rinv, h->d_ctl);
  TSEM_HIP(hipGetLastError());
  return TSEM_OK;
}

int tsem_rowpass_grid(tsem_ctx* h);
// the fp32 diagnostic pass (option "em_precision" = 1): same outputs as tsem_em_pass, fp32 arithmetic
static int em_pass_f32(tsem_ctx* h) {
  const int K = h->K;
  if (int rc = tsem_ensure_indices(h)) return rc;
  if (!h->d_c32) {
    TSEM_ALLOC(h->d_c32, K); TSEM_ALLOC(h->d_cs32, K); TSEM_ALLOC(h->d_lut32, h->lut_len);
    k_lut32<<<cdiv64(h->lut_len, 256), 256, 0, h->stream>>>(h->lut_len, h->d_lut, h->d_lut32);
  }
  unsigned long long* cmax = reinterpret_cast<unsigned long long*>(h->d_lnl_part + 12000);
  TSEM_HIP(hipMemsetAsync(cmax, 0, 8, h->stream));
  TSEM_HIP(hipMemsetAsync(h->d_cs32, 0, sizeof(float) * K, h->stream));
  k_cmax<<<cdiv64(K, 256), 256, 0, h->stream>>>(K, h->d_pi, h->d_theta, cmax);
  k_make_c32<<<cdiv64(K, 256), 256, 0, h->stream>>>(K, h->d_pi, h->d_theta, cmax, h->d_c32);
  if (h->N) k_em_rows_f32<<<tsem_rowpass_grid(h), 256, 0, h->stream>>>(h->N, h->d_indptr, h->d_indices, h->d_raw, h->d_lut32, h->d_c32, h->d_cs32);
  k_red_from_f32<<<cdiv64(K, 256), 256, 0, h->stream>>>(K, h->d_cs32, h->d_red);
  TSEM_HIP(hipGetLastError());
  return TSEM_OK;
}

// option "reproducible".  Every contribution to a column sum is cut into a high and a low piece on a per-slot grid (tsem_fused.h,
// phase 2), one pass each; sums of such pieces are exact in fp64, so they do not depend on the order of the LDS atomics.  The grid
// hangs on the slot's bound 2^E (ebias = E + 1023): pieces are multiples of 2^(E-30) and 2^(E-60).  All decisions below are functions
// of exact sums, hence identical in every run.
//   k_bin_check, after the high pass: a contribution reached the bound -> raise it;  the high sum lies more than BIN_SLACK bits
//     under what the bound was chosen for (or is zero although the column has entries and a non-zero pi * theta) -> lower it.  Either
//     way the high pass is repeated (one pass lost, the low pass has not run yet).
//   k_bin_finish, after the low pass: S = high + low, and the bound of the NEXT iteration = 16 x the sum this column is expected to
//     have then: columns that fall by d bits per iteration (linear EM convergence) or by d, 2d, 4d, ... bits (a column dying under a
//     zero prior) are followed, so that later iterations rarely repeat a pass.
// Worst-case relative error of a column sum: (entries of the column) x 2^-(57 - BIN_SLACK); typically the fp64 rounding of S.
constexpr int BIN_SLACK = 16;
__device__ __forceinline__ int bin_slot(uint32_t cm, int Kp, int* copies) {
  *copies = 1 << ((cm >> CM_LS) & 7u);
  return (int)(cm >> CM_PS) * Kp + (int)(cm & CM_SM);
}
__global__ void k_bin_check(int K, const double* __restrict__ red, const uint32_t* __restrict__ colmap, int Kp,
                            const unsigned long long* __restrict__ colcount, const uint32_t* __restrict__ ucount,
                            const double* __restrict__ pi, const double* __restrict__ theta,
                            uint16_t* __restrict__ ebias, uint8_t* __restrict__ ovf, uint32_t* __restrict__ flag) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= K || red[K] > 0.0) return;                       // (a pass that timed out: the sums mean nothing, the update kernel refuses them)
  int copies;
  const int pc = bin_slot(colmap[j], Kp, &copies);
  const int eb = ebias[pc];
  bool over = false;
  for (int c = 0; c < copies; ++c) { over |= ovf[pc + c] != 0; ovf[pc + c] = 0; }
  const double S = red[j];
  int eb_new = eb;
  if (over) eb_new = eb + 12;
  else if (S == 0.0) {
    // nothing arrived: too coarse a grid, unless the column has no entry in a row of several (those of single-entry rows feed pi
    // through pisum0, not through this pass) or pi * theta is zero
    if (colcount && colcount[j] > (ucount ? ucount[j] : 0u) && pi[j] * theta[j] != 0.0 && eb > 120) eb_new = eb - 24;
  }
  else {
    // (a high sum a few grid steps large says little about S: move by at most 24 bits and keep 6 bits in hand)
    const int ex4 = (int)((__double2hiint(S) >> 20) & 0x7FF) + 4;
    if (eb - ex4 > BIN_SLACK) eb_new = max(ex4 + 6, eb - 24);
    else if (ex4 - eb > 20) eb_new = ex4;                     // a sum 2^16 bounds large: more would not be exact (53 - 30 bits of room)
  }
  eb_new = min(2000, max(64, eb_new));
  if (eb_new != eb) {
    for (int c = 0; c < copies; ++c) ebias[pc + c] = (uint16_t)eb_new;
    atomicOr(flag, 1u);
  }
}
__global__ void k_bin_finish(int K, const double* __restrict__ red_hi, double* __restrict__ red, const uint32_t* __restrict__ colmap, int Kp,
                             uint16_t* __restrict__ ebias, uint8_t* __restrict__ ovf, int16_t* __restrict__ hist) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const double bad = fmax(red[K], red_hi[K]);               // a time-out of either pass: the sums mean nothing, the update kernel refuses
  if (j == 0) red[K] = bad;                                 // them — and the bounds / the drop history must not learn from them
  if (j >= K || bad > 0.0) return;
  const double S = red_hi[j] + red[j];
  red[j] = S;
  int copies;
  const int pc = bin_slot(colmap[j], Kp, &copies);
  for (int c = 0; c < copies; ++c) ovf[pc + c] = 0;          // (the low pass sees the same contributions as the high pass: nothing new)
  if (S == 0.0) return;                                       // no information: keep the bound
  const int ex4 = (int)((__double2hiint(S) >> 20) & 0x7FF) + 4;
  const int eprev = hist[j], dprev = hist[K + j];
  const int drop = eprev ? eprev - ex4 : 0;
  int pred = drop;
  if (drop >= 4 && dprev >= 2) pred = min(drop * drop / dprev, 2 * drop + 2);
  pred = max(-10, min(40, pred));
  const int shift = pred > 3 ? pred - 3 : (pred < 0 ? pred : 0);
  const int eb_new = min(2000, max(64, ex4 - shift));
  for (int c = 0; c < copies; ++c) ebias[pc + c] = (uint16_t)eb_new;
  hist[j] = (int16_t)ex4; hist[K + j] = (int16_t)max(-1000, min(1000, drop));
}

// `lag`: the pass also sums the log-likelihood of the iteration committed last (fused kernel MODE 4; its value goes to slot K+1 of the
// reduce buffer).  Only tsem_em_chunk asks for it, and only when lag_capable().
static bool lag_capable(const tsem_ctx* h) { return h->lnl3 && h->use_fused && h->nb > 0 && !h->opt_reproducible && h->opt_precision == 0; }
static int em_pass(tsem_ctx* h, bool lag);
int tsem_em_pass(tsem_ctx* h) {
  if (!h || !h->have_model) return TSEM_ERR_ARG;
  h->pev_iter = h->opt_phase > 0 ? 0 : -1;                 // host-driven iterations (fall-back transports): marks 0-2 here, 3-4 in tsem_em_update
  const int rc = em_pass(h, false);
  h->pev_iter = -1;
  return rc;
}

static int em_pass_impl(tsem_ctx* h, bool lag);
static int em_pass(tsem_ctx* h, bool lag) {
  if (int rc = phase_mark(h, 0)) return rc;
  if (int rc = em_pass_impl(h, lag)) return rc;
  return phase_mark(h, 2);
}
static int em_pass_impl(tsem_ctx* h, bool lag) {
  if (int rc = ensure_device(h)) return rc;
  if (h->opt_precision == 1) return em_pass_f32(h);
  hipEvent_t* pair = nullptr;
  if (int rc = begin_timing(h, &pair)) return rc;
  if (h->em_rows) {
    // K beyond 64 column parts: the identity column map makes d_ctab pi*theta in column order
    TSEM_HIP(hipMemsetAsync(h->d_red, 0, sizeof(double) * (h->K + 2), h->stream));
    if (h->N) k_em_rows<<<tsem_rowpass_grid(h), 256, 0, h->stream>>>(h->N, h->d_indptr, h->d_indices, h->d_raw, h->d_lut, h->d_ctab, h->d_red, h->d_ctl);
    TSEM_HIP(hipGetLastError());
    if (pair) TSEM_HIP(hipEventRecord(pair[1], h->stream));
    h->em_launches += 1;
    return TSEM_OK;
  }
  if (h->nb > 0 && h->use_fused && h->opt_reproducible) {
    // two exact passes (high and low pieces of every contribution); the high pass is repeated while a column's bound has to move:
    // the first iteration of a run takes a few repeats (the bounds start at the largest fragment weight), later ones rarely any
    if (h->d_ctl) {                                          // a chunk that has stopped: nothing to compute (the passes would return at once)
      uint32_t st = 0;
      TSEM_HIP(hipMemcpyAsync(&st, h->d_ctl, 4, hipMemcpyDeviceToHost, h->stream));
      TSEM_HIP(hipStreamSynchronize(h->stream));
      if (st) { h->em_launches += 1; return TSEM_OK; }
    }
    auto reduce = [&](const double* partial) -> int {
      k_colreduce<<<cdiv64(h->Kpad, 32), 256, 0, h->stream>>>(h->Kpad, h->fz_teams, partial, h->d_col_of_pc, h->d_colmap, h->d_red, h->K,
                                                              h->d_xflags, h->P, h->d_ctl,
                                                              h->P > 1 ? reinterpret_cast<unsigned long long*>(h->d_xchg) : nullptr,
                                                              h->P > 1 ? (int64_t)h->fz_teams * FZ_XS * h->P * h->R : 0, nullptr, nullptr, 0);
      // (k_colreduce cleared the exchange ring; the sync words are cleared by the memset of the next launch)
      TSEM_HIP(hipGetLastError());
      return TSEM_OK;
    };
    // exact_single: ONE launch leaves the team partials of both pieces (d_fpartial: high, d_fpartial2: low)
    auto pass = [&](int bin, hipEvent_t* ev) -> int {
      if (h->exact_single) {
        if (bin == 1) { if (int rc = launch_fused(h, 3, ev, 0)) return rc; }
        return reduce(bin == 1 ? h->d_fpartial : h->d_fpartial2);
      }
      if (int rc = launch_fused(h, 2, ev, bin)) return rc;
      return reduce(h->d_fpartial);
    };
    for (int attempt = 0;; ++attempt) {
      if (int rc = pass(1, attempt == 0 ? pair : nullptr)) return rc;
      TSEM_HIP(hipMemsetAsync(h->d_binflag, 0, 4, h->stream));
      k_bin_check<<<cdiv64(h->K, 256), 256, 0, h->stream>>>(h->K, h->d_red, h->d_colmap, h->Kp, h->d_colcount, h->d_ucount, h->d_pi, h->d_theta,
                                                            h->d_ebias, h->d_ovf, h->d_binflag);
      TSEM_HIP(hipGetLastError());
      uint32_t redo = 0;
      TSEM_HIP(hipMemcpyAsync(&redo, h->d_binflag, 4, hipMemcpyDeviceToHost, h->stream));
      TSEM_HIP(hipStreamSynchronize(h->stream));
      if (!redo) break;
      if (attempt >= 40) {                                   // (40 x 24 bits: from the largest weight down to the smallest normal number)
        h->bin_inexact = true;                               // the sums of this pass are NOT guaranteed exact: tsem_layout_info[21] says so
        break;
      }
      h->n_bin_repeats += 1;
    }
    TSEM_HIP(hipMemcpyAsync(h->d_red_hi, h->d_red, sizeof(double) * (h->K + 2), hipMemcpyDeviceToDevice, h->stream));
    if (int rc = pass(2, nullptr)) return rc;
    k_bin_finish<<<cdiv64(h->K, 256), 256, 0, h->stream>>>(h->K, h->d_red_hi, h->d_red, h->d_colmap, h->Kp, h->d_ebias, h->d_ovf, h->d_ehist);
    TSEM_HIP(hipGetLastError());
    if (pair) TSEM_HIP(hipEventRecord(pair[1], h->stream));
    h->em_launches += 1;
    return TSEM_OK;
  } else if (h->nb > 0 && h->use_fused) {
    int nu = 0;
    if (lag && h->lag_valid && h->N_uni > 0) {             // the unique rows' share of that log-likelihood (model.py:755-758 with Y = 0)
      nu = (int)std::min<int64_t>(2048, (h->N_uni + 255) / 256);
      k_lnl_unique<<<nu, 256, 0, h->stream>>>(h->N_uni, h->d_uni_col, h->d_uni_code, h->d_lut, h->d_pi_prev, h->d_pi, h->d_lnl_part + 4096);
      TSEM_HIP(hipGetLastError());
    }
    if (h->split) {
      // two light passes, one LDS table each: the row factors w_i * recip0(rowsum_i) through HBM (8 B per row), then
      // acc[j] += Q_ij s_i; k_colreduce multiplies by pi_j theta_j.  The error word of the first launch is kept by k_keep_err.
      if (!h->d_fz_aux) {
        TSEM_ALLOC(h->d_fz_aux, 4);
        TSEM_HIP(hipMemsetAsync(h->d_fz_aux, 0, sizeof(uint32_t) * 4, h->stream));
      }
      if (int rc = launch_fused(h, 5, pair)) return rc;
      if (int rc = launch_row_factors(h, 1)) return rc;
      if (int rc = launch_fused(h, 7, nullptr)) return rc;
      if (pair) TSEM_HIP(hipEventRecord(pair[1], h->stream));
      h->em_launches += 1;
      if (int rc = phase_mark(h, 1)) return rc;
      k_colreduce<<<cdiv64(h->Kpad, 32), 256, 0, h->stream>>>(h->Kpad, h->fz_teams, h->d_fpartial, h->d_col_of_pc, h->d_colmap, h->d_red, h->K,
                                                              h->d_xflags, h->P, h->d_ctl,
                                                              reinterpret_cast<unsigned long long*>(h->d_xchg), (int64_t)h->fz_teams * FZ_XS * h->P * h->R,
                                                              nullptr, nullptr, 0, h->d_ctab, h->d_fz_aux + 2);
      TSEM_HIP(hipGetLastError());
      return TSEM_OK;
    }
    if (int rc = launch_fused(h, lag ? 4 : 0, pair)) return rc;
    if (pair) TSEM_HIP(hipEventRecord(pair[1], h->stream));
    h->em_launches += 1;
    if (int rc = phase_mark(h, 1)) return rc;
    k_colreduce<<<cdiv64(h->Kpad, 32), 256, 0, h->stream>>>(h->Kpad, h->fz_teams, h->d_fpartial, h->d_col_of_pc, h->d_colmap, h->d_red, h->K,
                                                            h->d_xflags, h->P, h->d_ctl,
                                                            h->P > 1 ? reinterpret_cast<unsigned long long*>(h->d_xchg) : nullptr,
                                                            h->P > 1 ? (int64_t)h->fz_teams * FZ_XS * h->P * h->R : 0,
                                                            lag ? h->d_lnl_part : nullptr, h->d_lnl_part + 4096, nu);
    TSEM_HIP(hipGetLastError());
    return TSEM_OK;
  } else if (h->nb > 0) {
    const size_t lds2 = (size_t)(2 * h->Kp + h->R) * 8;
    const int64_t chunk = h->opt_chunk > 0 ? h->opt_chunk : h->nb;
    for (int64_t b0 = 0; b0 < h->nb; b0 += chunk) {
      const int64_t b1 = std::min(h->nb, b0 + chunk);
      if (int rc = launch_phase1(h, h->d_ctab, b0, b1, true)) return rc;
      k_phase2_em<1024><<<h->G2 * h->P, 1024, lds2, h->stream>>>(h->P, h->Kp, h->R, b0, b1, h->G2, b0 > 0 ? 1 : 0,
          h->N_amb_pad, h->d_sb_off, h->d_pval, h->d_prc, h->d_ctab, h->d_ypart, h->d_amb_wcode, h->d_lut, h->d_partial, h->d_ctl);
    }
    TSEM_HIP(hipGetLastError());
  }
  if (pair) TSEM_HIP(hipEventRecord(pair[1], h->stream));
  h->em_launches += 1;
  if (int rc = phase_mark(h, 1)) return rc;
  if (h->nb > 0) {
    k_colreduce<<<cdiv64(h->Kpad, 32), 256, 0, h->stream>>>(h->Kpad, h->G2, h->d_partial, h->d_col_of_pc, h->d_colmap, h->d_red, h->K,
                                                            nullptr, h->P, h->d_ctl, nullptr, 0, nullptr, nullptr, 0);
  } else {
    TSEM_HIP(hipMemsetAsync(h->d_red, 0, sizeof(double) * (h->K + 2), h->stream));
  }
  TSEM_HIP(hipGetLastError());
  return TSEM_OK;
}

static int launch_update(tsem_ctx* h, double* d_diff_slot, bool chunked = false, double eps = 0.0, int use_lnl = 0,
                         double* lag_lnl_slot = nullptr) {
  const double tpw = h->theta_prior * h->w_max, ppw = h->pi_prior * h->w_max;   // model.py:696-697
  const double tden = h->W_amb + tpw * h->K, pden = h->W_tot + ppw * h->K;      // model.py:732,738
  const int nblk = std::min(1024, cdiv64(h->K, 256));
  double* part = h->d_lnl_part + 10000;   // scratch for the per-block |pi_hat - pi| partials
  if (!h->d_fz_aux) {
    TSEM_ALLOC(h->d_fz_aux, 4);
    TSEM_HIP(hipMemsetAsync(h->d_fz_aux, 0, sizeof(uint32_t) * 4, h->stream));
  }
  const bool clean = h->use_fused && h->d_xflags && h->fused_launched;
  UpdCtl C;
  C.ctl = chunked ? h->d_ctl : nullptr; C.eps = eps; C.use_lnl = use_lnl;
  C.pi_first = h->first_pending ? h->d_pi_first : nullptr; C.theta_first = h->first_pending ? h->d_theta_first : nullptr;
  C.lag_check = lag_lnl_slot ? 1 : 0; C.ctld = h->d_ctld; C.lnl_slot = lag_lnl_slot;
  k_update<<<nblk, 256, 0, h->stream>>>(C, h->K, h->d_red, h->d_pisum0, tpw, tden, ppw, pden, h->d_pi, h->d_theta,
                                        h->d_pi_prev, h->d_theta_prev, h->d_colmap, h->Kp, h->d_ctab, h->d_ctab_prev,
                                        h->d_twin_rep, part, d_diff_slot, h->d_fz_aux,
                                        clean ? h->d_xflags : nullptr, h->d_fz_aux + 2,
                                        reinterpret_cast<unsigned long long*>(h->d_xchg),
                                        h->P > 1 ? (int64_t)h->fz_teams * FZ_XS * h->P * h->R : 0);
  TSEM_HIP(hipGetLastError());
  if (clean) h->fz_clean = true;
  h->em_prev = h->em_cur; h->em_cur = true;                // (a skipped update — stop flag, time-out — leaves older, equally valid M-step values)
  return TSEM_OK;
}

// the error word of the fused kernel: live, or what k_update / k_keep_err saved before zeroing it.  Clears it.
int tsem_twopass_attributes(tsem_ctx* h) {
  TSEM_HIP(hipFuncSetAttribute((const void*)k_phase1<512>, hipFuncAttributeMaxDynamicSharedMemorySize, TS_LDS_MAX));
  TSEM_HIP(hipFuncSetAttribute((const void*)k_phase2_em<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, TS_LDS_MAX));
  TSEM_HIP(hipFuncSetAttribute((const void*)k_phase2_lnl<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, TS_LDS_MAX - 1024));
  return TSEM_OK;
}

int tsem_sum_parts(tsem_ctx* h, const double* a, int na, const double* b, int nb, double* out) {
  k_sum_parts<<<1, 256, 0, h->stream>>>(a, na, b, nb, out);
  TSEM_HIP(hipGetLastError());
  return TSEM_OK;
}

int tsem_take_fused_error(tsem_ctx* h, uint32_t* word) {
  *word = 0;
  if (!h->d_xflags || !h->fused_launched) return TSEM_OK;
  uint32_t ee[2] = {0, 0}, kept[2] = {0, 0};
  TSEM_HIP(hipMemcpyAsync(ee, h->d_xflags + 9, 8, hipMemcpyDeviceToHost, h->stream));
  if (h->d_fz_aux) TSEM_HIP(hipMemcpyAsync(kept, h->d_fz_aux + 2, 8, hipMemcpyDeviceToHost, h->stream));
  TSEM_HIP(hipStreamSynchronize(h->stream));
  *word = ee[0] | kept[0];
  h->last_slow_path = h->fz_clean ? kept[1] : ee[1];
  if (*word) {
    TSEM_HIP(hipMemsetAsync(h->d_xflags + 9, 0, 4, h->stream));
    if (h->d_fz_aux) TSEM_HIP(hipMemsetAsync(h->d_fz_aux + 2, 0, 4, h->stream));
  }
  return TSEM_OK;
}

int tsem_fallback_twopass(tsem_ctx* h) {
  if (!h || !h->have_model) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  if (!h->use_fused) return TSEM_OK;
  TSEM_HIP(hipStreamSynchronize(h->stream));
  uint32_t e = 0;
  (void)tsem_take_fused_error(h, &e);
  if (h->opt_reproducible) {
    // the exact sums exist only in the fused kernel: keep the layout and let the caller redo the pass on it (every rank of a
    // row-sharded run sees the time-out through slot K and counts the same retries, so repeated time-outs end the run on
    // all of them together with TSEM_ERR_TIMEOUT — nobody is left in a collective, nothing is half torn down)
    h->opt_dbg &= ~(int64_t)(32 | 64);
    h->n_fallbacks += 1;
    fprintf(stderr, "libtelescope_em: the persistent EM kernel missed a hand-off (watchdog code %u); option `reproducible` keeps the "
                    "fused kernel: redoing the pass\n", e);
    return TSEM_OK;
  }
  h->em_kernel = TSEM_EMK_TWOPASS;
  h->opt_format = 1;                                       // the two-pass kernels read fp64 entries
  h->opt_dbg &= ~(int64_t)(32 | 64);
  if (int rc = tsem_choose_geometry(h)) return rc;
  if (int rc = tsem_build_layout(h)) return rc;
  // the permuted pi*theta tables follow the new column map
  TSEM_ALLOC(h->d_ctab, h->Kpad); TSEM_ALLOC(h->d_ctab_prev, h->Kpad);
  TSEM_HIP(hipMemsetAsync(h->d_ctab, 0, sizeof(double) * h->Kpad, h->stream));
  TSEM_HIP(hipMemsetAsync(h->d_ctab_prev, 0, sizeof(double) * h->Kpad, h->stream));
  if (int rc = tsem_make_ctabs(h)) return rc;
  TSEM_HIP(hipStreamSynchronize(h->stream));
  h->n_fallbacks += 1;
  fprintf(stderr, "libtelescope_em: the persistent EM kernel could not keep its workgroups co-resident (watchdog code %u); "
                  "continuing with the two-pass kernels\n", e);
  return TSEM_OK;
}

int tsem_recover_timeout(tsem_ctx* h, int32_t* switched) {
  if (!h || !h->have_model) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  if (switched) *switched = 0;
  uint32_t mine = 0;
  if (int rc = tsem_take_fused_error(h, &mine)) return rc;
  if (mine && h->use_fused) {
    if (int rc = tsem_fallback_twopass(h)) return rc;
    if (switched) *switched = 1;
  }
  return TSEM_OK;
}

int tsem_em_update(tsem_ctx* h, double* diff_est) {
  if (!h || !h->have_model) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  h->pev_iter = h->opt_phase > 0 ? 0 : -1;
  if (int rc = phase_mark(h, 3)) return rc;                // (whatever moved the reduce buffer between the pass and this call is the "all-reduce" phase)
  if (int rc = launch_update(h, h->d_diffs)) { h->pev_iter = -1; return rc; }
  if (int rc = phase_mark(h, 4)) return rc;
  h->pev_iter = -1;
  h->first_pending = false;
  h->lag_valid = false;
  if (diff_est) {
    TSEM_HIP(hipMemcpyAsync(diff_est, h->d_diffs, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    TSEM_HIP(hipStreamSynchronize(h->stream));
    phase_harvest(h, 1);
    // negative: the (all-reduced) error flag was up, so no rank committed this iteration
    if (*diff_est < 0.0)
      TSEM_FAIL(TSEM_ERR_TIMEOUT, "EM pass: the hand-off watchdog of the fused kernel fired on some rank; parameters "
                                  "left untouched (tsem_fallback_twopass, then redo the pass)");
  }
  return TSEM_OK;
}

static int launch_lnl(tsem_ctx* h) {
  int na = 0, nu = 0;
  if (h->em_rows) {
    na = h->N ? std::min(2048, tsem_rowpass_grid(h)) : 0;
    if (na) k_lnl_rows_amb<<<na, 256, 0, h->stream>>>(h->N, h->d_indptr, h->d_indices, h->d_raw, h->d_lut, h->d_ctab_prev, h->d_ctab, h->d_lnl_part);
    TSEM_HIP(hipGetLastError());
  } else if (h->nb > 0 && h->use_fused && h->split) {
    // the previous parameters' row factors (unweighted) through HBM, then sum z log1p(Q c) over the two halves of every part's columns
    if (!h->d_fz_aux) {
      TSEM_ALLOC(h->d_fz_aux, 4);
      TSEM_HIP(hipMemsetAsync(h->d_fz_aux, 0, sizeof(uint32_t) * 4, h->stream));
    }
    if (int rc = launch_fused(h, 5, nullptr, 1)) return rc;
    if (int rc = launch_row_factors(h, 0)) return rc;
    if (int rc = launch_fused(h, 8, nullptr, 0)) return rc;
    if (int rc = launch_fused(h, 8, nullptr, 1)) return rc;
    na = 2 * h->fz_grid;
  } else if (h->nb > 0 && h->use_fused) {
    if (int rc = launch_fused(h, 1, nullptr)) return rc;
    na = h->fz_grid;
  } else if (h->nb > 0) {
    if (int rc = launch_phase1(h, h->d_ctab_prev)) return rc;
    const size_t lds2 = (size_t)(2 * h->Kp + h->R) * 8;
    na = h->G2 * h->P;
    k_phase2_lnl<1024><<<na, 1024, lds2, h->stream>>>(h->P, h->Kp, h->R, h->nb, h->G2, h->N_amb_pad, h->d_sb_off,
        h->d_pval, h->d_prc, h->d_ctab_prev, h->d_ctab, h->d_ypart, h->d_lnl_part);
    TSEM_HIP(hipGetLastError());
  }
  if (h->N_uni > 0) {
    nu = (int)std::min<int64_t>(2048, (h->N_uni + 255) / 256);
    k_lnl_unique<<<nu, 256, 0, h->stream>>>(h->N_uni, h->d_uni_col, h->d_uni_code, h->d_lut, h->d_pi_prev, h->d_pi,
                                           h->d_lnl_part + 4096);
    TSEM_HIP(hipGetLastError());
  }
  k_sum_parts<<<1, 256, 0, h->stream>>>(h->d_lnl_part, na, h->d_lnl_part + 4096, nu, h->d_red + h->K);
  TSEM_HIP(hipGetLastError());
  return TSEM_OK;
}

int tsem_lnl_pass(tsem_ctx* h) {
  if (!h || !h->have_model) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  return launch_lnl(h);
}

int tsem_read_reduce(tsem_ctx* h, double* out, int64_t offset, int64_t count) {
  if (!h || !h->have_model || !out || offset < 0 || offset + count > h->K + 2) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  TSEM_HIP(hipMemcpyAsync(out, h->d_red + offset, sizeof(double) * count, hipMemcpyDeviceToHost, h->stream));
  TSEM_HIP(hipStreamSynchronize(h->stream));
  return TSEM_OK;
}

static int ensure_ctl(tsem_ctx* h) {
  if (h->d_ctl) return TSEM_OK;
  TSEM_ALLOC(h->d_ctl, 8); TSEM_ALLOC(h->d_ctld, 8); TSEM_ALLOC(h->d_lnls, TS_DIFF_RING);
  TSEM_HIP(hipMemsetAsync(h->d_ctl, 0, 32, h->stream));
  TSEM_HIP(hipMemsetAsync(h->d_ctld, 0, 64, h->stream));
  return TSEM_OK;
}

// the lnl of the iteration just committed, all-reduced, into d_ctld[1] (value) / d_ctld[2] (error flag)
static int enqueue_lnl_reduce(tsem_ctx* h) {
  if (int rc = launch_lnl(h)) return rc;
  k_lnl_slots<<<1, 1, 0, h->stream>>>(h->d_red + h->K, (h->use_fused && h->nb > 0) ? h->d_xflags : nullptr, h->d_ctld + 1,
                                      (h->split && h->use_fused && h->nb > 0) ? h->d_fz_aux + 2 : nullptr);
  TSEM_HIP(hipGetLastError());
  if (tsem_comm_on(h)) { if (int rc = tsem_comm_allreduce_dev(h->comm, h->d_ctld + 1, 2, 0, h->stream, h->err)) return rc; }
  return TSEM_OK;
}

// flags: bit 0 = first chunk of a run (model.py:771-778, 786), bit 1 = last chunk of the run (flush the log-likelihood that is still
// owed).  With `use_likelihood` and a layout built for it (option "use_likelihood", lag_capable) the lnl of iteration t is summed by the
// EM pass of iteration t+1 (fused kernel MODE 4) and tested by that iteration's update kernel BEFORE it commits anything: a run that
// converges in iteration t ends with the state of iteration t, like the reference's.  The value of a chunk's LAST iteration is then
// not known when the chunk returns (lnls_out[n_done - 1] = NaN): it comes with the next chunk as *lnl_carry (which may be all that
// chunk does: n_done = 0, stopped = 1), or is flushed by the dedicated lnl pass when bit 1 is set / by tsem_em_flush_lnl.
int tsem_em_chunk(tsem_ctx* h, int32_t n_max, double epsilon, int32_t use_likelihood, int32_t flags,
                  int32_t* n_done, int32_t* stopped, double* diffs_out, double* lnls_out, double* lnl_carry) {
  if (!h || !h->have_model || n_max < 0 || n_max > TS_DIFF_RING) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  if (int rc = ensure_ctl(h)) return rc;
  const bool first = (flags & 1) != 0, flush = (flags & 2) != 0;
  const double nan = std::nan("");
  if (lnl_carry) *lnl_carry = nan;
  if (first) {
    // model.py:786 compares the first iteration's lnl with self.lnl as the previous run left it (inf on a fresh model,
    // model.py:683): tsem_set_prev_lnl / the end of tsem_em_run keep that value for the next run
    const double seed = h->lnl_prev_seed;
    TSEM_HIP(hipMemcpy(h->d_ctld, &seed, sizeof(double), hipMemcpyHostToDevice));
    if (!h->d_pi_first) { TSEM_ALLOC(h->d_pi_first, h->K); TSEM_ALLOC(h->d_theta_first, h->K); }
    h->first_pending = true;
    h->lag_valid = false;
    if (use_likelihood && tsem_comm_on(h)) {
      // Row-sharded: the ranks' sequences of collectives must agree, so the carried lnl is used only if EVERY rank can (a rank
      // without ambiguous rows, or one that fell back to the two-pass kernels, cannot) — one tiny max all-reduce per run
      int64_t cannot = lag_capable(h) ? 0 : 1;
      if (int rc = tsem_comm_allreduce_host(h->comm, &cannot, 1, 3)) TSEM_FAIL(rc, std::string("all-reduce (lnl scheme): ") + tsem_comm_last_error());
      h->lag_agreed = cannot == 0;
    }
  }
  if (!use_likelihood) h->lag_valid = false;
  // h->lag_valid: the iteration committed last still OWES its log-likelihood, and d_rinv / d_ctab_prev are what the next MODE 4 pass
  // needs to sum it
  double* const d_carry = h->d_ctld + 3;                   // ... of the iteration the PREVIOUS chunk committed last
  const bool owed_at_entry = h->lag_valid;
  int done = 0, retries = 0;
  bool stop = false, lnl_pending = false;
  // (an lnl pass that timed out in the LAST iteration of the chunk is redone before returning: its value is the
  // iteration's log-likelihood and may end the run — ADVICE r2)
  while ((done < n_max || lnl_pending) && !stop) {
    // enqueue everything that is left; the device stops itself
    TSEM_HIP(hipMemsetAsync(h->d_ctl, 0, 8, h->stream));
    const int base = done, want = n_max - done;
    const bool lag = use_likelihood && lag_capable(h) && (!tsem_comm_on(h) || h->lag_agreed);
    if (lnl_pending || (h->lag_valid && !lag)) {           // the lnl of the last committed iteration by the dedicated pass: its own pass
      if (int rc = enqueue_lnl_reduce(h)) return rc;       // timed out, or the lagged scheme cannot deliver it (time-out / fall-back in between)
      k_lnl_check<<<1, 1, 0, h->stream>>>(h->d_ctl, h->d_ctld, h->d_ctld + 1, epsilon, base > 0 ? h->d_lnls + base - 1 : d_carry);
      TSEM_HIP(hipGetLastError());
      h->lag_valid = false;
    }
    const bool owed_before = h->lag_valid;
    for (int i = 0; i < want; ++i) {
      h->pev_iter = (h->opt_phase > 0 && i < 256) ? i : -1;
      if (int rc = em_pass(h, lag)) { h->pev_iter = -1; return rc; }
      if (int rc = tsem_comm_allreduce_red(h, 0, h->K + 2)) { h->pev_iter = -1; return rc; }
      if (int rc = phase_mark(h, 3)) return rc;
      double* lag_slot = (lag && h->lag_valid) ? (base + i > 0 ? h->d_lnls + base + i - 1 : d_carry) : nullptr;
      if (int rc = launch_update(h, h->d_diffs + base + i, true, epsilon, use_likelihood, lag_slot)) { h->pev_iter = -1; return rc; }
      if (int rc = phase_mark(h, 4)) return rc;
      h->pev_iter = -1;
      h->first_pending = false;                            // (a failed first update is redone below with the flag restored)
      if (lag) h->lag_valid = true;
      else if (use_likelihood) {
        if (int rc = enqueue_lnl_reduce(h)) return rc;
        k_lnl_check<<<1, 1, 0, h->stream>>>(h->d_ctl, h->d_ctld, h->d_ctld + 1, epsilon, h->d_lnls + base + i);
        TSEM_HIP(hipGetLastError());
      }
    }
    uint32_t ctl[2] = {0, 0};
    TSEM_HIP(hipMemcpyAsync(ctl, h->d_ctl, 8, hipMemcpyDeviceToHost, h->stream));
    TSEM_HIP(hipStreamSynchronize(h->stream));
    phase_harvest(h, std::min(want, 256));
    done = base + (int)ctl[1];
    lnl_pending = false;
    if (ctl[0] == 1u) {                                    // converged (lagged scheme: in the iteration committed last, whose lnl is known now)
      stop = true; h->lag_valid = false;
      break;
    }
    if (ctl[0] == 2u || ctl[0] == 3u) {                    // some rank's fused pass timed out: nobody committed that step
      if (++retries > 3) TSEM_FAIL(TSEM_ERR_TIMEOUT, "EM pass: repeated hand-off time-outs");
      const bool owed = lag ? (ctl[1] > 0 || owed_before) : false;   // (before the layout may change under us)
      uint32_t mine = 0;
      if (int rc = tsem_take_fused_error(h, &mine)) return rc;
      if (mine) { if (int rc = tsem_fallback_twopass(h)) return rc; }
      if (first && done == 0 && ctl[0] == 2u) h->first_pending = true;
      // lagged scheme: the pass that failed has rewritten part of d_rinv, so the lnl of the last committed iteration — if it is
      // still owed — comes from the dedicated pass at the top of the loop, and the next EM pass starts a new lag
      h->lag_valid = false;
      h->lag_agreed = false;                               // (every rank sees the time-out: from here on all run the lnl pass per iteration)
      lnl_pending = ctl[0] == 3u || owed;
      continue;
    }
    if (h->use_fused && !tsem_comm_on(h)) {                     // belt and braces: an error word the flags did not carry.  (Row-sharded
      uint32_t mine = 0;                                   //  runs rely on slot K alone: failing on ONE rank would leave the others in the next collective.)
      if (int rc = tsem_take_fused_error(h, &mine)) return rc;
      if (mine) TSEM_FAIL(TSEM_ERR_TIMEOUT, "fused EM kernel: hand-off watchdog fired (code " + std::to_string(mine) + ")");
    }
  }
  if (!stop && flush && use_likelihood && h->lag_valid) {
    // end of the run without convergence so far: the last iteration's log-likelihood by the dedicated pass, and its test
    for (int attempt = 0;; ++attempt) {
      TSEM_HIP(hipMemsetAsync(h->d_ctl, 0, 8, h->stream));
      if (int rc = enqueue_lnl_reduce(h)) return rc;
      k_lnl_check<<<1, 1, 0, h->stream>>>(h->d_ctl, h->d_ctld, h->d_ctld + 1, epsilon, done > 0 ? h->d_lnls + done - 1 : d_carry);
      TSEM_HIP(hipGetLastError());
      uint32_t c0 = 0;
      TSEM_HIP(hipMemcpyAsync(&c0, h->d_ctl, 4, hipMemcpyDeviceToHost, h->stream));
      TSEM_HIP(hipStreamSynchronize(h->stream));
      if (c0 == 3u) {
        if (attempt >= 2) TSEM_FAIL(TSEM_ERR_TIMEOUT, "lnl pass: repeated hand-off time-outs");
        uint32_t mine = 0;
        if (int rc = tsem_take_fused_error(h, &mine)) return rc;
        if (mine) { if (int rc = tsem_fallback_twopass(h)) return rc; }
        continue;
      }
      stop = c0 == 1u;
      break;
    }
    h->lag_valid = false;
  }
  if (n_done) *n_done = done;
  if (stopped) *stopped = stop ? 1 : 0;
  if (done && diffs_out) TSEM_HIP(hipMemcpy(diffs_out, h->d_diffs, sizeof(double) * done, hipMemcpyDeviceToHost));
  if (done && lnls_out && use_likelihood) {
    TSEM_HIP(hipMemcpy(lnls_out, h->d_lnls, sizeof(double) * done, hipMemcpyDeviceToHost));
    if (h->lag_valid) lnls_out[done - 1] = nan;            // comes with the next chunk / the flush
  }
  if (lnl_carry && use_likelihood && owed_at_entry && (done > 0 || !h->lag_valid))
    TSEM_HIP(hipMemcpy(lnl_carry, d_carry, sizeof(double), hipMemcpyDeviceToHost));
  TSEM_HIP(hipMemsetAsync(h->d_ctl, 0, 8, h->stream));     // passes launched outside a chunk must not see a stale stop flag
  return TSEM_OK;
}

int tsem_em_steps(tsem_ctx* h, int32_t n, double* diffs_out) {
  int32_t done = 0;
  return tsem_em_chunk(h, n, 0.0, 0, 0, &done, nullptr, diffs_out, nullptr, nullptr);
}

// the final log-likelihood (model.py:800-801), all-reduced; redone on the two-pass kernels after a time-out
static int final_lnl(tsem_ctx* h, double* lnl) {
  if (int rc = ensure_ctl(h)) return rc;
  for (int attempt = 0;; ++attempt) {
    if (int rc = enqueue_lnl_reduce(h)) return rc;
    double v[2] = {0, 0};
    TSEM_HIP(hipMemcpyAsync(v, h->d_ctld + 1, 16, hipMemcpyDeviceToHost, h->stream));
    TSEM_HIP(hipStreamSynchronize(h->stream));
    if (v[1] > 0.0) {
      if (attempt >= 2) TSEM_FAIL(TSEM_ERR_TIMEOUT, "lnl pass: repeated hand-off time-outs");
      uint32_t mine = 0;
      if (int rc = tsem_take_fused_error(h, &mine)) return rc;
      if (mine) { if (int rc = tsem_fallback_twopass(h)) return rc; }
      continue;
    }
    *lnl = v[0];
    return TSEM_OK;
  }
}

int tsem_final_lnl(tsem_ctx* h, double* lnl) {
  if (!h || !h->have_model || !lnl) return TSEM_ERR_ARG;
  if (int rc = ensure_device(h)) return rc;
  return final_lnl(h, lnl);
}

int tsem_em_run(tsem_ctx* h, double epsilon, int32_t max_iter, int32_t use_likelihood, int32_t* n_iter,
                int32_t* converged, double* lnl_out, double* diffs, double* lnls, double* pi_init,
                double* theta_init) {
  if (!h || !h->have_model) return TSEM_ERR_ARG;
  int inum = 0;
  bool conv = false;
  double lnl = INFINITY;
  do {                                        // model.py:771-797: at least one iteration
    const int want = std::max(1, std::min(8, max_iter - inum));
    int32_t done = 0, stopped = 0;
    std::vector<double> d(want), l(want);
    double carry = 0.0;
    const int flags = (inum == 0 ? 1 : 0) | (inum + want >= max_iter ? 2 : 0);
    if (int rc = tsem_em_chunk(h, want, epsilon, use_likelihood, flags, &done, &stopped, d.data(), l.data(), &carry)) return rc;
    if (use_likelihood && carry == carry && inum > 0) {    // the lnl the previous chunk's last iteration owed
      lnl = carry;
      if (lnls && inum - 1 < std::max(1, max_iter)) lnls[inum - 1] = carry;
    }
    for (int i = 0; i < done; ++i) {
      if (diffs && inum + i < std::max(1, max_iter)) diffs[inum + i] = d[i];
      if (lnls && use_likelihood && inum + i < std::max(1, max_iter)) lnls[inum + i] = l[i];
      if (use_likelihood && l[i] == l[i]) lnl = l[i];
    }
    inum += done;
    conv = stopped != 0;
  } while (!conv && inum < max_iter);
  if (pi_init || theta_init) { if (int rc = tsem_get_params(h, TSEM_Z_FIRST, pi_init, theta_init)) return rc; }
  if (!use_likelihood) {                      // model.py:800-801
    if (int rc = final_lnl(h, &lnl)) return rc;
  }
  h->lnl_prev_seed = lnl;                     // what the next run's first lnl is compared with (model.py:786)
  if (n_iter) *n_iter = inum;
  if (converged) *converged = conv ? 1 : 0;
  if (lnl_out) *lnl_out = lnl;
  return TSEM_OK;
}

int tsem_set_prev_lnl(tsem_ctx* h, double lnl) {
  if (!h) return TSEM_ERR_ARG;
  h->lnl_prev_seed = lnl;
  return TSEM_OK;
}

}  // extern "C"
