// libtelescope_em.so: csr_matrix_plus primitives on a caller's fp64 CSR (sparse_plus.py:26-129: norm, scale, binmax), stateless;
// numpy's legacy MT19937 bounded draw for choose_random (sparse_plus.py:140-154).
#include "tsem_internal.h"

// ---- csr_matrix_plus primitives on fp64 CSR --------------------------------
__global__ __launch_bounds__(256) void k_norm_rows(int64_t N, const int64_t* __restrict__ indptr,
                                                   const double* __restrict__ data, double* __restrict__ out) {
  const int sub = threadIdx.x / RP_SUB, lane = threadIdx.x % RP_SUB, subs = blockDim.x / RP_SUB;
  for (int64_t row = (int64_t)blockIdx.x * subs + sub; row < N; row += (int64_t)gridDim.x * subs) {
    int64_t s = indptr[row], e = indptr[row + 1];
    double y = 0.0;
    for (int64_t k = s + lane; k < e; k += RP_SUB) y += data[k];
    y = sg_sum<RP_SUB>(y);
    double r = recip0(y);
    for (int64_t k = s + lane; k < e; k += RP_SUB) out[k] = data[k] * r;
  }
}
__global__ __launch_bounds__(256) void k_binmax_rows(int64_t N, int32_t K, const int64_t* __restrict__ indptr,
                                                     const double* __restrict__ data, int8_t* __restrict__ out) {
  const int sub = threadIdx.x / RP_SUB, lane = threadIdx.x % RP_SUB, subs = blockDim.x / RP_SUB;
  for (int64_t row = (int64_t)blockIdx.x * subs + sub; row < N; row += (int64_t)gridDim.x * subs) {
    int64_t s = indptr[row], e = indptr[row + 1];
    bool any = false;
    double m = 0.0;
    for (int64_t k = s + lane; k < e; k += RP_SUB) { m = any ? fmax(m, data[k]) : data[k]; any = true; }
    // combine: lanes without entries must not contribute
    double mm = any ? m : -INFINITY;
    mm = sg_max<RP_SUB>(mm);
    if ((e - s) < K) mm = fmax(mm, 0.0);  // implicit zeros take part in max(1)
    for (int64_t k = s + lane; k < e; k += RP_SUB) out[k] = (data[k] == mm) ? 1 : 0;
  }
}

// whole-matrix reductions for csr_matrix_plus.norm() / scale() (sparse_plus.py:46-48, 93-95)
__global__ __launch_bounds__(256) void k_reduce_all(const double* __restrict__ v, int64_t n, int want_max,
                                                    double* __restrict__ part) {
  __shared__ double scratch[16];
  double acc = want_max ? -INFINITY : 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    acc = want_max ? fmax(acc, v[i]) : acc + v[i];
  if (want_max) {
    acc = sg_max<64>(acc);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { double m = scratch[0]; for (int i = 1; i < (int)(blockDim.x >> 6); ++i) m = fmax(m, scratch[i]); part[blockIdx.x] = m; }
  } else {
    double t = block_sum(acc, scratch);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
  }
}
__global__ void k_scale_all(const double* __restrict__ v, int64_t n, double f, double* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = v[i] * f;
}
__global__ __launch_bounds__(256) void k_scale_rows(int64_t N, int32_t K, const int64_t* __restrict__ indptr,
                                                    const double* __restrict__ data, double* __restrict__ out) {
  const int sub = threadIdx.x / RP_SUB, lane = threadIdx.x % RP_SUB, subs = blockDim.x / RP_SUB;
  for (int64_t row = (int64_t)blockIdx.x * subs + sub; row < N; row += (int64_t)gridDim.x * subs) {
    int64_t s = indptr[row], e = indptr[row + 1];
    double m = -INFINITY;
    for (int64_t k = s + lane; k < e; k += RP_SUB) m = fmax(m, data[k]);
    m = sg_max<RP_SUB>(m);
    if ((e - s) < K) m = fmax(m, 0.0);                 // implicit zeros take part in max(1)
    double r = recip0(m);
    for (int64_t k = s + lane; k < e; k += RP_SUB) out[k] = data[k] * r;
  }
}

extern "C" {

// ---------------------------------------------------------------------------
// numpy's LEGACY random stream for `choose` (sparse_plus.py:140-154)
// ---------------------------------------------------------------------------
// choose_random draws one np.random.choice per row with several best hits, on numpy's global legacy RandomState — the
// stream `telescope assign` seeds (telescope_assign.py:429-431).  One draw below a bound c is, in numpy's C
// (legacy-distributions / _bounded_integers, masked rejection): mask = the smallest 2^b - 1 >= c - 1, then 32-bit
// Mersenne-Twister outputs until (output & mask) <= c - 1.  np.random.randint(0, counts) on an array does exactly that
// per element (46 ms for the 5.6e6 tied rows of the 50M-row benchmark: it was the largest item of the whole report);
// this is the same loop in C on the caller's MT19937 state (np.random.get_state() -> here -> np.random.set_state()),
// bit for bit the same picks and the same state afterwards (tests/test_host_logic.py).  Host code: the stream is
// sequential by definition.
int tsem_legacy_randint(uint32_t* key624, int32_t* pos, const int32_t* counts, int64_t n, int32_t* out) {
  if (!key624 || !pos || (!counts && n) || (!out && n) || n < 0 || *pos < 0 || *pos > 624) return TSEM_ERR_ARG;
  constexpr int NN = 624, MM = 397;
  constexpr uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MATRIX_A = 0x9908b0dfu;
  uint32_t* mt = key624;
  int p = *pos;
  // the 624 outputs of the current state, tempered in one vectorisable sweep (the draw loop then only masks and compares:
  // 2.0 -> ~1 ns per draw; the state array itself stays untempered, as numpy keeps it)
  uint32_t buf[NN];
  auto temper_all = [&]() {
    for (int kk = 0; kk < NN; ++kk) {
      uint32_t y = mt[kk];
      y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
      buf[kk] = y;
    }
  };
  auto refill = [&]() {
    int kk = 0;
    for (; kk < NN - MM; ++kk) { const uint32_t y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER); mt[kk] = mt[kk + MM] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u); }
    for (; kk < NN - 1; ++kk) { const uint32_t y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER); mt[kk] = mt[kk + (MM - NN)] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u); }
    const uint32_t y = (mt[NN - 1] & UPPER) | (mt[0] & LOWER);
    mt[NN - 1] = mt[MM - 1] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u);
    p = 0;
    temper_all();
  };
  if (p < NN) temper_all();
  for (int64_t i = 0; i < n; ++i) {
    const int32_t c = counts[i];
    if (c <= 0) return TSEM_ERR_ARG;                       // (numpy raises "low >= high")
    const uint32_t rng = (uint32_t)c - 1u;
    if (rng == 0) { out[i] = 0; continue; }                // no random number is consumed
    const uint32_t mask = 0xFFFFFFFFu >> __builtin_clz(rng);   // the smallest 2^b - 1 >= rng
    uint32_t v;
    do {
      if (p == NN) refill();
      v = buf[p++] & mask;
    } while (v > rng);
    out[i] = (int32_t)v;
  }
  *pos = p;
  return TSEM_OK;
}

// ---------------------------------------------------------------------------
// csr_matrix_plus primitives (stateless)
// ---------------------------------------------------------------------------
static int csr_prim(int device, int64_t n_rows, int32_t n_cols, const int64_t* indptr, const double* data,
                    double* out_d, int8_t* out_b) {
  if (hipSetDevice(device) != hipSuccess) { g_create_err = "hipSetDevice failed (no usable HIP device)"; return TSEM_ERR_HIP; }
  if (n_rows < 0 || !indptr) return TSEM_ERR_ARG;
  int64_t nnz = indptr[n_rows];
  int64_t* d_ip = nullptr; double *d_in = nullptr, *d_od = nullptr; int8_t* d_ob = nullptr;
  bool ok = hipMalloc((void**)&d_ip, sizeof(int64_t) * (n_rows + 1)) == hipSuccess &&
            hipMalloc((void**)&d_in, sizeof(double) * std::max<int64_t>(1, nnz)) == hipSuccess;
  if (ok && out_d) ok = hipMalloc((void**)&d_od, sizeof(double) * std::max<int64_t>(1, nnz)) == hipSuccess;
  if (ok && out_b) ok = hipMalloc((void**)&d_ob, std::max<int64_t>(1, nnz)) == hipSuccess;
  int rc = TSEM_OK;
  if (!ok) { g_create_err = "hipMalloc failed"; rc = TSEM_ERR_NOMEM; }
  if (ok) {
    (void)hipMemcpy(d_ip, indptr, sizeof(int64_t) * (n_rows + 1), hipMemcpyHostToDevice);
    if (nnz) (void)hipMemcpy(d_in, data, sizeof(double) * nnz, hipMemcpyHostToDevice);
    int grid = (int)std::min<int64_t>(8192, std::max<int64_t>(1, (n_rows + 15) / 16));
    if (n_rows) {
      if (out_d) k_norm_rows<<<grid, 256>>>(n_rows, d_ip, d_in, d_od);
      if (out_b) k_binmax_rows<<<grid, 256>>>(n_rows, n_cols, d_ip, d_in, d_ob);
    }
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { g_create_err = hipGetErrorString(e); rc = TSEM_ERR_HIP; }
    if (rc == TSEM_OK && nnz) {
      if (out_d) (void)hipMemcpy(out_d, d_od, sizeof(double) * nnz, hipMemcpyDeviceToHost);
      if (out_b) (void)hipMemcpy(out_b, d_ob, nnz, hipMemcpyDeviceToHost);
    }
  }
  if (d_ip) (void)hipFree(d_ip);
  if (d_in) (void)hipFree(d_in);
  if (d_od) (void)hipFree(d_od);
  if (d_ob) (void)hipFree(d_ob);
  return rc;
}

// mode 0: out = data * (1 / sum(data))   norm()   sparse_plus.py:47-48
// mode 1: out = data * (1 / max(data))   scale()  sparse_plus.py:94-95   (max over the matrix incl. implicit zeros)
// mode 2: out = data * recip0(row max)   scale(1) sparse_plus.py:96-97
int tsem_csr_scale(int device, int mode, int64_t n_rows, int32_t n_cols, const int64_t* indptr, const double* data,
                   double* out) {
  if (hipSetDevice(device) != hipSuccess) { g_create_err = "hipSetDevice failed (no usable HIP device)"; return TSEM_ERR_HIP; }
  if (n_rows < 0 || !indptr || mode < 0 || mode > 2) return TSEM_ERR_ARG;
  const int64_t nnz = indptr[n_rows];
  int64_t* d_ip = nullptr; double *d_in = nullptr, *d_out = nullptr, *d_part = nullptr;
  const int G = 512;
  bool ok = hipMalloc((void**)&d_ip, sizeof(int64_t) * (n_rows + 1)) == hipSuccess &&
            hipMalloc((void**)&d_in, sizeof(double) * std::max<int64_t>(1, nnz)) == hipSuccess &&
            hipMalloc((void**)&d_out, sizeof(double) * std::max<int64_t>(1, nnz)) == hipSuccess &&
            hipMalloc((void**)&d_part, sizeof(double) * G) == hipSuccess;
  int rc = TSEM_OK;
  if (!ok) { g_create_err = "hipMalloc failed"; rc = TSEM_ERR_NOMEM; }
  if (ok) {
    (void)hipMemcpy(d_ip, indptr, sizeof(int64_t) * (n_rows + 1), hipMemcpyHostToDevice);
    if (nnz) (void)hipMemcpy(d_in, data, sizeof(double) * nnz, hipMemcpyHostToDevice);
    if (mode == 2) {
      int grid = (int)std::min<int64_t>(8192, std::max<int64_t>(1, (n_rows + 15) / 16));
      if (n_rows) k_scale_rows<<<grid, 256>>>(n_rows, n_cols, d_ip, d_in, d_out);
    } else if (nnz) {
      k_reduce_all<<<G, 256>>>(d_in, nnz, mode == 1, d_part);
      std::vector<double> part(G);
      (void)hipMemcpy(part.data(), d_part, sizeof(double) * G, hipMemcpyDeviceToHost);
      double r = mode == 1 ? -INFINITY : 0.0;
      for (int i = 0; i < G; ++i) r = mode == 1 ? std::max(r, part[i]) : r + part[i];
      if (mode == 1 && nnz < n_rows * (int64_t)n_cols) r = std::max(r, 0.0);
      k_scale_all<<<cdiv64(nnz, 256), 256>>>(d_in, nnz, 1.0 / r, d_out);
    }
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { g_create_err = hipGetErrorString(e); rc = TSEM_ERR_HIP; }
    if (rc == TSEM_OK && nnz) (void)hipMemcpy(out, d_out, sizeof(double) * nnz, hipMemcpyDeviceToHost);
  }
  if (d_ip) (void)hipFree(d_ip);
  if (d_in) (void)hipFree(d_in);
  if (d_out) (void)hipFree(d_out);
  if (d_part) (void)hipFree(d_part);
  return rc;
}

int tsem_csr_norm_rows(int device, int64_t n_rows, const int64_t* indptr, const double* data, double* out) {
  return csr_prim(device, n_rows, 0, indptr, data, out, nullptr);
}
int tsem_csr_binmax_rows(int device, int64_t n_rows, int32_t n_cols, const int64_t* indptr, const double* data,
                         int8_t* out) {
  return csr_prim(device, n_rows, n_cols, indptr, data, nullptr, out);
}


}  // extern "C"
