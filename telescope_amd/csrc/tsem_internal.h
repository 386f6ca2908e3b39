// Internal to libtelescope_em.so: what the translation units of the library share besides the context (tsem_common.h).
//
//   tsem_host.hip    handle, options, matrix load / synthetic generator, score table, instrumentation
//   tsem_setup.hip   row statistics, column signatures, the blocked layout (build_layout), model / parameter set-up
//   tsem_em.hip      EM pass (fused launch, two-pass kernels), column reduce, update, log-likelihood, the chunked loop
//   tsem_report.hip  CSR row passes: z export, best hits, reassign, the streaming report pass, per-barcode sums
//   tsem_comm.hip    collectives: RCCL resolved at run time, the in-process transport, the communicator ABI
//   tsem_csr.hip     csr_matrix_plus primitives on fp64 CSR, numpy's legacy random draw
//   tsem_fz_p*.hip   instantiations of the fused kernel (tsem_fused.h), one team size per unit
//
// A kernel is launched only from the unit that defines it; other units go through the host functions declared here.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <type_traits>

#include "tsem_common.h"
#include "tsem_device.h"
#include "tsem_fused.h"

constexpr int RS_SUB = 16, RP_SUB = 16;     // lanes per row of the 16-lane row passes (set-up / report)
extern std::string g_create_err;           // why the last tsem_create / stateless primitive failed (tsem_last_error(NULL))

// ---- small device helpers ---------------------------------------------------------------------------------------
template <int W>
__device__ __forceinline__ double sg_sum(double v) {
#pragma unroll
  for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, W);
  return v;
}
template <int W>
__device__ __forceinline__ double sg_max(double v) {
#pragma unroll
  for (int o = W / 2; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, W));
  return v;
}
template <int W>
__device__ __forceinline__ int sg_sum_i(int v) {
#pragma unroll
  for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, W);
  return v;
}
template <int W>
__device__ __forceinline__ int sg_max_i(int v) {
#pragma unroll
  for (int o = W / 2; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, W));
  return v;
}

// block-wide sum of one double per thread; result valid in thread 0
__device__ __forceinline__ double block_sum(double v, double* scratch /* >= 16 doubles */) {
  v = sg_sum<64>(v);
  int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) {
    int nw = (blockDim.x + 63) >> 6;
    for (int i = 0; i < nw; ++i) t += scratch[i];
  }
  return t;
}

// ---- host helpers -----------------------------------------------------------------------------------------------
template <typename T>
static int dalloc(tsem_ctx* h, T** p, size_t n) {
  if (*p) { (void)hipFree(*p); *p = nullptr; }
  if (n == 0) n = 1;
  hipError_t e = hipMalloc((void**)p, n * sizeof(T));
  if (e != hipSuccess) {
    h->err = std::string("hipMalloc(") + std::to_string(n * sizeof(T)) + " B): " + hipGetErrorString(e);
    return TSEM_ERR_NOMEM;
  }
  return TSEM_OK;
}
#define TSEM_ALLOC(ptr, n) do { int rc_ = dalloc(h, &(ptr), (size_t)(n)); if (rc_) return rc_; } while (0)

template <typename T>
static void dfree(T*& p) { if (p) { (void)hipFree(p); p = nullptr; } }

// Device memory freed when the scope ends: the temporaries of a call, on every return path.
struct DevTmp {
  void* p = nullptr;
  ~DevTmp() { if (p) (void)hipFree(p); }
  template <typename T> T* as() const { return static_cast<T*>(p); }
};
// ... and the same for a plain local pointer filled by TSEM_ALLOC / hipMalloc:  T* d_x = nullptr; TSEM_SCOPED(d_x);  — freed at scope
// exit on every return path (the manual hipFree at the end of a function is skipped by every early `return rc`)
template <typename T> struct DevScope {
  T*& p;
  explicit DevScope(T*& r) : p(r) {}
  ~DevScope() { if (p) { (void)hipFree(p); p = nullptr; } }
};
#define TSEM_SCOPED(ptr) DevScope<std::remove_pointer<decltype(ptr)>::type> tsem_scope_##ptr(ptr)
#define TSEM_TMP(tmp, bytes) do { if (hipMalloc(&(tmp).p, std::max<size_t>(1, (size_t)(bytes))) != hipSuccess) { \
    (tmp).p = nullptr; TSEM_FAIL(TSEM_ERR_NOMEM, "hipMalloc(" + std::to_string((size_t)(bytes)) + " B) failed"); } } while (0)

static inline int cdiv64(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// TSEM_TRACE=1: host wall clock of the set-up phases on stderr (each lap synchronises the stream; a diagnostic, not a product path)
struct PhaseTimer {
  bool on; hipStream_t s; std::chrono::steady_clock::time_point t;
  explicit PhaseTimer(hipStream_t st) : on(getenv("TSEM_TRACE") != nullptr), s(st), t(std::chrono::steady_clock::now()) {}
  void lap(const char* name) {
    if (!on) return;
    (void)hipStreamSynchronize(s);
    const auto n = std::chrono::steady_clock::now();
    fprintf(stderr, "[tsem] %-34s %8.3f ms\n", name, std::chrono::duration<double, std::milli>(n - t).count());
    t = n;
  }
};

static inline int ensure_device(tsem_ctx* h) {
  TSEM_HIP(hipSetDevice(h->device));
  return TSEM_OK;
}

// The fused kernel's instantiations (team size x mode x entry format x geometry: ~200 kernels, most of the library's build time)
// live in eight translation units compiled in parallel (tsem_fz_p1.hip ... tsem_fz_p8.hip, tsem_fused_inst.h); each exports one
// look-up function.  The host launches through the pointer.
typedef void (*fz_fn)(FusedArgs);
fz_fn tsem_fz_kernel_p1(int P, int mode, int fmt, int geo);
fz_fn tsem_fz_kernel_p2(int P, int mode, int fmt, int geo);
fz_fn tsem_fz_kernel_p3(int P, int mode, int fmt, int geo);
fz_fn tsem_fz_kernel_p4(int P, int mode, int fmt, int geo);
fz_fn tsem_fz_kernel_p5(int P, int mode, int fmt, int geo);
fz_fn tsem_fz_kernel_p6(int P, int mode, int fmt, int geo);
fz_fn tsem_fz_kernel_p7(int P, int mode, int fmt, int geo);
fz_fn tsem_fz_kernel_p8(int P, int mode, int fmt, int geo);
static inline int fz_fmt(const tsem_ctx* h) { return h->fmt_code ? 1 : (h->fmt_wcode ? 2 : 0); }
static inline fz_fn fz_kernel(int P, int mode, int fmt, int geo) {
  switch (P) {
    case 1: return tsem_fz_kernel_p1(P, mode, fmt, geo); case 2: return tsem_fz_kernel_p2(P, mode, fmt, geo);
    case 3: return tsem_fz_kernel_p3(P, mode, fmt, geo); case 4: return tsem_fz_kernel_p4(P, mode, fmt, geo);
    case 5: return tsem_fz_kernel_p5(P, mode, fmt, geo); case 6: return tsem_fz_kernel_p6(P, mode, fmt, geo);
    case 7: return tsem_fz_kernel_p7(P, mode, fmt, geo); case 8: return tsem_fz_kernel_p8(P, mode, fmt, geo);
    default: return nullptr;
  }
}

// code16 entry format: only with the fused kernel, and only while the score table is small enough to
// sit in LDS beside the column tables (uint16 scores allow 65536 entries; alignments give a few hundred).
// With the round-2 exchange (branch-free, partner loads after the combine for short rows) codes in row order win
// at every row length measured — 10 / 14 / 20 / 28 / 40 / 100 entries per row: 1.47 / 1.61 / 1.98 / 2.55 / 3.60 /
// 3.76 ms against 1.71 / 1.96 / 2.57 / 3.26 / 4.23 / 4.39 ms with fp64 entries (profiles/r02_sweep_short.txt).
static inline bool fz_wants_codes(const tsem_ctx* h) {
  return h->opt_format != 1 && h->lut_len > 0 && h->lut_len <= 2048;
}
static inline size_t fz_lds_bytes(const tsem_ctx* h, bool codes) {
  // (split layout: one table of Kp entries, or — the lnl pass — two of (Kp + 1) / 2)
  return (size_t)((h->split ? h->Kp + 2 : ((h->exact_single || h->lnl3) ? 3 : 2) * h->Kp) + (fz_yr(h->geo) + (h->lnl3 ? 4 : 2)) * h->R) * 8 + 192 + 512 + (codes ? (size_t)h->lut_len * 8 : 0) +
         std::max<size_t>(h->opt_reproducible ? (size_t)h->Kp * 2 + 16 : 0, FZ_LOGTAB * 16 + 16);   // (+ the slots' exponent table | the lnl pass's log table)
}

// ---- host functions shared between the units (defined in the unit named) ---------------------------------------------
extern "C" {
// tsem_host.hip
void tsem_free_layout(tsem_ctx* h);
void tsem_free_matrix(tsem_ctx* h);
// tsem_setup.hip
int tsem_choose_geometry(tsem_ctx* h);
int tsem_build_layout(tsem_ctx* h);
int tsem_bin_reset(tsem_ctx* h);                           // option "reproducible": the slots' bounds as a run finds them
int tsem_make_ctabs(tsem_ctx* h);                          // the permuted pi * theta tables (current and previous) from the parameters
void tsem_report_preload(void);                            // the same for the report unit (build_layout, behind the fill)
void tsem_setup_preload(void);                             // load the set-up unit's code object now (behind a kernel that is running anyway)
int tsem_ensure_indices(tsem_ctx* h);                      // the CSR column ids, rebuilt from the popularity ids if option "drop_csr_indices" freed them
// tsem_em.hip
int tsem_take_fused_error(tsem_ctx* h, uint32_t* word);
int tsem_twopass_attributes(tsem_ctx* h);                 // dynamic-LDS limits of the two-pass kernels
int tsem_sum_parts(tsem_ctx* h, const double* a, int na, const double* b, int nb, double* out);   // out[0] = sum a + sum b, fixed order
// tsem_report.hip
int tsem_rowpass_grid(tsem_ctx* h);
// tsem_comm.hip
bool tsem_comm_on(const tsem_ctx* h);
int tsem_comm_allreduce_dev(tsem_comm* c, void* buf, size_t count, int dtype, hipStream_t s, std::string& err);
int tsem_comm_allreduce_red(tsem_ctx* h, int64_t offset, int64_t count);
}  // extern "C"
