// Data-path ceiling of the fused EM design: persistent WGs stream 12 B entries into a ring of
// register sets (prefetch distance DL steps), phase 1 (LDS gather + ds_add_f64 into y[R]) at step i,
// phase 2 (LDS gather + ds_add_f64 into acc[Kp]) LAG steps later, one barrier per step.  No exchange.
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics ring.hip -o ring
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP %s @%d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__host__ __device__ inline uint64_t mix64(uint64_t z) { z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
__global__ void gen(uint32_t* rc, double* val, int64_t n, int R, int Kp) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t h = mix64(i);
    rc[i] = ((uint32_t)(h % R) << 16) | (uint32_t)((h >> 20) % Kp);
    val[i] = 1.0 + (double)(h & 1023) * 1e-3;
  }
}
__device__ __forceinline__ void lds_add(double* p, double v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

template <int Q> struct Set { uint4 rc[Q]; double2 v0[Q], v1[Q]; };

template <int NT, int Q, int NS, int DL, int LAG>
__global__ __launch_bounds__(NT) void k_ring(const uint32_t* __restrict__ rc, const double* __restrict__ val,
                                             int64_t nblocks, int R, int Kp, double* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* c = reinterpret_cast<double*>(smem);
  double* acc = c + Kp;
  double* y = acc + Kp;          // [4][R]
  double* s = y + 4 * R;         // [R]
  const int tid = threadIdx.x;
  for (int t = tid; t < Kp; t += NT) { c[t] = 0.5; acc[t] = 0.0; }
  for (int t = tid; t < 4 * R; t += NT) y[t] = 0.0;
  for (int t = tid; t < R; t += NT) s[t] = 1.0;
  __syncthreads();
  constexpr int BLK = NT * Q * 4;                 // entries per block
  Set<Q> ring[NS];
  const int64_t nb = (nblocks - blockIdx.x + gridDim.x - 1) / gridDim.x;   // my blocks: blockIdx.x + k*gridDim.x
  auto load = [&](Set<Q>& st, int64_t k) {
    if (k >= nb) return;
    const int64_t q0 = ((int64_t)(blockIdx.x + k * gridDim.x) * BLK) >> 2;
#pragma unroll
    for (int i = 0; i < Q; ++i) {
      int64_t q = q0 + (int64_t)i * NT + tid;
      st.rc[i] = reinterpret_cast<const uint4*>(rc)[q];
      st.v0[i] = reinterpret_cast<const double2*>(val)[2 * q];
      st.v1[i] = reinterpret_cast<const double2*>(val)[2 * q + 1];
    }
  };
  auto p1 = [&](Set<Q>& st, int64_t k) {
    if (k >= nb || k < 0) return;
    double* yb = y + (k & 3) * R;
#pragma unroll
    for (int i = 0; i < Q; ++i) {
      st.v0[i].x *= c[st.rc[i].x & 0xFFFF]; lds_add(&yb[st.rc[i].x >> 16], st.v0[i].x);
      st.v0[i].y *= c[st.rc[i].y & 0xFFFF]; lds_add(&yb[st.rc[i].y >> 16], st.v0[i].y);
      st.v1[i].x *= c[st.rc[i].z & 0xFFFF]; lds_add(&yb[st.rc[i].z >> 16], st.v1[i].x);
      st.v1[i].y *= c[st.rc[i].w & 0xFFFF]; lds_add(&yb[st.rc[i].w >> 16], st.v1[i].y);
    }
  };
  auto p2 = [&](Set<Q>& st, int64_t k) {
    if (k >= nb || k < 0) return;
#pragma unroll
    for (int i = 0; i < Q; ++i) {
      lds_add(&acc[st.rc[i].x & 0xFFFF], st.v0[i].x * s[st.rc[i].x >> 16]);
      lds_add(&acc[st.rc[i].y & 0xFFFF], st.v0[i].y * s[st.rc[i].y >> 16]);
      lds_add(&acc[st.rc[i].z & 0xFFFF], st.v1[i].x * s[st.rc[i].z >> 16]);
      lds_add(&acc[st.rc[i].w & 0xFFFF], st.v1[i].y * s[st.rc[i].w >> 16]);
    }
  };
  // prologue: loads for blocks 0..DL-1
#pragma unroll
  for (int k = 0; k < DL; ++k) load(ring[k % NS], k);
  // steps in groups of NS so ring indices are compile-time
  for (int64_t base = 0; base < nb + LAG; base += NS) {
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      const int64_t i = base + j;
      p2(ring[(j + NS - LAG) % NS], i - LAG);               // block i-LAG lives in set (i-LAG) % NS
      load(ring[(j + DL) % NS], i + DL);                     // requires DL + LAG == NS - ... see host check
      p1(ring[j], i);
      __syncthreads();
    }
  }
  double t = 0.0;
  for (int k = tid; k < Kp; k += NT) t += acc[k];
  for (int k = tid; k < 4 * R; k += NT) t += y[k];
  if (t == 123.456) out[0] = t;
}

template <int NT, int Q, int NS, int DL, int LAG>
void run(const uint32_t* rc, const double* val, int64_t n, int R, int Kp, double* out, int grid) {
  static_assert((DL + LAG) % NS == 0 || DL + LAG < NS, "ring too small");
  const int BLK = NT * Q * 4;
  int64_t nblocks = n / BLK;
  size_t lds = (size_t)(2 * Kp + 5 * R) * 8;
  CK(hipFuncSetAttribute((const void*)k_ring<NT, Q, NS, DL, LAG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  k_ring<NT, Q, NS, DL, LAG><<<grid, NT, lds>>>(rc, val, nblocks, R, Kp, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int r = 0; r < 3; ++r) k_ring<NT, Q, NS, DL, LAG><<<grid, NT, lds>>>(rc, val, nblocks, R, Kp, out);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 3;
  double bytes = 12.0 * nblocks * BLK;
  printf("NT=%4d Q=%d NS=%d DL=%d LAG=%d blk=%5d  %7.3f ms  %7.1f GB/s  %6.1f Gnnz/s\n", NT, Q, NS, DL, LAG, BLK, ms,
         bytes / ms * 1e-6, bytes / 12 / ms * 1e-6);
  fflush(stdout);
}

int main() {
  int64_t n = (int64_t)1 << 30;   // 1.07e9 entries = 12.9 GB
  int R = 640, Kp = 7500;
  uint32_t* rc; double *val, *out;
  CK(hipMalloc(&rc, n * 4)); CK(hipMalloc(&val, n * 8)); CK(hipMalloc(&out, 64));
  gen<<<8192, 256>>>(rc, val, n, R, Kp);
  CK(hipDeviceSynchronize());
  const int G = 256;
  for (int RR : {640, 320, 160}) {
    R = RR;
    gen<<<8192, 256>>>(rc, val, n, R, Kp);
    CK(hipDeviceSynchronize());
    printf("R=%d\n", R);
    run<896, 1, 6, 2, 4>(rc, val, n, R, Kp, out, G);
    run<896, 1, 7, 2, 5>(rc, val, n, R, Kp, out, G);
    run<1024, 1, 6, 2, 4>(rc, val, n, R, Kp, out, G);
  }
  return 0;
}
