// Primitive-throughput microbenchmarks for the EM kernel design (gfx950).
// Measures, on a synthetic (idx:int32, val:f64) stream of NNZ entries:
//   stream   : 12 B/nnz coalesced read only
//   gatherG  : + c[idx] gather from a K-entry f64 table in global memory
//   atomD    : + atomicAdd(f64) device scope into acc[K]
//   atomX    : + workgroup-scope atomic into per-XCD private acc copies
//   ldsA     : + ds_add_f64 into an LDS table of H entries (idx % H)
//   ldsG     : + LDS gather from an H-entry table
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics prim.hip -o prim
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__host__ __device__ inline uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__global__ void gen(int32_t* idx, double* val, int64_t n, int K, int zipf) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint64_t h = mix64(i * 0x9E3779B97F4A7C15ull + 12345);
    double u = (double)(h >> 11) * (1.0 / 9007199254740992.0);
    int j = zipf ? 1 + (int)((K - 1) * u * u * u) : 1 + (int)((K - 1) * u);
    idx[i] = j;
    val[i] = 1.0 + (double)(h & 1023) * 1e-3;
  }
}

constexpr int TPB = 256;
// each thread handles 4 consecutive entries per step (16 B idx load, 2x16B val loads)
template <int MODE>
__global__ __launch_bounds__(TPB) void k_stream(const int32_t* __restrict__ idx,
    const double* __restrict__ val, int64_t n, const double* __restrict__ c,
    double* __restrict__ acc, int K, int H, double* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* lds = reinterpret_cast<double*>(smem);
  if (MODE == 4 || MODE == 5 || MODE == 6) {
    for (int t = threadIdx.x; t < H; t += TPB) lds[t] = (MODE == 4) ? 0.0 : c[t % K];
    __syncthreads();
  }
  unsigned xcc = 0;
  if (MODE == 3) {
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7;
    acc += (size_t)xcc * K;
  }
  double s = 0.0;
  int64_t nq = n / 4;
  for (int64_t q = (int64_t)blockIdx.x * TPB + threadIdx.x; q < nq; q += (int64_t)gridDim.x * TPB) {
    int4 j = reinterpret_cast<const int4*>(idx)[q];
    double2 v0 = reinterpret_cast<const double2*>(val)[2 * q];
    double2 v1 = reinterpret_cast<const double2*>(val)[2 * q + 1];
    int jj[4] = {j.x, j.y, j.z, j.w};
    double vv[4] = {v0.x, v0.y, v1.x, v1.y};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (MODE == 0) s += vv[e] + jj[e];
      if (MODE == 1) s += vv[e] * c[jj[e]];
      if (MODE == 2) unsafeAtomicAdd(&acc[jj[e]], vv[e]);
      if (MODE == 3) __hip_atomic_fetch_add(&acc[jj[e]], vv[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (MODE == 4) __hip_atomic_fetch_add(&lds[jj[e] % H], vv[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (MODE == 5) s += vv[e] * lds[jj[e] % H];
      if (MODE == 6) { double cc = lds[jj[e] % H]; unsafeAtomicAdd(&acc[jj[e]], vv[e] * cc); }
      if (MODE == 7) { double cc = c[jj[e]]; unsafeAtomicAdd(&acc[jj[e]], vv[e] * cc); }
    }
  }
  if (MODE == 4) {
    __syncthreads();
    for (int t = threadIdx.x; t < H; t += TPB) s += lds[t];
  }
  if (s == 123.456) out[0] = s;
}

template <int MODE>
float run(const char* name, const int32_t* idx, const double* val, int64_t n, const double* c,
          double* acc, int K, int H, double* out, int grid, size_t lds_bytes, int reps) {
  CK(hipFuncSetAttribute((const void*)k_stream<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  k_stream<MODE><<<grid, TPB, lds_bytes>>>(idx, val, n, c, acc, K, H, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int r = 0; r < reps; ++r) k_stream<MODE><<<grid, TPB, lds_bytes>>>(idx, val, n, c, acc, K, H, out);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
  printf("%-34s grid=%5d lds=%6zu  %8.3f ms  %7.1f Gnnz/s  %7.1f GB/s(12B)\n", name, grid, lds_bytes, ms,
         n / ms * 1e-6, 12.0 * n / ms * 1e-6);
  fflush(stdout);
  return ms;
}

int main(int argc, char** argv) {
  int64_t n = (argc > 1) ? atoll(argv[1]) : (int64_t)1 << 28;
  int K = (argc > 2) ? atoi(argv[2]) : 30000;
  int32_t* idx; double *val, *c, *acc, *out;
  CK(hipMalloc(&idx, n * 4)); CK(hipMalloc(&val, n * 8));
  CK(hipMalloc(&c, K * 8)); CK(hipMalloc(&acc, (size_t)K * 8 * 8)); CK(hipMalloc(&out, 64));
  std::vector<double> hc(K, 0.5);
  CK(hipMemcpy(c, hc.data(), K * 8, hipMemcpyHostToDevice));
  CK(hipMemset(acc, 0, (size_t)K * 8 * 8));
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs=%d nnz=%lld K=%d\n", p.name, p.multiProcessorCount, (long long)n, K);
  for (int zipf = 0; zipf < 2; ++zipf) {
    gen<<<4096, 256>>>(idx, val, n, K, zipf);
    CK(hipDeviceSynchronize());
    printf("---- columns: %s ----\n", zipf ? "zipf u^3" : "uniform");
    for (int grid : {2048, 8192}) {
      run<0>("stream (12B/nnz)", idx, val, n, c, acc, K, 0, out, grid, 0, 5);
      run<1>("gather global c[idx]", idx, val, n, c, acc, K, 0, out, grid, 0, 3);
      run<2>("atomic f64 device scope", idx, val, n, c, acc, K, 0, out, grid, 0, 2);
      run<3>("atomic f64 wg scope XCD-private", idx, val, n, c, acc, K, 0, out, grid, 0, 2);
      run<7>("gather global + atomic device", idx, val, n, c, acc, K, 0, out, grid, 0, 2);
    }
    for (int H : {8192, 16384}) {
      int grid = 256 * (H == 8192 ? 2 : 1);
      char nm[64];
      snprintf(nm, 64, "LDS atomic ds_add_f64 H=%d", H);
      run<4>(nm, idx, val, n, c, acc, K, H, out, grid, (size_t)H * 8, 3);
      snprintf(nm, 64, "LDS gather H=%d", H);
      run<5>(nm, idx, val, n, c, acc, K, H, out, grid, (size_t)H * 8, 3);
      snprintf(nm, 64, "LDS gather + atomic device H=%d", H);
      run<6>(nm, idx, val, n, c, acc, K, H, out, grid, (size_t)H * 8, 2);
    }
  }
  return 0;
}
