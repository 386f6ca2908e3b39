// How fast can gfx950 read HBM?  Variants of a pure streaming read (sum into a register, one store per
// thread) over a 24 GB buffer: load width, loads in flight per thread, cache policy, workgroup->address map.
// Build: hipcc --offload-arch=gfx950 -O3 stream.hip -o stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP %s @%d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// MAP 0: grid-stride over 16-byte words (wave-contiguous 1 KB per instruction, WGs interleaved at 16 KB)
// MAP 1: every WG owns one contiguous chunk of the buffer
// MAP 2: persistent WGs (one per CU), chunks of CH bytes dealt round-robin (the fused kernel's pattern)
template <int UNROLL, int POLICY, int MAP>
__global__ __launch_bounds__(1024) void k_stream(const u32x4* __restrict__ p, int64_t n16, int64_t chunk16, uint32_t* out) {
  u32x4 acc = {0, 0, 0, 0};
  const int64_t tid = threadIdx.x, nt = blockDim.x;
  auto ld = [&](int64_t i) -> u32x4 {
    if (POLICY == 1) return __builtin_nontemporal_load(p + i);
    return p[i];
  };
  if (MAP == 0) {
    const int64_t stride = (int64_t)gridDim.x * nt;
    int64_t i = (int64_t)blockIdx.x * nt + tid;
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
      u32x4 v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) v[u] = ld(i + u * stride);
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) acc += v[u];
    }
  } else {
    for (int64_t c = blockIdx.x; c * chunk16 < n16; c += (MAP == 1 ? (int64_t)1 << 60 : gridDim.x)) {
      const int64_t b = (MAP == 1 ? (int64_t)blockIdx.x * chunk16 : c * chunk16), e = min(n16, b + chunk16);
      int64_t i = b + tid;
      for (; i + (UNROLL - 1) * nt < e; i += UNROLL * nt) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = ld(i + u * nt);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u];
      }
      if (MAP == 1) break;
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = acc.x;
}

template <int UNROLL, int POLICY, int MAP>
void run(const char* name, const u32x4* p, int64_t bytes, int grid, int64_t chunk_bytes, uint32_t* out) {
  const int64_t n16 = bytes / 16;
  int64_t chunk16 = chunk_bytes / 16;
  if (MAP == 1) chunk16 = (n16 + grid - 1) / grid;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  k_stream<UNROLL, POLICY, MAP><<<grid, 1024>>>(p, n16, chunk16, out);
  CK(hipDeviceSynchronize());
  float best = 1e9;
  for (int r = 0; r < 3; ++r) {
    CK(hipEventRecord(a));
    k_stream<UNROLL, POLICY, MAP><<<grid, 1024>>>(p, n16, chunk16, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
  }
  printf("%-58s grid %6d  %7.3f ms  %7.1f GB/s\n", name, grid, best, bytes / best / 1e6);
}

int main() {
  const int64_t bytes = 24ll << 30;
  u32x4* p; uint32_t* out;
  CK(hipMalloc(&p, bytes)); CK(hipMalloc(&out, 64));
  CK(hipMemset(p, 1, bytes));
  run<4, 0, 0>("grid-stride, 4 x 16 B in flight", p, bytes, 4096, 0, out);
  run<8, 0, 0>("grid-stride, 8 x 16 B in flight", p, bytes, 4096, 0, out);
  run<8, 1, 0>("grid-stride, 8 x 16 B, nt", p, bytes, 4096, 0, out);
  run<8, 1, 0>("grid-stride, 8 x 16 B, nt", p, bytes, 1024, 0, out);
  run<8, 1, 0>("grid-stride, 8 x 16 B, nt, 256 WGs", p, bytes, 256, 0, out);
  run<16, 1, 0>("grid-stride, 16 x 16 B, nt, 256 WGs", p, bytes, 256, 0, out);
  run<8, 1, 1>("one contiguous chunk per WG, 8 x 16 B, nt", p, bytes, 2048, 0, out);
  run<8, 1, 1>("one contiguous chunk per WG, 8 x 16 B, nt, 256 WGs", p, bytes, 256, 0, out);
  run<3, 1, 2>("persistent 256 WGs, 48 KB chunks round-robin, 3 x 16 B, nt", p, bytes, 256, 48 << 10, out);
  run<8, 1, 2>("persistent 256 WGs, 1 MB chunks round-robin, 8 x 16 B, nt", p, bytes, 256, 1 << 20, out);
  run<8, 1, 2>("persistent 512 WGs, 1 MB chunks round-robin, 8 x 16 B, nt", p, bytes, 512, 1 << 20, out);
  return 0;
}
