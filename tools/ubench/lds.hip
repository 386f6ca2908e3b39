// LDS cost model on gfx950 for the EM kernel's access patterns: random fp64 gathers and
// ds_add_f64 over a table of H doubles, one 1024-thread workgroup per CU.
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics lds.hip -o lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP %s @%d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
__device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__device__ __forceinline__ void lds_add(double* p, double v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// MODE 0: 4 random gathers (ds_read_b64)          MODE 1: 4 random ds_add_f64
// MODE 2: ds_add_f64, `act`/64 lanes active        MODE 3: ds_add_f64, runs of `act` adjacent lanes share an address
// MODE 4: 4 gathers from a 300-entry table          MODE 5: 4 random ds_add_f32
// MODE 6: gathers, `act`/64 lanes active           MODE 7: 2 x ds_read2_b64-style (two addresses per instruction, via inline asm)
template <int MODE>
__global__ __launch_bounds__(1024) void k(int H, int iters, int act, double* out, unsigned long long* cyc) {
  extern __shared__ double tab[];
  float* tabf = reinterpret_cast<float*>(tab);
  const int tid = threadIdx.x, lane = tid & 63;
  for (int t = tid; t < H; t += 1024) tab[t] = 1.0;
  __syncthreads();
  uint32_t idx[4];
  for (int j = 0; j < 4; ++j) {
    uint32_t key = (MODE == 3) ? (uint32_t)((tid / act) * 4 + j) : (uint32_t)(tid * 4 + j);
    idx[j] = mix(key * 2654435761u + 12345u) % (uint32_t)(MODE == 4 ? 300 : H);
  }
  if (MODE == 8 || MODE == 9) {                      // conflict-free: the lanes of one instruction touch consecutive doubles
    for (int j = 0; j < 4; ++j) idx[j] = (uint32_t)(((tid >> 6) * 256 + j * 64 + (tid & 63)) % H);
  }
  if (MODE == 10) {                                  // random rows of 64 consecutive doubles per instruction, lanes shuffled inside
    for (int j = 0; j < 4; ++j) idx[j] = (uint32_t)((mix((tid >> 6) * 4 + j) % (H / 64)) * 64 + (mix(tid * 7 + j) & 63));
  }
  if (MODE == 11 || MODE == 12) {                    // bank class = lane % 16 (MODE 11) or lane % 32 (MODE 12); the row of the table is random
    const int md = MODE == 11 ? 16 : 32;
    for (int j = 0; j < 4; ++j) idx[j] = (uint32_t)(((mix(tid * 4 + j) % (H / 32)) * 32 + (lane % md)) % H);
  }
  if (MODE == 13) {                                  // 2 lanes of every 16 share a class, otherwise distinct
    for (int j = 0; j < 4; ++j) idx[j] = (uint32_t)(((mix(tid * 4 + j) % (H / 32)) * 32 + ((lane % 16) / 2) * 2 + 16 * ((lane / 16) & 1)) % H);
  }
  if (MODE == 14) {                                  // 16 lanes: distinct mod 32, but lanes l and l+8 collide mod 16
    for (int j = 0; j < 4; ++j) idx[j] = (uint32_t)(((mix(tid * 4 + j) % (H / 32)) * 32 + ((lane % 16) < 8 ? (lane % 8) : (lane % 8) + 16)) % H);
  }
  if (MODE == 15 || MODE == 16) {                    // gathers: class = lane % 32 (MODE 15) / lane % 16 (MODE 16: lanes l, l+16 collide), random row
    const int md = MODE == 15 ? 32 : 16;
    for (int j = 0; j < 4; ++j) idx[j] = (uint32_t)(((mix(tid * 4 + j) % (H / 32)) * 32 + (lane % md)) % H);
  }
  const bool on = (MODE == 2 || MODE == 6) ? ((mix(lane * 77u + 5u) & 63u) < (unsigned)act) : true;
  double a = 0.0;
  __syncthreads();
  unsigned long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0 || MODE == 4) {
      a += tab[idx[0]] + tab[idx[1]] + tab[idx[2]] + tab[idx[3]];
    } else if (MODE == 6) {
      if (on) a += tab[idx[0]] + tab[idx[1]] + tab[idx[2]] + tab[idx[3]];
    } else if (MODE == 9 || MODE == 15 || MODE == 16) {
      a += tab[idx[0]] + tab[idx[1]] + tab[idx[2]] + tab[idx[3]];
    } else if (MODE == 1 || MODE == 3 || MODE == 8 || MODE == 10 || MODE == 11 || MODE == 12 || MODE == 13 || MODE == 14) {
      lds_add(&tab[idx[0]], 1.0); lds_add(&tab[idx[1]], 1.0); lds_add(&tab[idx[2]], 1.0); lds_add(&tab[idx[3]], 1.0);
    } else if (MODE == 2) {
      if (on) { lds_add(&tab[idx[0]], 1.0); lds_add(&tab[idx[1]], 1.0); lds_add(&tab[idx[2]], 1.0); lds_add(&tab[idx[3]], 1.0); }
    } else if (MODE == 5) {
      __hip_atomic_fetch_add(&tabf[idx[0]], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(&tabf[idx[1]], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(&tabf[idx[2]], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(&tabf[idx[3]], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (MODE == 7) {
      // two independent 8-byte addresses per instruction: offsets are in units of 8 bytes relative to one base,
      // so pair (idx0, idx1) needs |idx1 - idx0| < 256 — emulate with a base and small deltas (cost model only)
      double2 r0 = *reinterpret_cast<double2*>(&tab[(idx[0] & ~1u)]);
      double2 r1 = *reinterpret_cast<double2*>(&tab[(idx[2] & ~1u)]);
      a += r0.x + r0.y + r1.x + r1.y;
    }
    // rotate the indices a little so the compiler cannot hoist anything
    if (MODE < 8) idx[0] = (idx[0] + 1 == (uint32_t)(MODE == 4 ? 300 : H)) ? 0 : idx[0] + 1;
    asm volatile("" : "+v"(idx[0]), "+v"(idx[1]), "+v"(idx[2]), "+v"(idx[3]));
  }
  __syncthreads();
  unsigned long long t1 = clock64();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
  if (a == 12345.678) out[0] = a;
}

template <int MODE> void run(const char* name, int H, int act) {
  double* out; unsigned long long* cyc;
  CK(hipMalloc(&out, 8)); CK(hipMalloc(&cyc, 8 * 256));
  const int iters = 2000;
  CK(hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
  k<MODE><<<256, 1024, (size_t)H * 8, 0>>>(H, iters, act, out, cyc);
  CK(hipDeviceSynchronize());
  unsigned long long c[256];
  CK(hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost));
  double m = 0; for (int i = 0; i < 256; ++i) m += (double)c[i]; m /= 256;
  // lane-ops per cycle per CU: 1024 threads * 4 ops * iters / cycles
  printf("%-44s H=%5d act=%2d  %8.1f cyc/iter  %6.2f lane-ops/clk/CU  (%5.1f clk per wave-instr)\n", name, H, act, m / iters,
         1024.0 * 4 * iters / m, m / iters / 64.0);
  CK(hipFree(out)); CK(hipFree(cyc));
}

int main() {
  run<0>("gather b64 random", 7500, 64);
  run<0>("gather b64 random", 384, 64);
  run<4>("gather b64 from 300-entry table", 7500, 64);
  run<1>("ds_add_f64 random", 7500, 64);
  run<1>("ds_add_f64 random", 384, 64);
  run<5>("ds_add_f32 random", 7500, 64);
  for (int a : {48, 32, 16, 8}) run<2>("ds_add_f64 random, lanes active", 7500, a);
  for (int a : {32, 16, 8}) run<6>("gather b64 random, lanes active", 7500, a);
  for (int a : {2, 3, 4, 8}) run<3>("ds_add_f64, runs of adjacent lanes same addr", 384, a);
  run<7>("gather b128 (2 x per thread) random", 7500, 64);
  run<8>("ds_add_f64 conflict-free (consecutive)", 7424, 64);
  run<9>("gather b64 conflict-free (consecutive)", 7424, 64);
  run<10>("ds_add_f64 64 random lanes within 64 doubles", 7424, 64);
  run<11>("ds_add_f64 class = lane % 16, random row", 7424, 64);
  run<12>("ds_add_f64 class = lane % 32, random row", 7424, 64);
  run<13>("ds_add_f64 pairs share a class in each 16 lanes", 7424, 64);
  run<14>("ds_add_f64 16 lanes distinct mod 32, pairs equal mod 16", 7424, 64);
  run<15>("gather b64 class = lane % 32, random row", 7424, 64);
  run<16>("gather b64 class = lane % 16 (l, l+16 collide)", 7424, 64);
  return 0;
}
