// Fused single-pass EM kernel (E-step row sums + M-step column scatter in ONE
// read of the matrix) for the column-partitioned blocked-COO layout.
//
// Why it looks like this (measurements: profiles/r01_primitives_ubench.log,
// profiles/r01_fused_timeline.txt; reasoning: DESIGN.md section 4):
//   * per-entry gathers of pi*theta and scatter-adds of the column sums only
//     keep up with the HBM stream when they hit LDS (global fp64 atomics: 22 G/s,
//     LDS: >500 G/s); K*16 B does not fit one CU's 160 KB, so columns are split
//     in P parts and a TEAM of P workgroups (one per CU, each owning one part's
//     tables) walks the same row blocks;
//   * a row's normaliser needs all P partial sums -> the team exchanges R partial
//     sums per block.  Teams are formed inside the launch from the hardware
//     XCC_ID so that all members share one XCD and the exchange is served by that
//     XCD's L2 (plain stores stay in L2; agent-scope/sc1 loads bypass the reader's
//     L1) instead of crossing the fabric.  Values travel as 8-byte tagged granules
//     (mantissa LSB = epoch parity): no flag, no drain, no fence;
//   * each member keeps its sub-block's numerators in REGISTERS between the two
//     phases and prefetches two blocks ahead, so every stored entry is read from
//     HBM exactly once;
//   * the last FZ_NX waves of each workgroup are exchange waves, the others hold
//     the data.  A CU's vector-memory path returns data in order, so an exchange
//     load issued behind a 60 KB prefetch burst waits for all of it (~7 us
//     measured): the schedule lets the exchange waves read y from a quiet LDS
//     (barriers A..B), then publish and load BEFORE the data waves issue the
//     next burst.
// No assumption is made about dispatch order or block->XCD placement: teams,
// their size and the block round-robin all derive from tickets taken at run
// time; every wait is bounded and reports through an error word.
#pragma once

constexpr int FZ_NT = 1024;            // threads per workgroup
constexpr int FZ_NX = 2;               // exchange waves (the last FZ_NX waves)
constexpr int FZ_DW = 16 - FZ_NX;      // data waves
constexpr int FZ_DT = FZ_DW * 64;      // data threads
constexpr int FZ_QUADS = 2;            // 16-B index loads per data thread per block
constexpr int FZ_CAP = FZ_DT * FZ_QUADS * 4;   // register-resident entries per sub-block
constexpr int FZ_RB = 5;               // rows per lane per combine batch
constexpr unsigned FZ_SPIN_LIMIT = 2000000u;
constexpr int FZ_PROF_SLOTS = 16;

// sync words (uint32): [0..7] per-XCD tickets, [8] registered WGs, [9] error
constexpr int FZ_SYNC_WORDS = 16;

struct FusedArgs {
  int P, Kp, R;
  int64_t nb, N_amb_pad;
  const int64_t* sb_off;
  const double* pval;
  const uint32_t* prc;
  const double* ctab;
  const double* wrow;   // [N_amb_pad] fragment weight w_i = max_j Q_ij (0 in the padding)
  double* partial;      // [team slots][P*Kp], zero-filled before launch
  double* xchg;         // [team][2][P][R] tagged granules, zero-filled before launch
  uint32_t* sync;       // zero-filled before launch
  int xcd_local;        // 1: plain stores (stay in the XCD's L2); 0: write-through
  int poll_delay;       // s_sleep(16) units between publish and the first partner load
  int dbg;              // bit0: skip partner loads (timing experiments only; wrong results)
                        // bit1: no prefetch (load block i+1 just before its P1)
  unsigned long long* prof;   // optional per-block timestamps of team 0 / member 0
  int prof_blocks;
};

__device__ __forceinline__ uint32_t fz_ld_u32(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // sc1: bypasses L1
}

struct FzRegs {
  uint4 rc[FZ_QUADS];
  double2 v0[FZ_QUADS], v1[FZ_QUADS];
};

__device__ __forceinline__ void fz_load(FzRegs& r, const FusedArgs& A, int64_t q0, int64_t q1, int tid) {
#pragma unroll
  for (int i = 0; i < FZ_QUADS; ++i) {
    int64_t q = q0 + (int64_t)i * FZ_DT + tid;
    if (q < q1) {
      r.rc[i] = reinterpret_cast<const uint4*>(A.prc)[q];
      r.v0[i] = reinterpret_cast<const double2*>(A.pval)[2 * q];
      r.v1[i] = reinterpret_cast<const double2*>(A.pval)[2 * q + 1];
    } else {
      r.rc[i] = make_uint4(0, 0, 0, 0);
      r.v0[i] = make_double2(0.0, 0.0);
      r.v1[i] = make_double2(0.0, 0.0);
    }
  }
}

template <int PT>
__global__ __launch_bounds__(FZ_NT) void k_em_fused(FusedArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int P = PT;
  const int Kp = A.Kp, R = A.R;
  double* c = reinterpret_cast<double*>(smem);
  double* acc = c + Kp;
  double* y = acc + Kp;                            // y[2][R]  partial row sums, double-buffered
  double* s = y + 2 * R;                           // s[2][R]  w_i / rowsum_i, double-buffered
  int* ibox = reinterpret_cast<int*>(s + 2 * R);   // [0]=ticket [1..8]=xcd counts
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  uint32_t* const sync = A.sync;
  uint32_t* const err = sync + 9;

  // ---- team formation from the hardware XCC id --------------------------------
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 7u;
  if (tid == 0) {
    ibox[0] = (int)atomicAdd(&sync[xcc], 1u);
    __threadfence();
    atomicAdd(&sync[8], 1u);
    unsigned spins = 0;
    while (fz_ld_u32(&sync[8]) < gridDim.x) {          // every workgroup has taken its ticket
      __builtin_amdgcn_s_sleep(8);
      if (++spins > FZ_SPIN_LIMIT) { atomicOr(err, 1u); break; }
      if (fz_ld_u32(err)) break;
    }
    for (int x = 0; x < 8; ++x) ibox[1 + x] = (int)fz_ld_u32(&sync[x]);
  }
  for (int t = tid; t < Kp; t += FZ_NT) acc[t] = 0.0;
  for (int t = tid; t < 2 * R; t += FZ_NT) { y[t] = 0.0; s[t] = 0.0; }
  __syncthreads();
  const int ticket = ibox[0];
  const int u = ticket / P, p = ticket % P;
  int T = 0, tbase = 0;
  for (int x = 0; x < 8; ++x) {
    int teams = ibox[1 + x] / P;
    if (x < (int)xcc) tbase += teams;
    T += teams;
  }
  const bool valid = (u + 1) * P <= ibox[1 + xcc] && fz_ld_u32(err) == 0;
  if (!valid || T == 0) return;                          // leftover workgroup of an incomplete team
  const int team = tbase + u;                            // 0..T-1
  for (int t = tid; t < Kp; t += FZ_NT) c[t] = A.ctab[p * Kp + t];
  double* const xbase = A.xchg + (int64_t)team * 2 * P * R;
  __syncthreads();

  // Software pipeline per team member; block j of this team is team + j*T.
  // Barrier pairs A(j)/B(j), j = 0..nblk+1, are executed by every wave.
  //   data waves : prologue load(0) load(1) P1(0) [A0][B0];
  //                step i = 0..nblk: P2(i-1) | load(i+2) | P1(i+1) | [A(i+1)][B(i+1)]
  //   exchange   : block j = 0..nblk-1: [A(j)] read y(j) [B(j)] publish, load partners,
  //                combine -> s(j), clear y(j);  then the remaining barrier pairs.
  // So the exchange of block j overlaps the data waves' step j (scatter of block j-1,
  // prefetch burst of block j+2, row sums of block j+1) and its loads enter the CU's
  // in-order memory pipe ahead of that burst.
  const int64_t nblk = (A.nb > team) ? (A.nb - team + T - 1) / T : 0;
  if (wave >= FZ_DW) {
    // =========================== exchange waves ==============================
    __builtin_amdgcn_s_setprio(3);
    const int xw = wave - FZ_DW;
    const int rows_x = (R + FZ_NX - 1) / FZ_NX;          // rows owned by this exchange wave
    const int rlo = xw * rows_x, rhi = min(R, rlo + rows_x);
    for (int64_t j = 0; j < nblk; ++j) {
      const int64_t b = team + j * T;
      const unsigned seq = (unsigned)j;
      double* yb = y + (seq & 1) * R;
      double* sb = s + (seq & 1) * R;
      const bool pr = A.prof && team == 0 && p == 0 && xw == 0 && lane == 0 && (int)seq < A.prof_blocks;
      // Tagged granules: each partial row sum travels as ONE naturally aligned 8-byte
      // store whose mantissa LSB carries the epoch parity of its slot (slot = seq & 1 is
      // rewritten every other block, so the expected tag alternates; the buffer is
      // zeroed before the launch and the first tag is 1).  A reader that sees the
      // expected tag has the value.  Every member (the owner included) sums the SAME
      // tag-stripped values in the same order, so all members compute identical s.
      const unsigned long long tag = (unsigned long long)(((seq >> 1) & 1u) ^ 1u);
      unsigned long long* mine = reinterpret_cast<unsigned long long*>(xbase + ((int64_t)(seq & 1) * P + p) * R);
      __syncthreads();                                   // barrier A(j): y(j) complete, LDS quiet
      if (pr) A.prof[seq * FZ_PROF_SLOTS + 2] = clock64();
      bool passed_b = false;
      for (int r0 = rlo; r0 < rhi; r0 += 64 * FZ_RB) {
        unsigned long long own[FZ_RB], part[P > 1 ? P - 1 : 1][FZ_RB];
        double wv[FZ_RB];
#pragma unroll
        for (int k = 0; k < FZ_RB; ++k) {
          const int r = r0 + k * 64 + lane;
          own[k] = (r < rhi) ? (unsigned long long)__double_as_longlong(yb[r]) : 0ull;
        }
        if (!passed_b) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __syncthreads();                               // barrier B(j): data waves go on
          passed_b = true;
        }
        if (pr && r0 == rlo) A.prof[seq * FZ_PROF_SLOTS + 8] = clock64();
        if (P > 1) {
#pragma unroll
          for (int k = 0; k < FZ_RB; ++k) {
            const int r = r0 + k * 64 + lane;
            own[k] = (own[k] & ~1ull) | tag;
            if (r < rhi) {
              if (A.xcd_local) __hip_atomic_store(&mine[r], own[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              else __hip_atomic_store(&mine[r], own[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
        }
#pragma unroll
        for (int k = 0; k < FZ_RB; ++k) {
          const int r = r0 + k * 64 + lane;
          wv[k] = (r < rhi) ? A.wrow[b * R + r] : 0.0;
        }
        if (pr && r0 == rlo) A.prof[seq * FZ_PROF_SLOTS + 3] = clock64();
        unsigned spins = 0;
        if (P > 1 && !(A.dbg & 1)) {
          for (int d = 0; d < A.poll_delay; ++d) __builtin_amdgcn_s_sleep(16);
          for (;;) {
            bool ok = true;
#pragma unroll
            for (int k = 0; k < FZ_RB; ++k) {
              const int r = r0 + k * 64 + lane;
#pragma unroll
              for (int q = 0; q < P; ++q) {
                if (q == p) continue;
                const int qi = q < p ? q : q - 1;
                if (r < rhi) {
                  const unsigned long long* src =
                      reinterpret_cast<const unsigned long long*>(xbase + ((int64_t)(seq & 1) * P + q) * R) + r;
                  part[qi][k] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  ok &= (part[qi][k] & 1ull) == tag;
                } else {
                  part[qi][k] = 0;
                }
              }
            }
            if (pr && r0 == rlo && spins == 0) A.prof[seq * FZ_PROF_SLOTS + 10] = clock64();
            if (__all(ok)) break;
            if (++spins > FZ_SPIN_LIMIT) { if (lane == 0) atomicOr(err, 2u); break; }
            if ((spins & 255u) == 0 && fz_ld_u32(err)) break;
          }
        }
        if (pr && r0 == rlo) { A.prof[seq * FZ_PROF_SLOTS + 4] = clock64(); A.prof[seq * FZ_PROF_SLOTS + 11] = spins; }
#pragma unroll
        for (int k = 0; k < FZ_RB; ++k) {
          const int r = r0 + k * 64 + lane;
          if (r < rhi) {
            double ys = 0.0;
#pragma unroll
            for (int q = 0; q < P; ++q) {                 // fixed order: every member computes the same bits
              unsigned long long g = (q == p || (A.dbg & 1)) ? own[k] : part[q < p ? q : (q > 0 ? q - 1 : 0)][k];
              ys += __longlong_as_double((long long)(P > 1 ? (g & ~1ull) : g));
            }
            // z = n * recip0(rowsum) (sparse_plus.py:52), weighted by w_i (model.py:730)
            sb[r] = recip0(ys) * wv[k];
            yb[r] = 0.0;
          }
        }
      }
      if (!passed_b) __syncthreads();                    // a wave without rows still joins barrier B(j)
      if (pr) A.prof[seq * FZ_PROF_SLOTS + 5] = clock64();
    }
    __syncthreads(); __syncthreads();                    // A(nblk),   B(nblk)
    __syncthreads(); __syncthreads();                    // A(nblk+1), B(nblk+1)
  } else {
    // ============================ data waves =================================
    FzRegs r0, r1, r2;
    auto load_blk = [&](FzRegs& rr, int64_t i) {
      if (i < nblk) {
        const int64_t b = team + i * T;
        fz_load(rr, A, A.sb_off[b * P + p] >> 2, A.sb_off[b * P + p + 1] >> 2, tid);
      }
    };
    // phase 1: numerators n = Q * (pi*theta) kept in the registers, partial row sums into y(i)
    auto phase1 = [&](FzRegs& rr, int64_t i) {
      if (i >= nblk) return;
      const int64_t b = team + i * T;
      double* yb = y + (i & 1) * R;
#pragma unroll
      for (int k = 0; k < FZ_QUADS; ++k) {
        rr.v0[k].x *= c[rr.rc[k].x & 0xFFFF]; lds_add(&yb[rr.rc[k].x >> 16], rr.v0[k].x);
        rr.v0[k].y *= c[rr.rc[k].y & 0xFFFF]; lds_add(&yb[rr.rc[k].y >> 16], rr.v0[k].y);
        rr.v1[k].x *= c[rr.rc[k].z & 0xFFFF]; lds_add(&yb[rr.rc[k].z >> 16], rr.v1[k].x);
        rr.v1[k].y *= c[rr.rc[k].w & 0xFFFF]; lds_add(&yb[rr.rc[k].w >> 16], rr.v1[k].y);
      }
      const int64_t e0 = A.sb_off[b * P + p], e1 = A.sb_off[b * P + p + 1];
      for (int64_t e = e0 + FZ_CAP + tid; e < e1; e += FZ_DT) {   // overflow beyond the register tile (rare)
        uint32_t rc = A.prc[e];
        lds_add(&yb[rc >> 16], A.pval[e] * c[rc & 0xFFFF]);
      }
    };
    // phase 2: scatter w * z into the part's column accumulators
    auto phase2 = [&](FzRegs& rr, int64_t i) {
      if (i < 0) return;
      const int64_t b = team + i * T;
      const double* sb = s + (i & 1) * R;
#pragma unroll
      for (int k = 0; k < FZ_QUADS; ++k) {
        lds_add(&acc[rr.rc[k].x & 0xFFFF], rr.v0[k].x * sb[rr.rc[k].x >> 16]);
        lds_add(&acc[rr.rc[k].y & 0xFFFF], rr.v0[k].y * sb[rr.rc[k].y >> 16]);
        lds_add(&acc[rr.rc[k].z & 0xFFFF], rr.v1[k].x * sb[rr.rc[k].z >> 16]);
        lds_add(&acc[rr.rc[k].w & 0xFFFF], rr.v1[k].y * sb[rr.rc[k].w >> 16]);
      }
      const int64_t e0 = A.sb_off[b * P + p], e1 = A.sb_off[b * P + p + 1];
      for (int64_t e = e0 + FZ_CAP + tid; e < e1; e += FZ_DT) {
        uint32_t rc = A.prc[e];
        lds_add(&acc[rc & 0xFFFF], (A.pval[e] * c[rc & 0xFFFF]) * sb[rc >> 16]);
      }
    };
    int64_t i = 0;
    // step i: scatter block i-1 (its set is then refilled with block i+2), row sums of block i+1
    auto step = [&](FzRegs& s_p2, FzRegs& s_p1) {
      const bool pr = A.prof && team == 0 && p == 0 && tid == 0 && (int)i < A.prof_blocks;
      if (pr) A.prof[i * FZ_PROF_SLOTS + 0] = clock64();
      phase2(s_p2, i - 1);
      if (pr) A.prof[i * FZ_PROF_SLOTS + 7] = clock64();
      if (A.dbg & 2) load_blk(s_p1, i + 1); else load_blk(s_p2, i + 2);
      phase1(s_p1, i + 1);
      if (pr) A.prof[i * FZ_PROF_SLOTS + 1] = clock64();
      __syncthreads();                                   // barrier A(i+1)
      __syncthreads();                                   // barrier B(i+1)
      if (pr) A.prof[i * FZ_PROF_SLOTS + 6] = clock64();
      ++i;
    };
    load_blk(r0, 0);
    load_blk(r1, 1);
    phase1(r0, 0);
    __syncthreads();                                     // barrier A(0)
    __syncthreads();                                     // barrier B(0)
    while (i <= nblk) {                                  // block k lives in register set k % 3
      step(r2, r1);                                      // i % 3 == 0
      if (i > nblk) break;
      step(r0, r2);                                      // i % 3 == 1
      if (i > nblk) break;
      step(r1, r0);                                      // i % 3 == 2
    }
  }
  __syncthreads();
  double* out = A.partial + (int64_t)team * (P * Kp) + p * Kp;
  for (int t = tid; t < Kp; t += FZ_NT) out[t] = acc[t];
}
