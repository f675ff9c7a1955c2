// Blocked Cholesky, triangular inverse and K^-1/gradient kernels (gfx950).  Included by kernels.hpp.
//
// Right-looking blocked Cholesky with NB = 128:
//   k_potf2        one workgroup factors the 128x128 diagonal block in LDS and leaves the inverses of
//                  its eight 16x16 diagonal sub-blocks in the diagonal block of Linv
//   k_trsm_panel   L21 = A21 L11^-T, register resident: each wave keeps 16 rows x 128 columns as eight
//                  transposed 16x16 MFMA accumulators and substitutes block column by block column
//   k_syrk_trail   A22 -= L21 L21^T on the lower tiles (128x128 or 64x64 tiles)
// Triangular inverse (needed by alpha, predictions and the gradient):
//   k_trti_diag    all diagonal 128x128 blocks at once (one workgroup each)
//   k_trtri_T/R    recursive doubling over 128-blocks, two GEMM launches per level
//   k_lauum_grad   K^-1 = Linv^T Linv tile by tile, fused with the d+1 gradient reductions
#pragma once

namespace bobe {

constexpr int PLD = 130;                          // LDS leading dimension of a 128x128 block (doubles)
constexpr int POTF2_DLD = 17;                     // leading dimension of a staged 16x16 inverse

// coalesced 128x128 block <-> LDS (16-byte accesses, one row per wave per step).  LOWER: only the lower
// triangle is needed; lanes right of the diagonal re-read the diagonal's 16-byte granule (same cache line,
// no branch), which halves the distinct lines fetched.  The strictly upper part of S is then unspecified.
template <bool LOWER = false>
__device__ __forceinline__ void block_load(double* S, const double* __restrict__ G, int64_t ld) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v2d v[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const int r = 4 * i + wave;
    const int c = LOWER ? ((2 * lane <= r) ? 2 * lane : (r & ~1)) : 2 * lane;
    v[i] = *reinterpret_cast<const v2d*>(G + (int64_t)r * ld + c);
  }
#pragma unroll
  for (int i = 0; i < 32; ++i) *reinterpret_cast<v2d*>(S + (4 * i + wave) * PLD + 2 * lane) = v[i];
}
// store the lower triangle (zeros above the diagonal)
__device__ __forceinline__ void block_store_lower(const double* S, double* __restrict__ G, int64_t ld) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll 8
  for (int i = 0; i < 32; ++i) {
    const int r = 4 * i + wave, c = 2 * lane;
    v2d v = *reinterpret_cast<const v2d*>(S + r * PLD + c);
    if (c > r) v[0] = 0.0;
    if (c + 1 > r) v[1] = 0.0;
    *reinterpret_cast<v2d*>(G + (int64_t)r * ld + c) = v;
  }
}

// diagnostic cycle stamps (template-disabled in the production instantiations)
#define BOBE_STAMP(idx)                                                        \
  do {                                                                         \
    if (STAMP && threadIdx.x == 0) stamps[(idx)] = __builtin_amdgcn_s_memtime(); \
  } while (0)

// 1/sqrt(a): hardware estimate + two Newton steps (error ~1 ulp); NaN for a < 0, +inf for a = 0.
// (One third-order step instead is 4 % faster but not accurate enough: with it the N = 40 fit of
// tests/test_gpu_bo.py stops at a different optimum — the pivots feed an ill-conditioned K.)
__device__ __forceinline__ double rsqrt_nr(double a) {
  double y = __builtin_amdgcn_rsq(a);
  double e = __builtin_fma(-a * y, y, 1.0);
  y = __builtin_fma(0.5 * y, e, y);
  e = __builtin_fma(-a * y, y, 1.0);
  y = __builtin_fma(0.5 * y, e, y);
  return y;
}

// ---- diagonal block -------------------------------------------------------------------------------
// FACTOR: Cholesky of A[blk][blk] in place (lower, zeros above).  Always: the inverses of the eight
// 16x16 diagonal sub-blocks go to the same positions of Linv[blk][blk] (rest of that block untouched).
// A non-positive pivot makes the result NaN (like XLA) and records the column in *info.
//
// Per 16-column step p:
//   A  wave 0 factors the 16x16 diagonal sub-block right-looking, one matrix row per lane; lanes 16..31
//      carry the rows of an identity matrix through the same column operations, which multiplies them by
//      Lpp^-T: the inverse of the sub-block comes for free.  Meanwhile waves 1..3 apply the deferred
//      (non-urgent) updates of step p-1.
//   B  rows below: X^T = inv(Lpp) A^T, four MFMAs per 16-row tile.
//   C  the one urgent update, the next diagonal tile (p+1,p+1), by wave 0 without a barrier; every other
//      tile of the step's in-block update is deferred to the next step's phase A.
constexpr int POTF2_SMEM_BYTES = (TILE * PLD + 8 * 16 * POTF2_DLD) * 8;   // 150,528 B

// integer-only lower-triangular decode for small indices (t = i(i+1)/2 + j, i < 8)
__device__ __forceinline__ void tri_decode_small(int t, int& i, int& j) {
  int ii = 0;
#pragma unroll
  for (int c = 1; c < 8; ++c) ii += (t >= c * (c + 1) / 2) ? 1 : 0;
  i = ii;
  j = t - ii * (ii + 1) / 2;
}

// S[ti][tj] -= L[ti][p] L[tj][p]^T for two 16x16 tiles at once (independent MFMA chains)
__device__ __forceinline__ void potf2_update2(double* S, int o, int ti0, int tj0, int ti1, int tj1, bool two, int lane) {
  v4d acc0, acc1;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    acc0[r] = S[(16 * ti0 + (lane >> 4) + 4 * r) * PLD + 16 * tj0 + (lane & 15)];
    acc1[r] = S[(16 * ti1 + (lane >> 4) + 4 * r) * PLD + 16 * tj1 + (lane & 15)];
  }
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const double av0 = -S[(16 * ti0 + (lane & 15)) * PLD + o + 4 * ks + (lane >> 4)];
    const double bv0 = S[(16 * tj0 + (lane & 15)) * PLD + o + 4 * ks + (lane >> 4)];
    const double av1 = -S[(16 * ti1 + (lane & 15)) * PLD + o + 4 * ks + (lane >> 4)];
    const double bv1 = S[(16 * tj1 + (lane & 15)) * PLD + o + 4 * ks + (lane >> 4)];
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av0, bv0, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av1, bv1, acc1, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) S[(16 * ti0 + (lane >> 4) + 4 * r) * PLD + 16 * tj0 + (lane & 15)] = acc0[r];
  if (two) {
#pragma unroll
    for (int r = 0; r < 4; ++r) S[(16 * ti1 + (lane >> 4) + 4 * r) * PLD + 16 * tj1 + (lane & 15)] = acc1[r];
  }
}

// Hand-off of the fused panel launch (k_panel_fused), in the write-through form of cdna_hip_programming.md
// Guideline 16 / MI355X_MICROARCH.md "Valid forms": EVERY published byte is stored with an agent-scope (sc1,
// write-through) store and loaded with an agent-scope (sc1, L1-bypassing) load; the one storing wave drains its
// stores (s_waitcnt vmcnt(0)) before its lane 0 stores the flag; consumers poll the flag with sc1 loads and load
// the data only after the poll has matched (plus a workgroup barrier for the non-polling waves).  No fences.
__device__ __forceinline__ void st_agent(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_agent(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Publish sub-panel p of the block being factored: rows 16p..16p+15, columns < 16p of L and the 16x16 inverse
// invD_p, then flag = p+1.  Called by ONE wave.
__device__ __forceinline__ void potf2_publish(const double* S, const double* Dall, double* __restrict__ Ab, int64_t lda,
                                              double* __restrict__ Ib, int64_t ldl, int p, int* flag, int lane) {
  const int o = 16 * p;
  for (int i = 0; i < 16; ++i)
    for (int c = lane; c < o; c += 64) st_agent(Ab + (int64_t)(o + i) * lda + c, S[(o + i) * PLD + c]);
  {
    const int rr = lane >> 2, c0 = (lane & 3) * 4;
    const double* src = Dall + (p * 16 + rr) * POTF2_DLD + c0;
    double* dst = Ib + (int64_t)(o + rr) * ldl + o + c0;
#pragma unroll
    for (int c = 0; c < 4; ++c) st_agent(dst + c, src[c]);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) __hip_atomic_store(flag, p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool FACTOR, bool STAMP, bool PUB>
__device__ __forceinline__ void potf2_body(double* __restrict__ A, int64_t lda, double* __restrict__ Linv, int64_t ldl,
                                           int blk, int* __restrict__ info, unsigned long long* __restrict__ stamps,
                                           int* flag, int nvalid = TILE) {
  extern __shared__ double S[];
  double* Dall = S + TILE * PLD;  // [8][16][POTF2_DLD]: inverses of the 16x16 diagonal sub-blocks
  // columns >= nvalid of a ragged last block are the identity padding: their sub-steps are skipped (L = I,
  // inverse = I, and the rows of the panel below them are zero, so no update is lost)
  const int nsteps = FACTOR ? ((nvalid + 15) >> 4) : 0;
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = t >> 6;
  double* Ab = A + ((int64_t)blk * TILE) * lda + (int64_t)blk * TILE;
  double* Ib = Linv + ((int64_t)blk * TILE) * ldl + (int64_t)blk * TILE;
  BOBE_STAMP(0);
  block_load<true>(S, Ab, lda);
  if (FACTOR && nsteps < 8)
    for (int e = t; e < (8 - nsteps) * 16 * 16; e += 256) {
      const int pp = nsteps + (e >> 8), rr = (e >> 4) & 15, cc = e & 15;
      Dall[(pp * 16 + rr) * POTF2_DLD + cc] = (rr == cc) ? 1.0 : 0.0;
    }
  __syncthreads();
  BOBE_STAMP(1);

  for (int p = 0; p < nsteps; ++p) {
    const int o = 16 * p;
    double* Dv = Dall + p * 16 * POTF2_DLD;
    BOBE_STAMP(2 + 3 * p);
    if (wave == 0) {
      // ---- phase A, wave 0: 16x16 Cholesky + inverse (upper half-wave idle: EXEC[63:32] = 0) ----
      if (lane < 32) {
      const int li = lane & 15;
      const bool ident = (lane >= 16) && (lane < 32);
      double r[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) r[c] = ident ? ((c == li) ? 1.0 : 0.0) : S[(o + li) * PLD + o + c];
      bool bad = false;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const double ajj = readlane_f64(r[j], j);
        if (!(ajj > 0.0)) bad = true;
        const double inv = rsqrt_nr(ajj);
        r[j] *= inv;
#pragma unroll
        for (int c = j + 1; c < 16; ++c) {
          const double lcj = readlane_f64(r[j], c);
          r[c] = __builtin_fma(-r[j], lcj, r[c]);
        }
      }
      if (lane < 16) {
#pragma unroll
        for (int c = 0; c < 16; ++c) S[(o + li) * PLD + o + c] = (c <= li) ? r[c] : 0.0;
      } else if (ident) {
        // lane 16+i holds row i of Lpp^-T: r[c] = inv(Lpp)[c][i]
#pragma unroll
        for (int c = 0; c < 16; ++c) Dv[c * POTF2_DLD + li] = r[c];
      }
      if (bad && lane == 0) atomicMin(info, blk * TILE + o + 1);
      }
    } else if (p > 0) {
      // fused launch: wave 3 first hands sub-panel p-1 (final since the last barrier) to the panel solvers
      if (PUB && wave == 3) potf2_publish(S, Dall, Ab, lda, Ib, ldl, p - 1, flag, lane);
      // ---- phase A, waves 1..3: deferred updates of step p-1: every tile (ti, tj), p <= tj <= ti <= 7,
      //      except the diagonal tile (p, p), which wave 0 updated right after the previous phase B ----
      const int op = o - 16;
      const int nt = 8 - p;                 // tiles p..7
      const int ntiles = nt * (nt + 1) / 2 - 1;
      for (int q = wave - 1; q < ntiles; q += 6) {
        int a0, b0, a1, b1;
        tri_decode_small(q + 1, a0, b0);    // index 0 is (p, p): skipped
        const bool two = (q + 3) < ntiles;
        tri_decode_small(two ? q + 4 : q + 1, a1, b1);
        potf2_update2(S, op, p + a0, p + b0, p + a1, p + b1, two, lane);
      }
    }
    __syncthreads();
    BOBE_STAMP(3 + 3 * p);
    // ---- phase B: rows below, X^T = inv(Lpp) * A^T per 16-row tile ----
    for (int tt = p + 1 + wave; tt < 8; tt += 4) {
      v4d y = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double av = Dv[(lane & 15) * POTF2_DLD + (lane >> 4) + 4 * r];
        const double bv = S[(16 * tt + (lane & 15)) * PLD + o + (lane >> 4) + 4 * r];
        y = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, y, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) S[(16 * tt + (lane & 15)) * PLD + o + (lane >> 4) + 4 * r] = y[r];
    }
    __syncthreads();
    BOBE_STAMP(4 + 3 * p);
    // ---- phase C: the only urgent update is the next diagonal tile (p+1, p+1); wave 0 does it and runs
    //      straight into the next phase A (same wave: no barrier needed), the other waves go on to the
    //      deferred tiles ----
    if (p + 1 < nsteps && wave == 0) potf2_update2(S, o, p + 1, p + 1, p + 1, p + 1, false, lane);
  }
  __syncthreads();
  BOBE_STAMP(26);
  if (FACTOR && PUB && wave == 3) potf2_publish(S, Dall, Ab, lda, Ib, ldl, 7, flag, lane);   // last sub-panel first
  if (FACTOR) {
    block_store_lower(S, Ab, lda);
    // the eight 16x16 inverses: thread t -> sub-block t>>5, row (t>>1)&15, half row t&1
    const int bb = t >> 5, rr = (t >> 1) & 15, hh = t & 1;
    const double* src = Dall + (bb * 16 + rr) * POTF2_DLD + 8 * hh;
    double* dst = Ib + (int64_t)(16 * bb + rr) * ldl + 16 * bb + 8 * hh;
#pragma unroll
    for (int c = 0; c < 8; ++c) dst[c] = src[c];
  }
  BOBE_STAMP(27);
  if (!FACTOR && t < 128) {
    // given L (restore path): inverses of the eight 16x16 diagonal sub-blocks, one column per thread
    const int bb = t >> 4, col = t & 15, o = 16 * bb;
    double x[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      double s = (r == col) ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < r; ++k) s = __builtin_fma(-S[(o + r) * PLD + o + k], x[k], s);
      x[r] = s / S[(o + r) * PLD + o + r];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) Ib[(int64_t)(o + r) * ldl + o + col] = x[r];
  }
  BOBE_STAMP(28);
}

template <bool FACTOR, bool STAMP = false>
__global__ __launch_bounds__(256) void k_potf2(double* __restrict__ A, int64_t lda, double* __restrict__ Linv,
                                               int64_t ldl, int blk, int* __restrict__ info,
                                               unsigned long long* __restrict__ stamps = nullptr, int nvalid = TILE) {
  potf2_body<FACTOR, STAMP, false>(A, lda, Linv, ldl, blk, info, stamps, nullptr, nvalid);
}

// ---- inverse of every diagonal 128x128 block (grid = nb) --------------------------------------------
// in: L[blk][blk] (lower) and the 16x16 diagonal inverses left in Linv[blk][blk] by k_potf2.
// out: Linv[blk][blk] = L[blk][blk]^-1 (lower, zeros above).
__global__ __launch_bounds__(256) void k_trti_diag(const double* __restrict__ L, int64_t lda,
                                                   double* __restrict__ Linv, int64_t ldl) {
  extern __shared__ double S[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int blk = blockIdx.x;
  const double* Lb = L + ((int64_t)blk * TILE) * lda + (int64_t)blk * TILE;
  double* Ib = Linv + ((int64_t)blk * TILE) * ldl + (int64_t)blk * TILE;
  block_load(S, Lb, lda);
  __syncthreads();
  if (t < 128) {   // overwrite the diagonal 16x16 sub-blocks with their inverses
    const int bb = t >> 4, col = t & 15, o = 16 * bb;
#pragma unroll
    for (int r = 0; r < 16; ++r) S[(o + r) * PLD + o + col] = Ib[(int64_t)(o + r) * ldl + o + col];
  }
  __syncthreads();
  // block rows 1..7:  inv[i][j] = -inv[i][i] * sum_{k=j}^{i-1} L[i][k] inv[k][j]
  for (int i = 1; i < 8; ++i) {
    v4d res0 = (v4d){0.0, 0.0, 0.0, 0.0}, res1 = res0;
#pragma unroll
    for (int slot = 0; slot < 2; ++slot) {
      const int j = wave + 4 * slot;
      if (j < i) {
        v4d tacc = (v4d){0.0, 0.0, 0.0, 0.0};
        for (int k = j; k < i; ++k) {
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            const double av = S[(16 * i + (lane & 15)) * PLD + 16 * k + 4 * ks + (lane >> 4)];
            const double bv = S[(16 * k + 4 * ks + (lane >> 4)) * PLD + 16 * j + (lane & 15)];
            tacc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, tacc, 0, 0, 0);
          }
        }
        // R = -inv[i][i] * T ; T's accumulator register r holds row (lane>>4)+4r, used as the k index
        v4d racc = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double av = -S[(16 * i + (lane & 15)) * PLD + 16 * i + (lane >> 4) + 4 * r];
          racc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, tacc[r], racc, 0, 0, 0);
        }
        if (slot == 0) res0 = racc; else res1 = racc;
      }
    }
    __syncthreads();
#pragma unroll
    for (int slot = 0; slot < 2; ++slot) {
      const int j = wave + 4 * slot;
      if (j < i) {
        const v4d racc = slot == 0 ? res0 : res1;
#pragma unroll
        for (int r = 0; r < 4; ++r) S[(16 * i + (lane >> 4) + 4 * r) * PLD + 16 * j + (lane & 15)] = racc[r];
      }
    }
    __syncthreads();
  }
  block_store_lower(S, Ib, ldl);
}

// ---- panel: A[r][k-block] <- A[r][k-block] * L_kk^-T for rows r >= (k+1)*128 ------------------------
// grid = rows/64; the workgroup stages L_kk (lower) and the eight 16x16 diagonal inverses in LDS; each
// wave owns 16 rows and keeps them as eight transposed 16x16 accumulators X^T_p:
//   X^T_p = invD_p * (A^T_p - sum_{q<p} L_kk[p][q] X^T_q)
// (an accumulator's register r holds row (lane>>4)+4r, which serves as the k index of the next MFMA).
constexpr int TRSM_DLD = 17;                                             // leading dim of a staged 16x16 inverse
constexpr int TRSM_SMEM_BYTES = (TILE * PLD + 8 * 16 * TRSM_DLD) * 8;    // 150,528 B
template <bool STAMP = false>
__global__ __launch_bounds__(256) void k_trsm_panel(double* __restrict__ A, int64_t lda,
                                                    const double* __restrict__ Dinv, int64_t ldl, int k,
                                                    unsigned long long* __restrict__ stamps = nullptr) {
  extern __shared__ double S[];
  BOBE_STAMP(0);
  double* D = S + TILE * PLD;   // [8][16][TRSM_DLD]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int64_t row0 = (int64_t)(k + 1) * TILE + (int64_t)blockIdx.x * 64 + wave * 16;
  const int64_t col0 = (int64_t)k * TILE;
  const double* Lkk = A + col0 * lda + col0;
  const double* Dk = Dinv + col0 * ldl + col0;
  const int g = lane >> 4, li = lane & 15;
  double* Aw = A + row0 * lda + col0;          // this wave's 16 rows x 128 columns
  double* Sw = S + (wave * 16) * PLD;          // and its private 16-row slab of the LDS block
  // global -> registers, everything in flight at once: the wave's rows (full 1-KiB rows, 16 B per lane),
  // L_kk (one row per wave per step) and this thread's share of the eight diagonal inverses
  v2d xr[16], lr[32];
#pragma unroll
  for (int i = 0; i < 16; ++i) xr[i] = *reinterpret_cast<const v2d*>(Aw + (int64_t)i * lda + 2 * lane);
#pragma unroll
  for (int i = 0; i < 32; ++i) {   // lower triangle only (lanes right of the diagonal re-read its granule)
    const int r = 4 * i + wave;
    const int c = (2 * lane <= r) ? 2 * lane : (r & ~1);
    lr[i] = *reinterpret_cast<const v2d*>(Lkk + (int64_t)r * lda + c);
  }
  double dr[8];
  {
    const int bb = t >> 5, rr = (t >> 1) & 15, hh = t & 1;
    const double* src = Dk + (int64_t)(16 * bb + rr) * ldl + 16 * bb + 8 * hh;
#pragma unroll
    for (int c = 0; c < 8; ++c) dr[c] = src[c];
  }
  // rows -> LDS slab -> transposed MFMA accumulators X^T_p (lane (li,g), reg r = X[row li][16p + g + 4r])
#pragma unroll
  for (int i = 0; i < 16; ++i) *reinterpret_cast<v2d*>(Sw + i * PLD + 2 * lane) = xr[i];
  v4d X[8];
#pragma unroll
  for (int p = 0; p < 8; ++p)
#pragma unroll
    for (int r = 0; r < 4; ++r) X[p][r] = Sw[li * PLD + 16 * p + g + 4 * r];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 32; ++i) *reinterpret_cast<v2d*>(S + (4 * i + wave) * PLD + 2 * lane) = lr[i];
  {
    const int bb = t >> 5, rr = (t >> 1) & 15, hh = t & 1;
    double* dst = D + (bb * 16 + rr) * TRSM_DLD + 8 * hh;
#pragma unroll
    for (int c = 0; c < 8; ++c) dst[c] = dr[c];
  }
  __syncthreads();
  BOBE_STAMP(1);
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    v4d x = X[p];
#pragma unroll
    for (int q = 0; q < p; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double av = -S[(16 * p + li) * PLD + 16 * q + g + 4 * r];
        x = __builtin_amdgcn_mfma_f64_16x16x4f64(av, X[q][r], x, 0, 0, 0);
      }
    v4d y = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double av = D[(p * 16 + li) * TRSM_DLD + g + 4 * r];
      y = __builtin_amdgcn_mfma_f64_16x16x4f64(av, x[r], y, 0, 0, 0);
    }
    X[p] = y;
  }
  __syncthreads();   // every wave is done with L_kk: the block is reused to transpose the results back
  BOBE_STAMP(2);
#pragma unroll
  for (int p = 0; p < 8; ++p)
#pragma unroll
    for (int r = 0; r < 4; ++r) Sw[li * PLD + 16 * p + g + 4 * r] = X[p][r];
#pragma unroll
  for (int i = 0; i < 16; ++i)
    *reinterpret_cast<v2d*>(Aw + (int64_t)i * lda + 2 * lane) = *reinterpret_cast<const v2d*>(Sw + i * PLD + 2 * lane);
  BOBE_STAMP(3);
}

// ---- fused panel launch: potf2(k) (workgroup 0) + streaming panel solve (workgroups 1..2*rem) -------------
// The solvers are the same register-resident algorithm as k_trsm_panel, but instead of waiting for the whole
// L_kk they consume its 16-column sub-panels as workgroup 0 publishes them (flag value p+1 = sub-panels 0..p
// are in global memory; write-through hand-off, see st_agent / ld_agent above), so only the last sub-step is
// left when the factorisation finishes.  All workgroups
// of the launch are co-resident (grid <= 1 + 2*(nb-1) <= #CUs, one workgroup per CU by LDS size), the
// producer never waits for a consumer, and every spin is bounded (timeout -> *info = -1, results invalid).
constexpr int FUSED_LROW = 16 * PLD;                                  // one staged 16-row tile of L_kk
constexpr int FUSED_CONS_DOUBLES = 64 * PLD + 2 * FUSED_LROW + 2 * 16 * POTF2_DLD;
constexpr int FUSED_SMEM_BYTES = (POTF2_SMEM_BYTES > FUSED_CONS_DOUBLES * 8) ? POTF2_SMEM_BYTES : FUSED_CONS_DOUBLES * 8;

__device__ __forceinline__ void trsm_stream_body(double* __restrict__ A, int64_t lda, const double* __restrict__ Dinv,
                                                 int64_t ldl, int k, int wg, int* flag, int* __restrict__ info) {
  extern __shared__ double S[];
  double* Xs = S;                              // [64][PLD]   row transposes (start / end)
  double* Lp = S + 64 * PLD;                   // [2][16][PLD] staged row tiles of L_kk (double-buffered)
  double* Dp = Lp + 2 * FUSED_LROW;            // [2][16][POTF2_DLD] staged invD_p
  __shared__ int ok;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int64_t row0 = (int64_t)(k + 1) * TILE + (int64_t)wg * 64 + wave * 16;
  const int64_t col0 = (int64_t)k * TILE;
  const double* Lkk = A + col0 * lda + col0;
  const double* Dk = Dinv + col0 * ldl + col0;
  const int g = lane >> 4, li = lane & 15;
  double* Aw = A + row0 * lda + col0;
  double* Sw = Xs + (wave * 16) * PLD;
  // this wave's 16 rows -> LDS slab -> transposed accumulators (no dependence on the factorisation)
  {
    v2d xr[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) xr[i] = *reinterpret_cast<const v2d*>(Aw + (int64_t)i * lda + 2 * lane);
#pragma unroll
    for (int i = 0; i < 16; ++i) *reinterpret_cast<v2d*>(Sw + i * PLD + 2 * lane) = xr[i];
  }
  v4d X[8];
#pragma unroll
  for (int p = 0; p < 8; ++p)
#pragma unroll
    for (int r = 0; r < 4; ++r) X[p][r] = Sw[li * PLD + 16 * p + g + 4 * r];
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    // wait for sub-panel p: lane 0 polls the flag with sc1 loads; the other waves load after the barrier
    if (t == 0) {
      int spins = 0;
      while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < p + 1 && spins < (1 << 24)) {
        __builtin_amdgcn_s_sleep(2);
        ++spins;
      }
      ok = spins < (1 << 24);
    }
    __syncthreads();
    if (!ok) {
      if (t == 0) atomicMin(info, -1);
      return;
    }
    double* Lb = Lp + (p & 1) * FUSED_LROW;
    double* Db = Dp + (p & 1) * 16 * POTF2_DLD;
    if (p > 0) {   // row tile p of L_kk, columns < 16p: thread t -> row t>>4, 8 columns from 8*(t&15)
      const int rr = t >> 4, c0 = 8 * (t & 15);
      if (c0 < 16 * p) {
        const double* src = Lkk + (int64_t)(16 * p + rr) * lda + c0;
        double v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = ld_agent(src + c);
#pragma unroll
        for (int c = 0; c < 8; ++c) Lb[rr * PLD + c0 + c] = v[c];
      }
    }
    Db[(t >> 4) * POTF2_DLD + (t & 15)] = ld_agent(Dk + (int64_t)(16 * p + (t >> 4)) * ldl + 16 * p + (t & 15));
    __syncthreads();
    v4d x = X[p];
#pragma unroll
    for (int q = 0; q < p; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double av = -Lb[li * PLD + 16 * q + g + 4 * r];
        x = __builtin_amdgcn_mfma_f64_16x16x4f64(av, X[q][r], x, 0, 0, 0);
      }
    v4d y = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double av = Db[li * POTF2_DLD + g + 4 * r];
      y = __builtin_amdgcn_mfma_f64_16x16x4f64(av, x[r], y, 0, 0, 0);
    }
    X[p] = y;
  }
  // results back through the slab (each wave touches only its own 16 rows of Xs)
#pragma unroll
  for (int p = 0; p < 8; ++p)
#pragma unroll
    for (int r = 0; r < 4; ++r) Sw[li * PLD + 16 * p + g + 4 * r] = X[p][r];
#pragma unroll
  for (int i = 0; i < 16; ++i)
    *reinterpret_cast<v2d*>(Aw + (int64_t)i * lda + 2 * lane) = *reinterpret_cast<const v2d*>(Sw + i * PLD + 2 * lane);
}

__global__ __launch_bounds__(256) void k_panel_fused(double* __restrict__ A, int64_t lda, double* __restrict__ Linv,
                                                     int64_t ldl, int k, int* __restrict__ info, int* flags) {
  if (blockIdx.x == 0)
    potf2_body<true, false, true>(A, lda, Linv, ldl, k, info, nullptr, flags + k);
  else
    trsm_stream_body(A, lda, Linv, ldl, k, (int)blockIdx.x - 1, flags + k, info);
}

// ---- trailing update: A[i][j] -= sum_{k0 <= k < k1} L[i][k] L[j][k]^T over lower T x T tiles ---------------
// The tiles start at 128-block `first`; n = trailing size in T-tiles.  colmode 0: every lower tile (grid =
// n(n+1)/2).  colmode 1: only the tiles of 128-block column `first` (the next block column of a super-panel):
// tile columns b < 128/T, rows b <= a < n, grid = S n - S(S-1)/2 with S = 128/T.
template <int T, int BK>
__global__ __launch_bounds__(256) void k_syrk_trail(double* __restrict__ A, int64_t lda, int k0, int k1, int first,
                                                    int colmode, int n) {
  extern __shared__ double smem[];
  int a, b;
  if (colmode == 0) {
    tri_decode(blockIdx.x, a, b);
  } else {
    constexpr int S = TILE / T;
    int idx = blockIdx.x;
    b = 0;
#pragma unroll
    for (int c = 0; c < S - 1; ++c)
      if (b == c && idx >= n - c) {
        idx -= n - c;
        b = c + 1;
      }
    a = b + idx;
  }
  const int64_t base = (int64_t)first * TILE;
  v4d acc[T / 32][T / 32];
  load_tile<T, T>(acc, A, lda, base + (int64_t)a * T, base + (int64_t)b * T);   // acc = C, then acc -= A B^T
  gemm_tile<KC, KC, T, T, BK, true>(acc, A, lda, base + (int64_t)a * T, A, lda, base + (int64_t)b * T,
                                    (int64_t)k0 * TILE, (int64_t)k1 * TILE, smem);
  store_tile<T, T>(acc, A, lda, base + (int64_t)a * T, base + (int64_t)b * T, 1.0, 0.0);
}

// ---- recursive triangular inverse ------------------------------------------------------------------
// problem {lo, mid, hi} in 128-block units: with inv[lo:mid) and inv[mid:hi) known,
//   Tm = L[mid:hi, lo:mid) * inv[lo:mid)            (k_trtri_T, written to Tmp)
//   inv[mid:hi, lo:mid) = -inv[mid:hi) * Tm          (k_trtri_R)
// `off` counts T x T tiles: a problem owns (hi-mid)(mid-lo)(128/T)^2 consecutive blocks.
struct TriProb { int lo, mid, hi, off; };

// Locate the problem of workgroup `bid`.  T = 128: one tile per workgroup.  T = 64: a workgroup owns the
// two tiles e and nt-1-e of its problem's heavy-first enumeration, whose K lengths are complementary,
// so every workgroup of a problem does the same amount of work.
template <int T>
__device__ __forceinline__ bool tri_find(const TriProb* __restrict__ probs, int nprob, int bid, TriProb& p, int& e0,
                                         int& e1) {
  constexpr int S = (TILE / T) * (TILE / T);
  for (int q = 0; q < nprob; ++q) {
    const TriProb c = probs[q];
    const int nt = (c.hi - c.mid) * (c.mid - c.lo) * S;
    const int nw = (T == 128) ? nt : nt / 2;
    const int o = (T == 128) ? c.off : c.off * S / 2;
    if (bid >= o && bid < o + nw) {
      p = c;
      e0 = bid - o;
      e1 = (T == 128) ? -1 : nt - 1 - e0;
      return true;
    }
  }
  return false;
}

template <int T>
__global__ __launch_bounds__(256, 2) void k_trtri_T(const double* __restrict__ L, int64_t ldl,
                                                    const double* __restrict__ Linv, int64_t ldi,
                                                    double* __restrict__ Tmp, int64_t ldt,
                                                    const TriProb* __restrict__ probs, int nprob) {
  extern __shared__ double smem[];
  TriProb p;
  int e[2];
  if (!tri_find<T>(probs, nprob, blockIdx.x, p, e[0], e[1])) return;
  const int rows = (p.hi - p.mid) * (TILE / T);
  for (int u = 0; u < 2; ++u) {
    if (e[u] < 0) continue;
    const int tj = e[u] / rows, ti = e[u] % rows;          // column-major: small tj (long K) first
    const int64_t m0 = (int64_t)p.mid * TILE + (int64_t)ti * T, n0 = (int64_t)p.lo * TILE + (int64_t)tj * T;
    v4d acc[T / 32][T / 32];
    acc_zero(acc);
    gemm_tile<KC, RC, T, T, TileCfg<T>::bk>(acc, L, ldl, m0, Linv, ldi, n0, n0, (int64_t)p.mid * TILE, smem);
    store_tile<T, T>(acc, Tmp, ldt, m0, n0, 1.0, 0.0);
  }
}

template <int T>
__global__ __launch_bounds__(256, 2) void k_trtri_R(double* __restrict__ Linv, int64_t ldi,
                                                    const double* __restrict__ Tmp, int64_t ldt,
                                                    const TriProb* __restrict__ probs, int nprob) {
  extern __shared__ double smem[];
  TriProb p;
  int e[2];
  if (!tri_find<T>(probs, nprob, blockIdx.x, p, e[0], e[1])) return;
  const int rows = (p.hi - p.mid) * (TILE / T), w = (p.mid - p.lo) * (TILE / T);
  for (int u = 0; u < 2; ++u) {
    if (e[u] < 0) continue;
    const int ti = rows - 1 - e[u] / w, tj = e[u] % w;     // bottom rows (long K) first
    const int64_t m0 = (int64_t)p.mid * TILE + (int64_t)ti * T, n0 = (int64_t)p.lo * TILE + (int64_t)tj * T;
    v4d acc[T / 32][T / 32];
    acc_zero(acc);
    gemm_tile<KC, RC, T, T, TileCfg<T>::bk>(acc, Linv, ldi, m0, Tmp, ldt, n0, (int64_t)p.mid * TILE, m0 + T, smem);
    store_tile<T, T>(acc, Linv, ldi, m0, n0, -1.0, 0.0);
  }
}

// ---- K^-1 = Linv^T Linv fused with the MLL gradient reduction ------------------------------------------
// lower T x T tile (ti >= tj): Kinv = sum_{k >= ti*T} Linv[k][ti]^T Linv[k][tj];  W = alpha alpha^T - Kinv.
// partial[(blockIdx.x)*(DCAP+1) + j] = sum_ab W_ab dK_ab/dlog ls_j (j < d), [DCAP] = sum_ab W_ab Kt_ab,
// off-diagonal tiles weighted x2.  Optionally stores Kinv (lower tiles) for tests.
template <int KERN, int DCAP, int T>
__global__ __launch_bounds__(256, 2) void k_lauum_grad(const double* __restrict__ Linv, int64_t ldi, int64_t np,
                                                       int64_t n, const double* __restrict__ alpha,
                                                       const double* __restrict__ XsT, int64_t ldx, Hyper h,
                                                       double* __restrict__ partial, double* __restrict__ Kinv,
                                                       int64_t ldk, const Hyper* __restrict__ hp = nullptr) {
  extern __shared__ double smem[];
  if (hp) h = *hp;
  int ti, tj;
  tri_decode(blockIdx.x, ti, tj);
  v4d acc[T / 32][T / 32];
  acc_zero(acc);
  gemm_tile<RC, RC, T, T, TileCfg<T>::bk>(acc, Linv, ldi, (int64_t)ti * T, Linv, ldi, (int64_t)tj * T, (int64_t)ti * T, np,
                                          smem);
  if (Kinv) store_tile<T, T>(acc, Kinv, ldk, (int64_t)ti * T, (int64_t)tj * T, 1.0, 0.0);
  // stage coordinates and alpha in the (now free) GEMM LDS
  double* xa = smem;                  // [d][T]
  double* xb = smem + MAX_D * T;      // [d][T]
  double* aa = smem + 2 * MAX_D * T;  // [T]
  double* ab = aa + T;                // [T]
  double* red = ab + T;               // [4][DCAP+1]
  const int t = threadIdx.x;
  for (int e = t; e < h.d * T; e += 256) {
    const int j = e / T, c = e % T;
    xa[j * T + c] = XsT[j * ldx + (int64_t)ti * T + c];
    xb[j * T + c] = XsT[j * ldx + (int64_t)tj * T + c];
  }
  if (t < T) {
    aa[t] = alpha[(int64_t)ti * T + t];
    ab[t] = alpha[(int64_t)tj * T + t];
  }
  __syncthreads();
  double g[DCAP + 1];
#pragma unroll
  for (int j = 0; j <= DCAP; ++j) g[j] = 0.0;
#pragma unroll
  for (int i = 0; i < T / 32; ++i)
#pragma unroll
    for (int jj = 0; jj < T / 32; ++jj)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int a = acc_row<T>(i, r), b = acc_col<T>(jj);
        const int64_t ga = (int64_t)ti * T + a, gb = (int64_t)tj * T + b;
        if (ga < n && gb < n) {
          const double w = aa[a] * ab[b] - acc[i][jj][r];
          double dsq[DCAP];
          double r2 = 0.0;
#pragma unroll
          for (int j = 0; j < DCAP; ++j) {
            if (j < h.d) {
              const double df = xa[j * T + a] - xb[j * T + b];
              dsq[j] = df * df;
              r2 += dsq[j];
            } else {
              dsq[j] = 0.0;
            }
          }
          const double kv = kern_eval<KERN>(r2, h.kvar);
          const double wf = w * kern_grad_factor<KERN>(r2, h.kvar, kv);
#pragma unroll
          for (int j = 0; j < DCAP; ++j) g[j] += wf * dsq[j];
          g[DCAP] += w * kv;
        }
      }
  const double wt = (ti == tj) ? 1.0 : 2.0;
  const int lane = t & 63, wave = t >> 6;
#pragma unroll
  for (int j = 0; j <= DCAP; ++j) {
    const double s = wave_sum(g[j]);
    if (lane == 0) red[wave * (DCAP + 1) + j] = s;
  }
  __syncthreads();
  if (t <= DCAP) {
    const double s = ((red[t] + red[(DCAP + 1) + t]) + red[2 * (DCAP + 1) + t]) + red[3 * (DCAP + 1) + t];
    partial[(int64_t)blockIdx.x * (DCAP + 1) + t] = wt * s;
  }
}

}  // namespace bobe
