// Blocked Cholesky, triangular inverse and K^-1/gradient kernels (gfx950).  Included by gp_factor.hip only.
//
// Right-looking blocked Cholesky with NB = 128, batched over slots:
//   k_chol_panel   panel k of every factorisation of the batch in ONE launch: every workgroup factors the diagonal
//                  block (k,k) in LDS (each one redundantly: no hand-off) and solves its 64 rows of block column k
//                  under the factor's leaves
//   k_syrk_trail   A22 -= L21 L21^T on lower tiles (one K = 256 pass per two panels where the update dominates)
//   k_potf2 / k_trsm_panel   the two halves of a panel as separate launches (restore path; batches or matrices whose
//                  panel workgroups do not all fit on the chip at once)
// Every kernel takes the slot of a batch from its last grid dimension and offsets its matrices by a slot stride.
// Triangular inverse (needed by alpha, predictions and the gradient):
//   k_trti_diag    all diagonal 128x128 blocks at once (one workgroup each)
//   k_trtri_T/R    recursive doubling over 128-blocks, two GEMM launches per level
//   k_lauum_grad   K^-1 = Linv^T Linv tile by tile, fused with the d+1 gradient reductions
#pragma once
#include "kernels_common.hpp"

namespace bobe {

constexpr int PLD = 130;                          // LDS leading dimension of a 128x128 block (doubles)
constexpr int POTF2_DLD = 17;                     // leading dimension of a staged 16x16 inverse

// coalesced 128x128 block <-> LDS (16-byte accesses, one row per wave per step).  LOWER: only the lower
// triangle is needed; lanes right of the diagonal re-read the diagonal's 16-byte granule (same cache line,
// no branch), which halves the distinct lines fetched.  The strictly upper part of S is then unspecified.
template <bool LOWER = false, int NWAVES = 4>
__device__ __forceinline__ void block_load(double* S, const double* __restrict__ G, int64_t ld) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int NI = TILE / NWAVES;
  v2d v[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int r = NWAVES * i + wave;
    const int c = LOWER ? ((2 * lane <= r) ? 2 * lane : (r & ~1)) : 2 * lane;
    v[i] = *reinterpret_cast<const v2d*>(G + (int64_t)r * ld + c);
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) *reinterpret_cast<v2d*>(S + (NWAVES * i + wave) * PLD + 2 * lane) = v[i];
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

// The in-block update of n <= 6 consecutive tiles of the row-major list (positions t, t+1, ... of the triangle of tiles
// p..7) in ONE pass: all accumulators, then the four K-steps with n independent MFMA chains, then the stores - the pipe
// sees n x 4 MFMAs back to back instead of pairs between LDS round trips.  Per tile the operands and the order of its
// four MFMAs are potf2_update2's.
template <int K>
__device__ __forceinline__ void potf2_update_run(double* S, int o, int p, int a, int b, int lane) {
  // (a, b, p, o are wave-uniform: the tile offsets stay in scalar registers, a lane adds them to its two fixed bases - the
  // accumulator layout (row (lane >> 4) + 4 r, column lane & 15) and the operand layout (row lane & 15, column 4 ks +
  // (lane >> 4)) - and every access of a tile is that address plus a compile-time offset)
  double* accb = S + (lane >> 4) * PLD + (lane & 15);
  const double* opb = S + (lane & 15) * PLD + (lane >> 4) + o;
  double* pc[K];
  const double *pa[K], *pb[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int ti = p + a, tj = p + b;
    pc[k] = accb + ti * (16 * PLD) + tj * 16;
    pa[k] = opb + ti * (16 * PLD);
    pb[k] = opb + tj * (16 * PLD);
    if (++b > a) {
      ++a;
      b = 0;
    }
  }
  // every LDS operand of the pass is requested before the first MFMA (one latency for all of them)
  v4d acc[K];
  double av[K][4], bv[K][4];
#pragma unroll
  for (int k = 0; k < K; ++k) {
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[k][r] = pc[k][r * 4 * PLD];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      av[k][ks] = pa[k][4 * ks];
      bv[k][ks] = pb[k][4 * ks];
    }
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(-av[k][ks], bv[k][ks], acc[k], 0, 0, 0);
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int r = 0; r < 4; ++r) pc[k][r * 4 * PLD] = acc[k][r];
}

// k_chol_panel's deferred tiles per helper wave and step.  Helpers in the order waves 1, 2, 3, 5, 6, 7; waves 1 and 5 share a
// SIMD and both carry a row-solve side job (4 p + 4 dependent MFMAs in phase A of step p), waves 2 / 6 and 3 / 7 pair a
// solver with a wave that has none.  A 16 x 16 x 4 fp64 MFMA holds a SIMD's pipe for 64 cycles and the leaf gives phase A
// ~4 k: the counts below level tiles x 4 + side-job MFMAs over the three SIMDs (per-wave cycle stamps, round 3: with an
// even split the solver SIMD ran 5.5-6.8 k cycles in steps 1-4).  Which wave updates a tile does not change its arithmetic.
constexpr unsigned pack4(int a, int b, int c, int d, int e, int f) {
  return (unsigned)a | ((unsigned)b << 4) | ((unsigned)c << 8) | ((unsigned)d << 12) | ((unsigned)e << 16) | ((unsigned)f << 20);
}
__device__ __forceinline__ void panel_tile_range(int strips, int p, int h, int& start, int& cnt) {
  unsigned c;
  if (strips == 3) {                                  // waves 1, 2, 3 solve, waves 5, 6, 7 update: one of each per SIMD
    switch (p) {
      case 1: c = pack4(3, 3, 3, 6, 6, 6); break;     // 27 tiles (the side job is still short: the solvers take some)
      case 2: c = pack4(2, 2, 2, 5, 5, 4); break;     // 20
      case 3: c = pack4(1, 1, 1, 4, 4, 3); break;     // 14
      case 4: c = pack4(0, 0, 0, 3, 3, 3); break;     //  9
      case 5: c = pack4(0, 0, 0, 2, 2, 1); break;     //  5
      case 6: c = pack4(0, 0, 0, 1, 1, 0); break;     //  2
      default: c = 0u; break;
    }
  } else {
    switch (p) {
      case 1: c = pack4(4, 4, 4, 3, 6, 6); break;
      case 2: c = pack4(2, 3, 3, 2, 5, 5); break;
      case 3: c = pack4(1, 2, 2, 1, 4, 4); break;
      case 4: c = pack4(0, 1, 1, 0, 4, 3); break;
      case 5: c = pack4(0, 0, 0, 0, 3, 2); break;
      case 6: c = pack4(0, 0, 0, 0, 1, 1); break;
      default: c = 0u; break;
    }
  }
  start = 0;
  for (int i = 0; i < 6; ++i)
    if (i < h) start += (int)((c >> (4 * i)) & 15u);
  cnt = (int)((c >> (4 * h)) & 15u);
}

// The factor loop proper, on a block already staged in LDS (S: [128][PLD], lower part valid; Dall: [8][16][POTF2_DLD]).
// Called by the first 256 threads of a workgroup (waves 0..3); colbase = global index of the block's first column
// (for *info).  Leaves L (lower; the diagonal 16x16 tiles zero-filled above the diagonal) in S and the inverses of
// the eight diagonal 16x16 sub-blocks in Dall.  Ends with a barrier.
// NW = waves of the workgroup that take part (4 for the 256-thread kernels, 8 in k_chol_panel): wave 0 owns the
// serial leaf, the others share the deferred updates and the row solves (which tile a wave gets does not change any
// tile's arithmetic).
struct NoSideJob {
  __device__ __forceinline__ void operator()(int) const {}
};
// SIDE: work the helper waves (1 .. NW-1) do in phase A of step p after their deferred updates, called as side(p - 1):
// it may read everything steps <= p-1 produced (the L tiles of block row p-1, its 16x16 inverse) - k_chol_panel solves
// its rows below the block there, under the leaf of wave 0, instead of after the factor.
// SKIPW > 0: that wave takes no phase-A work (with more than four waves, wave SKIPW = 4 shares the leaf wave's SIMD, and
// MFMAs issued there slow the leaf down by half: measured 4.4-4.9 k -> 5.7-6.4 k cycles per step).
template <int NW = 4, class SIDE = NoSideJob, int SKIPW = -1, int STRIPS = 4>
__device__ __forceinline__ void potf2_factor_lds(double* S, double* Dall, int nsteps, int colbase, int* __restrict__ info,
                                                 SIDE side = SIDE()) {
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = t >> 6;
  for (int p = 0; p < nsteps; ++p) {
    const int o = 16 * p;
    double* Dv = Dall + p * 16 * POTF2_DLD;
    if (wave == 0) {
      // ---- phase A, wave 0: 16x16 Cholesky + inverse (upper half-wave idle: EXEC[63:32] = 0) ----
      if (lane < 32) {
      const int li = lane & 15;
      const bool ident = (lane >= 16) && (lane < 32);
      double r[16];
      // (every lane loads its row li - the identity lanes read the same 128 bytes as their twins and discard them:
      // eight 16-byte LDS reads with no EXEC games, instead of sixteen conditional 8-byte ones)
#pragma unroll
      for (int c = 0; c < 16; c += 2) {
        const v2d v = *reinterpret_cast<const v2d*>(S + (o + li) * PLD + o + c);
        r[c] = ident ? ((c == li) ? 1.0 : 0.0) : v[0];
        r[c + 1] = ident ? ((c + 1 == li) ? 1.0 : 0.0) : v[1];
      }
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
        // (the whole row: what lands right of the diagonal is never read as an operand, and block_store_lower writes
        // zeros there when the block leaves LDS)
#pragma unroll
        for (int c = 0; c < 16; c += 2) *reinterpret_cast<v2d*>(S + (o + li) * PLD + o + c) = (v2d){r[c], r[c + 1]};
      } else if (ident) {
        // lane 16+i holds row i of Lpp^-T: r[c] = inv(Lpp)[c][i]
#pragma unroll
        for (int c = 0; c < 16; ++c) Dv[c * POTF2_DLD + li] = r[c];
      }
      if (bad && lane == 0) atomicMin(info, colbase + o + 1);
      }
    } else if (p > 0 && wave != SKIPW) {
      // ---- phase A, waves 1..3: deferred updates of step p-1: every tile (ti, tj), p <= tj <= ti <= 7,
      //      except the diagonal tile (p, p), which wave 0 updated right after the previous phase B ----
      const int op = o - 16;
      const int nt = 8 - p;                 // tiles p..7
      const int ntiles = nt * (nt + 1) / 2 - 1;
      if (NW == 8) {
        // (eight waves: a contiguous run of the tile list per helper, sized by panel_tile_range)
        int q0, nq;
        const int wu = __builtin_amdgcn_readfirstlane(wave);       // (scalar: the run's tile offsets never touch a VGPR)
        panel_tile_range(STRIPS, p, wu < 4 ? wu - 1 : wu - 2, q0, nq);
        int a0 = 0, b0 = 0;
        if (nq > 0) tri_decode_small(q0 + 1, a0, b0);               // index 0 is (p, p): skipped

        while (nq > 0) {
          const int m = nq > 6 ? (nq + 1) / 2 : nq;                 // (a long run in two passes of similar size)
          switch (m) {
            case 1: potf2_update_run<1>(S, op, p, a0, b0, lane); break;
            case 2: potf2_update_run<2>(S, op, p, a0, b0, lane); break;
            case 3: potf2_update_run<3>(S, op, p, a0, b0, lane); break;
            case 4: potf2_update_run<4>(S, op, p, a0, b0, lane); break;
            case 5: potf2_update_run<5>(S, op, p, a0, b0, lane); break;
            default: potf2_update_run<6>(S, op, p, a0, b0, lane); break;
          }
          for (int i = 0; i < m; ++i)
            if (++b0 > a0) {
              ++a0;
              b0 = 0;
            }
          nq -= m;
        }
      } else {
        constexpr int NU = NW - 1 - (SKIPW > 0 ? 1 : 0);            // updater waves
        const int hw = wave - 1 - ((SKIPW > 0 && wave > SKIPW) ? 1 : 0);
        for (int q = hw; q < ntiles; q += 2 * NU) {
          int a0, b0, a1, b1;
          tri_decode_small(q + 1, a0, b0);    // index 0 is (p, p): skipped
          const bool two = (q + NU) < ntiles;
          tri_decode_small(two ? q + NU + 1 : q + 1, a1, b1);
          potf2_update2(S, op, p + a0, p + b0, p + a1, p + b1, two, lane);
        }
      }
      side(p - 1);
    }
    __syncthreads();
    // ---- phase B: rows below, X^T = inv(Lpp) * A^T per 16-row tile ----
    for (int tt = p + 1 + wave; tt < 8; tt += NW) {
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
    // ---- phase C: the only urgent update is the next diagonal tile (p+1, p+1); wave 0 does it and runs
    //      straight into the next phase A (same wave: no barrier needed), the other waves go on to the
    //      deferred tiles ----
    if (p + 1 < nsteps && wave == 0) potf2_update2(S, o, p + 1, p + 1, p + 1, p + 1, false, lane);
  }
  __syncthreads();
}

// stage-in of a diagonal block for the factor loop: lower part of the block, identity inverses for the padding steps
template <int NWAVES = 4>
__device__ __forceinline__ void potf2_stage_in(double* S, double* Dall, const double* __restrict__ Ab, int64_t lda,
                                               int nsteps) {
  const int t = threadIdx.x;
  block_load<true, NWAVES>(S, Ab, lda);
  if (nsteps < 8)
    for (int e = t; e < (8 - nsteps) * 16 * 16; e += 64 * NWAVES) {
      const int pp = nsteps + (e >> 8), rr = (e >> 4) & 15, cc = e & 15;
      Dall[(pp * 16 + rr) * POTF2_DLD + cc] = (rr == cc) ? 1.0 : 0.0;
    }
}

// write-back: L (lower, zeros above) to the block, the eight 16x16 inverses to the same positions of Linv's block
__device__ __forceinline__ void potf2_stage_out(const double* S, const double* Dall, double* __restrict__ Ab, int64_t lda,
                                                double* __restrict__ Ib, int64_t ldl) {
  const int t = threadIdx.x;
  block_store_lower(S, Ab, lda);
  // thread t -> sub-block t>>5, row (t>>1)&15, half row t&1
  const int bb = t >> 5, rr = (t >> 1) & 15, hh = t & 1;
  const double* src = Dall + (bb * 16 + rr) * POTF2_DLD + 8 * hh;
  double* dst = Ib + (int64_t)(16 * bb + rr) * ldl + 16 * bb + 8 * hh;
#pragma unroll
  for (int c = 0; c < 8; ++c) dst[c] = src[c];
}

template <bool FACTOR>
__device__ __forceinline__ void potf2_body(double* __restrict__ A, int64_t lda, double* __restrict__ Linv, int64_t ldl,
                                           int blk, int* __restrict__ info, int nvalid = TILE) {
  extern __shared__ double S[];
  double* Dall = S + TILE * PLD;  // [8][16][POTF2_DLD]: inverses of the 16x16 diagonal sub-blocks
  // columns >= nvalid of a ragged last block are the identity padding: their sub-steps are skipped (L = I,
  // inverse = I, and the rows of the panel below them are zero, so no update is lost)
  const int nsteps = FACTOR ? ((nvalid + 15) >> 4) : 0;
  const int t = threadIdx.x;
  double* Ab = A + ((int64_t)blk * TILE) * lda + (int64_t)blk * TILE;
  double* Ib = Linv + ((int64_t)blk * TILE) * ldl + (int64_t)blk * TILE;
  if (FACTOR) potf2_stage_in(S, Dall, Ab, lda, nsteps);
  else block_load<true>(S, Ab, lda);
  __syncthreads();
  if (FACTOR) potf2_factor_lds<>(S, Dall, nsteps, blk * TILE, info);
  if (FACTOR) potf2_stage_out(S, Dall, Ab, lda, Ib, ldl);
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
}

template <bool FACTOR>
__global__ __launch_bounds__(256) void k_potf2(double* __restrict__ A, int64_t lda, double* __restrict__ Linv,
                                               int64_t ldl, int blk, int* __restrict__ info, int nvalid = TILE,
                                               int64_t bsA = 0, int64_t bsL = 0) {
  const int slot = blockIdx.x;
  potf2_body<FACTOR>(A + slot * bsA, lda, Linv + slot * bsL, ldl, blk, info + slot, nvalid);
}

// ---- inverse of every diagonal 128x128 block (grid = nb) --------------------------------------------
// in: L[blk][blk] (lower) and the 16x16 diagonal inverses left in Linv[blk][blk] by k_potf2.
// out: Linv[blk][blk] = L[blk][blk]^-1 (lower, zeros above).
// diag / first_aside: the factorisation left the L_kk of blocks >= first_aside in its scratch blocks (k_chol_panel); they
// are read from there and put in place on the way (what k_copy_diag does when no inverse follows the factorisation).
// (the block's L is in S[128][PLD]; Ib = the block of Linv holding the eight 16x16 diagonal inverses; called by 256 threads)
__device__ __forceinline__ void trti_block(double* S, double* __restrict__ Ib, int64_t ldl) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
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

__global__ __launch_bounds__(256) void k_trti_diag(double* __restrict__ L, int64_t lda,
                                                   double* __restrict__ Linv, int64_t ldl, int64_t bsA = 0,
                                                   int64_t bsL = 0, const double* __restrict__ diag = nullptr,
                                                   int64_t bsD = 0, int first_aside = 1 << 30) {
  extern __shared__ double S[];
  const int blk = blockIdx.x;
  L += blockIdx.y * bsA;
  Linv += blockIdx.y * bsL;
  double* Lb = L + ((int64_t)blk * TILE) * lda + (int64_t)blk * TILE;
  double* Ib = Linv + ((int64_t)blk * TILE) * ldl + (int64_t)blk * TILE;
  if (diag && blk >= first_aside) {
    block_load(S, diag + blockIdx.y * bsD + (int64_t)blk * TILE * TILE, TILE);
    __syncthreads();
    block_store_lower(S, Lb, lda);
  } else {
    block_load(S, Lb, lda);
  }
  __syncthreads();
  trti_block(S, Ib, ldl);
}

// ---- panel: A[r][k-block] <- A[r][k-block] * L_kk^-T for rows r >= (k+1)*128 ------------------------
// grid = rows/64; the workgroup stages L_kk (lower) and the eight 16x16 diagonal inverses in LDS; each
// wave owns 16 rows and keeps them as eight transposed 16x16 accumulators X^T_p:
//   X^T_p = invD_p * (A^T_p - sum_{q<p} L_kk[p][q] X^T_q)
// (an accumulator's register r holds row (lane>>4)+4r, which serves as the k index of the next MFMA).
constexpr int TRSM_DLD = 17;                                             // leading dim of a staged 16x16 inverse
constexpr int TRSM_SMEM_BYTES = (TILE * PLD + 8 * 16 * TRSM_DLD) * 8;    // 150,528 B
__global__ __launch_bounds__(256) void k_trsm_panel(double* __restrict__ A, int64_t lda,
                                                    const double* __restrict__ Dinv, int64_t ldl, int k,
                                                    int64_t bsA = 0, int64_t bsL = 0) {
  extern __shared__ double S[];
  A += blockIdx.y * bsA;
  Dinv += blockIdx.y * bsL;
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
#pragma unroll
  for (int p = 0; p < 8; ++p)
#pragma unroll
    for (int r = 0; r < 4; ++r) Sw[li * PLD + 16 * p + g + 4 * r] = X[p][r];
#pragma unroll
  for (int i = 0; i < 16; ++i)
    *reinterpret_cast<v2d*>(Aw + (int64_t)i * lda + 2 * lane) = *reinterpret_cast<const v2d*>(Sw + i * PLD + 2 * lane);
}

// ---- panel k of the batched factorisation ------------------------------------------------------------------------
// Every panel workgroup stages block (k,k) in LDS and factors it itself (the factor is needed by all of them and takes as
// long on one CU as on sixty: recomputing it replaces a kernel boundary and a global round trip).  L_kk goes to a dense
// 128 x 128 scratch block, NOT to the diagonal block itself: the workgroups of the launch read A_kk whenever they get a
// CU - on a busy GPU possibly after others have finished - so the block must stay intact until the launch is over;
// k_copy_diag moves the scratch blocks into place afterwards.
// (A fused form - panel k side by side with the rest of the update by panel k-1 in one 1024-thread launch, k_chol_step -
// served the lone factorisation until the panel below made the separate launches as fast: 1.58 ms either way at
// N = 4096, 0.63 vs 0.66 ms at 2048; removed.  Its update half ran at 33-36 TFLOP/s against 48 for four 256-thread
// workgroups per CU, profiles/r02_a_ubench_update_variants.txt.)
// The panel as its own launch: EIGHT waves.  Wave 0 is the factor's leaf wave; wave 4 (same SIMD) does nothing while the
// leaf runs; waves 1, 2, 3, 5, 6, 7 are the helpers of the factor (deferred tile updates, shared out by panel_tile_range),
// and 1, 2, 3 (and 5 with four strips) each own 16 of the workgroup's 48 (64) rows below the block.  The
// solve of those rows by sub-block p (X^T_p = invD_p (A^T_p - sum_{q<p} L_kk[p][q] X^T_q), the MFMA sequence of
// k_trsm_panel, same bits) only needs what factor step p produced, so the helpers run it in phase A of step p+1, under
// wave 0's leaf, instead of after the factor: of the 144 MFMAs per wave (12 k cycles after a 52 k factor) only the last
// sub-block's four stay exposed.
constexpr int PANEL_THREADS = 512;
// 16-row strips a panel workgroup solves, one per solver wave.  STRIPS = 3 (48 rows, waves 1, 2, 3): every helper SIMD has
// one solver and one wave that only updates tiles - phase A of every step from 4 on is then the leaf's own 4.2 k cycles,
// against 4.6-5.7 k with STRIPS = 4 (64 rows, waves 1, 2, 3, 5: two solvers' dependent MFMA chains on one SIMD).  A third
// more workgroups: the host takes 3 strips wherever the launch still fits the chip, else 4 (cycle stamps, round 3: panel
// 64.3 k -> 60.2 k cycles).  Which workgroup solves a strip does not change its bits.
__host__ __device__ inline int panel_workgroups(int blocks_below, int strips) {
  return blocks_below > 0 ? (blocks_below * TILE + 16 * strips - 1) / (16 * strips) : 1;
}
template <int STRIPS>
__device__ __forceinline__ void chol_panel_body5(double* __restrict__ A, int64_t lda, double* __restrict__ Linv,
                                                 int64_t ldl, int k, int pw, int npanel, int* __restrict__ info,
                                                 int nvalid, double* __restrict__ Lkk_out, int gridDim_rows_below) {
  const bool has_rows = gridDim_rows_below > 0;
  extern __shared__ double S[];
  double* Dall = S + TILE * PLD;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int nsteps = (nvalid + 15) >> 4;
  const int64_t col0 = (int64_t)k * TILE;
  double* Ab = A + col0 * lda + col0;
  double* Ib = Linv + col0 * ldl + col0;
  // Stage-in under the first leaf: wave 0 fetches only what its first leaf reads - the 16 x 16 tile (0, 0) - and starts
  // factoring; waves 1..7 (idle in phase A of step 0) bring in rows 16..127 meanwhile (lower part: rows 0..15 hold nothing
  // else that is read).  The barrier that ends phase A of step 0 publishes the block (it used to cost ~10 k cycles before
  // the first leaf could start).
  if (wave == 0) {
    const int r = lane >> 2, c = (lane & 3) * 4;
    const v2d u0 = *reinterpret_cast<const v2d*>(Ab + (int64_t)r * lda + c);
    const v2d u1 = *reinterpret_cast<const v2d*>(Ab + (int64_t)r * lda + c + 2);
    *reinterpret_cast<v2d*>(S + r * PLD + c) = u0;
    *reinterpret_cast<v2d*>(S + r * PLD + c + 2) = u1;
  } else {
    v2d v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = 16 + (wave - 1) + 7 * i;
      const int c = (2 * lane <= r) ? 2 * lane : (r & ~1);
      v[i] = *reinterpret_cast<const v2d*>(Ab + (int64_t)r * lda + c);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) *reinterpret_cast<v2d*>(S + (16 + (wave - 1) + 7 * i) * PLD + 2 * lane) = v[i];
  }
  if (nsteps < 8)
    for (int e = t; e < (8 - nsteps) * 16 * 16; e += PANEL_THREADS) {
      const int pp = nsteps + (e >> 8), rr = (e >> 4) & 15, cc = e & 15;
      Dall[(pp * 16 + rr) * POTF2_DLD + cc] = (rr == cc) ? 1.0 : 0.0;
    }
  __builtin_amdgcn_sched_barrier(0);

  // solver wave (1, 2, 3, 5 = strip 0..3): rows (k+1)*128 + 16*STRIPS*pw + 16*strip .. +15 of block column k, straight into
  // transposed-accumulator layout (lane (li, g), register r of tile p = A[row li][16p + g + 4r]); in flight during the
  // first factor steps
  const int strip = wave < 4 ? wave - 1 : wave - 2;                        // waves 1, 2, 3, 5, 6, 7 -> strips 0 .. 5
  const int64_t rows_below = (int64_t)(gridDim_rows_below);
  constexpr int PANEL_ROWS = 16 * STRIPS;
  const bool solver = has_rows && wave != 0 && wave != 4 && strip < STRIPS &&
                      (int64_t)pw * PANEL_ROWS + strip * 16 < rows_below;      // (the last workgroup may own fewer strips)
  const int64_t row0 = (int64_t)(k + 1) * TILE + (int64_t)pw * PANEL_ROWS + strip * 16;
  double* Aw = A + row0 * lda + col0;
  // Only the first two sub-blocks now; sub-block p + 2 is requested at the end of the side job of step p + 1, a leaf
  // before it is needed (all 32 scattered loads per lane at once kept the stage-in waiting: phase A of step 0 7.9 k
  // cycles with three solver waves, 9.6-10.3 k with four)
  v4d X[8];
  auto fetch = [&](auto pc) {
    constexpr int p = decltype(pc)::value;
#pragma unroll
    for (int r = 0; r < 4; ++r) X[p][r] = Aw[(int64_t)li * lda + 16 * p + g + 4 * r];
  };
  if (solver) {
    fetch(std::integral_constant<int, 0>());
    fetch(std::integral_constant<int, 1>());
  }
  // (no barrier here: phase A of step 0 touches only wave 0's own tile; the scattered loads of X pass under the first leaf)
  // sub-block p in two parts: accumulate<p> (x = A^T_p - sum_{q<p} L_kk[p][q] X^T_q: needs the factor's steps < p only)
  // and finish<p> (X^T_p = invD_p x: needs leaf p).  Every LDS operand of a part is read first (one latency for all).
  auto accumulate = [&](auto pc) {
    constexpr int p = decltype(pc)::value;
    double lv[(p > 0 ? p : 1) * 4];
#pragma unroll
    for (int q = 0; q < p; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) lv[4 * q + r] = -S[(16 * p + li) * PLD + 16 * q + g + 4 * r];
    __builtin_amdgcn_sched_barrier(0);
    v4d x = X[p];
#pragma unroll
    for (int q = 0; q < p; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) x = __builtin_amdgcn_mfma_f64_16x16x4f64(lv[4 * q + r], X[q][r], x, 0, 0, 0);
    X[p] = x;
  };
  auto finish = [&](auto pc) {
    constexpr int p = decltype(pc)::value;
    double dv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) dv[r] = Dall[(p * 16 + li) * POTF2_DLD + g + 4 * r];
    v4d y = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 4; ++r) y = __builtin_amdgcn_mfma_f64_16x16x4f64(dv[r], X[p][r], y, 0, 0, 0);
    X[p] = y;
  };
  // phase A of factor step pd + 1: finish sub-block pd, then the whole accumulation of sub-block pd + 1
  // (X is indexed with constants only: it must stay in registers)
#define BOBE_SIDE_CASE(PD)                                                   \
  case PD:                                                                   \
    finish(std::integral_constant<int, PD>());                               \
    accumulate(std::integral_constant<int, PD + 1>());                       \
    if (PD + 2 < 8) fetch(std::integral_constant<int, (PD + 2) & 7>());      \
    break
  auto side = [&](int pd) {
    if (!solver) return;
    switch (pd) {
      BOBE_SIDE_CASE(0);
      BOBE_SIDE_CASE(1);
      BOBE_SIDE_CASE(2);
      BOBE_SIDE_CASE(3);
      BOBE_SIDE_CASE(4);
      BOBE_SIDE_CASE(5);
      BOBE_SIDE_CASE(6);
      default: break;
    }
  };
#undef BOBE_SIDE_CASE
  potf2_factor_lds<8, decltype(side), 4, STRIPS>(S, Dall, nsteps, (int)col0, info, side);
  // L_kk and the 16x16 inverses leave LDS once per slot: every workgroup of the launch holds the same factor, so
  // workgroup pw writes the rows pw, pw + npanel, ... of L_kk (zeros above the diagonal) and pw = 0 the inverses
  {
    const int np_eff = npanel < TILE ? npanel : TILE;
    for (int r = pw + np_eff * wave; r < TILE; r += np_eff * 8) {
      const int c = 2 * lane;
      v2d v = *reinterpret_cast<const v2d*>(S + r * PLD + c);
      if (c > r) v[0] = 0.0;
      if (c + 1 > r) v[1] = 0.0;
      *reinterpret_cast<v2d*>(Lkk_out + (int64_t)r * TILE + c) = v;
    }
    if (pw == 0 && t < 256) {
      const int bb = t >> 5, rr = (t >> 1) & 15, hh = t & 1;
      const double* src = Dall + (bb * 16 + rr) * POTF2_DLD + 8 * hh;
      double* dst = Ib + (int64_t)(16 * bb + rr) * ldl + 16 * bb + 8 * hh;
#pragma unroll
      for (int c = 0; c < 8; ++c) dst[c] = src[c];
    }
  }
  if (!has_rows) return;
  // (a block with rows below it is never the ragged last one: nsteps = 8 and the steps 0..6 ran under the leaves)
  if (solver) finish(std::integral_constant<int, 7>());
  __syncthreads();   // every wave is done with L_kk (and pw 0 with writing it back): reuse the block as transposer
  if (!solver) return;
  double* Sw = S + (strip * 16) * PLD;
#pragma unroll
  for (int p = 0; p < 8; ++p)
#pragma unroll
    for (int r = 0; r < 4; ++r) Sw[li * PLD + 16 * p + g + 4 * r] = X[p][r];
  // (each wave reads back only its own slab: no barrier needed, the wave's LDS accesses are ordered)
#pragma unroll
  for (int i = 0; i < 16; ++i)
    *reinterpret_cast<v2d*>(Aw + (int64_t)i * lda + 2 * lane) = *reinterpret_cast<const v2d*>(Sw + i * PLD + 2 * lane);
}

// panel k of every slot as ONE launch (grid = npanel x nbatch, PANEL_THREADS threads).  Replaces the k_potf2 +
// k_trsm_panel pair when all nbatch*npanel workgroups fit on the chip at once (the redundant factorisations then cost
// nothing and one kernel boundary plus the L_kk round trip through global memory go away).
// FILL: the launch carries extra workgroups (blockIdx.x >= npanel) that do NOT belong to the panel: they run trailing-
// update tiles that are not needed yet ("deferred" far columns of the trailing matrix, see potrf() in bobe_gp.hip) on the
// CUs the panel chain leaves empty.  A filler workgroup = two 256-thread groups, one 64 x 64 tile each, both with the same
// K range (panels [k0, k1) of its job: columns < k are final while panel k is being factored), on the tile core of
// k_syrk_trail - the same arithmetic per element as the separate update launch, only earlier and for free.
// (Measured with stand-in MFMA workgroups first, profiles/r03_filler_standin.txt: up to ~8 MFLOP per filler workgroup the
// factorisation takes exactly as long as without them; beyond that the launch lasts as long as its slowest filler.)
// (FillJob: gp_types.hpp)
constexpr int FILL_BK = 32;
constexpr int FILL_SMEM_DOUBLES = gemm_smem_doubles_exact<KC, KC, 64, 64, FILL_BK>();   // per group; two groups fit the panel's 150 KB
static_assert(2 * FILL_SMEM_DOUBLES * 8 <= POTF2_SMEM_BYTES, "update fillers exceed the panel's LDS");
template <bool FILL = false, int STRIPS = 4>
__global__ __launch_bounds__(PANEL_THREADS) void k_chol_panel(double* __restrict__ A, int64_t lda, int64_t bsA,
                                                              double* __restrict__ Linv, int64_t ldl, int64_t bsL, int k,
                                                              int npanel, int* __restrict__ info, int nvalid,
                                                              double* __restrict__ diag, int64_t bsD,
                                                              const FillJob* __restrict__ jobs = nullptr, int njobs = 0,
                                                              int rows_below = 0) {
  if (FILL && (int)blockIdx.x >= npanel) {
    extern __shared__ double S[];
    const int grp = threadIdx.x >> 8, tid = threadIdx.x & 255;
    const int idx = 2 * ((int)blockIdx.x - npanel) + grp;
    if (idx >= njobs) return;                 // (never: the job lists have even length, a workgroup = one pair)
    const FillJob jb = jobs[idx];
    double* As = A + blockIdx.y * bsA;
    const bool live = !(jb.flags & FILL_TWIN);
    const int64_t r0 = (int64_t)jb.ti * 64, c0 = (int64_t)jb.tj * 64;
    const int64_t kb = (int64_t)jb.k0 * 64, ke = (int64_t)jb.k1 * 64;
    v4d acc[2][2];
    load_tile<64, 64>(acc, As, lda, r0, c0, tid);
    gemm_tile<KC, KC, 64, 64, FILL_BK, true>(acc, As, lda, r0, As, lda, c0, kb, ke, S + grp * FILL_SMEM_DOUBLES, tid);
    if (live) store_tile<64, 64>(acc, As, lda, r0, c0, 1.0, 0.0, tid);
    return;
  }
  const int slot = blockIdx.y;
  chol_panel_body5<STRIPS>(A + slot * bsA, lda, Linv + slot * bsL, ldl, k, (int)blockIdx.x, npanel, info + slot, nvalid,
                           diag + slot * bsD + (int64_t)k * TILE * TILE, rows_below);
}

// A[blk][blk] <- scratch block blk for blk = first + blockIdx.x (slot = blockIdx.y): the L_kk the panel launches left aside
__global__ __launch_bounds__(256) void k_copy_diag(double* __restrict__ A, int64_t lda, int64_t bsA,
                                                   const double* __restrict__ diag, int64_t bsD, int first) {
  const int blk = first + blockIdx.x;
  const double* src = diag + blockIdx.y * bsD + (int64_t)blk * TILE * TILE;
  double* dst = A + blockIdx.y * bsA + ((int64_t)blk * TILE) * lda + (int64_t)blk * TILE;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll 8
  for (int i = 0; i < 32; ++i) {
    const int r = 4 * i + wave;
    *reinterpret_cast<v2d*>(dst + (int64_t)r * lda + 2 * lane) = *reinterpret_cast<const v2d*>(src + r * TILE + 2 * lane);
  }
}

// ---- trailing update: A[i][j] -= sum_{k0 <= k < k1} L[i][k] L[j][k]^T over lower T x T tiles ---------------
// The tiles start at 128-block `first`; n = trailing size in T-tiles.  colmode 0: every lower tile (grid =
// n(n+1)/2).  colmode 1: only the tiles of 128-block column `first` (the next block column of a super-panel):
// tile columns b < 128/T, rows b <= a < n, grid = S n - S(S-1)/2 with S = 128/T.
// colmode 2: the tiles of the first `ncol` T-columns only (rows from the diagonal down) - the few block columns an update
// touches when everything behind them is deferred.
// colk0 / far_col (optional, 64x64 tiles only): block columns >= far_col read their first pending panel from colk0[column]
// (deferred columns: the K range [colk0[c], k1) differs from column to column; colk0[c] >= k1 skips the column); columns
// before far_col take the scalar k0 without touching memory.
template <int T, int BK>
__global__ __launch_bounds__(256) void k_syrk_trail(double* __restrict__ A, int64_t lda, int k0, int k1, int first,
                                                    int colmode, int n, int64_t bsA = 0, int per = 0,
                                                    const int* __restrict__ colk0 = nullptr, int far_col = 0, int ncol = 0) {
  extern __shared__ double smem[];
  A += blockIdx.y * bsA;
  int a, b;
  if (colmode == 0) {
    if (per > 0) {                                        // one contiguous share of 8 x 8 tile groups per XCD
      const int t = xcd_share(blockIdx.x, per);
      if (t >= n * (n + 1) / 2) return;
      tri_grouped(t, n, a, b);
    } else {
      tri_decode(blockIdx.x, a, b);
    }
  } else {
    const int S = colmode == 1 ? TILE / T : ncol;
    int idx = blockIdx.x;
    b = 0;
    for (int c = 0; c < S - 1; ++c)
      if (b == c && idx >= n - c) {
        idx -= n - c;
        b = c + 1;
      }
    a = b + idx;
  }
  if (colk0) {
    const int col = first + (b * T) / TILE;
    if (col >= far_col) k0 = colk0[col];
    if (k0 >= k1) return;
  }
  const int64_t base = (int64_t)first * TILE;
  v4d acc[T / 32][T / 32];
  load_tile<T, T>(acc, A, lda, base + (int64_t)a * T, base + (int64_t)b * T);   // acc = C, then acc -= A B^T
  gemm_tile<KC, KC, T, T, BK, true>(acc, A, lda, base + (int64_t)a * T, A, lda, base + (int64_t)b * T,
                                    (int64_t)k0 * TILE, (int64_t)k1 * TILE, smem);
  store_tile<T, T>(acc, A, lda, base + (int64_t)a * T, base + (int64_t)b * T, 1.0, 0.0);
}

// ---- recursive triangular inverse ------------------------------------------------------------------
// problems {lo, mid, hi}: TriProb in gp_types.hpp

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
                                                    const TriProb* __restrict__ probs, int nprob, int64_t bsA = 0,
                                                    int64_t bsL = 0, int64_t bsT = 0, int per = 0) {
  extern __shared__ double smem[];
  L += blockIdx.y * bsA;
  Linv += blockIdx.y * bsL;
  Tmp += blockIdx.y * bsT;
  TriProb p;
  int e[2];
  if (!tri_find<T>(probs, nprob, xcd_share(blockIdx.x, per), p, e[0], e[1])) return;
  const int rows = (p.hi - p.mid) * (TILE / T), w = (p.mid - p.lo) * (TILE / T);
  for (int u = 0; u < 2; ++u) {
    if (e[u] < 0) continue;
    int ti, tj;
    if (per > 0) {
      rect_grouped(e[u], w, rows, tj, ti);                 // bands of 8 tile columns: small tj (long K) first
    } else {
      tj = e[u] / rows, ti = e[u] % rows;                  // column-major: small tj (long K) first
    }
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
                                                    const TriProb* __restrict__ probs, int nprob, int64_t bsL = 0,
                                                    int64_t bsT = 0, int per = 0) {
  extern __shared__ double smem[];
  Linv += blockIdx.y * bsL;
  Tmp += blockIdx.y * bsT;
  TriProb p;
  int e[2];
  if (!tri_find<T>(probs, nprob, xcd_share(blockIdx.x, per), p, e[0], e[1])) return;
  const int rows = (p.hi - p.mid) * (TILE / T), w = (p.mid - p.lo) * (TILE / T);
  for (int u = 0; u < 2; ++u) {
    if (e[u] < 0) continue;
    int ti, tj;
    if (per > 0) {
      rect_grouped(e[u], rows, w, ti, tj);                 // bands of 8 tile rows: bottom rows (long K) first
      ti = rows - 1 - ti;
    } else {
      ti = rows - 1 - e[u] / w, tj = e[u] % w;             // bottom rows (long K) first
    }
    const int64_t m0 = (int64_t)p.mid * TILE + (int64_t)ti * T, n0 = (int64_t)p.lo * TILE + (int64_t)tj * T;
    v4d acc[T / 32][T / 32];
    acc_zero(acc);
    gemm_tile<KC, RC, T, T, TileCfg<T>::bk>(acc, Linv, ldi, m0, Tmp, ldt, n0, (int64_t)p.mid * TILE, m0 + T, smem);
    store_tile<T, T>(acc, Linv, ldi, m0, n0, -1.0, 0.0);
  }
}

// ---- K^-1 = Linv^T Linv on 32 x 32 tiles, stored (no gradient) ------------------------------------------------------
// The GEMM half of k_lauum_grad<., ., 64> as a launch of its own, for matrices so small that the fused launch is as long as
// its longest 64 x 64 tile on ONE CU: the four 32 x 32 quarters of every lower 64 x 64 tile (all four of a diagonal tile too:
// the gradient epilogue reads the whole tile) are four times as many workgroups with a quarter of the MFMAs per wave and
// K-step.  An element's K range starts at the later of its row and column (before that one factor is a zero of the
// triangular Linv) and runs in the same groups of four: the bits of the 64 x 64 tiles.  Tile-major grid like k_lauum_grad.
__global__ __launch_bounds__(256, 2) void k_lauum_tiles32(const double* __restrict__ Linv, int64_t ldi, int64_t np,
                                                          double* __restrict__ Kinv, int64_t ldk, int64_t bsL, int64_t bsK,
                                                          int nbatch) {
  extern __shared__ double smem[];
  const int tile = (int)(blockIdx.x / nbatch), slot = (int)(blockIdx.x % nbatch);
  Linv += slot * bsL;
  Kinv += slot * bsK;
  int ti, tj;
  tri_decode(tile >> 2, ti, tj);                                   // the 64 x 64 tile, then its quarter
  const int r32 = 2 * ti + ((tile >> 1) & 1), c32 = 2 * tj + (tile & 1);
  v4d acc[1][1];
  acc_zero(acc);
  gemm_tile<RC, RC, 32, 32, BK32>(acc, Linv, ldi, (int64_t)r32 * 32, Linv, ldi, (int64_t)c32 * 32,
                                  (int64_t)(r32 > c32 ? r32 : c32) * 32, np, smem);
  store_tile<32, 32>(acc, Kinv, ldk, (int64_t)r32 * 32, (int64_t)c32 * 32, 1.0, 0.0);
}

// ---- K^-1 = Linv^T Linv fused with the MLL gradient reduction ------------------------------------------
// lower T x T tile (ti >= tj): Kinv = sum_{k >= ti*T} Linv[k][ti]^T Linv[k][tj];  W = alpha alpha^T - Kinv.
// partial[tile*(DCAP+1) + j] = sum_ab W_ab dK_ab/dlog ls_j (j < d), [DCAP] = sum_ab W_ab Kt_ab,
// off-diagonal tiles weighted x2.  Optionally stores Kinv (lower tiles) for tests.
// from_kinv: the tiles of K^-1 are already in Kinv (k_lauum_tiles, slot stride bsK): only the gradient epilogue runs, each
// thread on the elements it would own after the GEMM - the partial sums are the fused kernel's, bit for bit.
template <int KERN, int DCAP, int T>
__global__ __launch_bounds__(256, 2) void k_lauum_grad(const double* __restrict__ Linv, int64_t ldi, int64_t np,
                                                       int64_t n, const double* __restrict__ alpha,
                                                       const double* __restrict__ XsT, int64_t ldx, Hyper h,
                                                       double* __restrict__ partial, double* __restrict__ Kinv,
                                                       int64_t ldk, const Hyper* __restrict__ hp = nullptr,
                                                       int64_t bsL = 0, int64_t bsV = 0, int64_t bsX = 0,
                                                       int64_t bsP = 0, int nbatch = 1, int from_kinv = 0, int64_t bsK = 0) {
  extern __shared__ double smem[];
  // One grid dimension, tile-major: workgroup id = tile * nbatch + slot.  A tile's K length falls with its row (the first
  // tile row runs over all of K), and a launch lasts as long as the CU holding the most long tiles: with the slot in a second
  // grid dimension every matrix's long tiles were dealt to CUs that already held those of the matrices before it; dealt
  // tile-major, the long tiles of ALL matrices go out first, each to a CU of its own.
  const int tile = (int)(blockIdx.x / nbatch), slot = (int)(blockIdx.x % nbatch);
  if (hp) h = hp[slot];
  Linv += slot * bsL;
  alpha += slot * bsV;
  XsT += slot * bsX;
  partial += slot * bsP;
  int ti, tj;
  tri_decode(tile, ti, tj);
  v4d acc[T / 32][T / 32];
  if (from_kinv) {
    load_tile<T, T>(acc, Kinv + slot * bsK, ldk, (int64_t)ti * T, (int64_t)tj * T);
  } else {
    acc_zero(acc);
    gemm_tile<RC, RC, T, T, TileCfg<T>::bk>(acc, Linv, ldi, (int64_t)ti * T, Linv, ldi, (int64_t)tj * T, (int64_t)ti * T, np,
                                            smem);
    if (Kinv) store_tile<T, T>(acc, Kinv, ldk, (int64_t)ti * T, (int64_t)tj * T, 1.0, 0.0);
  }
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
    partial[(int64_t)tile * (DCAP + 1) + t] = wt * s;
  }
}

}  // namespace bobe
