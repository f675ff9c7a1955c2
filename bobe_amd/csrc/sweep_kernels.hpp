// Kernels of the acquisition sweep and of the posterior / score gradients (gfx950).  Included by gp_sweep.hip only.
#pragma once
#include "kernels_common.hpp"

namespace bobe {

// ---- the sweep's one big GEMM launch ------------------------------------------------------------
// grid.x = column (candidate) tile; grid.y enumerates row tiles, heaviest first:
//   y <  nzt : cross tile   G[z][c]  = sum_n  VZ[n][z] Bx[n][c]     (full K)   -> crossT (optional, nzt may be 0)
//   y >= nzt : V tile       V[i][c]  = sum_{k<=i} Linv[i][k] B[k][c] (lower-triangular K range), row tile
//              ti = nb-1-(y-nzt); optional store to V, optional qpart[ti*ldq + c] = sum over the tile's rows of V^2
// B is [Np x ncols] row-major (RC).  The cross tiles are k_cross_vv's product (see there: both factors are SOLVED ones,
// VZ = L^-1 K(X,Z) and Bx = the V this kernel stored for the PREVIOUS candidate chunk): they ride in the launch that
// solves the next chunk, where their full-K tiles fill the tails of the triangular ones, instead of a launch of their own
// (2.62 ms per chunk of 8192 candidates at N = 4096, M = 512 against 2.17 + 0.59).  ncv / ncx: column tiles of the two
// parts (the grid spans the larger; a chunk's last tiles may be missing in one of them).
__global__ __launch_bounds__(256, 2) void k_trimul(const double* __restrict__ Linv, int64_t ldi, int nb,
                                                   const double* __restrict__ B, int64_t ldb, double* __restrict__ V,
                                                   int64_t ldv, double* __restrict__ qpart, int64_t ldq,
                                                   const double* __restrict__ VZ, int64_t ldw, int nzt,
                                                   double* __restrict__ crossT, int64_t ldx,
                                                   const double* __restrict__ Bx = nullptr, int64_t ldbx = 0, int ncx = 0,
                                                   int ncv = 1 << 30) {
  extern __shared__ double smem[];
  // (with a grid of 8 m column tiles the workgroup ids go round-robin over the 8 XCDs: XCD k holds the columns k, k + 8, ...
  // of the 8 row tiles in flight - an 8 x 8 tile group per XCD, the fewest operand panels 64 tiles can share; contiguous
  // columns per XCD instead measured the same traffic and time, profiles/r06_traffic_k_trimul_contig.json)
  const int tc = blockIdx.x;
  v4d acc[4][4];
  acc_zero(acc);
  if ((int)blockIdx.y < nzt) {
    if (tc >= ncx) return;
    const int tz = blockIdx.y;
    gemm_tile<RC, RC>(acc, VZ, ldw, (int64_t)tz * TILE, Bx, ldbx, (int64_t)tc * TILE, 0, (int64_t)nb * TILE, smem);
    store_tile(acc, crossT, ldx, (int64_t)tz * TILE, (int64_t)tc * TILE, 1.0, 0.0);
    return;
  }
  if (tc >= ncv) return;
  const int ti = nb - 1 - ((int)blockIdx.y - nzt);
  // (the K range of a row tile ends with its diagonal block of the lower-triangular Linv: the zeros above the diagonal are
  // skipped, 1.2 % of the launch)
  gemm_tile<KC, RC, TILE, TILE, BK128, false, WgSync, true>(acc, Linv, ldi, (int64_t)ti * TILE, B, ldb, (int64_t)tc * TILE, 0,
                                                            (int64_t)(ti + 1) * TILE, smem);
  if (V) store_tile(acc, V, ldv, (int64_t)ti * TILE, (int64_t)tc * TILE, 1.0, 0.0);
  if (qpart) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    double* red = smem;  // [2][128]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double s = 0.0;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += acc[i][j][r] * acc[i][j][r];
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      if (lane < 16) red[(wave >> 1) * TILE + (wave & 1) * 64 + 16 * j + lane] = s;
    }
    __syncthreads();
    if (t < TILE) qpart[(int64_t)ti * ldq + (int64_t)tc * TILE + t] = red[t] + red[TILE + t];
  }
}

// ---- V = L^-1 B as a BLOCKED FORWARD SUBSTITUTION (the reference's solve_triangular, gp.py:462, 484, 571) -----------------
// The product Linv * B (k_trimul) is not a backward-stable solve: its error is eps |Linv| |B|, far above eps |V| when K(X, .)
// lies in the span of K's large eigenvectors - which it does for a smooth kernel.  The posterior variance tolerates that, the
// fantasy variance base_z - cross^2 / s_c does not: it is a difference of quantities that are each exact only for a
// CONSISTENTLY perturbed factor.  Measured at the reference's default noise of 1e-8 on a BO design of 1800 points
// (profiles/r06_conditioning.txt): kernel variance 4.67e4 (cond K ~3e14) WIPV 1.3e-2 off the extended-precision value, 3.5e5
// 1.0 off; LAPACK's triangular solve with its own factor 1.8e-4 / 1.5e-2.  So where the factor is ill conditioned
// (bobe_gp::refine_v: (kvar + noise) / smallest pivot above BOBE_REFINE_KAPPA, default 1e6 - where the plain product's error
// in the scores, ~8e-14 x that ratio, reaches the 1e-7 of the stated fp64 tolerance; the benchmark configurations sit at 7 ...
// 3e5) V is SOLVED for, block row by block row, top to bottom:
//     V_t = inv(L_tt) (B_t - sum_{j<t} L_tj V_j),      diagonal blocks of `bs` = 128 * bt rows.
// The diagonal bs-blocks of the inverse factor the handle already holds ARE inv(L_tt); everything off the diagonal blocks is
// multiplied by L itself, so the error is eps |L| |V| plus eps cond(L_tt) per block: with bs = 128 4.8e-5 / 2.5e-3 on the two
// rungs above, at or below LAPACK's on every rung and quantity of the ladder (256 and 512 lose the fantasy variance from
// kernel variances of 8.5e5 / 3.5e5 on: cond(L_tt) grows with the block).  N^2 flops per column like ONE k_trimul.  Round 5's
// form of the same repair - one step of iterative refinement V += Linv (B - L V), three triangular GEMMs - was 60.6 ms per
// headline-sized sweep against 29.8 this way (profiles/r06_solve_block_ab_headline.txt) and no more accurate; removed.
// One launch of this kernel carries up to two kinds of row tiles (grid.y; grid.x = column tile):
//   y <  u_rows : update tile ti = u_r0 + y:            B[ti] <- B[ti] - L[ti, u_k0:u_k1] V[u_k0:u_k1]     (in place)
//   y >= u_rows : solve tile  ti = s_r0 + s_rows-1-(..): V[ti] <- Linv[ti, s_r0*128 : (ti+1)*128] B[same rows]
//                 + k_trimul's epilogue (qpart[ti*ldq + c] = the tile's column sums of squares)
// The update tiles of a launch never touch the rows its solve tiles read (host: bobe_gp::solve_v).
__global__ __launch_bounds__(256, 2) void k_blk_step(const double* __restrict__ L, const double* __restrict__ Linv,
                                                     int64_t ld, double* __restrict__ B, int64_t ldb,
                                                     double* __restrict__ V, int64_t ldv, double* __restrict__ qpart,
                                                     int64_t ldq, int u_r0, int u_rows, int u_k0, int u_k1, int s_r0,
                                                     int s_rows) {
  extern __shared__ double smem[];
  const int tc = blockIdx.x;
  v4d acc[4][4];
  acc_zero(acc);
  if ((int)blockIdx.y < u_rows) {
    const int ti = u_r0 + (int)blockIdx.y;
    gemm_tile<KC, RC>(acc, L, ld, (int64_t)ti * TILE, V, ldv, (int64_t)tc * TILE, (int64_t)u_k0 * TILE,
                      (int64_t)u_k1 * TILE, smem);
    store_tile(acc, B, ldb, (int64_t)ti * TILE, (int64_t)tc * TILE, -1.0, 1.0);
    return;
  }
  const int ti = s_r0 + s_rows - 1 - ((int)blockIdx.y - u_rows);
  gemm_tile<KC, RC, TILE, TILE, BK128, false, WgSync, true>(acc, Linv, ld, (int64_t)ti * TILE, B, ldb, (int64_t)tc * TILE,
                                                            (int64_t)s_r0 * TILE, (int64_t)(ti + 1) * TILE, smem);
  store_tile(acc, V, ldv, (int64_t)ti * TILE, (int64_t)tc * TILE, 1.0, 0.0);
  if (qpart) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    double* red = smem;  // [2][128]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double s = 0.0;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += acc[i][j][r] * acc[i][j][r];
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      if (lane < 16) red[(wave >> 1) * TILE + (wave & 1) * 64 + 16 * j + lane] = s;
    }
    __syncthreads();
    if (t < TILE) qpart[(int64_t)ti * ldq + (int64_t)tc * TILE + t] = red[t] + red[TILE + t];
  }
}

// v[c*ldv + i] += dv[c*ldv + i]  /  r[c*ldv + i] = k[c*ldv + i] - lv[c*ldv + i]   (the vector forms of the same step)
__global__ void k_vec_axpy(double* __restrict__ y, const double* __restrict__ a, const double* __restrict__ b, double sb,
                           int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = a[i] + sb * b[i];
}

// qpart[ti*ldq + c] = the sum of squares of column c over the 128 rows of row tile ti of V, in EXACTLY the association of
// k_trimul's epilogue (per 64-row half: four lane groups q = row mod 4 within a 16-row fragment, each summing its fragments
// i = 0..3 and registers r = 0..3 in that order, then (s0 + s1) + (s2 + s3); then half 0 + half 1) - so that a V obtained
// elsewhere (the integration points' own solve, when the candidates ARE the integration points) yields the bits the
// candidates' solve would have left.  grid (ceil(ncols / 256), nb).
__global__ __launch_bounds__(256) void k_colsq_tile_parts(const double* __restrict__ V, int64_t ldv, int64_t ncols,
                                                          double* __restrict__ qpart, int64_t ldq) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int ti = blockIdx.y;
  if (c >= ncols) return;
  const double* col = V + ((int64_t)ti * TILE) * ldv + c;
  double half[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    double sq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      double s = 0.0;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double v = col[(int64_t)(64 * h + 16 * i + q + 4 * r) * ldv];
          s += v * v;
        }
      sq[q] = s;
    }
    half[h] = (sq[0] + sq[1]) + (sq[2] + sq[3]);
  }
  qpart[(int64_t)ti * ldq + c] = half[0] + half[1];
}

// ---- the sweep's cross-covariance product: G[z][c] = sum_n VZ[n][z] V[n][c], VZ = L^-1 K(X,Z), V = L^-1 K(X,C) -------
// cross(c, z) = k(x_c, z) - G[z][c] is the reference's own form: fantasy_var solves with the (N+1)-row factor
// (gp.py:552-576), whose last row is L^-1 k_c, so the cross term is an inner product of two triangular-solve results.
// Algebraically G = K(X,Z)^T K^-1 K(X,C) as well, and until round 5 the sweep formed it that way (W_Z = K^-1 K(X,Z) once per
// sweep, then W_Z^T K(X,C) as extra row tiles of k_trimul, V never stored) - but K^-1 K(X,Z) carries cond(K) eps where the
// solved factors carry sqrt(cond(K)) eps: at the reference's default noise of 1e-8 and kernel variances from ~5e4 the WIPV
// scores came out 1e-2 ... 1 (relative) off an extended-precision evaluation where the triangular-solve form is 1e-4 ...
// 1e-1 off (profiles/r05_conditioning.txt, tests/test_gpu_conditioning.py).  Same 2 N M flops per candidate.
// T = 128: grid (ncols / 128, Mp / 128); T = 64: grid (ncols / 64, Mp / 64) - few integration points fill the chip only
// with the small tile.
template <int T>
__global__ __launch_bounds__(256, 2) void k_cross_vv(const double* __restrict__ VZ, int64_t ldz,
                                                     const double* __restrict__ V, int64_t ldv, int64_t kend,
                                                     double* __restrict__ crossT, int64_t ldx) {
  extern __shared__ double smem[];
  const int tc = blockIdx.x, tz = blockIdx.y;
  v4d acc[T / 32][T / 32];
  acc_zero(acc);
  if constexpr (T == 128) {
    gemm_tile<RC, RC>(acc, VZ, ldz, (int64_t)tz * TILE, V, ldv, (int64_t)tc * TILE, 0, kend, smem);
    store_tile(acc, crossT, ldx, (int64_t)tz * TILE, (int64_t)tc * TILE, 1.0, 0.0);
  } else {
    gemm_tile<RC, RC, 64, 64, BK64>(acc, VZ, ldz, (int64_t)tz * 64, V, ldv, (int64_t)tc * 64, 0, kend, smem);
    store_tile<64, 64>(acc, crossT, ldx, (int64_t)tz * 64, (int64_t)tc * 64, 1.0, 0.0);
  }
}

// ---- W = Linv^T * V  (upper-triangular times dense):  W[m][z] = sum_{k >= m} Linv[k][m] V[k][z]
__global__ __launch_bounds__(256, 2) void k_trimul_t(const double* __restrict__ Linv, int64_t ldi, int nb,
                                                  const double* __restrict__ V, int64_t ldv, double* __restrict__ W,
                                                  int64_t ldw) {
  extern __shared__ double smem[];
  const int ti = blockIdx.y;
  const int tc = blockIdx.x;
  v4d acc[4][4];
  acc_zero(acc);
  gemm_tile<RC, RC>(acc, Linv, ldi, (int64_t)ti * TILE, V, ldv, (int64_t)tc * TILE, (int64_t)ti * TILE,
                    (int64_t)nb * TILE, smem);
  store_tile(acc, W, ldw, (int64_t)ti * TILE, (int64_t)tc * TILE, 1.0, 0.0);
}

// ---- the same two products on 64 x 64 tiles, for right-hand sides with FEW columns (the M = 512 integration points of a
// sweep: 4 x nb tiles of 128 x 128 leave half the chip idle for two launches of ~0.57 ms each at N = 4096; 8 x 2nb tiles
// of 64 x 64 fill it).  V = Linv B with the column sums of squares per 64-row tile (qpart[2 nb][.]), W = Linv^T V.
__global__ __launch_bounds__(256, 2) void k_trimul_v64(const double* __restrict__ Linv, int64_t ldi, int nt,
                                                       const double* __restrict__ B, int64_t ldb, double* __restrict__ V,
                                                       int64_t ldv, double* __restrict__ qpart, int64_t ldq) {
  extern __shared__ double smem[];
  const int tc = blockIdx.x;
  const int ti = nt - 1 - (int)blockIdx.y;                      // long K first
  v4d acc[2][2];
  acc_zero(acc);
  gemm_tile<KC, RC, 64, 64, BK64>(acc, Linv, ldi, (int64_t)ti * 64, B, ldb, (int64_t)tc * 64, 0, (int64_t)(ti + 1) * 64, smem);
  store_tile<64, 64>(acc, V, ldv, (int64_t)ti * 64, (int64_t)tc * 64, 1.0, 0.0);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  double* red = smem;  // [2][64]
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) s += acc[i][j][r] * acc[i][j][r];
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (lane < 16) red[(wave >> 1) * 64 + (wave & 1) * 32 + 16 * j + lane] = s;
  }
  __syncthreads();
  if (t < 64) qpart[(int64_t)ti * ldq + (int64_t)tc * 64 + t] = red[t] + red[64 + t];
}

__global__ __launch_bounds__(256, 2) void k_trimul_t64(const double* __restrict__ Linv, int64_t ldi, int nt,
                                                       const double* __restrict__ V, int64_t ldv, double* __restrict__ W,
                                                       int64_t ldw) {
  extern __shared__ double smem[];
  const int ti = blockIdx.y;                                    // (row tile 0 has the longest K and starts first)
  const int tc = blockIdx.x;
  v4d acc[2][2];
  acc_zero(acc);
  gemm_tile<RC, RC, 64, 64, BK64>(acc, Linv, ldi, (int64_t)ti * 64, V, ldv, (int64_t)tc * 64, (int64_t)ti * 64,
                                  (int64_t)nt * 64, smem);
  store_tile<64, 64>(acc, W, ldw, (int64_t)ti * 64, (int64_t)tc * 64, 1.0, 0.0);
}

// ---- WIPV / WIPStd scoring of every candidate (BOBE/gp.py:552-576, acquisition.py:438-465) -------
// crossT[z*ldx + c] = sum_n VZ[n][z] V[n][c] (from k_cross_vv).  For candidate c and integration point z:
//   cross = k(x_c, z) - crossT;  var+ = base_z - cross^2 / s_c  -> NaN / < 1e-12 -> 1e-12 -> * ystd2
// wipv[c] = mean_z var+, wipstd[c] = mean_z sqrt(var+).  One workgroup = 64 candidates x 4 interleaved
// z-slices; per-candidate sums are combined in a fixed order, so results do not depend on chunking.
template <int KERN, int DCAP>
__global__ __launch_bounds__(256) void k_wip_score(const double* __restrict__ crossT, int64_t ldx,
                                                   const double* __restrict__ CsT, int64_t ldc,
                                                   const double* __restrict__ ZsT, int64_t ldz, int64_t m,
                                                   const double* __restrict__ sc, const double* __restrict__ basez,
                                                   int64_t ncols, Hyper h, double ystd2, double* __restrict__ wipv,
                                                   double* __restrict__ wipstd, double* __restrict__ var_out,
                                                   int64_t ldvo) {
  extern __shared__ double zsm[];          // [d][ZT] + base[ZT]
  constexpr int ZT = 128;
  __shared__ double red[2][4][64];
  const int t = threadIdx.x, cx = t & 63, sl = t >> 6;
  const int64_t c = (int64_t)blockIdx.x * 64 + cx;
  const bool live = c < ncols;
  double xc[DCAP];
#pragma unroll
  for (int j = 0; j < DCAP; ++j) xc[j] = (live && j < h.d) ? CsT[j * ldc + c] : 0.0;
  const double s = live ? sc[c] : 1.0;
  const bool sbad = !(s >= 0.0);           // sqrt(negative) = NaN in fast_update_cholesky (gp.py:187)
  double sv = 0.0, ss = 0.0;
  for (int64_t z0 = 0; z0 < m; z0 += ZT) {
    __syncthreads();
    for (int e = t; e < h.d * ZT; e += 256) {
      const int j = e / ZT, zz = e % ZT;
      zsm[j * ZT + zz] = (z0 + zz < m) ? ZsT[j * ldz + z0 + zz] : 0.0;
    }
    if (t < ZT) zsm[h.d * ZT + t] = (z0 + t < m) ? basez[z0 + t] : 0.0;
    __syncthreads();
    const int zn = (m - z0 < ZT) ? (int)(m - z0) : ZT;
    if (live) {
      // one integration point of this thread's slice: cross-covariance from crossT (already loaded), fantasy variance, sums
      auto score = [&](int zz, double ct) {
        double r2 = 0.0;
#pragma unroll
        for (int j = 0; j < DCAP; ++j) {
          if (j < h.d) {
            const double df = xc[j] - zsm[j * ZT + zz];
            r2 += df * df;
          }
        }
        const double cross = kern_eval<KERN>(r2, h.kvar) - ct;
        double v = zsm[h.d * ZT + zz] - (cross * cross) / s;
        if (sbad) v = NOISE_FLOOR;
        if (v != v) v = NOISE_FLOOR;           // gp.py:574
        if (v < NOISE_FLOOR) v = NOISE_FLOOR;  // gp.py:575
        v *= ystd2;                            // gp.py:576
        sv += v;
        ss += sqrt(v);
        if (var_out) var_out[c * ldvo + z0 + zz] = v;
      };
      // (eight rows of crossT in flight per thread: one load per iteration left the loop waiting on memory - 83 us for
      // 8192 candidates x 512 points; the sums keep their order)
      const double* cp = crossT + z0 * ldx + c;
      int zz = sl;
      for (; zz + 28 < zn; zz += 32) {
        double ct[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) ct[q] = cp[(int64_t)(zz + 4 * q) * ldx];
#pragma unroll
        for (int q = 0; q < 8; ++q) score(zz + 4 * q, ct[q]);
      }
      for (; zz < zn; zz += 4) score(zz, cp[(int64_t)zz * ldx]);
    }
  }
  red[0][sl][cx] = sv;
  red[1][sl][cx] = ss;
  __syncthreads();
  if (sl == 0 && live) {
    if (wipv) wipv[c] = ((red[0][0][cx] + red[0][1][cx]) + (red[0][2][cx] + red[0][3][cx])) / (double)m;
    if (wipstd) wipstd[c] = ((red[1][0][cx] + red[1][1][cx]) + (red[1][2][cx] + red[1][3][cx])) / (double)m;
  }
}


// ---- sweep finalisers -------------------------------------------------------------------------------
// q = sum of row-tile partials; s = kself - q.  var policy: 0 -> clip(s, floor) keeps NaN (gp.py:465),
// 1 -> NaN and < floor -> floor (gp.py:487-488).  s_out keeps the raw s for the fantasy scoring.
__global__ void k_predict_finalize(const double* __restrict__ qpart, int64_t ldq, int nb, int64_t ncols, double kself,
                                   int policy, double* __restrict__ s_out, double* __restrict__ var_out) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncols) return;
  double q = 0.0;
  for (int rb = 0; rb < nb; ++rb) q += qpart[(int64_t)rb * ldq + c];
  const double s = kself - q;
  if (s_out) s_out[c] = s;
  if (var_out) {
    double v = s;
    if (policy == 1 && v != v) v = NOISE_FLOOR;
    if (v < NOISE_FLOOR) v = NOISE_FLOOR;
    var_out[c] = v;
  }
}

// argmin with first-occurrence tie-break (jnp.argmin, acquisition.py:397); NaN counts as minimal.
__global__ __launch_bounds__(1024) void k_argmin(const double* __restrict__ v, int64_t n, double* __restrict__ best_val,
                                                 int64_t* __restrict__ best_idx) {
  __shared__ double sv[1024];
  __shared__ int64_t si[1024];
  double bv = 0.0;
  int64_t bi = -1;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    const double x = v[i];
    const bool xnan = (x != x);
    const bool bnan = (bi >= 0) && (bv != bv);
    if (bi < 0 || (!bnan && (xnan || x < bv))) {
      bv = x;
      bi = i;
    }
  }
  sv[threadIdx.x] = bv;
  si[threadIdx.x] = bi;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      const double x = sv[threadIdx.x + o];
      const int64_t xi = si[threadIdx.x + o];
      const double b = sv[threadIdx.x];
      const int64_t bi2 = si[threadIdx.x];
      bool take = false;
      if (xi >= 0) {
        if (bi2 < 0) take = true;
        else {
          const bool xnan = (x != x), bnan = (b != b);
          if (xnan && bnan) take = xi < bi2;
          else if (xnan) take = true;
          else if (bnan) take = false;
          else take = (x < b) || (x == b && xi < bi2);
        }
      }
      if (take) {
        sv[threadIdx.x] = x;
        si[threadIdx.x] = xi;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    *best_val = sv[0];
    *best_idx = si[0];
  }
}


// ---- rank-b append (bobe_gp_append; the reference's fast_update_cholesky, gp.py:181-197, b rows at a time) ------
// G[i*b + j] = sum_{k < n} V[k*ldv + i] * V[k*ldv + j]   (b <= 64; one workgroup per (i, j), fixed summation order)
__global__ __launch_bounds__(256) void k_gram_small(const double* __restrict__ V, int64_t ldv, int64_t n, int b,
                                                    double* __restrict__ G) {
  __shared__ double red[4];
  const int i = blockIdx.x, j = blockIdx.y;
  double s = 0.0;
  for (int64_t k = threadIdx.x; k < n; k += 256) s += V[k * ldv + i] * V[k * ldv + j];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) G[i * b + j] = ((red[0] + red[1]) + red[2]) + red[3];
}

// rows n0 .. n0+b-1 of the padded factor and of its inverse:
//   L[n0+i][c]    = V[c][i]                         (c < n0)      L[n0+i][n0+j]    = L22[i][j]  (0 above the diagonal)
//   Linv[n0+i][c] = -sum_j L22inv[i][j] W[c][j]     (c < n0)      Linv[n0+i][n0+j] = L22inv[i][j]
// with V = Linv_old K(X, X_new), W = Linv_old^T V; s22 holds L22 (b*b) followed by L22inv (b*b), row-major.
__global__ void k_append_rows(double* __restrict__ L, double* __restrict__ Linv, int64_t ld, int64_t n0, int b,
                              const double* __restrict__ V, const double* __restrict__ W, int64_t ldv,
                              const double* __restrict__ s22) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n0 + b) return;
  const double* L22 = s22;
  const double* Li = s22 + b * b;
  for (int i = 0; i < b; ++i) {
    double lv, iv;
    if (c < n0) {
      lv = V[c * ldv + i];
      double a = 0.0;
      for (int j = 0; j <= i; ++j) a += Li[i * b + j] * W[c * ldv + j];
      iv = -a;
    } else {
      const int j = (int)(c - n0);
      lv = (j <= i) ? L22[i * b + j] : 0.0;
      iv = (j <= i) ? Li[i * b + j] : 0.0;
    }
    L[(n0 + i) * ld + c] = lv;
    Linv[(n0 + i) * ld + c] = iv;
  }
}


// ---- input gradients of the posterior mean and variance (for gradient-based consumers) ---------------
// dmean[c][j] = sum_n alpha_n dk(x_n, x_c)/dx_cj ;  dvar[c][j] = -2 sum_n u_nc dk(x_n, x_c)/dx_cj, u = K^-1 k_c
// with dk/dx_cj = G(r2) (s_nj - s_cj) / ls_j  (s = x / ls; G = k for RBF, the Matern-5/2 factor otherwise).
// One workgroup = 64 candidates x 4 interleaved slices over the training points; fixed summation order.
// dvar is zeroed where the (standardised) variance sits at its 1e-12 floor, like the gradient of jnp.where.
// U == nullptr: mean-only mode (the HMC sampler's call): dvar is not touched, and mean_out (optional) receives
// the posterior mean sum_n alpha_n k(x_n, x_c) itself, so that no K(X, C) matrix is needed at all.
template <int KERN, int DCAP>
__global__ __launch_bounds__(256) void k_predict_grad(const double* __restrict__ XsT, int64_t ldx, int64_t n,
                                                      const double* __restrict__ CsT, int64_t ldc, int64_t ncols,
                                                      const double* __restrict__ alpha, const double* __restrict__ U,
                                                      int64_t ldu, const double* __restrict__ svar, Hyper h,
                                                      double* __restrict__ dmean, double* __restrict__ dvar,
                                                      double* __restrict__ mean_out = nullptr) {
  extern __shared__ double psm[];            // [d][128] training coordinates + alpha[128]
  constexpr int NT = 128;
  __shared__ double red[2][4][64];
  const int t = threadIdx.x, cx = t & 63, sl = t >> 6;
  const int64_t c = (int64_t)blockIdx.x * 64 + cx;
  const bool live = c < ncols;
  double xc[DCAP], gm[DCAP], gv[DCAP];
  double ms = 0.0;
#pragma unroll
  for (int j = 0; j < DCAP; ++j) {
    xc[j] = (live && j < h.d) ? CsT[j * ldc + c] : 0.0;
    gm[j] = 0.0;
    gv[j] = 0.0;
  }
  for (int64_t n0 = 0; n0 < n; n0 += NT) {
    __syncthreads();
    for (int e = t; e < h.d * NT; e += 256) {
      const int j = e / NT, nn = e % NT;
      psm[j * NT + nn] = (n0 + nn < n) ? XsT[j * ldx + n0 + nn] : 0.0;
    }
    if (t < NT) psm[h.d * NT + t] = (n0 + t < n) ? alpha[n0 + t] : 0.0;
    __syncthreads();
    const int nn_end = (n - n0 < NT) ? (int)(n - n0) : NT;
    if (live) {
      for (int nn = sl; nn < nn_end; nn += 4) {
        double df[DCAP];
        double r2 = 0.0;
#pragma unroll
        for (int j = 0; j < DCAP; ++j) {
          df[j] = (j < h.d) ? psm[j * NT + nn] - xc[j] : 0.0;
          r2 += df[j] * df[j];
        }
        const double kv = kern_eval<KERN>(r2, h.kvar);
        const double gfac = kern_grad_factor<KERN>(r2, h.kvar, kv);
        const double a = psm[h.d * NT + nn] * gfac;
        const double u = U ? -2.0 * U[(n0 + nn) * ldu + c] * gfac : 0.0;
        ms += psm[h.d * NT + nn] * kv;
#pragma unroll
        for (int j = 0; j < DCAP; ++j) {
          gm[j] += a * df[j];
          gv[j] += u * df[j];
        }
      }
    }
  }
  const bool floored = (live && svar) ? !(svar[c] >= NOISE_FLOOR) : true;
  if (mean_out) {
    __syncthreads();
    red[0][sl][cx] = ms;
    __syncthreads();
    if (sl == 0 && live) mean_out[c] = (red[0][0][cx] + red[0][1][cx]) + (red[0][2][cx] + red[0][3][cx]);
  }
#pragma unroll
  for (int j = 0; j < DCAP; ++j) {
    if (j < h.d) {                       // uniform condition: barriers inside are safe
      __syncthreads();
      red[0][sl][cx] = gm[j];
      red[1][sl][cx] = gv[j];
      __syncthreads();
      if (sl == 0 && live) {
        const double m = (red[0][0][cx] + red[0][1][cx]) + (red[0][2][cx] + red[0][3][cx]);
        const double v = (red[1][0][cx] + red[1][1][cx]) + (red[1][2][cx] + red[1][3][cx]);
        dmean[c * h.d + j] = m / h.ls[j];
        if (dvar) dvar[c * h.d + j] = floored ? 0.0 : v / h.ls[j];
      }
    }
  }
}


// ---- WIPV / WIPStd and their gradients w.r.t. the candidate coordinates ---------------------------------------
// (what the reference gets from jax.grad of WIPV.fun / WIPStd.fun in the local refinement, acquisition.py:403-412)
// One workgroup per candidate c.  With s = kself - k_c^T K^-1 k_c, u = K^-1 k_c, W = K^-1 K(X,Z), v = L^-1 k_c, VZ = L^-1 K(X,Z):
//   cross_z      = k(z,x) - sum_n VZ[n][z] v_n       (the VALUE from the solved factors, as in the sweep: k_cross_vv)
//   var+_z       = base_z - cross_z^2 / s                       (floors of gp.py:574-575; floored terms have zero gradient)
//   d cross_z/dx_j = [G_z (s_zj - s_xj) - sum_n W[n][z] G_n (s_nj - s_xj)] / ls_j,   d s/dx_j = -2 sum_n u_n G_n (s_nj - s_xj) / ls_j
//   d var+_z/dx_j = -2 cross_z/s * d cross_z/dx_j + cross_z^2/s^2 * d s/dx_j
// (s_a = a / ls, G = dk/d(-r^2/2): k itself for RBF, the Matern-5/2 factor otherwise).  Fixed summation order.
template <int KERN, int DCAP>
__global__ __launch_bounds__(256) void k_wip_grad(const double* __restrict__ XsT, int64_t ldx, int64_t n,
                                                  const double* __restrict__ CsT, int64_t ldc,
                                                  const double* __restrict__ ZsT, int64_t ldz, int64_t m,
                                                  const double* __restrict__ W, int64_t ldw,
                                                  const double* __restrict__ U, int64_t ldu,
                                                  const double* __restrict__ VZ, const double* __restrict__ Vc, int64_t ldv,
                                                  const double* __restrict__ sc, const double* __restrict__ basez, Hyper h,
                                                  double ystd2, double* __restrict__ wipv, double* __restrict__ wipstd,
                                                  double* __restrict__ dwipv, double* __restrict__ dwipstd) {
  constexpr int NT = 128;
  __shared__ double T[DCAP][NT];      // G_n (s_nj - s_xj) of the staged training points
  __shared__ double kcv[NT];          // v_n = (L^-1 k_c)_n of the staged training points
  __shared__ double dsred[NT];
  __shared__ double ds[DCAP];         // sum_n u_n T[j][n]
  __shared__ double red[4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int64_t c = blockIdx.x;
  double xc[DCAP];
#pragma unroll
  for (int j = 0; j < DCAP; ++j) xc[j] = (j < h.d) ? CsT[j * ldc + c] : 0.0;
  const double s = sc[c];
  const bool sbad = !(s >= 0.0);

  // stage one chunk of training points: T, kcv (threads 0..127), returns this thread's u_n
  auto stage = [&](int64_t n0) -> double {
    double un = 0.0;
    if (t < NT) {
      const int64_t nn = n0 + t;
      double df[DCAP];
      double r2 = 0.0;
#pragma unroll
      for (int j = 0; j < DCAP; ++j) {
        df[j] = (j < h.d && nn < n) ? XsT[j * ldx + nn] - xc[j] : 0.0;
        r2 += df[j] * df[j];
      }
      const double kv = (nn < n) ? kern_eval<KERN>(r2, h.kvar) : 0.0;
      const double g = (nn < n) ? kern_grad_factor<KERN>(r2, h.kvar, kv) : 0.0;
#pragma unroll
      for (int j = 0; j < DCAP; ++j) T[j][t] = g * df[j];
      kcv[t] = (nn < n) ? Vc[nn * ldv + c] : 0.0;
      un = (nn < n) ? U[nn * ldu + c] : 0.0;
    }
    return un;
  };

  // pass 0: ds_j = sum_n u_n T[j][n]
  double dsa[DCAP];
#pragma unroll
  for (int j = 0; j < DCAP; ++j) dsa[j] = 0.0;
  for (int64_t n0 = 0; n0 < n; n0 += NT) {
    __syncthreads();
    const double un = stage(n0);
    if (t < NT) {
#pragma unroll
      for (int j = 0; j < DCAP; ++j) dsa[j] += un * T[j][t];
    }
  }
#pragma unroll
  for (int j = 0; j < DCAP; ++j) {
    if (j < h.d) {                     // uniform
      __syncthreads();
      if (t < NT) dsred[t] = dsa[j];
      __syncthreads();
      if (t == 0) {
        double a = 0.0;
        for (int q = 0; q < NT; ++q) a += dsred[q];
        ds[j] = a;
      }
    }
  }
  __syncthreads();

  // z passes: one integration point per thread and pass
  double sv = 0.0, ss = 0.0, gv[DCAP], gs[DCAP];
#pragma unroll
  for (int j = 0; j < DCAP; ++j) gv[j] = gs[j] = 0.0;
  for (int64_t zb = 0; zb < m; zb += 256) {
    const int64_t z = zb + t;
    const bool live = z < m;
    double acck = 0.0, accw[DCAP];
#pragma unroll
    for (int j = 0; j < DCAP; ++j) accw[j] = 0.0;
    for (int64_t n0 = 0; n0 < n; n0 += NT) {
      __syncthreads();
      (void)stage(n0);
      __syncthreads();
      if (live) {
        const int nn_end = (n - n0 < NT) ? (int)(n - n0) : NT;
        // (eight rows of W in flight per thread: one load per iteration left the loop waiting on memory, ~1000 cycles
        // per training point; the sums keep their order)
        const double* wp = W + n0 * ldw + z;
        const double* vp = VZ + n0 * ldw + z;          // (VZ has W's shape and leading dimension)
        int nn = 0;
        for (; nn + 8 <= nn_end; nn += 8) {
          double w8[8], v8[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            w8[q] = wp[(int64_t)(nn + q) * ldw];
            v8[q] = vp[(int64_t)(nn + q) * ldw];
          }
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            acck = __builtin_fma(v8[q], kcv[nn + q], acck);
#pragma unroll
            for (int j = 0; j < DCAP; ++j) accw[j] = __builtin_fma(w8[q], T[j][nn + q], accw[j]);
          }
        }
        for (; nn < nn_end; ++nn) {
          const double w = wp[(int64_t)nn * ldw];
          acck = __builtin_fma(vp[(int64_t)nn * ldw], kcv[nn], acck);
#pragma unroll
          for (int j = 0; j < DCAP; ++j) accw[j] = __builtin_fma(w, T[j][nn], accw[j]);
        }
      }
    }
    if (live) {
      double dz[DCAP];
      double r2 = 0.0;
#pragma unroll
      for (int j = 0; j < DCAP; ++j) {
        dz[j] = (j < h.d) ? ZsT[j * ldz + z] - xc[j] : 0.0;
        r2 += dz[j] * dz[j];
      }
      const double kz = kern_eval<KERN>(r2, h.kvar);
      const double gz = kern_grad_factor<KERN>(r2, h.kvar, kz);
      const double cross = kz - acck;
      double v = basez[z] - (cross * cross) / s;
      bool floored = sbad;
      if (v != v) floored = true;
      if (v < NOISE_FLOOR) floored = true;
      if (floored) v = NOISE_FLOOR;
      const double vs = v * ystd2;
      const double sq = sqrt(vs);
      sv += vs;
      ss += sq;
      if (!floored) {
        const double a1 = -2.0 * cross / s, a2 = (cross * cross) / (s * s);
#pragma unroll
        for (int j = 0; j < DCAP; ++j) {
          if (j < h.d) {
            const double dcross = (gz * dz[j] - accw[j]) / h.ls[j];
            const double dsj = -2.0 * ds[j] / h.ls[j];
            const double dv = (a1 * dcross + a2 * dsj) * ystd2;
            gv[j] += dv;
            gs[j] += dv / (2.0 * sq);
          }
        }
      }
    }
  }
  // block reductions in a fixed order: lanes by butterfly, then the four waves
  auto block_sum = [&](double x) -> double {
    const double wsum = wave_sum(x);
    __syncthreads();
    if (lane == 0) red[wave] = wsum;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
  };
  const double inv_m = 1.0 / (double)m;
  const double tv = block_sum(sv), ts = block_sum(ss);
  if (t == 0) {
    if (wipv) wipv[c] = tv * inv_m;
    if (wipstd) wipstd[c] = ts * inv_m;
  }
#pragma unroll
  for (int j = 0; j < DCAP; ++j) {
    if (j < h.d) {                     // uniform
      const double a = block_sum(gv[j]), b = block_sum(gs[j]);
      if (t == 0) {
        if (dwipv) dwipv[c * h.d + j] = a * inv_m;
        if (dwipstd) dwipstd[c * h.d + j] = b * inv_m;
      }
    }
  }
}

// ---- the same for a HANDFUL of candidates (the L-BFGS refinement of an acquisition point asks for one at a time) ----
// The batched path above spends a 128-column MFMA tile pass on each triangular solve of a single column and ONE workgroup
// on a candidate's whole N x M x d contraction.  Here every stage is matrix-vector work spread over the chip, and the
// gradient is rearranged so that no stage costs more than O(N M):
//   sum_z a_z sum_n W[n][z] T[j][n]  =  sum_n T[j][n] q_n,   q = W a      (a_z: the weight of d cross_z in the score)
// Stages (blockIdx.y / .z = candidate): k_wg_col -> k_gemv_lower (v = Linv k_c) -> k_gemv_t_part + k_colsum_parts
// (u = Linv^T v) -> k_wg_cross (s, cross_z, the weights, the z-side sums) -> k_wg_rows (q, q', the n-side sums) ->
// k_wg_final.  All sums in a fixed order.
template <int KERN, int DCAP>
__global__ __launch_bounds__(256) void k_wg_col(const double* __restrict__ XsT, int64_t ldx, int64_t n, int64_t np,
                                                const double* __restrict__ cand, Hyper h, double* __restrict__ kc,
                                                double* __restrict__ cand_dev = nullptr) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
  // (cand may be the caller's coordinates in PINNED HOST memory, read here over the fabric once: the device copy for the
  // later stages is left on the way instead of a host -> device copy command in front of the chain)
  if (cand_dev && blockIdx.x == 0 && (int)threadIdx.x < h.d) cand_dev[c * h.d + threadIdx.x] = cand[c * h.d + threadIdx.x];
  if (i >= np) return;
  double r2 = 0.0;
#pragma unroll
  for (int j = 0; j < DCAP; ++j) {
    const double df = (j < h.d && i < n) ? XsT[j * ldx + i] - cand[c * h.d + j] / h.ls[j] : 0.0;
    r2 += df * df;
  }
  kc[c * np + i] = (i < n) ? kern_eval<KERN>(r2, h.kvar) : 0.0;
}

constexpr int WG_ZS = 4 + 2 * MAX_D;     // doubles per k_wg_cross partial: sum var+, sum sqrt, sum a2, sum b2, Dv[j], Ds[j]
constexpr int WG_NS = 3 * MAX_D;         // doubles per k_wg_rows partial: Qv[j], Qs[j], ds[j]

// one workgroup = 64 integration points (lane) x 4 interleaved quarters of the training points (wave)
template <int KERN, int DCAP>
__global__ __launch_bounds__(256) void k_wg_cross(const double* __restrict__ ZsT, int64_t ldz, int64_t m,
                                                  const double* __restrict__ VZ, int64_t ldw, int64_t n, int64_t np,
                                                  const double* __restrict__ kc, const double* __restrict__ v,
                                                  const double* __restrict__ cand, Hyper h, double kself,
                                                  const double* __restrict__ basez, double ystd2,
                                                  double* __restrict__ a1, double* __restrict__ b1, int64_t lda,
                                                  double* __restrict__ partZ) {
  extern __shared__ double kcs[];                 // [n]: v = L^-1 k_c (the cross term is VZ^T v, as in the sweep)
  __shared__ double red[4], redz[4][64];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int64_t c = blockIdx.y;
  v += c * np;
  double sq = 0.0;
  for (int64_t i = t; i < n; i += 256) {
    const double vi = v[i];
    kcs[i] = vi;
    sq += vi * vi;
  }
  sq = wave_sum(sq);
  if (lane == 0) red[wave] = sq;
  __syncthreads();
  const double s = kself - ((red[0] + red[1]) + (red[2] + red[3]));
  const bool sbad = !(s >= 0.0);
  const int64_t z = (int64_t)blockIdx.x * 64 + lane;
  double acc = 0.0;
  {
    const double* wp = VZ + z;
    int64_t i = wave;
    for (; i + 28 < n; i += 32) {                 // eight rows in flight per thread
      double w8[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) w8[q] = wp[(i + 4 * q) * ldw];
#pragma unroll
      for (int q = 0; q < 8; ++q) acc = __builtin_fma(w8[q], kcs[i + 4 * q], acc);
    }
    for (; i < n; i += 4) acc = __builtin_fma(wp[i * ldw], kcs[i], acc);
  }
  redz[wave][lane] = acc;
  __syncthreads();
  if (wave != 0) return;
  const double wk = (redz[0][lane] + redz[1][lane]) + (redz[2][lane] + redz[3][lane]);
  double sv = 0.0, ss = 0.0, w1 = 0.0, w2 = 0.0, u1 = 0.0, u2 = 0.0, gz = 0.0, dz[DCAP];
#pragma unroll
  for (int j = 0; j < DCAP; ++j) dz[j] = 0.0;
  if (z < m) {
    double r2 = 0.0;
#pragma unroll
    for (int j = 0; j < DCAP; ++j) {
      dz[j] = (j < h.d) ? ZsT[j * ldz + z] - cand[c * h.d + j] / h.ls[j] : 0.0;
      r2 += dz[j] * dz[j];
    }
    const double kz = kern_eval<KERN>(r2, h.kvar);
    gz = kern_grad_factor<KERN>(r2, h.kvar, kz);
    const double cross = kz - wk;
    double vp = basez[z] - (cross * cross) / s;
    bool floored = sbad;
    if (vp != vp) floored = true;
    if (vp < NOISE_FLOOR) floored = true;
    if (floored) vp = NOISE_FLOOR;
    const double vs = vp * ystd2, sr = sqrt(vs);
    sv = vs;
    ss = sr;
    if (!floored) {
      w1 = -2.0 * cross / s * ystd2;
      w2 = (cross * cross) / (s * s) * ystd2;
      u1 = w1 / (2.0 * sr);
      u2 = w2 / (2.0 * sr);
    }
  }
  a1[c * lda + z] = w1;
  b1[c * lda + z] = u1;
  double* pz = partZ + (c * gridDim.x + blockIdx.x) * WG_ZS;
  const double t0 = chain_wave_sum(sv), t1 = chain_wave_sum(ss), t2 = chain_wave_sum(w2), t3 = chain_wave_sum(u2);
  if (lane == 0) {
    pz[0] = t0;
    pz[1] = t1;
    pz[2] = t2;
    pz[3] = t3;
  }
#pragma unroll
  for (int j = 0; j < DCAP; ++j) {
    if (j < h.d) {                                // uniform
      const double a = chain_wave_sum(w1 * gz * dz[j]), b = chain_wave_sum(u1 * gz * dz[j]);
      if (lane == 0) {
        pz[4 + j] = a;
        pz[4 + MAX_D + j] = b;
      }
    }
  }
}

// one workgroup = WG_ROWS training points, one per wave; lanes run over the integration points.
// (Sixteen per wave - 64 per workgroup - left a refinement step at N = 400 on seven CUs for 41 of its 90 us.)
constexpr int WG_ROWS = 4;
template <int KERN, int DCAP>
__global__ __launch_bounds__(256) void k_wg_rows(const double* __restrict__ XsT, int64_t ldx, int64_t n, int64_t np,
                                                 const double* __restrict__ cand, Hyper h,
                                                 const double* __restrict__ W, int64_t ldw, int64_t mp,
                                                 const double* __restrict__ a1, const double* __restrict__ b1,
                                                 int64_t lda, const double* __restrict__ u,
                                                 double* __restrict__ partN) {
  __shared__ double red[4][3][DCAP];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int64_t c = blockIdx.y;
  a1 += c * lda;
  b1 += c * lda;
  u += c * np;
  const double xl = (lane < h.d) ? cand[c * h.d + lane] / h.ls[lane] : 0.0;      // this lane's coordinate (lane = j)
  double xc[DCAP];                                                                 // the candidate, scaled (every lane)
#pragma unroll
  for (int j = 0; j < DCAP; ++j) xc[j] = (j < h.d) ? cand[c * h.d + j] / h.ls[j] : 0.0;
  double aq = 0.0, as = 0.0, au = 0.0;
  for (int r = 0; r < WG_ROWS / 4; ++r) {
    const int64_t i = (int64_t)blockIdx.x * WG_ROWS + wave + 4 * r;
    if (i >= n) break;                            // uniform per wave
    double q = 0.0, qs = 0.0;
    for (int64_t z = lane; z < mp; z += 64) {
      const double w = W[i * ldw + z];
      q = __builtin_fma(a1[z], w, q);
      qs = __builtin_fma(b1[z], w, qs);
    }
    q = chain_wave_sum(q);
    qs = chain_wave_sum(qs);
    double r2 = 0.0;
#pragma unroll
    for (int j = 0; j < DCAP; ++j) {
      const double df = (j < h.d) ? XsT[j * ldx + i] - xc[j] : 0.0;
      r2 += df * df;
    }
    const double kv = kern_eval<KERN>(r2, h.kvar);
    const double tj = (lane < h.d) ? kern_grad_factor<KERN>(r2, h.kvar, kv) * (XsT[lane * ldx + i] - xl) : 0.0;
    aq = __builtin_fma(q, tj, aq);
    as = __builtin_fma(qs, tj, as);
    au = __builtin_fma(u[i], tj, au);
  }
  if (lane < DCAP) {
    red[wave][0][lane] = aq;
    red[wave][1][lane] = as;
    red[wave][2][lane] = au;
  }
  __syncthreads();
  if (t < 3 * DCAP) {
    const int k = t / DCAP, j = t % DCAP;
    partN[(c * gridDim.x + blockIdx.x) * WG_NS + k * MAX_D + j] =
        (red[0][k][j] + red[1][k][j]) + (red[2][k][j] + red[3][k][j]);
  }
}

// the partial sums of k_wg_cross (nzw of them) and k_wg_rows (nnw) -> scores and gradients of candidate blockIdx.x.
// 256 threads: thread (j = t & 31, part = t >> 5) adds the partials part, part + 8, ... of coordinate j, the eight parts are
// combined in order.
__global__ __launch_bounds__(256) void k_wg_final(const double* __restrict__ partZ, int nzw,
                                                  const double* __restrict__ partN, int nnw, Hyper h, int64_t m,
                                                  double* __restrict__ wipv, double* __restrict__ wipstd,
                                                  double* __restrict__ dwipv, double* __restrict__ dwipstd) {
  __shared__ double acc[8][9][32];
  const int t = threadIdx.x, j = t & 31, part = t >> 5;
  const int64_t c = blockIdx.x;
  const double* pz = partZ + c * nzw * WG_ZS;
  const double* pn = partN + c * nnw * WG_NS;
  double sv = 0.0, ss = 0.0, a2 = 0.0, b2 = 0.0, dv = 0.0, dsd = 0.0, qv = 0.0, qs = 0.0, ds = 0.0;
  for (int w = part; w < nzw; w += 8) {
    sv += pz[w * WG_ZS];
    ss += pz[w * WG_ZS + 1];
    a2 += pz[w * WG_ZS + 2];
    b2 += pz[w * WG_ZS + 3];
    dv += pz[w * WG_ZS + 4 + j];
    dsd += pz[w * WG_ZS + 4 + MAX_D + j];
  }
  for (int w = part; w < nnw; w += 8) {
    qv += pn[w * WG_NS + j];
    qs += pn[w * WG_NS + MAX_D + j];
    ds += pn[w * WG_NS + 2 * MAX_D + j];
  }
  const double mine[9] = {sv, ss, a2, b2, dv, dsd, qv, qs, ds};
#pragma unroll
  for (int k = 0; k < 9; ++k) acc[part][k][j] = mine[k];
  __syncthreads();
  if (part != 0) return;
  double tot[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    double v = acc[0][k][j];
#pragma unroll
    for (int q = 1; q < 8; ++q) v += acc[q][k][j];
    tot[k] = v;
  }
  const double inv_m = 1.0 / (double)m;
  if (j == 0) {
    if (wipv) wipv[c] = tot[0] * inv_m;
    if (wipstd) wipstd[c] = tot[1] * inv_m;
  }
  if (j >= h.d) return;
  const double dsj = -2.0 * tot[8] / h.ls[j];
  if (dwipv) dwipv[c * h.d + j] = ((tot[4] - tot[6]) / h.ls[j] + tot[2] * dsj) * inv_m;
  if (dwipstd) dwipstd[c * h.d + j] = ((tot[5] - tot[7]) / h.ls[j] + tot[3] * dsj) * inv_m;
}

}  // namespace bobe
