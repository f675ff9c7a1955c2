// fp64 MFMA tile GEMM core for gfx950 (MI355X / CDNA4).
//
// One workgroup = 256 threads = 4 waves (2x2), output tile TM x TN (128 or 64 each), K-step BK (16 or 32).
// Each wave owns a (TM/2)x(TN/2) sub-tile = (TM/32)x(TN/32) fragments of v_mfma_f64_16x16x4_f64.
// Operands are staged global -> registers -> LDS (double-buffered, one barrier per K-step) and
// read back as MFMA fragments with conflict-free ds_read_b64:
//
//   layout KC ("k contiguous"):   element (r,k) at p[r*ld + k]   LDS image [R][BK+2]
//   layout RC ("row contiguous"): element (r,k) at p[k*ld + r]   LDS image [BK][R+16]
//
// where r is the m index for the A operand (R = TM) and the n index for the B operand (R = TN), so
// C[m][n] = sum_k A(m,k) * B(n,k) in both cases.
//
// v_mfma_f64_16x16x4_f64 fragment maps (cdna_hip_programming.md section 3):
//   A: lane l holds A[l&15][l>>4]        B: lane l holds B[k=l>>4][n=l&15]
//   D: lane l, reg r holds D[(l>>4)+4r][l&15]
//
// The 128x128 tile has the best arithmetic intensity (16 flop per LDS-staged byte); the 64x64 tile
// quarters the work per workgroup and is used where a launch has too few or too unequal tiles to
// fill 256 CUs (one CU delivers only ~0.3 TFLOP/s of fp64 MFMA, so balance beats tile efficiency).
//
// All matrices handled by this core are padded to multiples of 128 (rows) and 16 (K) with 16-byte
// aligned bases and even leading dimensions, so there are no bounds checks in the inner loop.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace bobe {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int TILE = 128;          // padding granule and default output tile edge
constexpr int TK = 16;             // K granule (every K range is a multiple of 16)
constexpr int GEMM_THREADS = 256;

enum Layout { KC = 0, RC = 1 };

// LDS image of one operand tile: R rows x BK k.  KC rows are padded by 2 doubles (stride = 4 mod 64
// dwords), RC k-rows by 16 doubles (stride = 32 mod 64 dwords): both make the fragment reads of a
// 32-lane ds_read_b64 group hit 32 distinct bank pairs.
template <int L, int R, int BK>
struct Img {
  static constexpr int stride = (L == KC) ? (BK + 2) : (R + 16);
  static constexpr int doubles = (L == KC) ? R * (BK + 2) : BK * (R + 16);
};
constexpr int imax(int a, int b) { return a > b ? a : b; }
// LDS doubles needed by gemm_tile<.,.,TM,TN,BK> (worst case over layouts), two buffers
template <int TM, int TN, int BK>
constexpr int gemm_smem_doubles() {
  return 2 * (imax(Img<KC, TM, BK>::doubles, Img<RC, TM, BK>::doubles) +
              imax(Img<KC, TN, BK>::doubles, Img<RC, TN, BK>::doubles));
}
// exact LDS doubles for a given layout pair (gemm_tile lays the two operand images out back to back)
template <int LA, int LB, int TM, int TN, int BK>
constexpr int gemm_smem_doubles_exact() {
  return 2 * (Img<LA, TM, BK>::doubles + Img<LB, TN, BK>::doubles);
}
#ifndef BOBE_BK128
#define BOBE_BK128 16
#endif
constexpr int BK128 = BOBE_BK128;   // 128x128 tiles: 73,728 B of LDS at BK = 16 -> two workgroups per CU
#ifndef BOBE_BK64
#define BOBE_BK64 16
#endif
constexpr int BK64 = BOBE_BK64;   // 64x64 tiles: 40,960 B of LDS -> FOUR workgroups per CU.  (Round 3: BK = 32 - 80 KB, two per CU -
                                  // made the inverse and K^-1 launches 5-8 % slower: fit 32.8 -> 31.8 ms; BK = 8: 32.5 ms)
constexpr int GEMM_SMEM_DOUBLES = gemm_smem_doubles<128, 128, BK128>();
constexpr int GEMM_SMEM_BYTES = GEMM_SMEM_DOUBLES * 8;
constexpr int GEMM64_SMEM_BYTES = gemm_smem_doubles<64, 64, BK64>() * 8;
// 32 x 32 tiles serve launches that are bound by the latency of ONE workgroup's K loop (few MFMAs per K-step): K-steps of 32
// halve the global -> LDS round trips (49,152 B of LDS, three workgroups per CU)
constexpr int BK32 = 32;
template <int T> struct TileCfg { static constexpr int bk = (T == 128) ? BK128 : (T == 64 ? BK64 : BK32); };
constexpr int GEMM32_SMEM_BYTES = gemm_smem_doubles<32, 32, BK32>() * 8;

// ---- global -> register staging (R*BK/512 x 16 B per thread per operand) -----------------------
template <int L, int R, int BK>
__device__ __forceinline__ void stage_load(v2d (&reg)[R * BK / 512], const double* __restrict__ p, int64_t ld,
                                           int64_t r0, int64_t k0, int t) {
  if (L == KC) {
    constexpr int VPR = BK / 2;            // 16-byte vectors per row
    const int kq = t % VPR;
    const int r = t / VPR;
#pragma unroll
    for (int i = 0; i < R * BK / 512; ++i)
      reg[i] = *reinterpret_cast<const v2d*>(p + (r0 + r + (256 / VPR) * i) * ld + k0 + 2 * kq);
  } else {
    constexpr int VPR = R / 2;             // 16-byte vectors per k-row
    const int c = t % VPR;
    const int k = t / VPR;
#pragma unroll
    for (int i = 0; i < R * BK / 512; ++i)
      reg[i] = *reinterpret_cast<const v2d*>(p + (k0 + k + (256 / VPR) * i) * ld + r0 + 2 * c);
  }
}

template <int L, int R, int BK>
__device__ __forceinline__ void stage_store(const v2d (&reg)[R * BK / 512], double* img, int t) {
  if (L == KC) {
    constexpr int VPR = BK / 2;
    const int kq = t % VPR;
    const int r = t / VPR;
#pragma unroll
    for (int i = 0; i < R * BK / 512; ++i)
      *reinterpret_cast<v2d*>(img + (r + (256 / VPR) * i) * (BK + 2) + 2 * kq) = reg[i];
  } else {
    constexpr int VPR = R / 2;
    const int c = t % VPR;
    const int k = t / VPR;
#pragma unroll
    for (int i = 0; i < R * BK / 512; ++i)
      *reinterpret_cast<v2d*>(img + (k + (256 / VPR) * i) * (R + 16) + 2 * c) = reg[i];
  }
}

// fragment read: 16-row sub-tile s of the wave's rows starting at woff, k-step ks (0..BK/4-1)
template <int L, int R, int BK>
__device__ __forceinline__ double frag_read(const double* img, int woff, int s, int ks, int lane) {
  if (L == KC)
    return img[(woff + 16 * s + (lane & 15)) * (BK + 2) + 4 * ks + (lane >> 4)];
  else
    return img[(4 * ks + (lane >> 4)) * (R + 16) + woff + 16 * s + (lane & 15)];
}

template <int FM, int FN>
__device__ __forceinline__ void acc_zero(v4d (&acc)[FM][FN]) {
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
}

// acc += A(m0.., k) * B(n0.., k) for k in [kbeg, kend); (kend - kbeg) must be a multiple of BK
// (K ranges are multiples of 128 for the 128x128 callers and of 64 for the 64x64 callers).
// smem: gemm_smem_doubles<TM,TN,BK>() doubles.  All 256 threads must call.  `tid` (0..255) is the thread's index
// inside its 256-thread tile group: threadIdx.x for the one-tile-per-workgroup kernels; the filler groups of the panel
// launch (k_chol_panel<true, .>) are two tile groups in one 512-thread workgroup, each with its own smem slice - the
// barriers inside are workgroup-wide, so both groups must run the same number of K-steps.
// NEGA: accumulate -A*B (the A fragment is negated on the way into the MFMA).
// SYNC: the barrier between K-steps; WgSync (a workgroup barrier) is the only policy in use.
struct WgSync {
  __device__ __forceinline__ void sync() { __syncthreads(); }
};

// TRIL: the A operand is LOWER TRIANGULAR in its last TM columns of K (A(m0 + r, kend - TM + c) = 0 for c > r: the
// diagonal block of a triangular matrix closes the K range).  In those K-steps the MFMAs of 4-column groups that lie wholly
// right of a 16-row fragment's last row are skipped - exact zeros times finite numbers: the accumulators keep their bits
// (up to the sign of a zero).  The steps before the block run the plain loop body (a predicate in every step cost the
// sweep GEMM 2.7 %; this form gains it 1.2 %; on 64 x 64 tiles - the inverse's launches - the skips cost more than they
// save: 4 MFMAs per K group and fragment row there).
template <int LA, int LB, int TM = 128, int TN = 128, int BK = BK128, bool NEGA = false, class SYNC = WgSync, bool TRIL = false>
__device__ __forceinline__ void gemm_tile(v4d (&acc)[TM / 32][TN / 32], const double* __restrict__ A, int64_t lda,
                                          int64_t m0, const double* __restrict__ B, int64_t ldb, int64_t n0,
                                          int64_t kbeg, int64_t kend, double* smem, int tid = threadIdx.x,
                                          SYNC* sy = nullptr) {
  WgSync wg_default;
  constexpr int FM = TM / 32, FN = TN / 32;
  constexpr int IA = Img<LA, TM, BK>::doubles;
  constexpr int IB = Img<LB, TN, BK>::doubles;
  const int t = tid;
  const int lane = t & 63;
  const int wave = t >> 6;
  const int wm = (wave >> 1) * (TM / 2);
  const int wn = (wave & 1) * (TN / 2);
  if (kend <= kbeg) return;
  v2d ra[TM * BK / 512], rb[TN * BK / 512];
  stage_load<LA, TM, BK>(ra, A, lda, m0, kbeg, t);
  stage_load<LB, TN, BK>(rb, B, ldb, n0, kbeg, t);
  stage_store<LA, TM, BK>(ra, smem, t);
  stage_store<LB, TN, BK>(rb, smem + IA, t);
  if (sy) sy->sync(); else wg_default.sync();
  int buf = 0;
  // one K-step: next step's operands on their way, this step's MFMAs, next step's LDS image.  IN (a constant): the step
  // lies inside the closing diagonal block of a TRIL operand, `rel` = its first column there
  const int wmu = TRIL ? __builtin_amdgcn_readfirstlane(wm) : 0;
  auto kstep = [&](int64_t k0, auto in_c, int rel) {
    constexpr bool IN = decltype(in_c)::value;
    const bool more = (k0 + BK) < kend;
    if (more) {
      stage_load<LA, TM, BK>(ra, A, lda, m0, k0 + BK, t);
      stage_load<LB, TN, BK>(rb, B, ldb, n0, k0 + BK, t);
    }
    const double* ia = smem + buf * (IA + IB);
    const double* ib = ia + IA;
#pragma unroll
    for (int ks = 0; ks < BK / 4; ++ks) {
      double a[FM], b[FN];
#pragma unroll
      for (int s = 0; s < FM; ++s) {
        a[s] = frag_read<LA, TM, BK>(ia, wm, s, ks, lane);
        if (NEGA) a[s] = -a[s];
      }
#pragma unroll
      for (int s = 0; s < FN; ++s) b[s] = frag_read<LB, TN, BK>(ib, wn, s, ks, lane);
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        if (IN && rel + 4 * ks > wmu + 16 * i + 15) continue;       // columns right of the fragment's rows: zeros
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
    if (more) {
      double* na = smem + (buf ^ 1) * (IA + IB);
      stage_store<LA, TM, BK>(ra, na, t);
      stage_store<LB, TN, BK>(rb, na + IA, t);
    }
    if (sy) sy->sync(); else wg_default.sync();
    buf ^= 1;
  };
  const int64_t kd = TRIL ? (kend - TM > kbeg ? kend - TM : kbeg) : kend;      // start of the closing diagonal block
  for (int64_t k0 = kbeg; k0 < kd; k0 += BK) kstep(k0, std::false_type(), 0);
  if (TRIL)
    for (int64_t k0 = kd; k0 < kend; k0 += BK) kstep(k0, std::true_type(), (int)(k0 - (kend - TM)));
}

// Coordinates of accumulator element (i, j, r) of this lane inside the TM x TN tile.
template <int TM = 128>
__device__ __forceinline__ int acc_row(int i, int r, int tid = threadIdx.x) {
  const int t = tid;
  return ((t >> 6) >> 1) * (TM / 2) + 16 * i + ((t & 63) >> 4) + 4 * r;
}
template <int TN = 128>
__device__ __forceinline__ int acc_col(int j, int tid = threadIdx.x) {
  const int t = tid;
  return ((t >> 6) & 1) * (TN / 2) + 16 * j + (t & 15);
}

// C[m0+row][n0+col] = alpha*acc + beta*C   (row-major C, ldc)
template <int TM = 128, int TN = 128>
__device__ __forceinline__ void store_tile(const v4d (&acc)[TM / 32][TN / 32], double* __restrict__ C, int64_t ldc,
                                           int64_t m0, int64_t n0, double alpha, double beta, int tid = threadIdx.x) {
#pragma unroll
  for (int i = 0; i < TM / 32; ++i)
#pragma unroll
    for (int j = 0; j < TN / 32; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double* q = C + (m0 + acc_row<TM>(i, r, tid)) * ldc + n0 + acc_col<TN>(j, tid);
        double v = alpha * acc[i][j][r];
        if (beta != 0.0) v += beta * (*q);
        *q = v;
      }
}

// acc = C[m0+row][n0+col]  (issued early so the loads overlap the operand staging)
template <int TM = 128, int TN = 128>
__device__ __forceinline__ void load_tile(v4d (&acc)[TM / 32][TN / 32], const double* __restrict__ C, int64_t ldc,
                                          int64_t m0, int64_t n0, int tid = threadIdx.x) {
#pragma unroll
  for (int i = 0; i < TM / 32; ++i)
#pragma unroll
    for (int j = 0; j < TN / 32; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = C[(m0 + acc_row<TM>(i, r, tid)) * ldc + n0 + acc_col<TN>(j, tid)];
}

// map a linear index to a lower-triangular tile (i >= j), row-major enumeration: t = i(i+1)/2 + j
__device__ __forceinline__ void tri_decode(int t, int& i, int& j) {
  int ii = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while ((ii + 1) * (ii + 2) / 2 <= t) ++ii;
  while (ii * (ii + 1) / 2 > t) --ii;
  i = ii;
  j = t - ii * (ii + 1) / 2;
}

// ---- workgroup -> tile maps that keep a launch's operand panels in ONE XCD's L2 ------------------------------
// Workgroups are dealt round-robin over the 8 XCDs (hardware ids b and b + 8 share an L2; MI355X_MICROARCH
// "Workgroup dispatch"), and an XCD starts its share in id order.  For launches whose workgroups all do the same
// amount of work, xcd_share() gives every XCD one CONTIGUOUS share of `per` logical ids (grid = 8 * per, ids past the
// tile count exit), and tri_grouped() / rect_grouped() enumerate tiles in 8 x 8 groups, so that the tiles an XCD has
// in flight share 8 A panels and 8 B panels instead of pulling one of each per tile through the fabric.  per <= 0:
// identity.  A pure speed choice: which workgroup computes a tile never changes the tile's bits.  Measured (round 2):
// nothing at N = 4096 (the panels of the default row-major order already stay in L2), 1.5-2.5 % at N = 8192.
// Launches with tiles of unequal length keep the row-major heavy-first order: runs of 64 logical ids dealt round-robin
// were 8-12 % SLOWER (a launch has only ~230 tiles per XCD, whole runs leave some XCDs a quarter more tiles than
// others), and equal-WORK contiguous shares cut on the host made the K^-1 launch 8 % slower (1.90 -> 2.05 ms for four
// matrices: the XCDs then hold tiles of very different length while ids are still handed out in order).
__device__ __forceinline__ int xcd_share(int b, int per) { return per > 0 ? (b & 7) * per + (b >> 3) : b; }

// lower triangle of an n x n tile grid: bands of 8 tile rows (top band first); inside a band the 8-column groups left
// to right, each row-major, then the band's diagonal triangle
__device__ __forceinline__ void tri_grouped(int t, int n, int& i, int& j) {
  int ii, jj;
  tri_decode(t, ii, jj);                       // (band boundaries coincide with row boundaries of the row-major order)
  const int R = ii & ~7;
  const int h = min(8, n - R);
  const int u = t - R * (R + 1) / 2;
  if (u < R * h) {
    const int g = u / (8 * h), w = u - g * 8 * h;
    i = R + (w >> 3);
    j = 8 * g + (w & 7);
  } else {
    tri_decode(u - R * h, ii, jj);
    i = R + ii;
    j = R + jj;
  }
}

// rows x cols tile grid: bands of 8 rows, inside a band groups of 8 columns, each group row-major
__device__ __forceinline__ void rect_grouped(int e, int rows, int cols, int& i, int& j) {
  const int r = e / (8 * cols);
  const int h = min(8, rows - 8 * r);
  const int u = e - r * 8 * cols;
  const int g = u / (8 * h);
  const int gw = min(8, cols - 8 * g);
  const int w = u - g * 8 * h;
  i = 8 * r + w / gw;
  j = 8 * g + w % gw;
}

}  // namespace bobe
