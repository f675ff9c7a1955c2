// fp64 MFMA tile GEMM core for gfx950 (MI355X / CDNA4).
//
// One workgroup = 256 threads = 4 waves (2x2), output tile 128x128, K-step 16.
// Each wave owns a 64x64 sub-tile = 4x4 fragments of v_mfma_f64_16x16x4_f64
// (64 f64 accumulators per lane).  Operands are staged global -> registers -> LDS
// (double-buffered, one barrier per K-step) and read back as MFMA fragments with
// conflict-free ds_read_b64:
//
//   layout KC ("k contiguous"):   element (r,k) at p[r*ld + k]   LDS image [128][18]
//   layout RC ("row contiguous"): element (r,k) at p[k*ld + r]   LDS image [16][144]
//
// where r is the m index for the A operand and the n index for the B operand, so
// C[m][n] = sum_k A(m,k) * B(n,k) in both cases.  Both LDS images take 2304 doubles.
//
// v_mfma_f64_16x16x4_f64 fragment maps (cdna_hip_programming.md section 3):
//   A: lane l holds A[l&15][l>>4]        B: lane l holds B[k=l>>4][n=l&15]
//   D: lane l, reg r holds D[(l>>4)+4r][l&15]
//
// All matrices handled by this core are padded to multiples of 128 (rows) and 16 (K)
// with 16-byte aligned bases and even leading dimensions, so there are no bounds
// checks in the inner loop.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bobe {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int TILE = 128;          // output tile edge
constexpr int TK = 16;             // K-step
constexpr int KC_STRIDE = 18;      // doubles per row of a KC LDS image (16 + 2 pad)
constexpr int RC_STRIDE = 144;     // doubles per k-row of an RC LDS image (128 + 16 pad)
constexpr int IMG = 2304;          // doubles per LDS operand image
constexpr int GEMM_THREADS = 256;
constexpr int GEMM_SMEM_DOUBLES = 4 * IMG;       // 2 operands x 2 buffers = 73,728 B
constexpr int GEMM_SMEM_BYTES = GEMM_SMEM_DOUBLES * 8;

enum Layout { KC = 0, RC = 1 };

// ---- global -> register staging (4 x 16 B per thread per operand) --------------------
template <int L>
__device__ __forceinline__ void stage_load(v2d (&reg)[4], const double* __restrict__ p, int64_t ld,
                                           int64_t r0, int64_t k0, int t) {
  if (L == KC) {
    const int kq = t & 7;
    const int r = t >> 3;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      reg[i] = *reinterpret_cast<const v2d*>(p + (r0 + r + 32 * i) * ld + k0 + 2 * kq);
  } else {
    const int k = t >> 4;
    const int c = t & 15;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      reg[i] = *reinterpret_cast<const v2d*>(p + (k0 + k) * ld + r0 + 2 * (c + 16 * i));
  }
}

template <int L>
__device__ __forceinline__ void stage_store(const v2d (&reg)[4], double* img, int t) {
  if (L == KC) {
    const int kq = t & 7;
    const int r = t >> 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<v2d*>(img + (r + 32 * i) * KC_STRIDE + 2 * kq) = reg[i];
  } else {
    const int k = t >> 4;
    const int c = t & 15;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<v2d*>(img + k * RC_STRIDE + 2 * (c + 16 * i)) = reg[i];
  }
}

// fragment read: sub-tile s (0..3) of the wave's 64 rows starting at w64, k-step ks (0..3)
template <int L>
__device__ __forceinline__ double frag_read(const double* img, int w64, int s, int ks, int lane) {
  if (L == KC)
    return img[(w64 + 16 * s + (lane & 15)) * KC_STRIDE + 4 * ks + (lane >> 4)];
  else
    return img[(4 * ks + (lane >> 4)) * RC_STRIDE + w64 + 16 * s + (lane & 15)];
}

__device__ __forceinline__ void acc_zero(v4d (&acc)[4][4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
}

// acc += A(m0.., k) * B(n0.., k) for k in [kbeg, kend), kbeg/kend multiples of 16.
// smem: GEMM_SMEM_DOUBLES doubles.  All 256 threads must call.
template <int LA, int LB>
__device__ __forceinline__ void gemm_tile(v4d (&acc)[4][4], const double* __restrict__ A, int64_t lda,
                                          int64_t m0, const double* __restrict__ B, int64_t ldb, int64_t n0,
                                          int64_t kbeg, int64_t kend, double* smem) {
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = t >> 6;
  const int wm = (wave >> 1) * 64;
  const int wn = (wave & 1) * 64;
  if (kend <= kbeg) return;
  v2d ra[4], rb[4];
  stage_load<LA>(ra, A, lda, m0, kbeg, t);
  stage_load<LB>(rb, B, ldb, n0, kbeg, t);
  stage_store<LA>(ra, smem, t);
  stage_store<LB>(rb, smem + IMG, t);
  __syncthreads();
  int buf = 0;
  for (int64_t k0 = kbeg; k0 < kend; k0 += TK) {
    const bool more = (k0 + TK) < kend;
    if (more) {
      stage_load<LA>(ra, A, lda, m0, k0 + TK, t);
      stage_load<LB>(rb, B, ldb, n0, k0 + TK, t);
    }
    const double* ia = smem + buf * 2 * IMG;
    const double* ib = ia + IMG;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      double a[4], b[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        a[s] = frag_read<LA>(ia, wm, s, ks, lane);
        b[s] = frag_read<LB>(ib, wn, s, ks, lane);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) {
      double* na = smem + (buf ^ 1) * 2 * IMG;
      stage_store<LA>(ra, na, t);
      stage_store<LB>(rb, na + IMG, t);
    }
    __syncthreads();
    buf ^= 1;
  }
}

// Coordinates of accumulator element (i, j, r) of this lane inside the 128x128 tile.
__device__ __forceinline__ int acc_row(int i, int r) {
  const int t = threadIdx.x;
  return ((t >> 6) >> 1) * 64 + 16 * i + ((t & 63) >> 4) + 4 * r;
}
__device__ __forceinline__ int acc_col(int j) {
  const int t = threadIdx.x;
  return ((t >> 6) & 1) * 64 + 16 * j + (t & 15);
}

// C[m0+row][n0+col] = alpha*acc + beta*C   (row-major C, ldc)
__device__ __forceinline__ void store_tile(const v4d (&acc)[4][4], double* __restrict__ C, int64_t ldc, int64_t m0,
                                           int64_t n0, double alpha, double beta) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double* q = C + (m0 + acc_row(i, r)) * ldc + n0 + acc_col(j);
        double v = alpha * acc[i][j][r];
        if (beta != 0.0) v += beta * (*q);
        *q = v;
      }
}

// map a linear index to a lower-triangular tile (i >= j), row-major enumeration: t = i(i+1)/2 + j
__device__ __forceinline__ void tri_decode(int t, int& i, int& j) {
  int ii = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while ((ii + 1) * (ii + 2) / 2 <= t) ++ii;
  while (ii * (ii + 1) / 2 > t) --ii;
  i = ii;
  j = t - ii * (ii + 1) / 2;
}

}  // namespace bobe
