// Kernels shared by the translation units of libbobe_gp.so (gfx950): kernel functions, coordinate scaling, kernel-matrix
// assembly, the fixed-order matrix-vector products and reductions.  See DESIGN.md for the data layout.
// (Non-template kernels here are `static`: every unit that includes the header gets its own copy.)
//
// Conventions
//   * Np = N rounded up to 128.  The padded kernel matrix is [[K,0],[0,I]], so its Cholesky
//     factor is [[L,0],[0,I]], its inverse factor [[L^-1,0],[0,I]], and padded y / alpha are 0.
//     Kernels that evaluate k(x_a, x_b) mask padded points by index.
//   * coordinates are stored SoA and pre-divided by the lengthscales: XsT[j*ld + i] = x_ij / ls_j
//     (reference op order: dist_sq(xa/ls, xb/ls), BOBE/gp.py:149, 161).
#pragma once
#include "gemm_f64.hpp"
#include "gp_types.hpp"

namespace bobe {

constexpr double SQRT5 = 2.23606797749978969641;
constexpr double NOISE_FLOOR = 1e-12;   // BOBE/gp.py:16

// ---- kernel functions (BOBE/gp.py:124-168) ---------------------------------------------
template <int KERN>
__device__ __forceinline__ double kern_eval(double r2, double kvar) {
  if (KERN == 0) {
    return kvar * exp(-0.5 * r2);
  } else if (KERN == 2) {
    return r2;                                // dist_sq (gp.py:80-96): the squared distance itself
  } else {
    const double dd = sqrt(r2 < 1e-30 ? 1e-30 : r2);
    const double e = exp(-SQRT5 * dd);
    const double poly = 1.0 + dd * (SQRT5 + (dd * 5.0) / 3.0);
    return kvar * poly * e;
  }
}

// d k / d log ls_j = grad_factor * D_j, with D_j the squared scaled difference in dim j
template <int KERN>
__device__ __forceinline__ double kern_grad_factor(double r2, double kvar, double kval) {
  if (KERN == 0) {
    return kval;
  } else {
    if (r2 < 1e-30) return 0.0;
    const double dd = sqrt(r2);
    return kvar * (5.0 / 3.0) * (1.0 + SQRT5 * dd) * exp(-SQRT5 * dd);
  }
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  union { double d; int i[2]; } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
  u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
  return u.d;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- wave-level sums without the LDS crossbar (the chain kernels of the samplers, the few-candidate gradient path) ----------
// __shfl_xor is a ds_bpermute (two per double, ~100+ cycles each, six in a row per sum); latency-bound kernels sit on that
// several times per step.  gfx950 exchanges lanes in the VALU instead: v_permlane32_swap / v_permlane16_swap
// trade the upper 32 (odd 16) lanes of one register for the lower 32 (even 16) of another, and DPP reads a neighbour within a
// row of 16 (row_ror:8, row_half_mirror, quad_perm).  Pairings per step: l ^ 32, l ^ 16, l ^ 8, 7 - l within eight, l ^ 2,
// l ^ 1 - every step joins two lanes that differ in the step's lane bit, so six steps cover the wave.  Fixed order.
// (A different pairing than wave_sum's, i.e. different last bits: used where no other kernel has to reproduce the sum - the
//  factorisation / sweep reductions keep wave_sum; the classifier gate uses this one in every kernel that evaluates it.)
constexpr int DPP_ROR8 = 0x128, DPP_HALF_MIRROR = 0x141, DPP_XOR2 = 0x4E /* quad_perm:[2,3,0,1] */,
              DPP_XOR1 = 0xB1 /* quad_perm:[1,0,3,2] */;
template <int CTRL>
__device__ __forceinline__ double dpp_read(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// a[l] + a[l ^ W] in the lanes whose bit W is clear, b[l] + b[l ^ W] in the others (W = 32, 16): one swap per register half
template <int W>
__device__ __forceinline__ double swap_add(double a, double b) {
  const unsigned al = (unsigned)__double2loint(a), ah = (unsigned)__double2hiint(a);
  const unsigned bl = (unsigned)__double2loint(b), bh = (unsigned)__double2hiint(b);
  if constexpr (W == 32) {
    const auto lo = __builtin_amdgcn_permlane32_swap(al, bl, false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(ah, bh, false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
  } else {
    const auto lo = __builtin_amdgcn_permlane16_swap(al, bl, false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(ah, bh, false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
  }
}
// step K (0 .. 5, lane bit 32 >> K) on the pair (a, b): the lane keeps a (bit clear) or b (bit set) and adds its partner's
template <int K>
__device__ __forceinline__ double chain_halve(double a, double b, int lane) {
  if constexpr (K == 0) return swap_add<32>(a, b);
  else if constexpr (K == 1) return swap_add<16>(a, b);
  else {
    constexpr int CTRL = K == 2 ? DPP_ROR8 : (K == 3 ? DPP_HALF_MIRROR : (K == 4 ? DPP_XOR2 : DPP_XOR1));
    const bool up = (lane & (32 >> K)) != 0;
    return (up ? b : a) + dpp_read<CTRL>(up ? a : b);
  }
}
// the same step on one value held by every lane
template <int K>
__device__ __forceinline__ double chain_fold(double r) {
  if constexpr (K == 0) return swap_add<32>(r, r);
  else if constexpr (K == 1) return swap_add<16>(r, r);
  else {
    constexpr int CTRL = K == 2 ? DPP_ROR8 : (K == 3 ? DPP_HALF_MIRROR : (K == 4 ? DPP_XOR2 : DPP_XOR1));
    return r + dpp_read<CTRL>(r);
  }
}
// the sum over the wave, in every lane
__device__ __forceinline__ double chain_wave_sum(double r) {
  r = chain_fold<0>(r);
  r = chain_fold<1>(r);
  r = chain_fold<2>(r);
  r = chain_fold<3>(r);
  r = chain_fold<4>(r);
  return chain_fold<5>(r);
}
// Sums of D per-lane values over the wave, all D at once: at every step the lane keeps the half of its values that its
// step bit selects and adds its partner's (log2 D steps: D - 1 exchanges), then folds the one value left over the remaining
// bits - one dependency chain of six steps instead of D of them.  Returns component (lane >> (6 - log2 D)), complete in
// every lane of that group.
template <int D, int HLF, int K>
__device__ __forceinline__ void chain_sum_halve(double (&v)[D], int lane) {
  if constexpr (HLF >= 1) {
#pragma unroll
    for (int i = 0; i < HLF; ++i) v[i] = chain_halve<K>(v[i], v[HLF + i], lane);
    chain_sum_halve<D, HLF / 2, K + 1>(v, lane);
  }
}
template <int D>
__device__ __forceinline__ double wave_sum_components(double (&v)[D], int lane) {
  static_assert(D == 8 || D == 16 || D == 32, "power of two");
  chain_sum_halve<D, D / 2, 0>(v, lane);
  double r = v[0];
  if constexpr (D == 8) r = chain_fold<3>(r);
  if constexpr (D <= 16) r = chain_fold<4>(r);
  return chain_fold<5>(r);
}

// ---- coordinate scaling: out[j*ldo + i] = in[i*d + j] / ls[j]  (0 for i >= n) ------------
// (hp, when given, overrides h with the device-resident hyper-parameters: a captured graph replays with new values)
// Batched launches: blockIdx.y = slot picks hp[slot] and offsets `out` by bsO doubles per slot.
static __global__ void k_scale_coords(const double* __restrict__ in, int64_t n, int64_t npad, Hyper h,
                               double* __restrict__ out, int64_t ldo, const Hyper* __restrict__ hp = nullptr,
                               int64_t bsO = 0, int* __restrict__ info_reset = nullptr) {
  if (hp) h = hp[blockIdx.y];
  out += blockIdx.y * bsO;
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  // (first kernel of an evaluation: it also arms the factorisation's info word - one memset launch less)
  if (info_reset && i == 0) info_reset[blockIdx.y] = 0x7f7f7f7f;
  if (i >= npad) return;
  // (constant indices into h.ls: a runtime index would send the by-value struct through scratch memory)
#pragma unroll
  for (int j = 0; j < MAX_D; ++j)
    if (j < h.d) out[j * ldo + i] = (i < n) ? in[i * h.d + j] / h.ls[j] : 0.0;
}

// ---- kernel-matrix assembly ---------------------------------------------------------------
// out[a*ldo + b] = k(A_a, B_b) for a < na, b < nb; padding is 0, or identity when SQUARE.
// SQUARE: blockIdx.x enumerates lower tiles (ti >= tj) and noise is added on the diagonal.
// else  : blockIdx.x = tile column, blockIdx.y = tile row.
// A thread owns one column b of the 128x128 tile (its d scaled coordinates stay in registers) and walks 64
// rows; a wave's lanes share the row, so the row's coordinates are wave-uniform and come through the scalar
// cache.  No LDS: the kernel is bound by the exp / pairwise-distance arithmetic and the coalesced 8-byte stores
// (512 B per wave per row).
// FULL: d == DCAP exactly (no per-dimension predication at all).
// wv / part (cross tiles only): the tile's share of out^T wv on the way, part[ti*ldp + column] = sum over the tile's 128 rows
// of out[row][column] wv[row] - the posterior-mean product K(X, C)^T alpha without reading K(X, C) back.  The thread
// halves own rows 0..63 / 64..127 and sum them in the order of k_gemv_t_part (four runs of 32 rows, then
// ((r0 + r1) + r2) + r3): the same bits as that kernel on the stored tile.
template <int KERN, bool SQUARE, int DCAP, bool FULL>
__global__ __launch_bounds__(256) void k_kernel_matrix(const double* __restrict__ AT, int64_t lda, int64_t na,
                                                       const double* __restrict__ BT, int64_t ldb, int64_t nb,
                                                       Hyper h, double* __restrict__ out, int64_t ldo,
                                                       const Hyper* __restrict__ hp = nullptr, int64_t bsX = 0,
                                                       int64_t bsO = 0, const double* __restrict__ wv = nullptr,
                                                       double* __restrict__ part = nullptr, int64_t ldp = 0) {
  if (hp) h = hp[SQUARE ? blockIdx.y : 0];
  int ti, tj;
  if (SQUARE) {
    // batched assembly of K(X,X): blockIdx.y = slot (its own scaled coordinates, hyper-parameters and output)
    AT += blockIdx.y * bsX;
    BT += blockIdx.y * bsX;
    out += blockIdx.y * bsO;
    // four workgroups per lower tile (32 rows each): n(n+1)/2 tiles alone are barely two per CU at N = 4096
    tri_decode(blockIdx.x >> 2, ti, tj);
  } else {
    ti = blockIdx.y;
    tj = blockIdx.x;
  }
  const int t = threadIdx.x;
  const int b = t & 127;
  const int64_t gb = (int64_t)tj * TILE + b;
  double xb[DCAP];
#pragma unroll
  for (int j = 0; j < DCAP; ++j) xb[j] = (FULL || j < h.d) ? BT[j * ldb + gb] : 0.0;
  const int a0 = __builtin_amdgcn_readfirstlane(t >> 7);   // wave-uniform (a wave spans 64 consecutive columns)
  const double* arow = AT + (int64_t)ti * TILE;
  // SQUARE: the halves interleave over the workgroup's 32 rows; cross tiles: half a0 owns rows 64 a0 .. 64 a0 + 63
  const int abeg = SQUARE ? (int)(blockIdx.x & 3) * (TILE / 4) + a0 : a0 * (TILE / 2);
  const int aend = SQUARE ? abeg - a0 + TILE / 4 : abeg + TILE / 2;
  const int astep = SQUARE ? 2 : 1;
  double s = 0.0, s_first = 0.0;
#pragma unroll 4
  for (int a = abeg; a < aend; a += astep) {
    const int64_t ga = (int64_t)ti * TILE + a;
    double r2 = 0.0;
    double xa[DCAP];   // unconditional (clamped) loads: all in flight at once, no branch per dimension
#pragma unroll
    for (int j = 0; j < DCAP; ++j) xa[j] = arow[(int64_t)((FULL || j < h.d) ? j : 0) * lda + a];
#pragma unroll
    for (int j = 0; j < DCAP; ++j) {
      const double df = (FULL || j < h.d) ? xa[j] - xb[j] : 0.0;
      r2 = __builtin_fma(df, df, r2);
    }
    double v;
    if (ga < na && gb < nb) {
      v = kern_eval<KERN>(r2, h.kvar);
      if (SQUARE && ga == gb) v += h.noise;
    } else {
      v = (SQUARE && ga == gb) ? 1.0 : 0.0;
    }
    out[ga * ldo + gb] = v;
    if (!SQUARE && wv) {
      if (a == abeg + TILE / 4) {      // (wave-uniform) second run of 32 rows
        s_first = s;
        s = 0.0;
      }
      s = __builtin_fma(v, wv[ga], s);
    }
  }
  if (!SQUARE && wv) {
    __shared__ double red[2][TILE];
    if (a0 == 1) {
      red[0][b] = s_first;
      red[1][b] = s;
    }
    __syncthreads();
    if (a0 == 0) part[(int64_t)ti * ldp + gb] = ((s_first + s) + red[0][b]) + red[1][b];
  }
}


// ---- matrix-vector products ---------------------------------------------------------------------
// w[i] = sum_{k <= i} M[i][k] y[k]   (one wave per row, fixed summation order)
static __global__ __launch_bounds__(256) void k_gemv_lower(const double* __restrict__ M, int64_t ld, int64_t np,
                                                    const double* __restrict__ y, double* __restrict__ w,
                                                    int64_t bsM = 0, int64_t bsW = 0, int64_t bsY = 0) {
  M += blockIdx.y * bsM;      // batched: blockIdx.y = slot (y is shared by the slots unless bsY is given)
  w += blockIdx.y * bsW;
  y += blockIdx.y * bsY;
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= np) return;
  double s = 0.0;
  for (int64_t k = lane; k <= i; k += 64) s += M[i * ld + k] * y[k];
  s = wave_sum(s);
  if (lane == 0) w[i] = s;
}

// part[rb*ldp + c] = sum_{k in row block rb (128 rows)} M[k][c] w[k], only row blocks rb >= rb_min(c)
// where rb_min = (lower ? c/128 : 0).  grid.x = column strips of 64, grid.y = row blocks.
static __global__ __launch_bounds__(256) void k_gemv_t_part(const double* __restrict__ M, int64_t ld, int lower,
                                                     const double* __restrict__ w, double* __restrict__ part,
                                                     int64_t ldp, int64_t bsM = 0, int64_t bsW = 0, int64_t bsP = 0) {
  M += blockIdx.z * bsM;      // batched: blockIdx.z = slot
  w += blockIdx.z * bsW;
  part += blockIdx.z * bsP;
  __shared__ double red[4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int64_t c = (int64_t)blockIdx.x * 64 + cx;
  const int rb = blockIdx.y;
  double s = 0.0;
  if (!lower || rb >= (int)(blockIdx.x * 64 / TILE)) {
    const int64_t k0 = (int64_t)rb * TILE + ry * 32;
#pragma unroll 8
    for (int k = 0; k < 32; ++k) s += M[(k0 + k) * ld + c] * w[k0 + k];
  }
  red[ry][cx] = s;
  __syncthreads();
  if (ry == 0) part[(int64_t)rb * ldp + c] = ((red[0][cx] + red[1][cx]) + red[2][cx]) + red[3][cx];
}

// out[c] = sum_{rb=rb0(c)}^{nrb-1} part[rb*ldp + c]   (fixed order)
static __global__ void k_colsum_parts(const double* __restrict__ part, int64_t ldp, int nrb, int lower, int64_t ncols,
                               double* __restrict__ out, int64_t bsP = 0, int64_t bsO = 0) {
  part += blockIdx.y * bsP;   // batched: blockIdx.y = slot
  out += blockIdx.y * bsO;
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncols) return;
  double s = 0.0;
  for (int rb = lower ? (int)(c / TILE) : 0; rb < nrb; ++rb) s += part[(int64_t)rb * ldp + c];
  out[c] = s;
}

// ---- scalar reductions ------------------------------------------------------------------------------
// res[0] = sum_i w_i^2 (0 without w) ; res[1] = sum_i log L_ii ; res[101] = min_i L_ii        (single workgroup, fixed order)
__device__ __forceinline__ void mll_terms_body(int slot, const double* __restrict__ w, const double* __restrict__ L, int64_t ld,
                                                   int64_t np, double* __restrict__ res, int64_t bsW,
                                                   int64_t bsL, int64_t bsR, const int* __restrict__ info) {
  if (w) w += slot * bsW;      // batched: blockIdx.x = slot
  L += slot * bsL;
  res += slot * bsR;
  // (the factorisation's info word rides along in res[100], so that one copy brings everything to the host)
  if (info && threadIdx.x == 0) reinterpret_cast<int*>(res + 100)[0] = info[slot];
  __shared__ double r0[4], r1[4], r2[4];
  double a = 0.0, b = 0.0, mn = 1.0;                  // (the padding's diagonal is 1)
  for (int64_t i = threadIdx.x; i < np; i += 256) {
    const double lii = L[i * ld + i];
    if (w) a += w[i] * w[i];
    b += log(lii);
    mn = (lii < mn || lii != lii) ? lii : mn;     // (fmin would drop a NaN diagonal: NaN counts as failed)
  }
  a = wave_sum(a);
  b = wave_sum(b);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double other = __shfl_xor(mn, o, 64);
    mn = (other < mn || other != other) ? other : mn;
  }
  if ((threadIdx.x & 63) == 0) {
    r0[threadIdx.x >> 6] = a;
    r1[threadIdx.x >> 6] = b;
    r2[threadIdx.x >> 6] = mn;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    res[0] = ((r0[0] + r0[1]) + r0[2]) + r0[3];
    res[1] = ((r1[0] + r1[1]) + r1[2]) + r1[3];
    double m = r2[0];                                             // smallest pivot's root: the host's rank test (pivot_floor)
#pragma unroll
    for (int q = 1; q < 4; ++q) m = (r2[q] < m || r2[q] != r2[q]) ? r2[q] : m;
    res[101] = m;
  }
}
static __global__ __launch_bounds__(256) void k_mll_terms(const double* __restrict__ w, const double* __restrict__ L, int64_t ld,
                                                   int64_t np, double* __restrict__ res, int64_t bsW = 0,
                                                   int64_t bsL = 0, int64_t bsR = 0,
                                                   const int* __restrict__ info = nullptr) {
  mll_terms_body((int)blockIdx.x, w, L, ld, np, res, bsW, bsL, bsR, info);      // batched: blockIdx.x = slot
}

// The two reductions that end an evaluation in ONE launch.  Workgroups 0..d: res[2 + j] = 0.5 * sum over the tiles of
// partial[tile * stride + src(j)], one wave per component, lane-strided partial sums combined by a fixed butterfly
// (deterministic).  Workgroup d+1 is k_mll_terms.  blockIdx.y = slot of a lock-step batch (strides in doubles).
static __global__ __launch_bounds__(256) void k_mll_grad_reduce(const double* __restrict__ partial, int ntiles, int stride, int d,
                                                         int dcap, double* __restrict__ res,
                                                         const double* __restrict__ w, const double* __restrict__ L,
                                                         int64_t ld, int64_t np, const int* __restrict__ info,
                                                         int64_t bsP = 0, int64_t bsR = 0, int64_t bsW = 0, int64_t bsL = 0) {
  const int slot = blockIdx.y;
  if ((int)blockIdx.x == d + 1) {
    mll_terms_body(slot, w, L, ld, np, res, bsW, bsL, bsR, info);
    return;
  }
  if (threadIdx.x >= 64) return;
  partial += slot * bsP;
  res += slot * bsR;
  const int j = blockIdx.x;
  const int src = (j == d) ? dcap : j;
  double s = 0.0;
  for (int q = threadIdx.x; q < ntiles; q += 64) s += partial[(int64_t)q * stride + src];
  s = wave_sum(s);
  if (threadIdx.x == 0) res[2 + j] = 0.5 * s;
}

// ---- misc ------------------------------------------------------------------------------------------
static __global__ void k_fill(double* __restrict__ p, int64_t n, double v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// dst[i*ldd + j] = (lower_only && j > i) ? 0 : src[i*lds + j]  for i < rows, j < cols
static __global__ void k_copy2d(const double* __restrict__ src, int64_t lds, double* __restrict__ dst, int64_t ldd, int64_t rows,
                         int64_t cols, int lower_only) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t i = blockIdx.y;
  if (i < rows && j < cols) dst[i * ldd + j] = (lower_only && j > i) ? 0.0 : src[i * lds + j];
}

// pad-aware load of a caller-provided N x N lower factor into the padded [[L,0],[0,I]] layout
static __global__ void k_load_padded_lower(const double* __restrict__ src, int64_t n, double* __restrict__ dst, int64_t ld,
                                    int64_t np, int whole = 0) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t i = blockIdx.y;
  if (i >= np || j >= np) return;
  double v;
  if (i < n && j < n) v = (j <= i || whole) ? src[i * n + j] : 0.0;
  else v = (i == j) ? 1.0 : 0.0;
  dst[i * ld + j] = v;
}

}  // namespace bobe
