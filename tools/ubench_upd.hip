// Update-half variants for the one-workgroup-per-CU step kernel (150 KB of LDS per workgroup), saturated, K = 128.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/bin/ubench_upd tools/ubench_upd.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../bobe_amd/csrc/kernels.hpp"
using namespace bobe;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// G groups of 256 threads, each a T x T tile, BK; PERSIST: grid-stride loop over tile quads
template <int G, int T, int BK, bool PERSIST, bool INTERLEAVE = false>
__global__ __launch_bounds__(256 * G) void k_upd(double* __restrict__ A, int64_t lda, int first, int n, int ntiles) {
  extern __shared__ double smem[];
  const int wv = threadIdx.x >> 6;
  const int grp = INTERLEAVE ? (wv % G) : (threadIdx.x >> 8);
  const int tid = INTERLEAVE ? ((wv / G) * 64 + (threadIdx.x & 63)) : (threadIdx.x & 255);
  constexpr int SL = gemm_smem_doubles_exact<KC, KC, T, T, BK>();
  const int nq = (ntiles + G - 1) / G;
  const int64_t base = (int64_t)first * TILE;
  for (int q = blockIdx.x; q < nq; q += gridDim.x) {
    int tile = q * G + grp;
    const bool live = tile < ntiles;
    if (!live) tile = ntiles - 1;
    int a, b;
    tri_decode(tile, a, b);
    v4d acc[T / 32][T / 32];
    load_tile<T, T>(acc, A, lda, base + (int64_t)a * T, base + (int64_t)b * T, tid);
    gemm_tile<KC, KC, T, T, BK, true>(acc, A, lda, base + (int64_t)a * T, A, lda, base + (int64_t)b * T, 0, 128,
                                      smem + grp * SL, tid);
    if (live) store_tile<T, T>(acc, A, lda, base + (int64_t)a * T, base + (int64_t)b * T, 1.0, 0.0, tid);
    if (!PERSIST) break;
  }
}

// G independent groups: group-local barriers, every group walks its own tile sequence
template <int G, int T, int BK>
__global__ __launch_bounds__(256 * G) void k_upd_dec(double* __restrict__ A, int64_t lda, int first, int n, int ntiles,
                                                     int* __restrict__ err) {
  extern __shared__ double smem[];
  __shared__ int ctr[G];
  __shared__ int failed;
  const int grp = threadIdx.x >> 8, tid = threadIdx.x & 255;
  if (threadIdx.x < G) ctr[threadIdx.x] = 0;
  if (threadIdx.x == 0) failed = 0;
  __syncthreads();
  constexpr int SL = gemm_smem_doubles_exact<KC, KC, T, T, BK>();
  GroupSync gs{&ctr[grp], 0, &failed};
  const int64_t base = (int64_t)first * TILE;
  for (int tile = blockIdx.x * G + grp; tile < ntiles; tile += gridDim.x * G) {
    int a, b;
    tri_decode(tile, a, b);
    v4d acc[T / 32][T / 32];
    load_tile<T, T>(acc, A, lda, base + (int64_t)a * T, base + (int64_t)b * T, tid);
    gemm_tile<KC, KC, T, T, BK, true, GroupSync>(acc, A, lda, base + (int64_t)a * T, A, lda, base + (int64_t)b * T, 0, 128,
                                                 smem + grp * SL, tid, &gs);
    store_tile<T, T>(acc, A, lda, base + (int64_t)a * T, base + (int64_t)b * T, 1.0, 0.0, tid);
  }
  if (failed && tid == 0) *err = 1;
}

template <int G, int T, int BK>
int run_dec(const char* name, double* A, int64_t n, int first, int smem, int grid) {
  CK(hipFuncSetAttribute((const void*)k_upd_dec<G, T, BK>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int* err; CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
  const int rem = (int)(n / 128) - first;
  const int nt = rem * (128 / T);
  const int tiles = nt * (nt + 1) / 2;
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_upd_dec<G, T, BK>), dim3(grid), dim3(256 * G), smem, 0, A, n, first, nt, tiles, err);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  int herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
  const double fl = 2.0 * tiles * (double)T * T * 128.0;
  printf("%-28s first=%2d tiles=%6d: %8.2f us  %6.2f TFLOP/s  (grid %d, spin timeout %d)\n", name, first, tiles, best * 1e3,
         fl / (best * 1e-3) / 1e12, grid, herr);
  return 0;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_upd_w4(double* __restrict__ A, int64_t lda, int first, int n, int ntiles) {
  extern __shared__ double smem[];
  const int64_t base = (int64_t)first * TILE;
  int a, b;
  tri_decode(blockIdx.x, a, b);
  v4d acc[2][2];
  load_tile<64, 64>(acc, A, lda, base + (int64_t)a * 64, base + (int64_t)b * 64);
  gemm_tile<KC, KC, 64, 64, 16, true>(acc, A, lda, base + (int64_t)a * 64, A, lda, base + (int64_t)b * 64, 0, 128, smem);
  store_tile<64, 64>(acc, A, lda, base + (int64_t)a * 64, base + (int64_t)b * 64, 1.0, 0.0);
}

template <int G, int T, int BK, bool PERSIST, bool INTERLEAVE = false>
int run(const char* name, double* A, int64_t n, int first, int smem) {
  CK(hipFuncSetAttribute((const void*)k_upd<G, T, BK, PERSIST, INTERLEAVE>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int rem = (int)(n / 128) - first;
  const int nt = rem * (128 / T);
  const int tiles = nt * (nt + 1) / 2;
  const int nq = (tiles + G - 1) / G;
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_upd<G, T, BK, PERSIST, INTERLEAVE>), dim3(PERSIST ? (G == 1 ? 1024 : 256) : nq), dim3(256 * G), smem, 0, A, n, first, nt, tiles);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double fl = 2.0 * tiles * (double)T * T * 128.0;
  printf("%-28s first=%2d tiles=%6d: %8.2f us  %6.2f TFLOP/s\n", name, first, tiles, best * 1e3, fl / (best * 1e-3) / 1e12);
  return 0;
}

int main() {
  const int64_t n = 8192;
  double* A;
  CK(hipMalloc(&A, n * n * 8));
  std::vector<double> h((size_t)n * n);
  for (size_t i = 0; i < h.size(); ++i) h[i] = ((double)((i * 2654435761u) % 1000) / 1000.0 - 0.5) * 1e-3;
  CK(hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  const int BIG = 150528;
  {
    CK(hipFuncSetAttribute((const void*)k_upd_w4, hipFuncAttributeMaxDynamicSharedMemorySize, 36864));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int nt = 112, tiles = nt * (nt + 1) / 2;
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(k_upd_w4, dim3(tiles), dim3(256), 36864, 0, A, n, 8, nt, tiles);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    printf("1x64 BK16 waves_per_eu(4,4)  first= 8 tiles=%6d: %8.2f us  %6.2f TFLOP/s\n", tiles, best * 1e3, 2.0 * tiles * 64.0 * 64 * 128 / (best * 1e-3) / 1e12);
  }
  for (int first : {8}) {
    if (run<2, 64, 16, false>("2x64 BK16 512thr 2WG/CU", A, n, first, 2 * 36864)) return 1;
    if (run<1, 64, 16, true>("1x64 BK16 persistent x1024", A, n, first, 36864)) return 1;
    if (run<1, 64, 32, false>("1x64 BK32 (2 WG/CU)", A, n, first, gemm_smem_doubles_exact<KC, KC, 64, 64, 32>() * 8)) return 1;
    if (run<1, 128, 32, false>("1x128 BK32 (1 WG/CU)", A, n, first, gemm_smem_doubles_exact<KC, KC, 128, 128, 32>() * 8)) return 1;   // 56 blocks (207 MB of lower tiles: HBM) / 31 blocks (63 MB: Infinity Cache)
    if (run<1, 64, 16, false>("1x64 BK16 (4 WG/CU)", A, n, first, gemm_smem_doubles_exact<KC, KC, 64, 64, 16>() * 8)) return 1;
    if (run<4, 64, 16, false>("4x64 BK16 1024thr", A, n, first, BIG)) return 1;
    if (run<4, 64, 16, true>("4x64 BK16 1024thr persist", A, n, first, BIG)) return 1;
    if (run<4, 64, 16, false, true>("4x64 BK16 1024thr interleaved", A, n, first, BIG)) return 1;
    if (run<4, 64, 16, true, true>("4x64 BK16 1024 persist intl", A, n, first, BIG)) return 1;
    if (run<2, 128, 16, false>("2x128 BK16 512thr", A, n, first, BIG)) return 1;
    if (run<2, 128, 16, true>("2x128 BK16 512thr persist", A, n, first, BIG)) return 1;
    if (run<1, 128, 16, false>("1x128 BK16 (2 WG/CU)", A, n, first, gemm_smem_doubles_exact<KC, KC, 128, 128, 16>() * 8)) return 1;
    if (run<2, 64, 32, false>("2x64 BK32 512thr", A, n, first, BIG)) return 1;
    if (run_dec<4, 64, 16>("4x64 BK16 decoupled persist", A, n, first, BIG, 256)) return 1;
    if (run_dec<4, 64, 16>("4x64 BK16 decoupled g=194", A, n, first, BIG, 194)) return 1;
    if (run_dec<2, 128, 16>("2x128 BK16 decoupled", A, n, first, BIG, 256)) return 1;
  }
  return 0;
}
