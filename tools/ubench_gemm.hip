// Tile-GEMM core per tile/BK variant: per-tile latency at K = 128 (the Cholesky trailing-update shape) and
// steady-state throughput at K = 4096 (the sweep's shape).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/bin/ubench_gemm tools/ubench_gemm.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../bobe_amd/csrc/kernels.hpp"
using namespace bobe;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int T, int BK>
__global__ __launch_bounds__(256) void k_t(const double* __restrict__ A, int64_t lda, double* __restrict__ C, int64_t ldc,
                                           int ntx, int64_t K, unsigned long long* st) {
  extern __shared__ double smem[];
  const int ty = blockIdx.x / ntx, tx = blockIdx.x % ntx;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  v4d acc[T / 32][T / 32];
  acc_zero(acc);
  gemm_tile<KC, KC, T, T, BK>(acc, A, lda, (int64_t)ty * T, A, lda, (int64_t)tx * T, 0, K, smem);
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  store_tile<T, T>(acc, C, ldc, (int64_t)ty * T, (int64_t)tx * T, 1.0, 0.0);
  unsigned long long t2 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { st[0] = t1 - t0; st[1] = t2 - t1; }
}

template <int T, int BK>
int run(const char* name, const double* A, double* C, int64_t n, unsigned long long* st, double* flush, size_t flush_n,
        int64_t K = 128) {
  const int smem = gemm_smem_doubles<T, T, BK>() * 8;
  CK(hipFuncSetAttribute((const void*)k_t<T, BK>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int tiles_side : {1, 8, 16, 23, 32}) {
    if (tiles_side * T > n) continue;
    const int nt = tiles_side * tiles_side;
    for (int cold = 0; cold < 2; ++cold) {
      float best = 1e9f;
      unsigned long long h[2] = {0, 0};
      for (int rep = 0; rep < 4; ++rep) {
        if (cold) CK(hipMemsetAsync(flush, rep, flush_n, 0));   // evict L2 / MALL
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_t<T, BK>), dim3(nt), dim3(256), smem, 0, A, n, C, n, tiles_side, K, st);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) { best = ms; CK(hipMemcpy(h, st, 16, hipMemcpyDeviceToHost)); }
      }
      const double fl = 2.0 * nt * T * T * (double)K;
      printf("%-10s K=%lld tiles=%5d %s: %7.2f us  %6.2f TFLOP/s | block0 cycles: gemm %llu store %llu\n", name, (long long)K, nt, cold ? "cold" : "warm",
             best * 1e3, fl / (best * 1e-3) / 1e12, h[0], h[1]);
    }
  }
  return 0;
}

int main() {
  const int64_t n = 4096;
  double *A, *C, *flush;
  unsigned long long* st;
  const size_t flush_n = (size_t)1 << 30;
  CK(hipMalloc(&A, n * n * 8)); CK(hipMalloc(&C, n * n * 8)); CK(hipMalloc(&flush, flush_n)); CK(hipMalloc(&st, 64));
  std::vector<double> h((size_t)n * n);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5;
  CK(hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  if (run<128, 16>("T128/BK16", A, C, n, st, flush, flush_n)) return 1;
  if (run<128, 32>("T128/BK32", A, C, n, st, flush, flush_n)) return 1;
  if (run<64, 16>("T64/BK16", A, C, n, st, flush, flush_n)) return 1;
  if (run<64, 32>("T64/BK32", A, C, n, st, flush, flush_n)) return 1;
  if (run<64, 64>("T64/BK64", A, C, n, st, flush, flush_n)) return 1;
  if (run<128, 16>("T128/BK16", A, C, n, st, flush, flush_n, 4096)) return 1;
  if (run<64, 16>("T64/BK16", A, C, n, st, flush, flush_n, 4096)) return 1;
  if (run<64, 32>("T64/BK32", A, C, n, st, flush, flush_n, 4096)) return 1;
  if (run<128, 32>("T128/BK32", A, C, n, st, flush, flush_n, 4096)) return 1;
  return 0;
}
