// Saturated throughput of the trailing-update kernel variants (lower tiles of an 8192^2 trailing matrix).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/bin/ubench_syrk tools/ubench_syrk.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../bobe_amd/csrc/kernels.hpp"
using namespace bobe;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int T, int BK>
int run(const char* name, double* A, int64_t n, int smem, int kmax) {
  CK(hipFuncSetAttribute((const void*)k_syrk_trail<T, BK>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int kb : {1, 2, 4, 8}) {
    if (kb > kmax) continue;
    const int first = 8;                       // panels in blocks [0, kb), trailing matrix from block 8
    const int rem = (int)(n / 128) - first;
    const int nt = rem * (128 / T);
    const int tiles = nt * (nt + 1) / 2;
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL((k_syrk_trail<T, BK>), dim3(tiles), dim3(256), smem, 0, A, n, 0, kb, first, 0, nt);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    const double fl = 2.0 * tiles * (double)T * T * 128.0 * kb;
    printf("%-12s K=%4d tiles=%6d: %8.2f us  %6.2f TFLOP/s\n", name, 128 * kb, tiles, best * 1e3, fl / (best * 1e-3) / 1e12);
  }
  return 0;
}

int main() {
  const int64_t n = 8192;
  double* A;
  CK(hipMalloc(&A, n * n * 8));
  std::vector<double> h((size_t)n * n);
  for (size_t i = 0; i < h.size(); ++i) h[i] = ((double)((i * 2654435761u) % 1000) / 1000.0 - 0.5) * 1e-3;
  CK(hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  if (run<32, 128>("T32/BK128x1", A, n, gemm_smem_doubles_exact<KC, KC, 32, 32, 128>() * 8 / 2, 1)) return 1;
  if (run<32, 32>("T32/BK32", A, n, gemm_smem_doubles_exact<KC, KC, 32, 32, 32>() * 8, 8)) return 1;
  if (run<64, 64>("T64/BK64", A, n, gemm_smem_doubles_exact<KC, KC, 64, 64, 64>() * 8, 8)) return 1;
  if (run<64, 32>("T64/BK32", A, n, gemm_smem_doubles_exact<KC, KC, 64, 64, 32>() * 8, 8)) return 1;
  if (run<64, 16>("T64/BK16", A, n, gemm_smem_doubles_exact<KC, KC, 64, 64, 16>() * 8, 8)) return 1;
  if (run<128, 32>("T128/BK32", A, n, gemm_smem_doubles_exact<KC, KC, 128, 128, 32>() * 8, 8)) return 1;
  if (run<128, 16>("T128/BK16", A, n, gemm_smem_doubles_exact<KC, KC, 128, 128, 16>() * 8, 8)) return 1;
  return 0;
}
