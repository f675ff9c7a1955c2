// Diagnostic micro-benchmarks for the Cholesky panel kernels (stamped builds; never the production path).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/bin/ubench tools/ubench.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "../bobe_amd/csrc/kernels.hpp"
using namespace bobe;
#ifndef UB_STRIPS
#define UB_STRIPS 3      // 16-row strips per panel workgroup (3 or 4)
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
  const int N = 1024;   // 8 blocks
  std::vector<double> K((size_t)N * N);
  // SPD matrix: RBF kernel of points on a line + noise
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) {
      double d = (i - j) * 0.37 / N * 40.0;
      K[(size_t)i * N + j] = exp(-0.5 * d * d) + (i == j ? 1e-3 : 0.0);
    }
  double *A, *Linv;
  unsigned long long* st;
  int* info;
  CK(hipMalloc(&A, K.size() * 8));
  CK(hipMalloc(&Linv, K.size() * 8));
  CK(hipMalloc(&st, 320 * 8));
  CK(hipMemset(st, 0, 320 * 8));
  CK(hipMalloc(&info, 4));
  CK(hipMemcpy(A, K.data(), K.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemset(Linv, 0, K.size() * 8));
  CK(hipMemset(info, 0x7f, 4));
  CK(hipFuncSetAttribute((const void*)k_potf2<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, POTF2_SMEM_BYTES));
  CK(hipFuncSetAttribute((const void*)k_potf2<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, POTF2_SMEM_BYTES));
  CK(hipFuncSetAttribute((const void*)k_trsm_panel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, TRSM_SMEM_BYTES));
  unsigned long long h[320];
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemcpy(A, K.data(), K.size() * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((k_potf2<true, true>), dim3(1), dim3(256), POTF2_SMEM_BYTES, 0, A, (int64_t)N, Linv, (int64_t)N, 0, info, st);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, st, 64 * 8, hipMemcpyDeviceToHost));
    printf("potf2 rep %d (cycles): load %llu |", rep, h[1] - h[0]);
    unsigned long long ta = 0, tb = 0, tc = 0;
    for (int p = 0; p < 8; ++p) {
      unsigned long long a = h[3 + 3 * p] - h[2 + 3 * p], b = h[4 + 3 * p] - h[3 + 3 * p];
      unsigned long long c = (p < 7 ? h[2 + 3 * (p + 1)] : h[26]) - h[4 + 3 * p];
      ta += a; tb += b; tc += c;
      if (rep == 2) printf(" p%d a=%llu b=%llu c=%llu |", p, a, b, c);
    }
    printf(" sum a=%llu b=%llu c=%llu | store %llu | diag-inv %llu | total %llu\n", ta, tb, tc, h[27] - h[26], h[28] - h[27], h[28] - h[0]);
    hipLaunchKernelGGL((k_trsm_panel<true>), dim3(2 * 7), dim3(256), TRSM_SMEM_BYTES, 0, A, (int64_t)N, (const double*)Linv, (int64_t)N, 0, st);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, st, 64 * 8, hipMemcpyDeviceToHost));
    printf("trsm  rep %d (cycles): load+stage %llu | compute %llu | store %llu | total %llu\n", rep, h[1] - h[0], h[2] - h[1], h[3] - h[2], h[3] - h[0]);
  }
  // wall-clock per kernel with events
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms;
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k_potf2<true, true>), dim3(1), dim3(256), POTF2_SMEM_BYTES, 0, A, (int64_t)N, Linv, (int64_t)N, 1, info, st);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  {
    CK(hipFuncSetAttribute((const void*)k_chol_panel<true, false, UB_STRIPS>, hipFuncAttributeMaxDynamicSharedMemorySize, POTF2_SMEM_BYTES));
    double* dg;
    CK(hipMalloc(&dg, (size_t)8 * 128 * 128 * 8));
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemcpy(A, K.data(), K.size() * 8, hipMemcpyHostToDevice));
      hipLaunchKernelGGL((k_chol_panel<true, false, UB_STRIPS>), dim3(panel_workgroups(7, UB_STRIPS), 1), dim3(PANEL_THREADS), POTF2_SMEM_BYTES, 0, A, (int64_t)N, (int64_t)0, Linv, (int64_t)N,
                         (int64_t)0, 0, panel_workgroups(7, UB_STRIPS), info, 128, dg, (int64_t)0, st, (const FillJob*)nullptr, 0, 0, (double*)nullptr, 7 * 128);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(h, st, 320 * 8, hipMemcpyDeviceToHost));
      if (rep == 2)          // per wave and step: start of phase A (after the loop-top stamp of wave 0), deferred tiles, side job
        for (int p = 1; p < 8; ++p) {
          printf("  waves p%d (start+, tiles, side):", p);
          for (int w = 0; w < 8; ++w) {
            const unsigned long long* q = h + 64 + 32 * w + 3 * p;
            if (w == 0) printf(" | w0 leaf %llu", q[2] - q[0]);
            else if (q[0]) printf(" | w%d +%lld %llu(%llu) %llu", w, (long long)(q[0] - h[2 + 3 * p]), q[1] - q[0], h[64 + 32 * w + 24 + p] ? h[64 + 32 * w + 24 + p] - q[0] : 0ull, q[2] - q[1]);
          }
          printf("\n");
        }
      unsigned long long ta = 0, tb = 0, tc = 0;
      for (int p = 0; p < 8; ++p) {
        ta += h[3 + 3 * p] - h[2 + 3 * p];
        tb += h[4 + 3 * p] - h[3 + 3 * p];
        tc += (p < 7 ? h[5 + 3 * p] : h[26]) - h[4 + 3 * p];
        if (rep == 1) printf("  panel p%d a=%llu b=%llu\n", p, h[3 + 3 * p] - h[2 + 3 * p], h[4 + 3 * p] - h[3 + 3 * p]);
      }
      printf("chol_panel rep %d (cycles): stage-in %llu | factor a=%llu b=%llu c=%llu | L_kk out %llu | row solve %llu | rows out %llu | total %llu\n",
             rep, h[1] - h[0], ta, tb, tc, h[27] - h[26], h[28] - h[27], h[29] - h[28], h[29] - h[0]);
    }
  }
  {  // block load with the block already in this XCD's L2 (second launch on untouched memory) vs first touch
    for (int rep = 0; rep < 2; ++rep) {
      if (rep == 0) CK(hipMemcpy(A, K.data(), K.size() * 8, hipMemcpyHostToDevice));
      hipLaunchKernelGGL((k_potf2<false, true>), dim3(1), dim3(256), POTF2_SMEM_BYTES, 0, A, (int64_t)N, Linv, (int64_t)N, 0, info, st);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(h, st, 64 * 8, hipMemcpyDeviceToHost));
      printf("block load (128 KB through registers into LDS, 4 waves), %s: %llu cycles\n", rep ? "again (L2 / MALL hot)" : "after a host upload", h[1] - h[0]);
    }
  }
  printf("potf2 wall %.2f us per launch (back-to-back)\n", ms * 1e3 / 20);
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k_trsm_panel<true>), dim3(2 * 7), dim3(256), TRSM_SMEM_BYTES, 0, A, (int64_t)N, (const double*)Linv, (int64_t)N, 0, st);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  printf("trsm  wall %.2f us per launch (back-to-back)\n", ms * 1e3 / 20);
  return 0;
}
