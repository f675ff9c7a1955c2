// Store-bandwidth ceiling of the kernel-assembly write pattern (tile of rows x cols per workgroup, row stride ld).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/bin/ubench_store tools/ubench_store.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// each thread owns VEC adjacent columns; the workgroup covers 256*VEC columns x ROWS rows
template <int VEC, int ROWS, int WORK>
__global__ __launch_bounds__(256) void k_store(double* __restrict__ out, int64_t ld, int tiles_x, double seed) {
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
  const int64_t col = (int64_t)tx * 256 * VEC + threadIdx.x * VEC;
  double v = seed + threadIdx.x;
  for (int r = 0; r < ROWS; ++r) {
    for (int w = 0; w < WORK; ++w) v = __builtin_fma(v, 1.0000001, 1e-9);   // dependent arithmetic per element
    double* p = out + ((int64_t)ty * ROWS + r) * ld + col;
    if (VEC == 1) p[0] = v;
    else { typedef double v2 __attribute__((ext_vector_type(2))); *reinterpret_cast<v2*>(p) = (v2){v, v + 1.0}; }
  }
}

template <int VEC, int ROWS, int WORK>
int run(double* out, int64_t rows, int64_t cols, const char* name) {
  const int tiles_x = (int)(cols / (256 * VEC)), tiles_y = (int)(rows / ROWS);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_store<VEC, ROWS, WORK>), dim3(tiles_x * tiles_y), dim3(256), 0, 0, out, cols, tiles_x, 1.0 + rep);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  printf("%-34s %lldx%lld: %7.1f us  %5.2f TB/s\n", name, (long long)rows, (long long)cols, best * 1e3, rows * cols * 8.0 / (best * 1e-3) / 1e12);
  return 0;
}

int main() {
  const int64_t rows = 4096, cols = 8192;
  double* out;
  CK(hipMalloc(&out, rows * cols * 8));
  if (run<1, 64, 0>(out, rows, cols, "8B/lane 64 rows/WG, no work")) return 1;
  if (run<1, 128, 0>(out, rows, cols, "8B/lane 128 rows/WG, no work")) return 1;
  if (run<1, 16, 0>(out, rows, cols, "8B/lane 16 rows/WG, no work")) return 1;
  if (run<2, 64, 0>(out, rows, cols, "16B/lane 64 rows/WG, no work")) return 1;
  if (run<2, 16, 0>(out, rows, cols, "16B/lane 16 rows/WG, no work")) return 1;
  if (run<1, 64, 40>(out, rows, cols, "8B/lane 64 rows/WG, 40 dep FMAs")) return 1;
  if (run<2, 64, 40>(out, rows, cols, "16B/lane 64 rows/WG, 40 dep FMAs")) return 1;
  if (run<1, 16, 40>(out, rows, cols, "8B/lane 16 rows/WG, 40 dep FMAs")) return 1;
  if (run<1, 64, 80>(out, rows, cols, "8B/lane 64 rows/WG, 80 dep FMAs")) return 1;
  if (run<1, 16, 80>(out, rows, cols, "8B/lane 16 rows/WG, 80 dep FMAs")) return 1;
  return 0;
}
