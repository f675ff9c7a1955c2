// Dependent-issue latencies of the instructions on the diagonal factorisation's pivot chain (one wave).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/bin/ubench_lat tools/ubench_lat.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// timestamp ordered after `dep` is produced and before it is consumed again
__device__ __forceinline__ unsigned long long stamp(double& dep) {
  unsigned long long t;
  asm volatile("s_nop 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(dep)::"memory");
  return t;
}

__global__ void k_lat(double* out, unsigned long long* cyc, double seed) {
  double x = seed + threadIdx.x * 1e-9, y = 1.0000001;
  unsigned long long t0 = stamp(x);
#pragma unroll
  for (int i = 0; i < 256; ++i) x = __builtin_fma(x, y, 1e-9);
  unsigned long long t1 = stamp(x);
  double r = x;
#pragma unroll
  for (int i = 0; i < 64; ++i) r = __builtin_amdgcn_rsq(r + 2.0);
  unsigned long long t2 = stamp(r);
  v4d acc = {r, x, r, x};
  double a = x * 1e-3;
#pragma unroll
  for (int i = 0; i < 64; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0);
  double a0 = acc[0];
  unsigned long long t3 = stamp(a0);
  acc[0] = a0;
  // mfma -> readlane -> valu -> mfma round trip
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    union { double d; int w[2]; } u;
    u.d = acc[0];
    u.w[0] = __builtin_amdgcn_readlane(u.w[0], 5);
    u.w[1] = __builtin_amdgcn_readlane(u.w[1], 5);
    a = a * u.d;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0);
  }
  a0 = acc[0];
  unsigned long long t4 = stamp(a0);
  acc[0] = a0;
  // 32-bit dependent chain for reference
  float f = (float)seed;
#pragma unroll
  for (int i = 0; i < 256; ++i) f = __builtin_fmaf(f, 1.0001f, 1e-6f);
  double fd = f;
  unsigned long long t5 = stamp(fd);
  f = (float)fd;
  out[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3] + s + f;
  if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3; cyc[4] = t5 - t4; }
}

int main() {
  double* out; unsigned long long* cyc;
  CK(hipMalloc(&out, 64 * 8)); CK(hipMalloc(&cyc, 8 * 8));
  unsigned long long h[8];
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k_lat, dim3(1), dim3(64), 0, 0, out, cyc, 1.0 + rep);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost));
  }
  printf("dependent v_fma_f64:            %.1f cycles each\n", h[0] / 256.0);
  printf("dependent v_rsq_f64 (+add):     %.1f cycles each\n", h[1] / 64.0);
  printf("dependent mfma_f64_16x16x4:     %.1f cycles each\n", h[2] / 64.0);
  printf("mfma -> readlane -> mul -> mfma: %.1f cycles per round\n", h[3] / 64.0);
  printf("dependent v_fma_f32:            %.1f cycles each\n", h[4] / 256.0);
  return 0;
}
