// Issue cost of the instruction kinds of the 16x16 leaf in ONE wave (independent instructions, no dependency stalls):
// fp64 FMA, v_readlane pair, readlane pair + FMA on the scalar (one rank-1 update element), LDS broadcast read + FMA.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/bin/ubench_issue tools/ubench_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__device__ __forceinline__ unsigned long long stamp(double& dep) {
  unsigned long long t;
  asm volatile("s_nop 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(dep)::"memory");
  return t;
}
__device__ __forceinline__ double rl(double v, int lane) {
  union { double d; int w[2]; } u;
  u.d = v;
  u.w[0] = __builtin_amdgcn_readlane(u.w[0], lane);
  u.w[1] = __builtin_amdgcn_readlane(u.w[1], lane);
  return u.d;
}
__global__ void k_issue(double* out, unsigned long long* cyc, double seed, int half) {
  __shared__ double lds[64];
  if (half && threadIdx.x >= 32) return;
  double r[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) r[i] = seed + i + threadIdx.x * 1e-9;
  lds[threadIdx.x] = seed * 0.5 + threadIdx.x;
  __syncthreads();
  double y = 1.0000001;
  unsigned long long t0 = stamp(r[0]);
#pragma unroll
  for (int i = 0; i < 256; ++i) r[i & 15] = __builtin_fma(r[i & 15], y, 1e-9);
  unsigned long long t1 = stamp(r[0]);
  double acc = 0.0;
#pragma unroll
  for (int i = 0; i < 128; ++i) acc += 0.0 * 0 + 0;   // (placeholder keeps register allocation similar)
  double s[8];
#pragma unroll
  for (int rep = 0; rep < 16; ++rep)
#pragma unroll
    for (int q = 0; q < 8; ++q) s[q] = rl(r[q], (rep + q) & 15);
  double ss = s[0] + s[1] + s[2] + s[3] + s[4] + s[5] + s[6] + s[7];
  unsigned long long t2 = stamp(ss);
#pragma unroll
  for (int rep = 0; rep < 8; ++rep)
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const double sc = rl(r[(c + 1) & 15], (rep + c) & 15);
      r[c] = __builtin_fma(-y, sc, r[c]);
    }
  unsigned long long t3 = stamp(r[0]);
#pragma unroll
  for (int rep = 0; rep < 8; ++rep)
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const double sc = lds[(rep * 16 + c) & 63];
      r[c] = __builtin_fma(-y, sc, r[c]);
    }
  unsigned long long t4 = stamp(r[0]);
  double o = ss + acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) o += r[i];
  out[threadIdx.x] = o;
  if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3; }
}
__device__ __forceinline__ double rsqrt_nr(double a) {
  double y = __builtin_amdgcn_rsq(a);
  double e = __builtin_fma(-a * y, y, 1.0);
  y = __builtin_fma(0.5 * y, e, y);
  e = __builtin_fma(-a * y, y, 1.0);
  y = __builtin_fma(0.5 * y, e, y);
  return y;
}
// the 16x16 leaf of k_potf2 (row per lane, lanes 16..31 carry identity rows), 8 times back to back on fresh data
__global__ void k_leaf(double* out, unsigned long long* cyc, const double* __restrict__ in, int spin) {
  __shared__ double S[16 * 17];
  const int lane = threadIdx.x;
  if (threadIdx.x >= 64) {                      // companion waves: idle (exit) or busy with VALU work for a while
    double a = 1.0 + threadIdx.x * 1e-9;
    for (int i = 0; i < spin; ++i) a = __builtin_fma(a, 0.999999, 1e-9);
    if (a == 123.456) out[threadIdx.x] = a;
    return;
  }
  if (lane >= 32) return;
  const int li = lane & 15;
  const bool ident = lane >= 16;
  double tot = 0.0;
  unsigned long long tsum = 0;
  for (int rep = 0; rep < 8; ++rep) {
    if (!ident)
      for (int c = 0; c < 16; ++c) S[li * 17 + c] = in[rep * 256 + li * 16 + c];
    __builtin_amdgcn_s_waitcnt(0);
    double r[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) r[c] = ident ? ((c == li) ? 1.0 : 0.0) : S[li * 17 + c];
    unsigned long long t0 = stamp(r[0]);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const double ajj = rl(r[j], j);
      const double inv = rsqrt_nr(ajj);
      r[j] *= inv;
#pragma unroll
      for (int c = j + 1; c < 16; ++c) {
        const double lcj = rl(r[j], c);
        r[c] = __builtin_fma(-r[j], lcj, r[c]);
      }
    }
    unsigned long long t1 = stamp(r[15]);
    tsum += t1 - t0;
#pragma unroll
    for (int c = 0; c < 16; ++c) tot += r[c];
  }
  out[lane] = tot;
  if (lane == 0) cyc[0] = tsum;
}

int main() {
  double* out; unsigned long long* cyc;
  CK(hipMalloc(&out, 64 * 8)); CK(hipMalloc(&cyc, 8 * 8));
  unsigned long long h[8];
  for (int half = 0; half < 2; ++half) {
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(k_issue, dim3(1), dim3(64), 0, 0, out, cyc, 1.0 + rep, half);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost));
    }
    printf("%s lanes active:\n", half ? "32" : "64");
    printf("  independent v_fma_f64 (16 chains):   %.1f cycles each\n", h[0] / 256.0);
    printf("  v_readlane pair (64-bit broadcast):   %.1f cycles per pair\n", h[1] / 128.0);
    printf("  readlane pair + fma (one update):     %.1f cycles each\n", h[2] / 128.0);
    printf("  LDS broadcast read + fma:             %.1f cycles each\n", h[3] / 128.0);
  }
  // leaf
  double hin[8 * 256];
  for (int rep = 0; rep < 8; ++rep)
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) hin[rep * 256 + i * 16 + j] = (i == j ? 20.0 + rep : 0.0) + 1.0 / (1.0 + i + j);
  double* din;
  CK(hipMalloc(&din, sizeof(hin)));
  CK(hipMemcpy(din, hin, sizeof(hin), hipMemcpyHostToDevice));
  for (int cfg = 0; cfg < 3; ++cfg) {
    const int threads = cfg == 0 ? 64 : 256, spin = cfg == 2 ? 20000 : 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(k_leaf, dim3(1), dim3(threads), 0, 0, out, cyc, (const double*)din, spin);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost));
    }
    printf("16x16 leaf, %s: %.0f cycles per leaf = %.1f per column\n",
           cfg == 0 ? "one wave alone" : (cfg == 1 ? "three companion waves that exit at once" : "three companion waves busy with FMAs"),
           h[0] / 8.0, h[0] / 128.0);
  }
  return 0;
}
