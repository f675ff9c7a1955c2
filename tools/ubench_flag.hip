// Cost of an in-launch producer -> consumers hand-off on gfx950 (research for a fused potf2+trsm / persistent
// Cholesky): WG0 publishes a 2-KiB payload + flag per round; 62 consumer workgroups poll, acquire, read.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/bin/ubench_flag tools/ubench_flag.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ROUNDS = 16, PAY = 256;   // 256 doubles = 2 KiB

__global__ __launch_bounds__(256) void k_handoff(double* payload, int* flag, unsigned long long* prod_t,
                                                 unsigned long long* cons_t, int* bad, int work_cycles) {
  const int t = threadIdx.x;
  if (blockIdx.x == 0) {
    for (int r = 1; r <= ROUNDS; ++r) {
      // stand-in for one potf2 sub-step
      unsigned long long t0 = __builtin_amdgcn_s_memtime();
      while ((long long)(__builtin_amdgcn_s_memtime() - t0) < work_cycles) {}
      unsigned long long a = __builtin_amdgcn_s_memrealtime();
      payload[(r & 1) * PAY + t] = (double)(r * 1000 + t);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (t == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(flag, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
      unsigned long long b = __builtin_amdgcn_s_memrealtime();
      if (t == 0) { prod_t[2 * r] = a; prod_t[2 * r + 1] = b; }
    }
  } else {
    __shared__ int ok;
    for (int r = 1; r <= ROUNDS; ++r) {
      if (t == 0) {
        int spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < r && spins < (1 << 22)) {
          __builtin_amdgcn_s_sleep(2);
          ++spins;
        }
        ok = spins < (1 << 22);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
      if (!ok) { if (t == 0) atomicAdd(bad, 1000000); return; }
      const double v = payload[(r & 1) * PAY + t];
      unsigned long long c = __builtin_amdgcn_s_memrealtime();
      if (v != (double)(r * 1000 + t)) atomicAdd(bad, 1);
      if (t == 0) cons_t[(blockIdx.x - 1) * (ROUNDS + 1) + r] = c;
      __syncthreads();
    }
  }
}

int main() {
  double* payload; int *flag, *bad; unsigned long long *pt, *ct;
  const int NC = 62;
  CK(hipMalloc(&payload, 2 * PAY * 8)); CK(hipMalloc(&flag, 4)); CK(hipMalloc(&bad, 4));
  CK(hipMalloc(&pt, (2 * ROUNDS + 2) * 8)); CK(hipMalloc(&ct, NC * (ROUNDS + 1) * 8));
  for (int work : {0, 2000, 10000}) {
    CK(hipMemset(flag, 0, 4)); CK(hipMemset(bad, 0, 4)); CK(hipMemset(payload, 0, 2 * PAY * 8));
    hipLaunchKernelGGL(k_handoff, dim3(1 + NC), dim3(256), 0, 0, payload, flag, pt, ct, bad, work);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> hp(2 * ROUNDS + 2), hc(NC * (ROUNDS + 1));
    int hb; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hp.data(), pt, hp.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hc.data(), ct, hc.size() * 8, hipMemcpyDeviceToHost));
    double pub = 0, lat_avg = 0, lat_max = 0;
    for (int r = 2; r <= ROUNDS; ++r) {
      pub += (hp[2 * r + 1] - hp[2 * r]) * 10.0;     // 100 MHz ticks -> ns
      double mx = 0, av = 0;
      for (int c = 0; c < NC; ++c) { double l = ((long long)(hc[c * (ROUNDS + 1) + r] - hp[2 * r])) * 10.0; av += l; if (l > mx) mx = l; }
      lat_avg += av / NC; lat_max += mx;
    }
    printf("work=%6d cycles: stale/timeouts=%d  producer publish cost %.0f ns  consumer sees data after %.0f ns (avg) %.0f ns (slowest)\n",
           work, hb, pub / (ROUNDS - 1), lat_avg / (ROUNDS - 1), lat_max / (ROUNDS - 1));
  }
  return 0;
}
