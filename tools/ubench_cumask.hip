// Diagnostic: which physical CUs (XCD, SE, CU) a CU-masked stream really runs on, per mask-bit pattern.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/bin/ubench_cumask tools/ubench_cumask.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <map>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_where(uint32_t* out, int spin) {
  uint32_t hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  // keep the CU busy a little so the dispatcher spreads the grid
  double x = threadIdx.x;
  for (int i = 0; i < spin; ++i) x = x * 1.0000001 + 1e-9;
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = (xcc & 0xf) | (x == 12345.678 ? 0x100 : 0);
  }
}

static int run(hipStream_t st, uint32_t* d, std::vector<uint32_t>& h, int nwg, const char* label) {
  hipLaunchKernelGGL(k_where, dim3(nwg), dim3(256), 0, st, d, 20000);
  CK(hipStreamSynchronize(st));
  CK(hipMemcpy(h.data(), d, nwg * 8, hipMemcpyDeviceToHost));
  std::map<int, std::set<int>> per_xcc;
  for (int i = 0; i < nwg; ++i) {
    const uint32_t hw = h[2 * i];
    const int cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
    per_xcc[h[2 * i + 1] & 0xf].insert(se * 100 + sh * 16 + cu);
  }
  int total = 0;
  printf("%-28s", label);
  for (auto& kv : per_xcc) { printf(" x%d:%zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
  printf("  => %d CUs\n", total);
  return 0;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
  printf("CUs %d\n", ncu);
  const int nwg = 8192;
  uint32_t* d;
  CK(hipMalloc(&d, nwg * 8));
  std::vector<uint32_t> h(2 * nwg);
  hipStream_t s0;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  run(s0, d, h, nwg, "no mask");
  struct Pat { const char* name; int kind, a, b; };
  const Pat pats[] = {{"bits 0..63", 0, 0, 64},      {"bits 0..31", 0, 0, 32},      {"bits 32..63", 0, 32, 64},
                      {"bits 0..7", 0, 0, 8},        {"bit 0", 0, 0, 1},            {"bit 1", 0, 1, 2},
                      {"bit 8", 0, 8, 9},            {"bit 9", 0, 9, 10},           {"bit 16", 0, 16, 17},
                      {"bits 192..255", 0, 192, 256}, {"c%4==0", 1, 4, 0},           {"c%8==0", 1, 8, 0},
                      {"c%8 in {0,1}", 2, 8, 2},     {"c%8 in {0..3}", 2, 8, 4},    {"c%16 in {0,1}", 2, 16, 2},
                      {"c%5==0", 1, 5, 0}};
  for (const Pat& p : pats) {
    std::vector<uint32_t> mask(words, 0u);
    for (int c = 0; c < ncu; ++c) {
      const bool on = p.kind == 0 ? (c >= p.a && c < p.b) : p.kind == 1 ? (c % p.a == p.b) : (c % p.a < p.b);
      if (on) mask[c / 32] |= 1u << (c % 32);
    }
    hipStream_t st;
    if (hipExtStreamCreateWithCUMask(&st, words, mask.data()) != hipSuccess) { printf("%s: create failed\n", p.name); continue; }
    run(st, d, h, nwg, p.name);
    CK(hipStreamDestroy(st));
  }
  return 0;
}
