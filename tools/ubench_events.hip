// Cross-stream event hop latency: two streams ping-pong tiny kernels through hipEventRecord / hipStreamWaitEvent.
// Plain streams vs streams created through the CU-mask entry point (own hardware queues).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/bin/ubench_events tools/ubench_events.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_tiny(int* p) { if (threadIdx.x == 0) atomicAdd(p, 1); }

static int run(hipStream_t a, hipStream_t b, int* d, const char* name) {
  const int hops = 400;
  std::vector<hipEvent_t> ev(2 * hops);
  for (auto& evt : ev) CK(hipEventCreateWithFlags(&evt, hipEventDisableTiming));
  // baseline: the same number of kernels on ONE stream
  CK(hipDeviceSynchronize());
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 2 * hops; ++i) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, a, d);
  CK(hipStreamSynchronize(a));
  auto t1 = std::chrono::steady_clock::now();
  for (int i = 0; i < hops; ++i) {
    hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, a, d);
    CK(hipEventRecord(ev[2 * i], a));
    CK(hipStreamWaitEvent(b, ev[2 * i], 0));
    hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, b, d);
    CK(hipEventRecord(ev[2 * i + 1], b));
    CK(hipStreamWaitEvent(a, ev[2 * i + 1], 0));
  }
  CK(hipStreamSynchronize(a));
  CK(hipStreamSynchronize(b));
  auto t2 = std::chrono::steady_clock::now();
  const double one = std::chrono::duration<double, std::micro>(t1 - t0).count() / (2 * hops);
  const double pp = std::chrono::duration<double, std::micro>(t2 - t1).count() / (2 * hops);
  printf("%-22s same-stream kernel-to-kernel %.2f us | cross-stream hop %.2f us\n", name, one, pp);
  for (auto& evt : ev) (void)hipEventDestroy(evt);
  return 0;
}

int main() {
  int* d;
  CK(hipMalloc(&d, 4));
  CK(hipMemset(d, 0, 4));
  hipStream_t a, b, c, q;
  CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  std::vector<uint32_t> mask((prop.multiProcessorCount + 31) / 32, 0xffffffffu);
  CK(hipExtStreamCreateWithCUMask(&c, (uint32_t)mask.size(), mask.data()));
  CK(hipExtStreamCreateWithCUMask(&q, (uint32_t)mask.size(), mask.data()));
  for (int rep = 0; rep < 2; ++rep) {
    if (run(a, b, d, "plain streams")) return 1;
    if (run(c, q, d, "own-queue streams")) return 1;
  }
  return 0;
}
