// Do hipGraphs shrink the ~4.5 us gap between dependent kernels of one stream?
// 240 kernels of ~20 us each: plain stream launches vs one captured graph.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ __launch_bounds__(256) void work(uint4* dst, const uint4* src, long long n16) {
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < n16; i += (long long)gridDim.x*256) dst[i] = src[i];
}
int main() {
  const long long bytes = 48LL << 20;            // ~20 us copy
  uint4 *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
  CK(hipMemset(a, 1, bytes));
  hipStream_t st; CK(hipStreamCreate(&st));
  const int N = 240;
  auto launch_all = [&]() {
    for (int i = 0; i < N; ++i)
      hipLaunchKernelGGL(work, dim3(1024), dim3(256), 0, st, (i & 1) ? a : b, (i & 1) ? b : a, bytes/16);
  };
  auto time_it = [&](auto fn, const char* name) {
    fn(); CK(hipStreamSynchronize(st));
    double best = 1e9;
    for (int r = 0; r < 5; ++r) {
      auto t0 = std::chrono::high_resolution_clock::now();
      fn(); CK(hipStreamSynchronize(st));
      auto t1 = std::chrono::high_resolution_clock::now();
      double us = std::chrono::duration<double, std::micro>(t1 - t0).count();
      if (us < best) best = us;
    }
    printf("%-28s %8.1f us total, %6.2f us per kernel\n", name, best, best/N);
    return best;
  };
  time_it(launch_all, "stream launches");
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  launch_all();
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  time_it([&]() { CK(hipGraphLaunch(ge, st)); }, "graph launch");
  // single big kernel equivalent for reference
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, st);
  hipLaunchKernelGGL(work, dim3(1024), dim3(256), 0, st, a, b, bytes/16);
  hipEventRecord(e1, st); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("one kernel alone (events): %.1f us\n", ms*1e3);
  return 0;
}
