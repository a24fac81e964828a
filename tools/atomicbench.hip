// Cost of contended global atomics on MI355X: n_wg workgroups, each wave-0 lane-0 issues
// `per` atomic adds at the END of a short kernel; addresses spread over `naddr` slots with
// a byte stride. Reports kernel time (events) minus the same kernel without atomics.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <typename T>
__global__ __launch_bounds__(256) void k(T* dst, int naddr, int stride_elems, int per, int waves, float* sink) {
  float x = threadIdx.x;
  for (int i = 0; i < 200; ++i) x = x*1.0001f + 0.5f;      // ~1 us of work
  if (x == 123.f) sink[0] = x;
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0 && wid < waves)
    for (int i = 0; i < per; ++i) {
      const int slot = (blockIdx.x*waves + wid + i*7) % naddr;
      if (sizeof(T) == 8) __hip_atomic_fetch_add(dst + (long long)slot*stride_elems, (T)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else __hip_atomic_fetch_add(dst + (long long)slot*stride_elems, (T)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
template <typename T>
float run(T* buf, int grid, int naddr, int stride_bytes, int per, int waves, float* sink) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int r = 0; r < 6; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<T>, dim3(grid), dim3(256), 0, 0, buf, naddr, stride_bytes/(int)sizeof(T), per, waves, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (r > 0 && ms < best) best = ms;
  }
  return best*1e3f;
}
int main() {
  double* d; float* f; float* sink;
  CK(hipMalloc(&d, 64 << 20)); CK(hipMalloc(&f, 64 << 20)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(d, 0, 64 << 20)); CK(hipMemset(f, 0, 64 << 20));
  for (int grid : {256, 512, 2048}) {
    const float base = run<double>(d, grid, 1, 8, 0, 0, sink);
    printf("grid %4d: no atomics %.1f us\n", grid, base);
    for (int waves : {1, 4})
      for (int naddr : {1, 2, 32, 256})
        for (int stride : {8, 128, 4096}) {
          if (naddr == 1 && stride != 8) continue;
          const float t64 = run<double>(d, grid, naddr, stride, 2, waves, sink);
          const float t32 = run<float>(f, grid, naddr, stride, 2, waves, sink);
          const int n = grid*waves*2;
          printf("  waves %d naddr %3d stride %4d B: f64 +%.1f us (%.1f ns/atomic)   f32 +%.1f us (%.1f ns/atomic)   [%d atomics]\n",
                 waves, naddr, stride, t64 - base, (t64 - base)*1e3/n, t32 - base, (t32 - base)*1e3/n, n);
        }
  }
  return 0;
}
