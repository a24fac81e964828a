// Measures the shader clock and the s_memtime tick rate: a dependent v_fma chain of known
// length (wave64 VALU op = 4 cycles issue; dependent chain ~ 4-8 cycles per op).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void chain(float* out, long long* ticks, int n, int heavy) {
  float x = threadIdx.x*1e-9f, y = 1.0000001f;
  long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
  long long c0 = clock64();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int j = 0; j < 64; ++j) x = __builtin_fmaf(x, y, 1e-7f);
  }
  long long c1 = clock64();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
  if (threadIdx.x == 0) { ticks[2*blockIdx.x] = t0; ticks[2*blockIdx.x+1] = t1; }
  out[blockIdx.x*blockDim.x + threadIdx.x] = x;
}
int main() {
  float* out; long long* ticks;
  hipMalloc(&out, 4096*1024*4); hipMalloc(&ticks, 4096*16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int grids[] = {1, 256, 1024, 2048};
  int thr[] = {64, 64, 256, 512};
  for (int rep = 0; rep < 2; ++rep)
  for (int g = 0; g < 4; ++g) for (int n : {100, 1000, 10000}) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(chain, dim3(grids[g]), dim3(thr[g]), 0, 0, out, ticks, n, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    static long long h[8192]; hipMemcpy(h, ticks, 16*grids[g], hipMemcpyDeviceToHost);
    // group blocks by counter base (XCDs have unsynchronised counters): cluster by entry within 1e8
    long long best_span = 0; double dur = 0;
    for (int i = 0; i < grids[g]; ++i) {
      dur += (double)(h[2*i+1] - h[2*i]);
      long long mn = h[2*i], mx = h[2*i+1];
      for (int j = 0; j < grids[g]; ++j) {
        long long d = h[2*j] - h[2*i]; if (d < 0) d = -d;
        if (d < 100000000LL) { if (h[2*j] < mn) mn = h[2*j]; if (h[2*j+1] > mx) mx = h[2*j+1]; }
      }
      if (mx - mn > best_span) best_span = mx - mn;
    }
    dur /= grids[g];
    printf("grid %4d x %3d n=%5d  %.1f us  mean WG ticks=%.0f  max same-XCD span=%lld ticks -> %.0f MHz (span/time)\n",
           grids[g], thr[g], n, ms*1e3, dur, best_span, best_span/(ms*1e3));
  }
  return 0;
}
