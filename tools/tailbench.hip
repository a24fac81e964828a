// How long after the last wave of a kernel exits does the next kernel start, as a
// function of what the kernel did (bytes written / read, LDS size, grid)?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__device__ __forceinline__ long long rt() {
  long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t;
}
__global__ void stamp(long long* out, int slot) {
  long long t = rt();
  if (threadIdx.x == 0) { out[slot] = t; if (slot == 0) { out[2] = 0x7fffffffffffffffLL; out[3] = 0; } }
}
template <int MODE>   // 0 write, 1 read, 2 copy (read+write), 3 write nontemporal
__global__ __launch_bounds__(256) void work(uint4* dst, const uint4* src, long long n16, long long* out, int lds_touch) {
  extern __shared__ unsigned char dyn[];
  long long t0 = rt();
  if (lds_touch && threadIdx.x == 0) dyn[0] = 1;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < n16; i += (long long)gridDim.x*256) {
    if (MODE == 0) dst[i] = make_uint4((unsigned)i, 1, 2, 3);
    else if (MODE == 3) { typedef unsigned int u4 __attribute__((ext_vector_type(4))); u4 v = {(unsigned)i, 1, 2, 3}; __builtin_nontemporal_store(v, (u4*)(dst + i)); }
    else if (MODE == 1) { uint4 v = src[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    else dst[i] = src[i];
  }
  if (MODE == 1 && acc.x == 0x12345678u && acc.y == 1) dst[0] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicMin((unsigned long long*)(out + 2), (unsigned long long)t0);
    atomicMax((unsigned long long*)(out + 3), (unsigned long long)rt());
  }
}
int main() {
  const long long maxb = 1LL << 30;
  uint4 *a, *b; long long* out;
  CK(hipMalloc(&a, maxb)); CK(hipMalloc(&b, maxb)); CK(hipMalloc(&out, 64));
  CK(hipMemset(a, 1, maxb)); CK(hipMemset(b, 1, maxb));
  const char* names[] = {"write", "read", "copy", "write-nt"};
  for (int mode = 0; mode < 4; ++mode)
    for (long long mb : {1, 16, 64, 128, 256, 1024})
      for (int lds : {0, 65536}) {
        if (mode == 2 && mb > 512) continue;
        const long long n16 = mb*1024*1024/16;
        int grid = 2048;
        double s_pre = 0, s_run = 0, s_tail = 0; const int reps = 5;
        for (int r = 0; r < reps + 1; ++r) {
          hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, 0, out, 0);
          if (mode == 0) hipLaunchKernelGGL(work<0>, dim3(grid), dim3(256), lds, 0, a, b, n16, out, lds > 0);
          if (mode == 1) hipLaunchKernelGGL(work<1>, dim3(grid), dim3(256), lds, 0, a, b, n16, out, lds > 0);
          if (mode == 2) hipLaunchKernelGGL(work<2>, dim3(grid), dim3(256), lds, 0, a, b, n16, out, lds > 0);
          if (mode == 3) hipLaunchKernelGGL(work<3>, dim3(grid), dim3(256), lds, 0, a, b, n16, out, lds > 0);
          hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, 0, out, 1);
          CK(hipDeviceSynchronize());
          long long h[4]; CK(hipMemcpy(h, out, 32, hipMemcpyDeviceToHost));
          if (r == 0) continue;
          s_pre += (h[2] - h[0])/100.0; s_run += (h[3] - h[2])/100.0; s_tail += (h[1] - h[3])/100.0;
        }
        printf("%-8s %5lld MB lds %5d: prev-end->first-entry %6.2f us  waves active %8.2f us (%6.0f GB/s)  last-exit->next-start %6.2f us\n",
               names[mode], mb, lds, s_pre/reps, s_run/reps, mb*1.048576e6/(s_run/reps*1e-6)/1e9*(mode == 2 ? 2 : 1), s_tail/reps);
      }
  return 0;
}
