// Are 16-byte global / buffer accesses at 4-byte-aligned (not 16-byte-aligned) addresses legal
// and how fast are they on gfx950? Copies n floats from src+off to dst+off with x4 accesses.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../brever_amd/csrc/common.cuh"
using namespace brv;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void copy_global(const float* s, float* d, long long n4) {
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < n4; i += (long long)gridDim.x*256) {
    f4 v; __builtin_memcpy(&v, s + 4*i, 16);          // alignment 4 assumed by the compiler
    __builtin_memcpy(d + 4*i, &v, 16);
  }
}
__global__ void copy_buffer(const float* s, float* d, long long n4, long long bytes) {
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(s, bytes), rd = make_rsrc(d, bytes);
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < n4; i += (long long)gridDim.x*256)
    buf_store16(rd, (unsigned int)(16*i), buf_load16(rs, (unsigned int)(16*i)));
}
int main() {
  const long long n = 64 << 20;
  float *s, *d; CK(hipMalloc(&s, n*4 + 64)); CK(hipMalloc(&d, n*4 + 64));
  float* h = (float*)malloc(n*4 + 64);
  for (long long i = 0; i < n + 16; ++i) h[i] = (float)(i % 9973);
  CK(hipMemcpy(s, h, n*4 + 64, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int off = 0; off < 4; ++off) for (int kind = 0; kind < 2; ++kind) {
    CK(hipMemset(d, 0, n*4 + 64));
    float best = 1e9;
    for (int r = 0; r < 4; ++r) {
      hipEventRecord(e0);
      if (kind == 0) hipLaunchKernelGGL(copy_global, dim3(4096), dim3(256), 0, 0, s + off, d + off, n/4);
      else hipLaunchKernelGGL(copy_buffer, dim3(4096), dim3(256), 0, 0, s + off, d + off, n/4, n*4);
      hipEventRecord(e1); CK(hipEventSynchronize(e1));
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    CK(hipMemcpy(h, d, n*4 + 64, hipMemcpyDeviceToHost));
    long long bad = 0;
    for (long long i = 0; i < n; ++i) if (h[i + off] != (float)((i + off) % 9973)) ++bad;
    printf("%s offset %d floats: %.1f us, %.2f TB/s, mismatches %lld\n", kind ? "buffer" : "global", off, best*1e3, 2.0*n*4/best/1e9, bad);
    for (long long i = 0; i < n + 16; ++i) h[i] = (float)(i % 9973);
  }
  return 0;
}
