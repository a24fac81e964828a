// What rate does a plain streaming kernel reach at the SIZE of one Conv-TasNet TCN kernel
// (65.5 MB in, 65.5 MB out; 24 different buffer pairs in a row, as the 24 blocks of a step)?
//   hipcc --offload-arch=gfx950 -O3 tools/sizebench.hip -o tools/sizebench && tools/sizebench
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void copy(const uint4* in, uint4* out, long long n) {
  for (long long i = (long long)blockIdx.x*blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x*blockDim.x) out[i] = in[i];
}
// 2 reads + 1 write (dz-like), 3 reads + 1 write (dwconv_bwd-like)
__global__ __launch_bounds__(256) void rrw(const uint4* a, const uint4* b, uint4* out, long long n) {
  for (long long i = (long long)blockIdx.x*blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x*blockDim.x) { uint4 x = a[i], y = b[i]; x.x ^= y.x; x.y += y.y; x.z ^= y.z; x.w += y.w; out[i] = x; }
}
__global__ __launch_bounds__(256) void rrrw(const uint4* a, const uint4* b, const uint4* c, uint4* out, long long n) {
  for (long long i = (long long)blockIdx.x*blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x*blockDim.x) { uint4 x = a[i], y = b[i], z = c[i]; x.x ^= y.x + z.x; x.y += y.y ^ z.y; x.z ^= y.z; x.w += z.w; out[i] = x; }
}
int main() {
  const long long bytes = 16LL*3999*512*2;      // one [B][T][512] bf16 tensor
  const long long n16 = bytes/16;
  const int NB = 24;
  uint4* buf[4][NB];
  for (int k = 0; k < 4; ++k) for (int i = 0; i < NB; ++i) { hipMalloc(&buf[k][i], bytes); hipMemset(buf[k][i], i + k, bytes); }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, double moved, auto launch) {
    for (int i = 0; i < NB; ++i) launch(i);
    hipEventRecord(e0);
    const int rounds = 5;
    for (int r = 0; r < rounds; ++r) for (int i = 0; i < NB; ++i) launch(i);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %7.1f us per launch  %7.1f GB/s (gaps included)\n", name, ms*1e3/(rounds*NB), moved*rounds*NB/ms/1e6);
  };
  for (int grid : {1024, 2048, 4096, 16384}) {
    printf("grid %d x 256\n", grid);
    run("copy  (1 read + 1 write)", 2.0*bytes, [&](int i) { copy<<<grid, 256>>>(buf[0][i], buf[1][i], n16); });
    run("rrw   (2 reads + 1 write)", 3.0*bytes, [&](int i) { rrw<<<grid, 256>>>(buf[0][i], buf[1][i], buf[2][i], n16); });
    run("rrrw  (3 reads + 1 write)", 4.0*bytes, [&](int i) { rrrw<<<grid, 256>>>(buf[0][i], buf[1][i], buf[2][i], buf[3][i], n16); });
  }
  return 0;
}
