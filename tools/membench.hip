// Write/copy bandwidth calibration for the store patterns used by the kernels.
//   hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o tools/membench && tools/membench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

// pattern 0: each wave stores whole 1 KiB rows (64 lanes x 16 B)
// pattern 1: each wave-instruction stores 8 rows x 128 B (8 lanes per row), rows 1 KiB apart
// pattern 2: 16 rows x 64 B
template <int PAT>
__global__ __launch_bounds__(256) void fill(uint4* out, long long rows) {   // rows of 1 KiB
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)blockIdx.x*blockDim.x + threadIdx.x) >> 6;
  const long long nw = ((long long)gridDim.x*blockDim.x) >> 6;
  const uint4 v = make_uint4(lane, 1, 2, 3);
  if (PAT == 0) {
    for (long long r = wave; r < rows; r += nw) out[r*64 + lane] = v;
  } else if (PAT == 1) {
    // a wave owns 8 consecutive rows and walks the 8 column groups of 128 B
    for (long long g = wave; g < rows/8; g += nw)
      for (int c = 0; c < 8; ++c)
        out[(g*8 + (lane >> 3))*64 + c*8 + (lane & 7)] = v;
  } else {
    for (long long g = wave; g < rows/16; g += nw)
      for (int c = 0; c < 16; ++c)
        out[(g*16 + (lane >> 2))*64 + c*4 + (lane & 3)] = v;
  }
}
__global__ __launch_bounds__(256) void copy(const uint4* in, uint4* out, long long n) {
  for (long long i = (long long)blockIdx.x*blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x*blockDim.x) out[i] = in[i];
}
__global__ __launch_bounds__(256) void rdsum(const uint4* in, uint4* out, long long n) {
  uint4 a = make_uint4(0, 0, 0, 0);
  for (long long i = (long long)blockIdx.x*blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x*blockDim.x) { uint4 v = in[i]; a.x ^= v.x; a.y ^= v.y; a.z ^= v.z; a.w ^= v.w; }
  if (a.x == 0x12345678u) out[0] = a;
}

int main() {
  const long long bytes = 256LL << 20;          // 256 MiB buffers
  const long long rows = bytes/1024, n16 = bytes/16;
  uint4 *a, *b;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes);
  hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, double moved, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    const int it = 20;
    for (int i = 0; i < it; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %8.1f us  %7.1f GB/s\n", name, ms*1e3/it, moved*it/ms/1e6);
  };
  for (int grid : {1024, 2048, 8192}) {
    printf("grid %d x 256\n", grid);
    run("fill 1KiB rows", bytes, [&] { fill<0><<<grid, 256>>>(a, rows); });
    run("fill 8 rows x 128B per instr", bytes, [&] { fill<1><<<grid, 256>>>(a, rows); });
    run("fill 16 rows x 64B per instr", bytes, [&] { fill<2><<<grid, 256>>>(a, rows); });
    run("copy (read + write)", 2.0*bytes, [&] { copy<<<grid, 256>>>(a, b, n16); });
    run("read only", bytes, [&] { rdsum<<<grid, 256>>>(a, b, n16); });
  }
  return 0;
}
