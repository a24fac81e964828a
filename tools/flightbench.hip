// How many bytes must a CU keep in flight to stream at the HBM rate when ONE persistent workgroup per CU does the
// loading (the organisation of the fused TCN kernels)? Wave w of NW reads the 1 KB rows w, w + NW, ... of 64-row tiles
// (16 bytes per lane: one row per instruction), D rows requested ahead, optionally writes each row back (z2-like).
//   hipcc --offload-arch=gfx950 -O3 tools/flightbench.hip -o tools/flightbench && tools/flightbench
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, long long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(bytes > 0xffffffffLL ? 0xffffffffu : (unsigned)bytes), 0x00020000);
}
template <int NW, int D, bool WR, int SYNC>
__global__ __launch_bounds__(64*NW) void stream(const uint4* in, uint4* out, int n_tiles, unsigned* sink) {
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const __amdgpu_buffer_rsrc_t ri = rsrc(in, (long long)n_tiles*65536), ro = rsrc(out, (long long)n_tiles*65536);
  constexpr int FPW = 64/NW;
  u32x4 ring[D];
  unsigned acc = 0;
  // rows of this wave in order: tile blockIdx.x + i gridDim.x, row wid + NW j
  auto off = [&](int q) -> unsigned {
    const int i = q / FPW, j = q % FPW;
    const long long tile = blockIdx.x + (long long)i*gridDim.x;
    return tile < n_tiles ? (unsigned)(tile*65536 + (wid + NW*j)*1024 + lane*16) : 0xfffffff0u;
  };
  const int nq = ((n_tiles - (int)blockIdx.x + (int)gridDim.x - 1)/(int)gridDim.x)*FPW;
#pragma unroll
  for (int d = 0; d < D; ++d) ring[d] = __builtin_amdgcn_raw_buffer_load_b128(ri, (int)off(d), 0, 0);
  for (int q = 0; q < nq; q += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      u32x4 v = ring[d];
      acc += v.x ^ v.y ^ v.z ^ v.w;
      if (WR) { v.x += 1; __builtin_amdgcn_raw_buffer_store_b128(v, ro, (int)off(q + d), 0, 0); }
      ring[d] = __builtin_amdgcn_raw_buffer_load_b128(ri, (int)off(q + d + D), 0, 0);
    }
    if (SYNC && ((q + D) % FPW) == 0) __syncthreads();      // one barrier per tile, as the fused kernels
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
int main() {
  const long long bytes = 16LL*3999*512*2/65536*65536;
  const int n_tiles = (int)(bytes/65536);
  const int NB = 24;
  uint4 *a[NB], *b[NB]; unsigned* sink; hipMalloc(&sink, 64);
  for (int i = 0; i < NB; ++i) { hipMalloc(&a[i], bytes); hipMalloc(&b[i], bytes); hipMemset(a[i], i + 1, bytes); hipMemset(b[i], 0, bytes); }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, double moved, auto launch) {
    for (int i = 0; i < NB; ++i) launch(i);
    hipEventRecord(e0);
    const int rounds = 4;
    for (int r = 0; r < rounds; ++r) for (int i = 0; i < NB; ++i) launch(i);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %7.1f us  %7.1f GB/s\n", name, ms*1e3/(rounds*NB), moved*rounds*NB/ms/1e6);
  };
#define RUN(NW, D, WR, SY, G) { char nm[96]; snprintf(nm, 96, "NW %d D %2d %s sync %d grid %4d: %3d KB/CU", NW, D, WR ? "copy" : "read", SY, G, NW*D*(G/256)); \
    run(nm, (WR ? 2.0 : 1.0)*bytes, [&](int i) { stream<NW, D, WR, SY><<<G, 64*NW>>>(a[i], b[i], n_tiles, sink); }); }
  RUN(4, 2, false, 1, 256) RUN(4, 4, false, 1, 256) RUN(4, 8, false, 1, 256) RUN(4, 16, false, 1, 256)
  RUN(8, 2, false, 1, 256) RUN(8, 4, false, 1, 256) RUN(8, 8, false, 1, 256)
  RUN(16, 2, false, 1, 256) RUN(16, 4, false, 1, 256)
  RUN(8, 4, false, 0, 256) RUN(8, 8, false, 0, 256)
  RUN(4, 4, false, 1, 512) RUN(4, 8, false, 1, 512) RUN(4, 4, false, 1, 1024) RUN(4, 2, false, 1, 2048)
  RUN(4, 8, true, 1, 256) RUN(4, 16, true, 1, 256) RUN(8, 4, true, 1, 256) RUN(8, 8, true, 1, 256) RUN(16, 4, true, 1, 256)
  RUN(4, 4, true, 1, 512) RUN(4, 8, true, 1, 512) RUN(4, 4, true, 1, 1024)
  return 0;
}
