// fp32 master weights -> padded bf16 GEMM operands (plain and transposed layouts).
#pragma once
#include "common.cuh"

namespace brv {

// ---------------------------------------------------------------------------
// dst (bf16, leading dim dst_ld) region [rows x cols] <- src fp32 [R x C]
// (transposed if tr), zero outside the source extent.
struct PrepJob {
  long long src_off, dst_off;
  int R, C, rows, cols, dst_ld, tr;
  long long scale_off;       // >= 0: source column c is multiplied by params[scale_off + c] first
};
// 256 threads; `part` of `nparts` workgroups share the job. 32-bit index arithmetic without divisions, pairs of
// columns per thread (4-byte stores) for the plain layout, 32 x 32 tiles through LDS for the transposed one (both
// sides coalesced). (One element per thread with a 64-bit division and remainder each: 38 us per step for the
// ~150 jobs of the default network, on an otherwise idle chip -- profiles/r05_step_boundary.txt.)
__device__ __forceinline__ void prep_job_run(const float* params, bf16_t* prepped,
                                             const PrepJob& j, int part, int nparts) {
  const float* src = params + j.src_off;
  const float* scale = j.scale_off >= 0 ? params + j.scale_off : nullptr;
  bf16_t* dst = prepped + j.dst_off;
  const int tid = threadIdx.x;
  if (!j.tr) {
    const int pairs = (j.cols + 1) >> 1;
    int lg = 0;
    while ((1 << lg) < pairs && lg < 8) ++lg;           // threads per row: a power of two <= 256
    const int tpr = 1 << lg, rpi = 256 >> lg;
    const int ty = tid >> lg, tx = tid & (tpr - 1);
    for (int r = part*rpi + ty; r < j.rows; r += nparts*rpi) {
      const bool rin = r < j.R;
      for (int cp = tx; cp < pairs; cp += tpr) {
        const int c = 2*cp;
        float v0 = 0.f, v1 = 0.f;
        if (rin && c < j.C) { v0 = src[r*j.C + c]; if (scale) v0 *= scale[c]; }
        if (rin && c + 1 < j.C) { v1 = src[r*j.C + c + 1]; if (scale) v1 *= scale[c + 1]; }
        bf16_t* d = dst + (long long)r*j.dst_ld + c;
        if (c + 1 < j.cols && !((unsigned long long)d & 3)) *reinterpret_cast<unsigned int*>(d) = pack2(v0, v1);
        else { d[0] = f2bf(v0); if (c + 1 < j.cols) d[1] = f2bf(v1); }
      }
    }
    return;
  }
  // dst[r][c] = src[c][r] * scale[r]
  __shared__ float tile[32][33];
  const int x = tid & 31, y = tid >> 5;
  const int tr_n = (j.rows + 31) >> 5, tc_n = (j.cols + 31) >> 5;
  int ti = part / tc_n, tj = part - ti*tc_n;            // (one division per workgroup)
  for (int t = part; t < tr_n*tc_n; t += nparts) {
    const int r0 = ti*32, c0 = tj*32;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int sr = c0 + y + 8*k, sc = r0 + x;
      float v = 0.f;
      if (sr < j.R && sc < j.C) { v = src[sr*j.C + sc]; if (scale) v *= scale[sc]; }
      tile[y + 8*k][x] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int r = r0 + y + 8*k, c = c0 + x;
      if (r < j.rows && c < j.cols) dst[(long long)r*j.dst_ld + c] = f2bf(tile[x][y + 8*k]);
    }
    __syncthreads();
    tj += nparts;
    while (tj >= tc_n) { tj -= tc_n; ++ti; }
  }
}

}  // namespace brv
