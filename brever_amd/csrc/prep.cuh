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
__device__ __forceinline__ void prep_job_run(const float* params, bf16_t* prepped,
                                             const PrepJob& j, int part, int nparts) {
  const long long total = (long long)j.rows*j.cols;
  for (long long i = (long long)part*256 + threadIdx.x; i < total;
       i += (long long)nparts*256) {
    const int r = (int)(i / j.cols), c = (int)(i % j.cols);
    const int sr = j.tr ? c : r, sc = j.tr ? r : c;
    float v = 0.f;
    if (sr < j.R && sc < j.C) {
      v = params[j.src_off + (long long)sr*j.C + sc];
      if (j.scale_off >= 0) v *= params[j.scale_off + sc];
    }
    prepped[j.dst_off + (long long)r*j.dst_ld + c] = f2bf(v);
  }
}

}  // namespace brv
