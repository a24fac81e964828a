// Large products of fp32 matrices at fp32 accuracy (fp32 MFMA, or split-bf16 MFMAs): host-side interface of
// gemm_f32_big.hip, shared by brv_gemm_f32 (stft.hip) and the fp32 Conv-TasNet path (ctn_f32.hip).
// Internal to libbrever_hip.so (hidden visibility): the C ABI stays include/brever_hip.h.
#pragma once
#include <hip/hip_runtime.h>

namespace brv {

// y = (prelu(z) - mean_t) rstd_t gain[c] + bias[c] applied to an operand on its way into LDS:
// the normalised tensor of a global / cumulative layer norm is never written out
// (reference: brever/models/convtasnet/convtasnet.py:267-281 norm -> next convolution).
struct NormPro {
  const float* table;      // [frames][2] = (mean_t, rstd_t); null: no transform
  const float* gain; const float* bias;
  const float* slope;      // null: no PReLU
};

struct BigGemm {
  // D[z] (M x N) = sum over (kb, k) op(A)[m][k] op(B)[k][n] + bias + add
  int M, N, K, kbatch, batch;
  const float* A; long long a_bs, a_kbs; int lda; int ta;   // ta: A[m][k] = a[k*lda + m]
  const float* B; long long b_bs, b_kbs; int ldb; int tb;   // tb: B[k][n] = b[n*ldb + k]
  float* D; long long d_bs; int ldd;
  // rows >= m_split of the result go to D2 (row m - m_split), same ldd: two weight gradients
  // from one product ([res | skip])
  float* D2; int m_split;
  const float* add; long long add_bs; int ldadd;            // may alias D (accumulate)
  const float* add2;                                        // for the D2 rows
  const float* bias; int col_bias;                          // per row m, or per column n
  // columns >= n_split of the result come from a second B (column n - n_split of B2) and go to D2 /
  // add2 / bias2 (column n - n_split): two 1x1 convolutions of one input ([res | skip]) in one pass
  // over A. n_split must be a multiple of the tile width; excludes m_split.
  const float* B2; const float* bias2; int n_split;
  NormPro pa;              // on A, !ta: frame = m, channel = k
  NormPro pb;              // on B, !tb: frame = k (kbatch == 1), channel = n
  // reduction split over workgroups (long-K, small M x N): partial tiles in `scratch`
  // ([split][batch][M][N] floats), summed in split order by a second kernel -- no atomics
  float* scratch; long long scratch_floats;
  // take the split-bf16 kernel (three bf16 pieces per operand, six MFMAs per k-step: fp32 accuracy at
  // 2.7x the rate) when the layout allows it (!ta, tb)
  // bit 1 (2): also for the (!ta, tb) form with a long reduction over few tiles, split over workgroups through
  // `scratch`; bit 2 (4): also for (!ta, !tb) and short-K (ta, !tb) products that fill the chip without a split
  int x3;
};

// true when the shape / alignment qualify for the big-tile kernel
__attribute__((visibility("hidden"))) bool gemm_f32_big_ok(const BigGemm& g);
// scratch floats the product needs (0: no reduction split for this shape)
__attribute__((visibility("hidden"))) long long gemm_f32_big_scratch(const BigGemm& g);
// returns a hipError_t value (0 = ok), -1 for a shape the kernel does not take
__attribute__((visibility("hidden"))) int gemm_f32_big(const BigGemm& g, hipStream_t st);

}  // namespace brv
