// STFT / inverse STFT / mel filterbank as small fp32 GEMMs on the exact-fp32 MFMA
// (v_mfma_f32_32x32x2_f32: bit-for-bit an fp32 fma chain, so the transform keeps fp32
// accuracy; the dense DFT costs 0.5 MFLOP per 512-sample frame, 4 GFLOP for a whole
// 16 x 4 s batch -- far below the HBM time of writing the spectrogram).
//
// Reference: brever/modules/stft.py:59-149 (STFT.forward / backward / pad /
// frame_count around torch.stft / torch.istft, center=True, pad_mode='constant') and
// :152-198 (MelFilterbank). The window-weighted DFT bases are built on the host in
// float64 (brever_amd/modules/stft.py) and passed in as fp32 matrices.
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>

#include "../../include/brever_hip.h"
#include "common.cuh"
#include <type_traits>
#include "gemm_f32_big.h"

using namespace brv;

namespace {

enum AMode { GA_PLAIN = 0, GA_SPEC_T = 1, GA_T = 2 };     // GA_T: A[i][k] = a[k*lda + i]
enum BMode { GB_PLAIN = 0, GB_FRAMES = 1, GB_WT = 2 };
enum SMode { GS_PLAIN = 0, GS_SPEC = 1 };

struct G32 {
  int M, N, K;                 // D[M][N] = A[M][K] * B[K][N], per batch item
  const float* A; long long a_bs; int lda;
  const float* B; long long b_bs; int ldb;
  float* D; long long d_bs; int ldd;
  // GB_FRAMES: B[k][j] = x[j*hop + k - pad_left], zero outside [0, len)
  int hop, pad_left, len;
  // GA_SPEC_T: A[i][c] = spectrum bin c/2, part c%2, frame i, of X[bins][frames] complex
  //            (frames = M), after undoing scale / compression
  int frames; float inv_scale, inv_comp;
  // GS_SPEC: D rows are (bin, part) pairs; store complex [bins][N] after compression
  float comp, scale; int bins;
  // general GEMM extras (brv_gemm_f32): the reduction also runs over kbatch operand pairs
  // (a + kb*a_kbs, b + kb*b_kbs); D = acc + row_bias[m] (+ D if accumulate)
  int kbatch; long long a_kbs, b_kbs;
  const float* row_bias; int accumulate;
  int col_bias;            // the bias is per output COLUMN (accumulate == 2 at the C ABI)
  int b_bf16, d_bf16, a_bf16;   // gemm_bf16 only: operands / result are bf16 in memory (strides in elements)
  // gemm64_kernel only: the DFT basis in double precision (replaces A for GA_PLAIN / B for
  // GB_PLAIN and GB_WT); the other operand is fp32 data converted on load
  const double* A64; const double* B64;
  // gemm_bf16 only: B is the COLUMN MATRIX of an image that is never written out (implicit GEMM of
  // the DCCRN convolutions). Row r = (c, i, j) of a kh x kw window, column = pixel (y, x) of a
  // cv_Ho x cv_Wo grid; B points at the image (cv_C, cv_H, cv_W) fp32, b_bs / b_kbs are image strides.
  //   b_conv 1 (im2col):            image[c][y*sh - ph + i][x*sw - pw + j]
  //   b_conv 2 (transposed gather): image[c][(y + ph - i)/sh][(x + pw - j)/sw] where divisible
  // zero outside the image. With TB the roles of B's two axes are swapped as for a real matrix.
  int b_conv, cv_C, cv_H, cv_W, cv_kh, cv_kw, cv_sh, cv_sw, cv_ph, cv_pw, cv_Ho, cv_Wo;
};

constexpr int TM = 64, TN = 64, TK = 32;

template <int AM, int BM, int SM>
__global__ __launch_bounds__(256) void gemm32_kernel(const G32 p) {
  __shared__ float As[TM][TK + 1];
  __shared__ float Bs[TK][TN + 1];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int m0 = blockIdx.y*TM, n0 = blockIdx.x*TN, b = blockIdx.z;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int nkb = p.kbatch > 1 ? p.kbatch : 1;
  for (int kb = 0; kb < nkb; ++kb) {
  const float* A = p.A + (long long)b*p.a_bs + (long long)kb*p.a_kbs;
  const float* B = p.B + (long long)b*p.b_bs + (long long)kb*p.b_kbs;
  for (int k0 = 0; k0 < p.K; k0 += TK) {
    // ---- stage A [TM][TK] ----
#pragma unroll
    for (int e = tid; e < TM*TK; e += 256) {
      const int i = e / TK, k = e % TK;
      const int m = m0 + i, kk = k0 + k;
      float v = 0.f;
      if (m < p.M && kk < p.K) {
        if (AM == GA_PLAIN) {
          v = A[(long long)m*p.lda + kk];
        } else if (AM == GA_T) {
          v = A[(long long)kk*p.lda + m];
        } else {
          // spectrum element (bin, part) of frame m; undo scale and magnitude compression
          const int bin = kk >> 1, part = kk & 1;
          const float2 z = *reinterpret_cast<const float2*>(A + ((long long)bin*p.frames + m)*2);
          float re = z.x*p.inv_scale, im = z.y*p.inv_scale;
          if (p.inv_comp != 1.f) {
            const float mag = sqrtf(re*re + im*im);
            const float f = mag > 0.f ? powf(mag, p.inv_comp - 1.f) : 0.f;
            re *= f; im *= f;
          }
          v = part ? im : re;
        }
      }
      As[i][k] = v;
    }
    // ---- stage B [TK][TN] ----
#pragma unroll
    for (int e = tid; e < TK*TN; e += 256) {
      const int k = e / TN, j = e % TN;
      const int kk = k0 + k, n = n0 + j;
      float v = 0.f;
      if (kk < p.K && n < p.N) {
        if (BM == GB_PLAIN) v = B[(long long)kk*p.ldb + n];
        else if (BM == GB_WT) v = B[(long long)n*p.ldb + kk];
        else {
          const long long idx = (long long)n*p.hop + kk - p.pad_left;
          if (idx >= 0 && idx < p.len) v = B[idx];
        }
      }
      Bs[k][j] = v;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < TK/2; ++s) {
      const float a = As[32*wm + (lane & 31)][2*s + (lane >> 5)];
      const float bv = Bs[2*s + (lane >> 5)][32*wn + (lane & 31)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  }
  // D element (row, col): col = lane & 31, row = (i & 3) + 8*(i >> 2) + 4*(lane >> 5)
  float* D = p.D + (long long)b*p.d_bs;
  const int col = n0 + 32*wn + (lane & 31);
  if (col >= p.N) return;
  if (SM == GS_PLAIN) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = m0 + 32*wm + (i & 3) + 8*(i >> 2) + 4*(lane >> 5);
      if (row < p.M) {
        float v = acc[i];
        if (p.row_bias) v += p.row_bias[p.col_bias ? col : row];
        if (p.accumulate) v += D[(long long)row*p.ldd + col];
        D[(long long)row*p.ldd + col] = v;
      }
    }
  } else {
    // rows come in (re, im) pairs: registers (2q, 2q+1) of one lane
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = 2*q;
      const int row = m0 + 32*wm + (i & 3) + 8*(i >> 2) + 4*(lane >> 5);
      const int bin = row >> 1;
      if (bin >= p.bins) continue;
      float re = acc[i], im = acc[i + 1];
      if (p.comp != 1.f) {
        const float mag = sqrtf(re*re + im*im);
        const float f = mag > 0.f ? powf(mag, p.comp - 1.f) : 0.f;
        re *= f; im *= f;
      }
      *reinterpret_cast<float2*>(D + ((long long)bin*p.N + col)*2) =
          make_float2(re*p.scale, im*p.scale);
    }
  }
}

// ---- general fp32 GEMM (brv_gemm_f32): 128 x 128 tile, 4 waves of 2 x 2 MFMA accumulators ----
// D[z] (M x N) (+)= sum over (kb, k) of op(A)[m][k] * op(B)[k][n] (+ row_bias[m]). The k-tiles of
// all kbatch operand pairs form one reduction stream; with ksplit > 1 it is divided over
// workgroups that add their partial tiles into a zeroed D with fp32 atomics (weight-gradient
// shapes: small M x N, very long reduction). Operands are staged k-major in LDS whatever
// their storage order, through registers so that the next tile's loads fly during the MFMAs.
constexpr int BM2 = 128, BN2 = 128, BK2 = 32, LD2 = BM2 + 4;

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const G32 p, int ksplit) {
  __shared__ float As[BK2][LD2];
  __shared__ float Bs[BK2][LD2];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int m0 = blockIdx.y*BM2, n0 = blockIdx.x*BN2;
  const int b = blockIdx.z / ksplit, split = blockIdx.z % ksplit;
  const int nkb = p.kbatch > 1 ? p.kbatch : 1;
  const int ktiles = (p.K + BK2 - 1)/BK2;
  const long long total = (long long)nkb*ktiles;
  const long long per = (total + ksplit - 1)/ksplit;
  const long long t_lo = split*per, t_hi = t_lo + per < total ? t_lo + per : total;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging maps: 128 x BK2 elements per operand, NR per thread; the fast thread index runs along
  // the operand's contiguous storage direction
  constexpr int NR = BM2*BK2/256;
  float ra[NR], rb[NR];
  auto fetch = [&](long long t) {
    const int kb = (int)(t / ktiles), k0 = (int)(t % ktiles)*BK2;
    const float* A = p.A + (long long)b*p.a_bs + (long long)kb*p.a_kbs;
    const float* B = p.B + (long long)b*p.b_bs + (long long)kb*p.b_kbs;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int e = tid + r*256;
      int i, k;
      if (TA) { i = e % BM2; k = e / BM2; } else { k = e % BK2; i = e / BK2; }
      const int m = m0 + i, kk = k0 + k;
      ra[r] = (m < p.M && kk < p.K) ? (TA ? A[(long long)kk*p.lda + m] : A[(long long)m*p.lda + kk]) : 0.f;
      int j, k2;
      if (TB) { k2 = e % BK2; j = e / BK2; } else { j = e % BN2; k2 = e / BN2; }
      const int n = n0 + j, kk2 = k0 + k2;
      rb[r] = (n < p.N && kk2 < p.K) ? (TB ? B[(long long)n*p.ldb + kk2] : B[(long long)kk2*p.ldb + n]) : 0.f;
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int e = tid + r*256;
      if (TA) As[e / BM2][e % BM2] = ra[r]; else As[e % BK2][e / BK2] = ra[r];
      if (TB) Bs[e % BK2][e / BK2] = rb[r]; else Bs[e / BN2][e % BN2] = rb[r];
    }
  };
  if (t_lo < t_hi) fetch(t_lo);
  for (long long t = t_lo; t < t_hi; ++t) {
    __syncthreads();
    stash();
    __syncthreads();
    if (t + 1 < t_hi) fetch(t + 1);
#pragma unroll
    for (int s = 0; s < BK2/2; ++s) {
      const int kk = 2*s + (lane >> 5), c = lane & 31;
      const float a0 = As[kk][64*wm + c], a1 = As[kk][64*wm + 32 + c];
      const float b0 = Bs[kk][64*wn + c], b1 = Bs[kk][64*wn + 32 + c];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
  }
  float* D = p.D + (long long)b*p.d_bs;
#pragma unroll
  for (int fi = 0; fi < 2; ++fi)
#pragma unroll
    for (int fj = 0; fj < 2; ++fj) {
      const int col = n0 + 64*wn + 32*fj + (lane & 31);
      if (col >= p.N) continue;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = m0 + 64*wm + 32*fi + (i & 3) + 8*(i >> 2) + 4*(lane >> 5);
        if (row >= p.M) continue;
        float v = acc[fi][fj][i];
        float* d = D + (long long)row*p.ldd + col;
        if (ksplit > 1) {
          if (split == 0 && p.row_bias) v += p.row_bias[p.col_bias ? col : row];
          atomicAdd(d, v);
        } else {
          if (p.row_bias) v += p.row_bias[p.col_bias ? col : row];
          if (p.accumulate) v += *d;
          *d = v;
        }
      }
    }
}

// ---- the same product with bf16 operands (brv_gemm_bf16): fp32 matrices in HBM are rounded to
// bf16 on their way into LDS ([row][k] images, 32 k per tile, rows padded to 80 bytes so that
// the 16-byte fragment reads are conflict-free), v_mfma_f32_32x32x16_bf16 accumulates in fp32.
// 16x the matrix rate of the fp32 MFMA: these products then run at the speed of their loads.
constexpr int BKL = 32, LDL = BKL + 8;            // k per tile, bf16 elements per LDS row

// VEC: every operand row / column the loader touches is a whole, 16-byte aligned float4 (checked
// by the host): 4 x 16-byte loads per operand, thread and tile instead of 16 scalar ones -- the
// instruction stream of the scalar loader, not the MFMA, is what bounds this kernel.
// CV: op_b is the column matrix of an image (p.b_conv); its own instantiation so that the plain
// kernel keeps its register budget (160 VGPRs = 3 waves per SIMD; the column loader needs 184)
template <bool TA, bool TB, bool VEC, bool CV = false>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(const G32 p, int ksplit) {
  __shared__ __attribute__((aligned(16))) bf16_t As[BM2][LDL];
  __shared__ __attribute__((aligned(16))) bf16_t Bs[BN2][LDL];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int m0 = blockIdx.y*BM2, n0 = blockIdx.x*BN2;
  const int b = blockIdx.z / ksplit, split = blockIdx.z % ksplit;
  const int nkb = p.kbatch > 1 ? p.kbatch : 1;
  const int ktiles = (p.K + BKL - 1)/BKL;
  const long long total = (long long)nkb*ktiles;
  const long long per = (total + ksplit - 1)/ksplit;
  const long long t_lo = split*per, t_hi = t_lo + per < total ? t_lo + per : total;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- virtual column matrix (p.b_conv): one element / four consecutive pixels of one row
  auto col_elem = [&](const float* img, int row, int pix) -> float {
    const int khw = p.cv_kh*p.cv_kw;
    const int c = row / khw, t = row - c*khw, i = t / p.cv_kw, j = t - i*p.cv_kw;
    const int y = pix / p.cv_Wo, x = pix - y*p.cv_Wo;
    int hi, wi;
    if (p.b_conv == 1) { hi = y*p.cv_sh - p.cv_ph + i; wi = x*p.cv_sw - p.cv_pw + j; }
    else {
      const int hn = y + p.cv_ph - i, wn = x + p.cv_pw - j;
      if (hn < 0 || wn < 0 || hn % p.cv_sh || wn % p.cv_sw) return 0.f;
      hi = hn/p.cv_sh; wi = wn/p.cv_sw;
    }
    if (hi < 0 || hi >= p.cv_H || wi < 0 || wi >= p.cv_W) return 0.f;
    return img[((long long)c*p.cv_H + hi)*p.cv_W + wi];
  };
  // the vector loader's form: row -> (c, i, j) and pixel -> (y, x) are split off so that whichever
  // of the two is fixed for a thread is decomposed ONCE per kernel; the divisions are float
  // multiplications with an exact correction (the loader's instruction stream bounds this kernel)
  const float inv_khw = CV ? 1.f/(float)(p.cv_kh*p.cv_kw) : 1.f, inv_kw = CV ? 1.f/(float)p.cv_kw : 1.f;
  const float inv_wo = CV ? 1.f/(float)p.cv_Wo : 1.f, inv_sh = CV ? 1.f/(float)p.cv_sh : 1.f;
  auto fdiv = [](int n, int d, float inv, int& q, int& r) {       // n >= 0
    q = (int)((float)n*inv);
    r = n - q*d;
    if (r < 0) { --q; r += d; } else if (r >= d) { ++q; r -= d; }
  };
  struct ColRC { int c, i, j; };
  struct ColYX { int y, x; };
  auto dec_row = [&](int row) { ColRC o; int t; fdiv(row, p.cv_kh*p.cv_kw, inv_khw, o.c, t); fdiv(t, p.cv_kw, inv_kw, o.i, o.j); return o; };
  auto dec_pix = [&](int pix) { ColYX o; fdiv(pix, p.cv_Wo, inv_wo, o.y, o.x); return o; };
  // one element from its decomposed coordinates (no integer division)
  auto col_one = [&](const float* img, const ColRC& rc, int y, int x) -> float {
    int hi, wi;
    if (p.b_conv == 1) { hi = y*p.cv_sh - p.cv_ph + rc.i; wi = x*p.cv_sw - p.cv_pw + rc.j; }
    else {
      const int hn = y + p.cv_ph - rc.i, wn = x + p.cv_pw - rc.j;
      if (hn < 0 || wn < 0) return 0.f;
      int r1, r2;
      fdiv(hn, p.cv_sh, inv_sh, hi, r1);
      fdiv(wn, p.cv_sw, 1.f/(float)p.cv_sw, wi, r2);
      if (r1 || r2) return 0.f;
    }
    if (hi < 0 || hi >= p.cv_H || wi < 0 || wi >= p.cv_W) return 0.f;
    return img[((long long)rc.c*p.cv_H + hi)*p.cv_W + wi];
  };
  auto col_vec4 = [&](const float* img, const ColRC& rc, const ColYX& yx, int row, int pix0) -> float4 {
    // unit stride along the image's contiguous axis: 4 pixels of one grid row read 4 consecutive
    // image elements (one 16-byte load, 4-byte aligned: legal on gfx950) or a row of padding
    if (p.cv_sw == 1 && yx.x + 3 < p.cv_Wo) {
      int hi;
      bool live = true;
      if (p.b_conv == 1) hi = yx.y*p.cv_sh - p.cv_ph + rc.i;
      else {
        const int hn = yx.y + p.cv_ph - rc.i;
        int rem = 0;
        hi = 0;
        if (hn >= 0) fdiv(hn, p.cv_sh, inv_sh, hi, rem);
        live = hn >= 0 && rem == 0;
      }
      if (!live || hi < 0 || hi >= p.cv_H) return make_float4(0.f, 0.f, 0.f, 0.f);
      const int wi = p.b_conv == 1 ? yx.x - p.cv_pw + rc.j : yx.x + p.cv_pw - rc.j;
      if (wi >= 0 && wi + 3 < p.cv_W) {
        float4 v;
        __builtin_memcpy(&v, img + ((long long)rc.c*p.cv_H + hi)*p.cv_W + wi, 16);
        return v;
      }
    }
    // image / grid edges: element by element, the pixel coordinates carried (a group may wrap
    // into the next grid row; a thread that owns such a group takes this path on every tile)
    float e[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int y = yx.y, x = yx.x + q;
      if (x >= p.cv_Wo) { x -= p.cv_Wo; ++y; }
      e[q] = y < p.cv_Ho ? col_one(img, rc, y, x) : 0.f;
    }
    return make_float4(e[0], e[1], e[2], e[3]);
  };

  // each thread stages 8 (row, k-pair) items per operand: 128 rows x 16 pairs; the fast thread
  // index follows the operand's contiguous axis (rows when transposed, k pairs otherwise)
  constexpr int NP = BM2*(BKL/2)/256;
  float2 ra[NP], rb[NP];
  auto item = [&](int e, bool trans, int& row, int& kp) {
    if (trans) { row = e % BM2; kp = e / BM2; } else { kp = e % (BKL/2); row = e / (BKL/2); }
  };
  auto fetch = [&](long long t) {
    const int kb = (int)(t / ktiles), k0 = (int)(t % ktiles)*BKL;
    const long long aoff = (long long)b*p.a_bs + (long long)kb*p.a_kbs;
    const float* A = p.A + aoff;
    const bf16_t* Ah = reinterpret_cast<const bf16_t*>(p.A) + aoff;
    auto lda_ = [&](long long i) { return p.a_bf16 ? bf2f(Ah[i]) : A[i]; };
    const long long boff = (long long)b*p.b_bs + (long long)kb*p.b_kbs;
    const float* B = p.B + boff;
    const bf16_t* Bh = reinterpret_cast<const bf16_t*>(p.B) + boff;
    auto ldb = [&](long long i) { return p.b_bf16 ? bf2f(Bh[i]) : B[i]; };
#pragma unroll
    for (int r = 0; r < NP; ++r) {
      int row, kp;
      item(tid + r*256, TA, row, kp);
      const int m = m0 + row, k = k0 + 2*kp;
      float2 v = make_float2(0.f, 0.f);
      if (m < p.M) {
        if (k < p.K) v.x = lda_(TA ? (long long)k*p.lda + m : (long long)m*p.lda + k);
        if (k + 1 < p.K) v.y = lda_(TA ? (long long)(k + 1)*p.lda + m : (long long)m*p.lda + k + 1);
      }
      ra[r] = v;
      item(tid + r*256, !TB, row, kp);
      const int n = n0 + row;
      v = make_float2(0.f, 0.f);
      if (CV && n < p.N) {
        // (row, pixel) of the virtual matrix: (n, k) with TB, (k, n) without
        if (k0 + 2*kp < p.K) v.x = TB ? col_elem(B, n, k0 + 2*kp) : col_elem(B, k0 + 2*kp, n);
        if (k0 + 2*kp + 1 < p.K) v.y = TB ? col_elem(B, n, k0 + 2*kp + 1) : col_elem(B, k0 + 2*kp + 1, n);
      } else if (n < p.N) {
        if (k0 + 2*kp < p.K)
          v.x = ldb(TB ? (long long)n*p.ldb + k0 + 2*kp : (long long)(k0 + 2*kp)*p.ldb + n);
        if (k0 + 2*kp + 1 < p.K)
          v.y = ldb(TB ? (long long)n*p.ldb + k0 + 2*kp + 1 : (long long)(k0 + 2*kp + 1)*p.ldb + n);
      }
      rb[r] = v;
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int r = 0; r < NP; ++r) {
      int row, kp;
      item(tid + r*256, TA, row, kp);
      *reinterpret_cast<unsigned int*>(&As[row][2*kp]) =
          pack2(ra[r].x, ra[r].y);
      item(tid + r*256, !TB, row, kp);
      *reinterpret_cast<unsigned int*>(&Bs[row][2*kp]) =
          pack2(rb[r].x, rb[r].y);
    }
  };
  // vector loader: an operand whose k axis is contiguous is cut into (row, 4 k) items, one whose
  // rows are contiguous into (4 rows, k pair) items -- 4 float4 loads per operand either way
  float4 va[4], vb[4];
  auto ld4 = [&](const float* X, long long i, bool half) {
    if (!half) return *reinterpret_cast<const float4*>(X + i);
    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(X) + i);
    return make_float4(bf2f((bf16_t)(u.x & 0xffffu)), bf2f((bf16_t)(u.x >> 16)),
                       bf2f((bf16_t)(u.y & 0xffffu)), bf2f((bf16_t)(u.y >> 16)));
  };
  auto vfetch1 = [&](const float* X, long long off, long long ld, bool kcontig, int r0, int R, int k0,
                     float4* v, bool half) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int e = tid + (r >> (kcontig ? 0 : 1))*256;
      float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
      if (kcontig) {
        const int kq = e & 7, row = e >> 3;
        if (r0 + row < R && k0 + 4*kq < p.K) q = ld4(X, off + (long long)(r0 + row)*ld + k0 + 4*kq, half);
      } else {
        const int rq = e & 31, kp = e >> 5, k = k0 + 2*kp + (r & 1);
        if (r0 + 4*rq < R && k < p.K) q = ld4(X, off + (long long)k*ld + r0 + 4*rq, half);
      }
      v[r] = q;
    }
  };
  auto vstash1 = [&](bf16_t (*S)[LDL], bool kcontig, const float4* v) {
    if (kcontig) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int e = tid + r*256, kq = e & 7, row = e >> 3;
        *reinterpret_cast<uint2*>(&S[row][4*kq]) =
            make_uint2(pack2(v[r].x, v[r].y),
                       pack2(v[r].z, v[r].w));
      }
    } else {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int e = tid + h*256, rq = e & 31, kp = e >> 5;
        const float4 lo = v[2*h], hi = v[2*h + 1];          // k = 2 kp and 2 kp + 1
        *reinterpret_cast<unsigned int*>(&S[4*rq][2*kp]) = pack2(lo.x, hi.x);
        *reinterpret_cast<unsigned int*>(&S[4*rq + 1][2*kp]) = pack2(lo.y, hi.y);
        *reinterpret_cast<unsigned int*>(&S[4*rq + 2][2*kp]) = pack2(lo.z, hi.z);
        *reinterpret_cast<unsigned int*>(&S[4*rq + 3][2*kp]) = pack2(lo.w, hi.w);
      }
    }
  };
  // per-thread invariants of the column-matrix loader: without TB the thread's 4-pixel group,
  // with TB its four rows
  ColYX my_yx = {0, 0};
  ColRC my_rc[4] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  if (VEC && CV) {
    if (TB) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int row = n0 + ((tid + r*256) >> 3); if (row < p.N) my_rc[r] = dec_row(row); }
    } else if (n0 + 4*(tid & 31) < p.N) my_yx = dec_pix(n0 + 4*(tid & 31));
  }
  // without TB the thread's four rows advance by one tile (BKL rows) per step: (c, i, j) are
  // carried instead of re-divided (the loader's instruction count bounds this kernel)
  ColRC run_rc[4] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  int run_t[4] = {0, 0, 0, 0}, run_k0 = -0x40000000;
  const int khw_ = CV ? p.cv_kh*p.cv_kw : 1;
  const int adv_c = BKL / khw_, adv_t = BKL - adv_c*khw_;
  auto vfetch_col = [&](const float* img, int k0) {
    // the items of vfetch1 with the column matrix behind them: TB -> (row n, 4 pixels along k),
    // else (4 pixels along n, rows k and k + 1)
    ColYX yx = my_yx;
    if (TB && k0 + 4*(tid & 7) < p.K) yx = dec_pix(k0 + 4*(tid & 7));
    const bool carried = !TB && k0 == run_k0 + BKL;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int e = tid + (r >> (TB ? 0 : 1))*256;
      float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
      if (TB) {
        const int kq = e & 7, row = e >> 3;
        if (n0 + row < p.N && k0 + 4*kq < p.K) q = col_vec4(img, my_rc[r], yx, n0 + row, k0 + 4*kq);
      } else {
        const int rq = e & 31, kp = e >> 5, k = k0 + 2*kp + (r & 1);
        if (carried) {
          run_rc[r].c += adv_c; run_t[r] += adv_t;
          if (run_t[r] >= khw_) { run_t[r] -= khw_; ++run_rc[r].c; }
        } else {
          fdiv(k, khw_, inv_khw, run_rc[r].c, run_t[r]);
        }
        fdiv(run_t[r], p.cv_kw, inv_kw, run_rc[r].i, run_rc[r].j);
        if (n0 + 4*rq < p.N && k < p.K) q = col_vec4(img, run_rc[r], yx, k, n0 + 4*rq);
      }
      vb[r] = q;
    }
    run_k0 = k0;
  };
  auto vfetch = [&](long long t) {
    const int kb = (int)(t / ktiles), k0 = (int)(t % ktiles)*BKL;
    vfetch1(p.A, (long long)b*p.a_bs + (long long)kb*p.a_kbs, p.lda, !TA, m0, p.M, k0, va, p.a_bf16 != 0);
    if constexpr (CV) vfetch_col(p.B + (long long)b*p.b_bs + (long long)kb*p.b_kbs, k0);
    else vfetch1(p.B, (long long)b*p.b_bs + (long long)kb*p.b_kbs, p.ldb, TB, n0, p.N, k0, vb, p.b_bf16 != 0);
  };
  auto vstash = [&]() { vstash1(As, !TA, va); vstash1(Bs, TB, vb); };
  if (t_lo < t_hi) { if (VEC) vfetch(t_lo); else fetch(t_lo); }
  for (long long t = t_lo; t < t_hi; ++t) {
    __syncthreads();
    if (VEC) vstash(); else stash();
    __syncthreads();
    if (t + 1 < t_hi) { if (VEC) vfetch(t + 1); else fetch(t + 1); }
    const int c = lane & 31, kh = (lane >> 5)*8;
#pragma unroll
    for (int s = 0; s < BKL/16; ++s) {
      const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(&As[64*wm + c][16*s + kh]);
      const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(&As[64*wm + 32 + c][16*s + kh]);
      const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(&Bs[64*wn + c][16*s + kh]);
      const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(&Bs[64*wn + 32 + c][16*s + kh]);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
    }
  }
  float* D = p.D + (long long)b*p.d_bs;
#pragma unroll
  for (int fi = 0; fi < 2; ++fi)
#pragma unroll
    for (int fj = 0; fj < 2; ++fj) {
      const int col = n0 + 64*wn + 32*fj + (lane & 31);
      if (col >= p.N) continue;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = m0 + 64*wm + 32*fi + (i & 3) + 8*(i >> 2) + 4*(lane >> 5);
        if (row >= p.M) continue;
        float v = acc[fi][fj][i];
        if (p.d_bf16) {                      // bf16 result (host guarantees ksplit == 1, no accumulate)
          if (p.row_bias) v += p.row_bias[p.col_bias ? col : row];
          (reinterpret_cast<bf16_t*>(p.D) + (long long)b*p.d_bs)[(long long)row*p.ldd + col] = f2bf(v);
          continue;
        }
        float* d = D + (long long)row*p.ldd + col;
        if (ksplit > 1) {
          if (split == 0 && p.row_bias) v += p.row_bias[p.col_bias ? col : row];
          atomicAdd(d, v);
        } else {
          if (p.row_bias) v += p.row_bias[p.col_bias ? col : row];
          if (p.accumulate) v += *d;
          *d = v;
        }
      }
    }
}

// Windowed overlap-add with window-envelope normalisation (torch.istft, center=True):
//   y[q] = sum_t frames[t][q + n/2 - t*hop] / sum_t w^2[q + n/2 - t*hop],  q < hop*(F-1)
struct OlaParams {
  const float* frames; float* y; const float* win; int F, n, hop, out_len;
  long long f_bs, y_bs;
  int normalize;             // 1: divide by the window-square envelope (torch.istft)
  int pad_left;              // frame t starts at sample t*hop - pad_left of the output
};
__global__ __launch_bounds__(256) void istft_ola_kernel(const OlaParams p) {
  const int b = blockIdx.y;
  const float* fr = p.frames + (long long)b*p.f_bs;
  float* y = p.y + (long long)b*p.y_bs;
  for (int q = blockIdx.x*256 + threadIdx.x; q < p.out_len; q += gridDim.x*256) {
    const int pos = q + p.pad_left;
    int t_hi = pos/p.hop; if (t_hi > p.F - 1) t_hi = p.F - 1;
    int t_lo = (pos - p.n + p.hop)/p.hop; if (pos - p.n + 1 <= 0) t_lo = 0;
    if (t_lo < 0) t_lo = 0;
    float s = 0.f, env = 0.f;
    for (int t = t_lo; t <= t_hi; ++t) {          // fixed order: deterministic
      const int m = pos - t*p.hop;
      if (m < 0 || m >= p.n) continue;
      s += fr[(long long)t*p.n + m];
      if (p.normalize) { const float w = p.win[m]; env += w*w; }
    }
    y[q] = p.normalize ? s/env : s;
  }
}

template <int AM, int BM, int SM>
int launch_g32(const G32& p, int batch, hipStream_t st) {
  if (p.M <= 0 || p.N <= 0 || batch <= 0) return 0;
  dim3 grid((p.N + TN - 1)/TN, (p.M + TM - 1)/TM, batch);
  hipLaunchKernelGGL((gemm32_kernel<AM, BM, SM>), grid, dim3(256), 0, st, p);
  return (int)hipGetLastError();
}

// ---- the same framed products with fp64 accumulation (STFT.forward / STFT.backward) ------------
// The reference's transforms are FFTs (torch.stft / torch.istft): log2(n) rounding stages, round
// trip within 1e-6 (tests/test_modules.py:319-326). A DFT as a plain fp32 product accumulates n
// terms in sequence and lands at ~2e-5. The DFT products of the STFT module therefore run on the
// double-precision matrix pipe: v_mfma_f64_16x16x4_f64, basis in double, fp32 data converted on
// load, result rounded to fp32 once. Lane maps (MI355X guide): A[row = lane & 15][k = lane >> 4],
// B[k = lane >> 4][col = lane & 15], D register r: row = (lane >> 4) + 4 r, col = lane & 15.
typedef __attribute__((ext_vector_type(4))) double f64x4;

template <int AM, int BM, int SM>
__global__ __launch_bounds__(256) void gemm64_kernel(const G32 p) {
  __shared__ double As[TM][TK + 1];
  __shared__ double Bs[TK][TN + 1];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int m0 = blockIdx.y*TM, n0 = blockIdx.x*TN, b = blockIdx.z;
  f64x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;
  const float* A = p.A + (long long)b*p.a_bs;
  const float* B = p.B + (long long)b*p.b_bs;
  // The operands of k tile k0 + TK are requested into registers before the MFMAs of k tile k0 and written to LDS
  // behind them. Every load is UNCONDITIONAL (indices clamped into the operand, the value replaced by zero
  // afterwards) and the fp64 / fp32 form of each operand is chosen once, outside the loop: with a load under a
  // lane condition, or two alternative loads into one register, hipcc waits for every single load before it
  // issues the next one (s_waitcnt vmcnt(0) x 24 per k tile: 18 TFLOP/s).
  constexpr int NE = TM*TK/256;
  static_assert(TM*TK == TK*TN, "one item count for both operands");
  auto run = [&](auto a64_, auto b64_) {
    constexpr bool A64 = decltype(a64_)::value, B64 = decltype(b64_)::value;
    double ra[NE], rb[NE];
    auto fetch = [&](int k0) {
#pragma unroll
      for (int r = 0; r < NE; ++r) {
        const int e = tid + 256*r;
        {
          const int i = e / TK, k = e % TK;
          const int m = m0 + i, kk = k0 + k;
          const bool ok = m < p.M && kk < p.K;
          const int mc = m < p.M ? m : p.M - 1, kc = kk < p.K ? kk : p.K - 1;
          double v;
          if (AM == GA_PLAIN) {
            if (A64) v = p.A64[(long long)mc*p.lda + kc]; else v = (double)A[(long long)mc*p.lda + kc];
          } else {                                   // GA_SPEC_T
            const int bin = kc >> 1, part = kc & 1;
            const float2 z = *reinterpret_cast<const float2*>(A + ((long long)bin*p.frames + mc)*2);
            double re = (double)z.x*p.inv_scale, im = (double)z.y*p.inv_scale;
            if (p.inv_comp != 1.f) {
              const double mag = sqrt(re*re + im*im);
              const double f = mag > 0.0 ? pow(mag, (double)p.inv_comp - 1.0) : 0.0;
              re *= f; im *= f;
            }
            v = part ? im : re;
          }
          ra[r] = ok ? v : 0.0;
        }
        {
          // lanes follow the operand's contiguous axis: columns for GB_PLAIN, k for the framed signal / W^T
          const int k = BM == GB_PLAIN ? e / TN : e % TK, j = BM == GB_PLAIN ? e % TN : e / TK;
          const int kk = k0 + k, n = n0 + j;
          bool ok = kk < p.K && n < p.N;
          const int kc = kk < p.K ? kk : p.K - 1, nc = n < p.N ? n : p.N - 1;
          double v;
          if (BM == GB_PLAIN) {
            if (B64) v = p.B64[(long long)kc*p.ldb + nc]; else v = (double)B[(long long)kc*p.ldb + nc];
          } else if (BM == GB_WT) {
            if (B64) v = p.B64[(long long)nc*p.ldb + kc]; else v = (double)B[(long long)nc*p.ldb + kc];
          } else {
            long long idx = (long long)nc*p.hop + kc - p.pad_left;
            ok = ok && idx >= 0 && idx < p.len;
            idx = idx < 0 ? 0 : (idx >= p.len ? p.len - 1 : idx);
            v = (double)B[idx];
          }
          rb[r] = ok ? v : 0.0;
        }
      }
    };
    auto stash = [&]() {
#pragma unroll
      for (int r = 0; r < NE; ++r) {
        const int e = tid + 256*r;
        As[e / TK][e % TK] = ra[r];
        if (BM == GB_PLAIN) Bs[e / TN][e % TN] = rb[r]; else Bs[e % TK][e / TK] = rb[r];
      }
    };
    fetch(0);
    for (int k0 = 0; k0 < p.K; k0 += TK) {
      stash();
      __syncthreads();
      fetch(k0 + TK < p.K ? k0 + TK : k0);          // (the last tile is requested once more: no branch around loads)
#pragma unroll
      for (int s = 0; s < TK/4; ++s) {
        const int kk = 4*s + (lane >> 4), c = lane & 15;
        const double a0 = As[32*wm + c][kk], a1 = As[32*wm + 16 + c][kk];
        const double b0 = Bs[kk][32*wn + c], b1 = Bs[kk][32*wn + 16 + c];
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
      }
      __syncthreads();
    }
  };
  if (p.A64) run(std::true_type{}, std::false_type{});
  else if (p.B64) run(std::false_type{}, std::true_type{});
  else run(std::false_type{}, std::false_type{});
  // accumulators -> LDS tile (rows of one (re, im) pair sit in different lanes), then row-wise out
  double (*Cs)[TK + 1] = As;                        // 64 x 33 doubles: half a tile at a time
  float* D = p.D + (long long)b*p.d_bs;
  for (int half = 0; half < 2; ++half) {            // columns [32 half, 32 half + 32) of the tile
    __syncthreads();
    if (wn == half) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            Cs[32*wm + 16*i + (lane >> 4) + 4*r][16*j + (lane & 15)] = acc[i][j][r];
    }
    __syncthreads();
    if (SM == GS_PLAIN) {
      for (int e = tid; e < TM*32; e += 256) {
        const int i = e / 32, j = e % 32;
        const int row = m0 + i, col = n0 + 32*half + j;
        if (row < p.M && col < p.N) D[(long long)row*p.ldd + col] = (float)Cs[i][j];
      }
    } else {
      for (int e = tid; e < (TM/2)*32; e += 256) {
        const int q = e / 32, j = e % 32;
        const int bin = (m0 >> 1) + q, col = n0 + 32*half + j;
        if (bin >= p.bins || col >= p.N) continue;
        double re = Cs[2*q][j], im = Cs[2*q + 1][j];
        if (p.comp != 1.f) {
          const double mag = sqrt(re*re + im*im);
          const double f = mag > 0.0 ? pow(mag, (double)p.comp - 1.0) : 0.0;
          re *= f; im *= f;
        }
        *reinterpret_cast<float2*>(D + ((long long)bin*p.N + col)*2) =
            make_float2((float)(re*p.scale), (float)(im*p.scale));
      }
    }
  }
}

// y[r][i] = x[r][reflect(i - left)], i < out_len: F.pad(mode='reflect' | 'replicate' | 'circular')
// of the STFT module's pad_mode (stft.py:140-149 and torch.stft's centre padding)
__global__ void pad_signal_kernel(const float* x, float* y, long long rows, long long L, long long left,
                                  long long out_len, int mode) {
  const long long n = rows*out_len;
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < n; i += (long long)gridDim.x*256) {
    const long long r = i / out_len;
    long long j = i % out_len - left;
    if (mode == 1) {                      // reflect (no edge repeat); valid for pads < L
      if (j < 0) j = -j;
      if (j >= L) j = 2*(L - 1) - j;
    } else if (mode == 2) {               // replicate
      j = j < 0 ? 0 : (j >= L ? L - 1 : j);
    } else if (mode == 3) {               // circular
      j %= L; if (j < 0) j += L;
    }
    y[i] = (j >= 0 && j < L) ? x[r*L + j] : 0.f;
  }
}

// complex <-> (magnitude, phase): the 'mag_phase' return / input types of the STFT module
__global__ void polar_kernel(const float* mag, const float* phase, float2* out, long long n) {
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < n; i += (long long)gridDim.x*256) {
    float sn, cs;
    sincosf(phase[i], &sn, &cs);
    out[i] = make_float2(mag[i]*cs, mag[i]*sn);
  }
}
__global__ void mag_phase_kernel(const float2* x, float* mag, float* phase, long long n) {
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < n; i += (long long)gridDim.x*256) {
    const float2 z = x[i];
    mag[i] = sqrtf(z.x*z.x + z.y*z.y);
    phase[i] = atan2f(z.y, z.x);
  }
}

template <int AM, int BM, int SM>
int launch_g64(const G32& p, int batch, hipStream_t st) {
  if (p.M <= 0 || p.N <= 0 || batch <= 0) return 0;
  dim3 grid((p.N + TN - 1)/TN, (p.M + TM - 1)/TM, batch);
  hipLaunchKernelGGL((gemm64_kernel<AM, BM, SM>), grid, dim3(256), 0, st, p);
  return (int)hipGetLastError();
}

// Y = scale |X|^(c-1) X (magnitude compression + scaling, stft.py:85-87) and its gradient:
// gX = scale r^(c-1) u (c Re(gY conj u) + j Im(gY conj u)), u = X/|X|, r = |X| (0 at r = 0)
__global__ void spec_compress_kernel(const float2* x, float2* y, long long n, float c, float scale) {
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < n; i += (long long)gridDim.x*256) {
    const float2 z = x[i];
    const float r = sqrtf(z.x*z.x + z.y*z.y);
    const float f = (c == 1.f) ? scale : (r > 0.f ? scale*powf(r, c - 1.f) : 0.f);
    y[i] = make_float2(z.x*f, z.y*f);
  }
}
__global__ void spec_compress_bwd_kernel(const float2* x, const float2* gy, float2* gx, long long n,
                                         float c, float scale) {
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < n; i += (long long)gridDim.x*256) {
    const float2 z = x[i], g = gy[i];
    const float r = sqrtf(z.x*z.x + z.y*z.y);
    float2 o = make_float2(0.f, 0.f);
    if (c == 1.f) {
      o = make_float2(scale*g.x, scale*g.y);
    } else if (r > 0.f) {
      const float ux = z.x/r, uy = z.y/r;
      const float a = g.x*ux + g.y*uy;            // Re(g conj u)
      const float b = g.y*ux - g.x*uy;            // Im(g conj u)
      const float f = scale*powf(r, c - 1.f);
      o = make_float2(f*(c*a*ux - b*uy), f*(c*a*uy + b*ux));
    }
    gx[i] = o;
  }
}

}  // namespace

extern "C" {

int64_t brv_stft_frames(int64_t length, int64_t frame_length, int64_t hop_length) {
  // STFT.frame_count + the n/2 centre padding on both sides (stft.py:140-149, :72)
  if (length < 0 || frame_length < 2 || hop_length < 1) return -1;
  const int64_t d = length > frame_length ? length - frame_length : 0;
  const int64_t nc = (d + hop_length - 1)/hop_length + 1;
  const int64_t padded = (nc - 1)*hop_length + frame_length;     // after STFT.pad
  return 1 + padded/hop_length;        // 1 + (padded + n - n)/hop with centre padding
}

int brv_stft_forward(const float* x, const float* basis, float* spec, int64_t rows,
                     int64_t length, int64_t frame_length, int64_t hop_length,
                     float compression, float scale, brv_stream_t stream) {
  const int64_t F = brv_stft_frames(length, frame_length, hop_length);
  if (rows < 1 || F < 1) return -1;
  const int bins = (int)(frame_length/2 + 1);
  G32 p; memset(&p, 0, sizeof(p));
  p.M = 2*bins; p.N = (int)F; p.K = (int)frame_length;
  p.A = basis; p.a_bs = 0; p.lda = (int)frame_length;
  p.B = x; p.b_bs = length; p.hop = (int)hop_length; p.pad_left = (int)(frame_length/2);
  p.len = (int)length;
  p.D = spec; p.d_bs = (long long)bins*F*2; p.bins = bins;
  p.comp = compression; p.scale = scale;
  return launch_g32<GA_PLAIN, GB_FRAMES, GS_SPEC>(p, (int)rows, (hipStream_t)stream);
}

int brv_istft_backward(const float* spec, const float* inv_basis, const float* window,
                       float* frames_scratch, float* y, int64_t rows, int64_t frames,
                       int64_t frame_length, int64_t hop_length, float compression,
                       float scale, brv_stream_t stream) {
  if (rows < 1 || frames < 1 || frame_length < 2 || hop_length < 1) return -1;
  hipStream_t st = (hipStream_t)stream;
  const int bins = (int)(frame_length/2 + 1);
  G32 p; memset(&p, 0, sizeof(p));
  p.M = (int)frames; p.N = (int)frame_length; p.K = 2*bins;
  p.A = spec; p.a_bs = (long long)bins*frames*2; p.frames = (int)frames;
  p.inv_scale = 1.f/scale; p.inv_comp = 1.f/compression;
  p.B = inv_basis; p.b_bs = 0; p.ldb = 2*bins;           // inv_basis[m][c] read transposed
  p.D = frames_scratch; p.d_bs = (long long)frames*frame_length; p.ldd = (int)frame_length;
  if (int r = launch_g32<GA_SPEC_T, GB_WT, GS_PLAIN>(p, (int)rows, st)) return r;
  OlaParams o;
  o.frames = frames_scratch; o.y = y; o.win = window; o.F = (int)frames;
  o.n = (int)frame_length; o.hop = (int)hop_length;
  o.out_len = (int)(hop_length*(frames - 1));
  o.f_bs = (long long)frames*frame_length; o.y_bs = o.out_len; o.normalize = 1;
  o.pad_left = (int)(frame_length/2);
  int gx = (o.out_len + 255)/256; if (gx > 1024) gx = 1024; if (gx < 1) gx = 1;
  hipLaunchKernelGGL(istft_ola_kernel, dim3(gx, (unsigned)rows), dim3(256), 0, st, o);
  return (int)hipGetLastError();
}

// Adjoint of brv_stft_forward with respect to x (compression 1): the gradient of a loss
// through the framed, zero-padded, windowed DFT,
//   dx[r][q] = scale * sum_t sum_c dspec[r][c][t] * basis[c][q + n/2 - t*hop],  q < length,
// as a GEMM (dspec^T x basis -> per-frame gradients) + plain overlap-add.
int brv_stft_adjoint(const float* dspec, const float* basis, float* frames_scratch, float* dx,
                     int64_t rows, int64_t length, int64_t frame_length, int64_t hop_length,
                     float scale, brv_stream_t stream) {
  const int64_t F = brv_stft_frames(length, frame_length, hop_length);
  if (rows < 1 || F < 1) return -1;
  hipStream_t st = (hipStream_t)stream;
  const int bins = (int)(frame_length/2 + 1);
  G32 p; memset(&p, 0, sizeof(p));
  p.M = (int)F; p.N = (int)frame_length; p.K = 2*bins;
  p.A = dspec; p.a_bs = (long long)bins*F*2; p.frames = (int)F;
  p.inv_scale = scale; p.inv_comp = 1.f;                  // GA_SPEC_T multiplies by inv_scale
  p.B = basis; p.b_bs = 0; p.ldb = (int)frame_length;     // basis[c][m], plain
  p.D = frames_scratch; p.d_bs = (long long)F*frame_length; p.ldd = (int)frame_length;
  if (int r = launch_g32<GA_SPEC_T, GB_PLAIN, GS_PLAIN>(p, (int)rows, st)) return r;
  OlaParams o;
  o.frames = frames_scratch; o.y = dx; o.win = nullptr; o.F = (int)F;
  o.n = (int)frame_length; o.hop = (int)hop_length;
  o.out_len = (int)length; o.normalize = 0; o.pad_left = (int)(frame_length/2);
  o.f_bs = (long long)F*frame_length; o.y_bs = length;
  int gx = (o.out_len + 255)/256; if (gx > 1024) gx = 1024; if (gx < 1) gx = 1;
  hipLaunchKernelGGL(istft_ola_kernel, dim3(gx, (unsigned)rows), dim3(256), 0, st, o);
  return (int)hipGetLastError();
}

// Framed DFT with explicit geometry (ConvSTFT, brever/modules/stft.py:201-319: the STFT as a
// strided convolution with sqrt-window DFT rows): frame t covers samples
// [t*hop - pad_left, t*hop - pad_left + n) of x (zeros outside [0, length)), `frames` frames.
int brv_framed_dft_forward(const float* x, const float* basis, float* spec, int64_t rows,
                           int64_t length, int64_t frame_length, int64_t hop_length,
                           int64_t pad_left, int64_t frames, float compression, float scale,
                           brv_stream_t stream) {
  if (rows < 1 || frames < 1 || frame_length < 2 || hop_length < 1) return -1;
  const int bins = (int)(frame_length/2 + 1);
  G32 p; memset(&p, 0, sizeof(p));
  p.M = 2*bins; p.N = (int)frames; p.K = (int)frame_length;
  p.A = basis; p.a_bs = 0; p.lda = (int)frame_length;
  p.B = x; p.b_bs = length; p.hop = (int)hop_length; p.pad_left = (int)pad_left;
  p.len = (int)length;
  p.D = spec; p.d_bs = (long long)bins*frames*2; p.bins = bins;
  p.comp = compression; p.scale = scale;
  return launch_g32<GA_PLAIN, GB_FRAMES, GS_SPEC>(p, (int)rows, (hipStream_t)stream);
}

// Transposed framed DFT (ConvSTFT.backward = conv_transpose1d with the same filters, and the
// adjoint of brv_framed_dft_forward): y[r][q] = post * sum_t sum_c (spec/scale)[c][t] *
// basis[c][q + pad_left - t*hop], q < out_len, after undoing the magnitude compression.
int brv_framed_dft_transpose(const float* spec, const float* basis, float* frames_scratch,
                             float* y, int64_t rows, int64_t frames, int64_t frame_length,
                             int64_t hop_length, int64_t pad_left, int64_t out_len,
                             float compression, float scale, brv_stream_t stream) {
  if (rows < 1 || frames < 1 || out_len < 1) return -1;
  hipStream_t st = (hipStream_t)stream;
  const int bins = (int)(frame_length/2 + 1);
  G32 p; memset(&p, 0, sizeof(p));
  p.M = (int)frames; p.N = (int)frame_length; p.K = 2*bins;
  p.A = spec; p.a_bs = (long long)bins*frames*2; p.frames = (int)frames;
  p.inv_scale = 1.f/scale; p.inv_comp = 1.f/compression;
  p.B = basis; p.b_bs = 0; p.ldb = (int)frame_length;
  p.D = frames_scratch; p.d_bs = (long long)frames*frame_length; p.ldd = (int)frame_length;
  if (int r = launch_g32<GA_SPEC_T, GB_PLAIN, GS_PLAIN>(p, (int)rows, st)) return r;
  OlaParams o;
  o.frames = frames_scratch; o.y = y; o.win = nullptr; o.F = (int)frames;
  o.n = (int)frame_length; o.hop = (int)hop_length;
  o.out_len = (int)out_len; o.normalize = 0; o.pad_left = (int)pad_left;
  o.f_bs = (long long)frames*frame_length; o.y_bs = out_len;
  int gx = (o.out_len + 255)/256; if (gx > 1024) gx = 1024; if (gx < 1) gx = 1;
  hipLaunchKernelGGL(istft_ola_kernel, dim3(gx, (unsigned)rows), dim3(256), 0, st, o);
  return (int)hipGetLastError();
}

// General fp32 GEMM on the exact-fp32 MFMA (the Linear layers of the FFNN model and their
// gradients, brever/models/ffnn/ffnn.py:151-171):
//   d[z] (M x N) (+)= sum_kb op_a(a[z, kb]) (M x K) @ op_b(b[z, kb]) (K x N) + row_bias[m]
// op_a = transpose iff trans_a (a stored K x M), op_b likewise (b stored N x K).
static int gemm_any(int lowp, const float* a, const float* b, float* d, int64_t batch, int64_t M,
                    int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldd,
                    int64_t a_batch_stride, int64_t b_batch_stride, int64_t d_batch_stride,
                    int trans_a, int trans_b, int64_t kbatch, int64_t a_kbatch_stride,
                    int64_t b_kbatch_stride, const float* row_bias, int accumulate,
                    brv_stream_t stream, int flags = 0, const int* conv = nullptr,
                    float* ws = nullptr, long long ws_floats = 0) {
  if (batch < 1 || M < 1 || N < 1 || K < 1) return -1;
  if (conv && !lowp) return -1;
  if (flags && (!lowp || ((flags & 2) && accumulate == 1))) return -1;
  G32 p; memset(&p, 0, sizeof(p));
  p.M = (int)M; p.N = (int)N; p.K = (int)K;
  p.A = a; p.a_bs = a_batch_stride; p.lda = (int)lda;
  p.B = b; p.b_bs = b_batch_stride; p.ldb = (int)ldb;
  p.D = d; p.d_bs = d_batch_stride; p.ldd = (int)ldd;
  p.kbatch = (int)kbatch; p.a_kbs = a_kbatch_stride; p.b_kbs = b_kbatch_stride;
  p.col_bias = accumulate == 2;
  p.b_bf16 = flags & 1; p.d_bf16 = (flags >> 1) & 1; p.a_bf16 = (flags >> 2) & 1;
  if (conv) {
    p.b_conv = conv[0]; p.cv_C = conv[1]; p.cv_H = conv[2]; p.cv_W = conv[3]; p.cv_kh = conv[4];
    p.cv_kw = conv[5]; p.cv_sh = conv[6]; p.cv_sw = conv[7]; p.cv_ph = conv[8]; p.cv_pw = conv[9];
    p.cv_Ho = conv[10]; p.cv_Wo = conv[11];
  }
  if (accumulate == 2) accumulate = 0;
  p.row_bias = row_bias; p.accumulate = accumulate;
  hipStream_t st = (hipStream_t)stream;
#ifndef BRV_GEMM_F32_SMALL       // diagnostic builds: every fp32 product on the 128 x 128 kernel below
  if (!lowp && M >= 32 && N >= 32 &&
      (double)M*(double)N*(double)K*(double)batch*(kbatch > 1 ? kbatch : 1) >= 3.0e7) {
    // 16-byte aligned operands of a product worth 256 x 128 tiles: gemm_f32_big.hip -- when it fills the
    // chip WITHOUT splitting the reduction. A split there needs scratch for the partial tiles; this entry
    // point has none (a stream-ordered allocation per call cost the SGMSE+ fp32 training step 8 %), so
    // long reductions over few tiles stay on the 128 x 128 kernel below, whose split adds with atomics.
    // (The fp32 Conv-TasNet path calls gemm_f32_big directly with scratch from its workspace: its
    // weight gradients are split in a fixed order.)
    brv::BigGemm g; memset(&g, 0, sizeof(g));
    g.M = (int)M; g.N = (int)N; g.K = (int)K; g.kbatch = kbatch > 1 ? (int)kbatch : 1; g.batch = (int)batch;
    g.A = a; g.a_bs = a_batch_stride; g.a_kbs = a_kbatch_stride; g.lda = (int)lda; g.ta = trans_a != 0;
    g.B = b; g.b_bs = b_batch_stride; g.b_kbs = b_kbatch_stride; g.ldb = (int)ldb; g.tb = trans_b != 0;
    g.D = d; g.d_bs = d_batch_stride; g.ldd = (int)ldd;
    g.bias = row_bias; g.col_bias = p.col_bias;
    if (accumulate) { g.add = d; g.add_bs = d_batch_stride; g.ldadd = (int)ldd; }
    g.x3 = 1 | 4;      // split-bf16 kernel where the layout allows it (every pair but a stored K x M with b stored N x K)
    if (ws) {
      // brv_gemm_f32_ws: the caller's scratch takes the partial tiles of a reduction split (summed in split
      // order by a second kernel), which also opens the split-bf16 form to long reductions over few tiles
      g.x3 = 1 | 2 | 4 | 8;
      if (brv::gemm_f32_big_ok(g)) {
        const long long need = brv::gemm_f32_big_scratch(g);
        if (need <= ws_floats) {
          if (need > 0) { g.scratch = ws; g.scratch_floats = ws_floats; }
          return brv::gemm_f32_big(g, st);
        }
      }
      g.x3 = 1 | 4;
    }
    if (brv::gemm_f32_big_ok(g) && brv::gemm_f32_big_scratch(g) == 0) return brv::gemm_f32_big(g, st);
  }
#endif
  // reduction split: fill the chip when the output has few tiles and the reduction is long
  const long long tiles = ((M + BM2 - 1)/BM2)*((N + BN2 - 1)/BN2)*batch;
  const long long red = (kbatch > 1 ? kbatch : 1)*((K + BK2 - 1)/BK2);
  long long ksplit = 1;
#ifndef BRV_GEMM_SPLIT_TARGET
#define BRV_GEMM_SPLIT_TARGET 512    // workgroups a split reduction aims for (diagnostic builds: 1024, 2048)
#endif
  if (tiles < 128 && red >= 16 && !(flags & 2)) {
    ksplit = BRV_GEMM_SPLIT_TARGET/tiles;
    if (ksplit > red/4) ksplit = red/4;
    if (ksplit < 1) ksplit = 1;
  }
  if (ksplit > 1 && !accumulate) {
    if (ldd == N && (batch == 1 || d_batch_stride == M*N)) {
      // (one fill for a contiguous batch: sixteen 5-us fills in a row stood in front of the LSTM weight-gradient
      // products of the DCCRN step -- profiles/r05_dccrn_trace.txt)
      if (hipMemsetAsync(d, 0, (size_t)batch*M*N*4, st) != hipSuccess) return -2;
    } else if (ldd == N) {
      for (int64_t z = 0; z < batch; ++z)
        if (hipMemsetAsync(d + z*d_batch_stride, 0, (size_t)M*N*4, st) != hipSuccess) return -2;
    } else {
      for (int64_t z = 0; z < batch; ++z)
        if (hipMemset2DAsync(d + z*d_batch_stride, (size_t)ldd*4, 0, (size_t)N*4, (size_t)M, st)
            != hipSuccess) return -2;
    }
  }
  const dim3 grid((unsigned)((N + BN2 - 1)/BN2), (unsigned)((M + BM2 - 1)/BM2),
                  (unsigned)(batch*ksplit));
  if (lowp) {
#ifndef BRV_GEMM_SCALAR
#define BRV_GEMM_SCALAR 0      // diagnostic builds: general bf16 GEMM without the vector loaders
#endif
    // vector loader: strides, bases and the extent along each operand's contiguous axis are
    // multiples of 4 floats
    auto q4 = [](long long v) { return (v & 3) == 0; };
    // (a column-matrix B: only its pixel count matters -- its loads are 4-byte aligned by design)
    const bool vec = q4(lda) && (conv || q4(ldb)) && q4(a_batch_stride) && (conv || q4(b_batch_stride)) &&
                     q4(a_kbatch_stride) && (conv || q4(b_kbatch_stride)) &&
                     ((uintptr_t)a & ((flags & 4) ? 7 : 15)) == 0 &&
                     (conv || ((uintptr_t)b & ((flags & 1) ? 7 : 15)) == 0) &&
                     q4(trans_a ? M : K) && q4(trans_b ? K : N) && !BRV_GEMM_SCALAR;
#define BRV_BF16_LAUNCH(TA_, TB_) \
    do { if (conv && vec) hipLaunchKernelGGL((gemm_bf16_kernel<TA_, TB_, true, true>), grid, dim3(256), 0, st, p, (int)ksplit); \
         else if (conv) hipLaunchKernelGGL((gemm_bf16_kernel<TA_, TB_, false, true>), grid, dim3(256), 0, st, p, (int)ksplit); \
         else if (vec) hipLaunchKernelGGL((gemm_bf16_kernel<TA_, TB_, true>), grid, dim3(256), 0, st, p, (int)ksplit); \
         else hipLaunchKernelGGL((gemm_bf16_kernel<TA_, TB_, false>), grid, dim3(256), 0, st, p, (int)ksplit); } while (0)
    if (trans_a && trans_b) BRV_BF16_LAUNCH(true, true);
    else if (trans_a) BRV_BF16_LAUNCH(true, false);
    else if (trans_b) BRV_BF16_LAUNCH(false, true);
    else BRV_BF16_LAUNCH(false, false);
#undef BRV_BF16_LAUNCH
    return (int)hipGetLastError();
  }
  if (trans_a && trans_b)
    hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, dim3(256), 0, st, p, (int)ksplit);
  else if (trans_a)
    hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, dim3(256), 0, st, p, (int)ksplit);
  else if (trans_b)
    hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, dim3(256), 0, st, p, (int)ksplit);
  else
    hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, dim3(256), 0, st, p, (int)ksplit);
  return (int)hipGetLastError();
}

int brv_gemm_f32(const float* a, const float* b, float* d, int64_t batch, int64_t M, int64_t N,
                 int64_t K, int64_t lda, int64_t ldb, int64_t ldd, int64_t a_batch_stride,
                 int64_t b_batch_stride, int64_t d_batch_stride, int trans_a, int trans_b,
                 int64_t kbatch, int64_t a_kbatch_stride, int64_t b_kbatch_stride,
                 const float* row_bias, int accumulate, brv_stream_t stream) {
  return gemm_any(0, a, b, d, batch, M, N, K, lda, ldb, ldd, a_batch_stride, b_batch_stride,
                  d_batch_stride, trans_a, trans_b, kbatch, a_kbatch_stride, b_kbatch_stride,
                  row_bias, accumulate, stream);
}
int64_t brv_gemm_f32_workspace_bytes(int64_t batch, int64_t M, int64_t N, int64_t K, int trans_a, int trans_b,
                                     int64_t kbatch) {
  if (batch < 1 || M < 32 || N < 32 || K < 1) return 0;
  if ((double)M*(double)N*(double)K*(double)batch*(kbatch > 1 ? kbatch : 1) < 3.0e7) return 0;
  brv::BigGemm g; memset(&g, 0, sizeof(g));
  g.M = (int)M; g.N = (int)N; g.K = (int)K; g.kbatch = kbatch > 1 ? (int)kbatch : 1; g.batch = (int)batch;
  g.ta = trans_a != 0; g.tb = trans_b != 0; g.lda = (int)(trans_a ? M : K); g.ldb = (int)(trans_b ? K : N);
  g.ldd = (int)N; g.x3 = 1 | 2 | 4 | 8;
  return 4*brv::gemm_f32_big_scratch(g);
}
int brv_gemm_f32_ws(const float* a, const float* b, float* d, int64_t batch, int64_t M, int64_t N,
                    int64_t K, int64_t lda, int64_t ldb, int64_t ldd, int64_t a_batch_stride,
                    int64_t b_batch_stride, int64_t d_batch_stride, int trans_a, int trans_b,
                    int64_t kbatch, int64_t a_kbatch_stride, int64_t b_kbatch_stride,
                    const float* row_bias, int accumulate, float* workspace, int64_t workspace_bytes,
                    brv_stream_t stream) {
  if (workspace && (((uintptr_t)workspace & 15) || workspace_bytes < 0)) return -1;
  return gemm_any(0, a, b, d, batch, M, N, K, lda, ldb, ldd, a_batch_stride, b_batch_stride,
                  d_batch_stride, trans_a, trans_b, kbatch, a_kbatch_stride, b_kbatch_stride,
                  row_bias, accumulate, stream, 0, nullptr, workspace, workspace ? workspace_bytes/4 : 0);
}
int brv_gemm_bf16_mixed(const void* a, const void* b, void* d, int64_t batch, int64_t M, int64_t N,
                        int64_t K, int64_t lda, int64_t ldb, int64_t ldd, int64_t a_batch_stride,
                        int64_t b_batch_stride, int64_t d_batch_stride, int trans_a, int trans_b,
                        int64_t kbatch, int64_t a_kbatch_stride, int64_t b_kbatch_stride,
                        const float* row_bias, int accumulate, int flags, brv_stream_t stream) {
  return gemm_any(1, (const float*)a, (const float*)b, (float*)d, batch, M, N, K, lda, ldb, ldd, a_batch_stride,
                  b_batch_stride, d_batch_stride, trans_a, trans_b, kbatch, a_kbatch_stride,
                  b_kbatch_stride, row_bias, accumulate, stream, flags & 7);
}
int brv_gemm_bf16_conv(const float* a, const float* image, float* d, int64_t batch, int64_t M,
                       int64_t N, int64_t K, int64_t lda, int64_t ldd, int64_t a_batch_stride,
                       int64_t image_batch_stride, int64_t d_batch_stride, int trans_a, int trans_b,
                       int64_t kbatch, int64_t a_kbatch_stride, int64_t image_kbatch_stride,
                       const float* row_bias, int accumulate, int mode, int64_t C, int64_t H,
                       int64_t W, int64_t kh, int64_t kw, int64_t sh, int64_t sw, int64_t ph,
                       int64_t pw, int64_t Ho, int64_t Wo, brv_stream_t stream) {
  if ((mode != 1 && mode != 2) || C < 1 || H < 1 || W < 1 || kh < 1 || kw < 1 || sh < 1 || sw < 1 ||
      Ho < 1 || Wo < 1) return -1;
  // the column matrix is (C*kh*kw) x (Ho*Wo): it is op_b's K x N (trans_b: N x K)
  const int64_t rows = C*kh*kw, pix = Ho*Wo;
  if ((trans_b ? N : K) != rows || (trans_b ? K : N) != pix || pix >= (1LL << 31)) return -1;
  const int conv[12] = {mode, (int)C, (int)H, (int)W, (int)kh, (int)kw, (int)sh, (int)sw, (int)ph,
                        (int)pw, (int)Ho, (int)Wo};
  return gemm_any(1, a, image, d, batch, M, N, K, lda, pix, ldd, a_batch_stride, image_batch_stride,
                  d_batch_stride, trans_a, trans_b, kbatch, a_kbatch_stride, image_kbatch_stride,
                  row_bias, accumulate, stream, 0, conv);
}
int brv_gemm_bf16(const float* a, const float* b, float* d, int64_t batch, int64_t M, int64_t N,
                  int64_t K, int64_t lda, int64_t ldb, int64_t ldd, int64_t a_batch_stride,
                  int64_t b_batch_stride, int64_t d_batch_stride, int trans_a, int trans_b,
                  int64_t kbatch, int64_t a_kbatch_stride, int64_t b_kbatch_stride,
                  const float* row_bias, int accumulate, brv_stream_t stream) {
  return gemm_any(1, a, b, d, batch, M, N, K, lda, ldb, ldd, a_batch_stride, b_batch_stride,
                  d_batch_stride, trans_a, trans_b, kbatch, a_kbatch_stride, b_kbatch_stride,
                  row_bias, accumulate, stream);
}

// Framed DFT / its transposes with every size explicit and the basis in DOUBLE precision (fp64
// MFMA accumulation): what STFT.forward / STFT.backward run on. `rows2` = basis rows = 2 x bins
// (bins = n_fft/2 + 1 one-sided, n_fft two-sided); frame t covers samples [t*hop - pad_left, + n).
int brv_dft64_forward(const float* x, const double* basis, float* spec, int64_t rows, int64_t length,
                      int64_t n, int64_t hop, int64_t pad_left, int64_t frames, int64_t bins,
                      float compression, float scale, brv_stream_t stream) {
  if (rows < 1 || frames < 1 || n < 2 || hop < 1 || bins < 1) return -1;
  G32 p; memset(&p, 0, sizeof(p));
  p.M = (int)(2*bins); p.N = (int)frames; p.K = (int)n;
  p.A64 = basis; p.A = nullptr; p.a_bs = 0; p.lda = (int)n;
  p.B = x; p.b_bs = length; p.hop = (int)hop; p.pad_left = (int)pad_left; p.len = (int)length;
  p.D = spec; p.d_bs = (long long)bins*frames*2; p.bins = (int)bins;
  p.comp = compression; p.scale = scale;
  return launch_g64<GA_PLAIN, GB_FRAMES, GS_SPEC>(p, (int)rows, (hipStream_t)stream);
}

// frames_out[r][t][m] = sum_c (spec/scale, decompressed)[c][t] * tbasis[c][m]  (tbasis: (2 bins, n)
// doubles, row-major): the synthesis product of the inverse transform (tbasis = inverse basis
// transposed) and of the forward transform's adjoint (tbasis = forward basis).
int brv_dft64_synthesis(const float* spec, const double* tbasis, float* frames_out, int64_t rows,
                        int64_t frames, int64_t n, int64_t bins, float compression, float scale,
                        brv_stream_t stream) {
  if (rows < 1 || frames < 1 || n < 2 || bins < 1) return -1;
  G32 p; memset(&p, 0, sizeof(p));
  p.M = (int)frames; p.N = (int)n; p.K = (int)(2*bins);
  p.A = spec; p.a_bs = (long long)bins*frames*2; p.frames = (int)frames;
  p.inv_scale = 1.f/scale; p.inv_comp = 1.f/compression;
  p.B64 = tbasis; p.B = nullptr; p.b_bs = 0; p.ldb = (int)n;
  p.D = frames_out; p.d_bs = (long long)frames*n; p.ldd = (int)n;
  return launch_g64<GA_SPEC_T, GB_PLAIN, GS_PLAIN>(p, (int)rows, (hipStream_t)stream);
}

// y[r][q] = sum_t frames[r][t][q + pad_left - t*hop] (/ window-square envelope iff window != NULL)
int brv_overlap_add(const float* frames_in, const float* window, float* y, int64_t rows,
                    int64_t frames, int64_t n, int64_t hop, int64_t pad_left, int64_t out_len,
                    brv_stream_t stream) {
  if (rows < 1 || frames < 1 || out_len < 1 || hop < 1) return -1;
  OlaParams o;
  o.frames = frames_in; o.y = y; o.win = window; o.F = (int)frames; o.n = (int)n; o.hop = (int)hop;
  o.out_len = (int)out_len; o.normalize = window != nullptr; o.pad_left = (int)pad_left;
  o.f_bs = (long long)frames*n; o.y_bs = out_len;
  int gx = (int)((out_len + 255)/256); if (gx > 1024) gx = 1024; if (gx < 1) gx = 1;
  hipLaunchKernelGGL(istft_ola_kernel, dim3(gx, (unsigned)rows), dim3(256), 0, (hipStream_t)stream, o);
  return (int)hipGetLastError();
}

int brv_pad_signal(const float* x, float* y, int64_t rows, int64_t length, int64_t left,
                   int64_t out_len, int mode, brv_stream_t stream) {
  if (rows < 1 || length < 1 || out_len < 1 || mode < 0 || mode > 3) return -1;
  if (mode == 1 && (left >= length || out_len - left - length >= length)) return -2;
  int gx = (int)((rows*out_len + 255)/256); if (gx > 4096) gx = 4096;
  hipLaunchKernelGGL(pad_signal_kernel, dim3(gx), dim3(256), 0, (hipStream_t)stream, x, y,
                     (long long)rows, (long long)length, (long long)left, (long long)out_len, mode);
  return (int)hipGetLastError();
}
int brv_polar(const float* mag, const float* phase, float* out, int64_t n, brv_stream_t stream) {
  if (n < 1) return -1;
  int gx = (int)((n + 255)/256); if (gx > 4096) gx = 4096;
  hipLaunchKernelGGL(polar_kernel, dim3(gx), dim3(256), 0, (hipStream_t)stream, mag, phase,
                     (float2*)out, (long long)n);
  return (int)hipGetLastError();
}
int brv_mag_phase(const float* x, float* mag, float* phase, int64_t n, brv_stream_t stream) {
  if (n < 1) return -1;
  int gx = (int)((n + 255)/256); if (gx > 4096) gx = 4096;
  hipLaunchKernelGGL(mag_phase_kernel, dim3(gx), dim3(256), 0, (hipStream_t)stream,
                     (const float2*)x, mag, phase, (long long)n);
  return (int)hipGetLastError();
}

int brv_spec_compress(const float* x, float* y, int64_t n, float compression, float scale,
                      brv_stream_t stream) {
  if (n < 1) return -1;
  int gx = (int)((n + 255)/256); if (gx > 4096) gx = 4096;
  hipLaunchKernelGGL(spec_compress_kernel, dim3(gx), dim3(256), 0, (hipStream_t)stream,
                     (const float2*)x, (float2*)y, (long long)n, compression, scale);
  return (int)hipGetLastError();
}
int brv_spec_compress_backward(const float* x, const float* gy, float* gx_out, int64_t n,
                               float compression, float scale, brv_stream_t stream) {
  if (n < 1) return -1;
  int gx = (int)((n + 255)/256); if (gx > 4096) gx = 4096;
  hipLaunchKernelGGL(spec_compress_bwd_kernel, dim3(gx), dim3(256), 0, (hipStream_t)stream,
                     (const float2*)x, (const float2*)gy, (float2*)gx_out, (long long)n,
                     compression, scale);
  return (int)hipGetLastError();
}

int brv_matmul_f32(const float* a, const float* b, float* d, int64_t batch, int64_t M,
                   int64_t N, int64_t K, int64_t a_batch_stride, brv_stream_t stream) {
  if (batch < 1 || M < 1 || N < 1 || K < 1) return -1;
  G32 p; memset(&p, 0, sizeof(p));
  p.M = (int)M; p.N = (int)N; p.K = (int)K;
  p.A = a; p.a_bs = a_batch_stride; p.lda = (int)K;
  p.B = b; p.b_bs = K*N; p.ldb = (int)N;
  p.D = d; p.d_bs = M*N; p.ldd = (int)N;
  return launch_g32<GA_PLAIN, GB_PLAIN, GS_PLAIN>(p, (int)batch, (hipStream_t)stream);
}

}  // extern "C"
