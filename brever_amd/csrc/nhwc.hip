// Channels-last fp16 companions of brv_conv_nhwc_forward: what the SGMSE+ score network does
// between its 3x3 convolutions under use_amp (reference brever/models/sgmse/net.py:180-477 --
// GroupNorm, FIR resampling, 1x1 convolutions, the progressive 4-channel side branch).
// Activations are (B, H, W, Cs) fp16, Cs a multiple of 8, channels >= C hold zeros; the 4-channel
// side branch and the network's input / output stay (B, C, H, W) fp32.
// All of these stream their tensors once; none is MFMA work.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include "../../include/brever_hip.h"
#include "common.cuh"

using namespace brv;

namespace {

#define NH_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return (int)e_; } while (0)

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float nh_silu(float v) {
  return v*__builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f*v));
}

// ---- layout conversion ------------------------------------------------------------------
// (B, C, HW) fp32 -> (B, HW, Cs) fp16, zero channels beyond C. One thread per (pixel, octet):
// reads are strided by HW per channel but consecutive threads take consecutive pixels.
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* x, _Float16* y, int C,
                                                           int Cs, long long HW) {
  const int oct = Cs >> 3;
  const long long b = blockIdx.z;
  const int o = blockIdx.y;
  if (o >= oct) return;
  const long long px = (long long)blockIdx.x*256 + threadIdx.x;
  if (px >= HW) return;
  h8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = o*8 + j;
    v[j] = c < C ? (_Float16)x[(b*C + c)*HW + px] : (_Float16)0.f;
  }
  *reinterpret_cast<h8*>(y + (b*HW + px)*Cs + o*8) = v;
}
// (B, HW, Cs) fp16 -> (B, C, HW) fp32
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const _Float16* x, float* y, int C, int Cs,
                                                           long long HW) {
  const long long b = blockIdx.z;
  const int o = blockIdx.y;
  const long long px = (long long)blockIdx.x*256 + threadIdx.x;
  if (px >= HW) return;
  const h8 v = *reinterpret_cast<const h8*>(x + (b*HW + px)*Cs + o*8);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = o*8 + j;
    if (c < C) y[(b*C + c)*HW + px] = (float)v[j];
  }
}

// ---- GroupNorm: per-channel sums, then the fold ---------------------------------------------
// sums[b][c][0..1] += (sum, sum of squares) of channel c over the pixels of this slice. Thread =
// (pixel lane, octet); fp32 partials over at most `per_thread` pixels, combined through LDS,
// one fp64 atomic pair per (workgroup, channel). Per-CHANNEL sums make the statistics of a channel
// concatenation the concatenation of the statistics, and a per-channel shift (the embedding
// term) an algebraic correction in the fold.
__global__ __launch_bounds__(256) void chan_stats_kernel(const _Float16* x, double* sums, int C,
                                                         int Cs, long long HW, int c_off, int Ctot,
                                                         long long slice) {
  __shared__ float red[256][17];
  const int oct = Cs >> 3;
  const int lanes = 256/oct;                        // pixels in flight
  const int o = threadIdx.x % oct, pl = threadIdx.x / oct;
  const long long b = blockIdx.y;
  const long long lo = (long long)blockIdx.x*slice;
  long long hi = lo + slice;
  if (hi > HW) hi = HW;
  float s[8], q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; }
  if (pl < lanes) {
    const _Float16* xb = x + b*HW*Cs + o*8;
    for (long long px = lo + pl; px < hi; px += lanes) {
      const f32x8 v = __builtin_convertvector(*reinterpret_cast<const h8*>(xb + px*Cs), f32x8);
#pragma unroll
      for (int j = 0; j < 8; ++j) { s[j] += v[j]; q[j] = fmaf(v[j], v[j], q[j]); }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { red[threadIdx.x][j] = s[j]; red[threadIdx.x][8 + j] = q[j]; }
  __syncthreads();
  // thread t < 16*oct: (octet, value k of 16) summed over the pixel lanes
  for (int t = threadIdx.x; t < oct*16; t += 256) {
    const int oo = t >> 4, k = t & 15;
    double a = 0.0;
    for (int l = 0; l < lanes; ++l) a += (double)red[l*oct + oo][k];
    const int c = oo*8 + (k & 7);
    if (c < C) atomicAdd(&sums[((b*Ctot + c_off + c) << 1) + (k >> 3)], a);
  }
}

// one workgroup (64 threads) per (item, group): statistics of x + add[b][c] from the channel
// sums, then scale = rstd*gamma*(1 + adm_scale), shift = (beta + (add - mean)*rstd*gamma)*(1 +
// adm_scale) + adm_shift (the arithmetic of gn_fold_kernel, sgmse.hip)
__global__ __launch_bounds__(64) void chan_fold_kernel(const double* sums, const double* sums2, int C1,
                                                       const float* add,
                                                       const float* gamma, const float* beta,
                                                       const float* adm_scale, const float* adm_shift,
                                                       float* scale, float* shift, int C, long long HW,
                                                       int groups, float eps) {
  __shared__ double red[2][64];
  const int bg = blockIdx.x, b = bg / groups, g = bg % groups;
  const int cpg = C/groups;
  double s1 = 0.0, s2 = 0.0;
  for (int j = threadIdx.x; j < cpg; j += 64) {
    const long long idx = (long long)b*C + g*cpg + j;
    const double e = add ? (double)add[idx] : 0.0;
    // channels [0, C1) from `sums` (B, C1, 2), the rest from `sums2` (B, C - C1, 2)
    const int ch = g*cpg + j;
    const double* sp = ch < C1 ? sums + (((long long)b*C1 + ch) << 1)
                               : sums2 + (((long long)b*(C - C1) + ch - C1) << 1);
    const double cs = sp[0], cq = sp[1];
    s1 += cs + (double)HW*e;
    s2 += cq + 2.0*e*cs + (double)HW*e*e;
  }
  red[0][threadIdx.x] = s1; red[1][threadIdx.x] = s2;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 64; ++i) { s1 += red[0][i]; s2 += red[1][i]; }
    red[0][0] = s1; red[1][0] = s2;
  }
  __syncthreads();
  const double n = (double)cpg*(double)HW;
  const double mean = red[0][0]/n;
  double var = red[1][0]/n - mean*mean;
  if (var < 0) var = 0;
  const float rstd = (float)(1.0/sqrt(var + (double)eps));
  for (int j = threadIdx.x; j < cpg; j += 64) {
    const int c = g*cpg + j;
    const long long idx = (long long)b*C + c;
    float sc = rstd*gamma[c];
    float sh = beta[c] + ((add ? add[idx] : 0.f) - (float)mean)*sc;
    if (adm_scale) { const float m = 1.f + adm_scale[idx]; sc *= m; sh = sh*m + adm_shift[idx]; }
    scale[idx] = sc; shift[idx] = sh;
  }
}

// y = act(scale[b][c]*x + shift[b][c]), channels-last fp16 both sides (zero beyond C)
__global__ __launch_bounds__(256) void nhwc_affine_act_kernel(const _Float16* x, const float* scale,
                                                              const float* shift, _Float16* y, int C,
                                                              int Cs, long long HW, int act) {
  const int oct = Cs >> 3;
  const long long b = blockIdx.y;
  const long long n = HW*oct;
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < n; i += (long long)gridDim.x*256) {
    const int o = (int)(i % oct);
    const f32x8 v = __builtin_convertvector(*reinterpret_cast<const h8*>(x + b*HW*Cs + i*8), f32x8);
    f32x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = o*8 + j;
      float t = 0.f;
      if (c < C) { t = scale[b*C + c]*v[j] + shift[b*C + c]; if (act) t = nh_silu(t); }
      r[j] = t;
    }
    *reinterpret_cast<h8*>(y + b*HW*Cs + i*8) = __builtin_convertvector(r, h8);
  }
}

// ---- depthwise FIR resampling by 2 (Resample.forward), channels-last fp16 ---------------------
// down = conv2d(stride 2, padding (ph, pw)); up = conv_transpose2d(stride 2, padding, output
// padding) with the kernel times `gain`. One thread per (output pixel, octet).
template <bool UP>
__global__ __launch_bounds__(256) void nhwc_fir_kernel(const _Float16* x, const float* k, _Float16* y,
                                                       int Cs, int H, int W, int Ho, int Wo, int K,
                                                       int ph, int pw, float gain) {
  const int oct = Cs >> 3;
  const long long b = blockIdx.y;
  const long long n = (long long)Ho*Wo*oct;
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < n; i += (long long)gridDim.x*256) {
    const int o = (int)(i % oct);
    const long long opx = i / oct;
    const int wo = (int)(opx % Wo), ho = (int)(opx / Wo);
    f32x8 acc;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int a = 0; a < K; ++a) {
      int hi;
      if (UP) { const int hn = ho + ph - a; if (hn < 0 || (hn & 1)) continue; hi = hn >> 1; }
      else hi = ho*2 - ph + a;
      if (hi < 0 || hi >= H) continue;
      for (int c = 0; c < K; ++c) {
        int wi;
        if (UP) { const int wn = wo + pw - c; if (wn < 0 || (wn & 1)) continue; wi = wn >> 1; }
        else wi = wo*2 - pw + c;
        if (wi < 0 || wi >= W) continue;
        const f32x8 v = __builtin_convertvector(
            *reinterpret_cast<const h8*>(x + ((b*H + hi)*W + wi)*Cs + o*8), f32x8);
        const float kv = k[a*K + c];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = fmaf(v[j], kv, acc[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] *= gain;
    *reinterpret_cast<h8*>(y + ((b*Ho + ho)*Wo + wo)*(long long)Cs + o*8) = __builtin_convertvector(acc, h8);
  }
}

// Both resamplings a UNetBlock needs of one tensor in ONE pass (models/sgmse.py UNetBlock.forward_h):
//   y_raw = FIR(x)   and   y_act = FIR(act(scale*x + shift))
// (the GroupNorm fold + SiLU in front of the second one used to be a pass of its own: x read three
// times and the activated copy written and read back). A workgroup stages the input window of an
// output tile x 32 channels in LDS, raw and activated (fp16 both: the activated copy has the rounding
// it had in HBM), then every output is a handful of LDS reads. K <= 4.
struct FirDualParams {
  const _Float16* x; const float* scale; const float* shift; const float* k;
  _Float16* y_raw; _Float16* y_act;
  int C, Cs, H, W, Ho, Wo, K, ph, pw, act, tiles_w;
  float gain;
};
template <bool UP>
__global__ __launch_bounds__(256) void nhwc_fir_dual_kernel(const FirDualParams p) {
  constexpr int TH = UP ? 16 : 8, TW = UP ? 32 : 16;
  constexpr int MAXPX = UP ? 10*18 : 18*34;          // window of a tile for K <= 4
  __shared__ h8 raw[MAXPX*4];
  __shared__ h8 act[MAXPX*4];
  __shared__ float ks[16];
  const int tid = threadIdx.x;
  const int ho0 = (blockIdx.x / p.tiles_w)*TH, wo0 = (blockIdx.x % p.tiles_w)*TW;
  const int o = tid & 3, oc = blockIdx.y*4 + o;        // channel octet of this thread (fixed: 256 % 4 == 0)
  const long long b = blockIdx.z;
  const int K = p.K;
  if (tid < K*K) ks[tid] = p.k[tid];
  int hi_lo, hi_hi, wi_lo, wi_hi;
  if (UP) {
    const int a = ho0 + p.ph - (K - 1), c = wo0 + p.pw - (K - 1);
    hi_lo = a <= 0 ? 0 : (a + 1) >> 1; hi_hi = (ho0 + TH - 1 + p.ph) >> 1;
    wi_lo = c <= 0 ? 0 : (c + 1) >> 1; wi_hi = (wo0 + TW - 1 + p.pw) >> 1;
  } else {
    hi_lo = ho0*2 - p.ph; hi_hi = (ho0 + TH - 1)*2 - p.ph + K - 1;
    wi_lo = wo0*2 - p.pw; wi_hi = (wo0 + TW - 1)*2 - p.pw + K - 1;
  }
  const int nrows = hi_hi - hi_lo + 1, ncols = wi_hi - wi_lo + 1;
  f32x8 sc, sh;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = oc*8 + j;
    const bool ok = c < p.C;
    sc[j] = ok ? p.scale[b*p.C + c] : 0.f; sh[j] = ok ? p.shift[b*p.C + c] : 0.f;
  }
  const bool oc_ok = oc*8 < p.Cs;
  // all window pieces of the thread are requested before the first is transformed (one at a time the loop was a
  // chain of up to ten memory round trips: 268 us for 413 MB at full resolution)
  constexpr int NPC = (MAXPX*4 + 255)/256;
  h8 pre[NPC];
  bool pok[NPC];
#pragma unroll
  for (int q = 0; q < NPC; ++q) {
    const int idx = tid + 256*q;
    const int px = idx >> 2;
    const int hi = hi_lo + px / ncols, wi = wi_lo + px % ncols;
    pok[q] = idx < nrows*ncols*4 && oc_ok && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;
    const h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    pre[q] = z;
    if (pok[q]) pre[q] = *reinterpret_cast<const h8*>(p.x + ((b*p.H + hi)*p.W + wi)*p.Cs + oc*8);
  }
#pragma unroll
  for (int q = 0; q < NPC; ++q) {
    const int idx = tid + 256*q;
    h8 v = pre[q], t;
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = (_Float16)0.f;
    if (pok[q]) {
      const f32x8 vf = __builtin_convertvector(v, f32x8);
      f32x8 r;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float u = 0.f;
        if (oc*8 + j < p.C) { u = sc[j]*vf[j] + sh[j]; if (p.act) u = nh_silu(u); }
        r[j] = u;
      }
      t = __builtin_convertvector(r, h8);
    }
    if (idx < nrows*ncols*4) { raw[idx] = v; act[idx] = t; }
  }
  __syncthreads();
  if (!oc_ok) return;
  for (int item = tid; item < TH*TW*4; item += 256) {
    const int opx = item >> 2;
    const int ho = ho0 + opx / TW, wo = wo0 + opx % TW;
    if (ho >= p.Ho || wo >= p.Wo) continue;
    f32x8 ar, aa;
#pragma unroll
    for (int j = 0; j < 8; ++j) { ar[j] = 0.f; aa[j] = 0.f; }
    for (int a = 0; a < K; ++a) {
      int hi;
      if (UP) { const int hn = ho + p.ph - a; if (hn < 0 || (hn & 1)) continue; hi = hn >> 1; }
      else hi = ho*2 - p.ph + a;
      if (hi < 0 || hi >= p.H) continue;
      for (int c = 0; c < K; ++c) {
        int wi;
        if (UP) { const int wn = wo + p.pw - c; if (wn < 0 || (wn & 1)) continue; wi = wn >> 1; }
        else wi = wo*2 - p.pw + c;
        if (wi < 0 || wi >= p.W) continue;
        const int li = ((hi - hi_lo)*ncols + (wi - wi_lo))*4 + o;
        const f32x8 v = __builtin_convertvector(raw[li], f32x8);
        const f32x8 t = __builtin_convertvector(act[li], f32x8);
        const float kv = ks[a*K + c];
#pragma unroll
        for (int j = 0; j < 8; ++j) { ar[j] = fmaf(v[j], kv, ar[j]); aa[j] = fmaf(t[j], kv, aa[j]); }
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { ar[j] *= p.gain; aa[j] *= p.gain; }
    const long long off = ((b*p.Ho + ho)*p.Wo + wo)*(long long)p.Cs + oc*8;
    *reinterpret_cast<h8*>(p.y_raw + off) = __builtin_convertvector(ar, h8);
    *reinterpret_cast<h8*>(p.y_act + off) = __builtin_convertvector(aa, h8);
  }
}

// out = alpha*a + beta*b on fp16 tensors (b nullable), fp32 arithmetic
__global__ __launch_bounds__(256) void nhwc_axpby_kernel(const _Float16* a, float alpha,
                                                         const _Float16* b, float beta, _Float16* out,
                                                         long long n8) {
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < n8; i += (long long)gridDim.x*256) {
    const f32x8 va = __builtin_convertvector(*reinterpret_cast<const h8*>(a + i*8), f32x8);
    f32x8 r;
    if (b) {
      const f32x8 vb = __builtin_convertvector(*reinterpret_cast<const h8*>(b + i*8), f32x8);
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] = alpha*va[j] + beta*vb[j];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] = alpha*va[j];
    }
    *reinterpret_cast<h8*>(out + i*8) = __builtin_convertvector(r, h8);
  }
}

// ---- 1x1 convolution, channels-last fp16 (UNetBlock.skip_conv): a row-major GEMM
//   Y[pixel][co] = out_scale*(bias[co] + sum_ci W[co][ci] X[pixel][ci]),  X = [x1 | x2]
// on v_mfma_f32_32x32x16_f16 with NO LDS staging of the operands: channels-last rows ARE the B
// fragment (lane (pixel n, k-half) reads its 16 contiguous bytes), the packed weights the A
// fragment (the same 1-KB pieces for every workgroup: L1 / L2 hits). A workgroup = 4 waves x 32
// pixels x up to 128 output channels; the result leaves through LDS as whole pixel rows.
// Memory-bound by construction (arithmetic intensity Cout/2 FLOP per byte).
struct Pw1Params {
  const _Float16* x1; const _Float16* x2; const h8* wp; const float* bias; _Float16* y;
  int C1, C1s, C2, C2s, Cout, Cys, n_ks1, n_ks;
  long long npx; float out_scale;
};
// NB = 128-channel output blocks per workgroup. With NB = 1 and Cout = 256 the two blocks of a pixel tile were two
// workgroups far apart in the dispatch order (blockIdx.y): the input was fetched from HBM twice -- 790 MB per launch
// at full resolution where 527 MB are algorithmic (profiles/r05_rows_sgmse_b8_pmc_hbm_traffic.json). NB = 2: one
// workgroup multiplies its 128 pixels against both blocks (8 accumulators per wave), the input is read once.
#ifndef PW1_ABL
#define PW1_ABL 0      // diagnostic builds (results wrong): 1 no weight loads, 2 no MFMAs, 4 no stores, 8 no input loads
#endif
template <int NB>
__global__ __launch_bounds__(256, 2) void nhwc_conv1x1_kernel(const Pw1Params p) {
  constexpr int ESTRIDE = 272, CF = 4*NB;
  __shared__ __attribute__((aligned(16))) unsigned char stage[128*ESTRIDE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n32 = lane & 31, khalf = lane >> 5;
  const int n_cob = (p.Cout + 127)/128;
  // 1-D grid (gridDim.y == 1): ids congruent mod 8 run on one XCD (one L2): XCD x takes the pixel tiles 8 j + x and
  // walks the output blocks of a tile in consecutive slots, so that a tile's input is fetched into that L2 once
  long long ptile = blockIdx.x;
  int cob0 = blockIdx.y*NB;
  if (gridDim.y == 1 && n_cob > NB) {
    const int nyb = (n_cob + NB - 1)/NB;
    const unsigned int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    cob0 = (int)(slot % nyb)*NB;
    ptile = (long long)(slot / nyb)*8 + xcd;
    if (ptile*128 >= p.npx) return;                // (padding of the grid: whole workgroups, before any barrier)
  }
  const long long px0 = ptile*128;
  const long long px = px0 + wave*32 + n32;
  const bool pok = px < p.npx;
  f32x16 acc[CF];
#pragma unroll
  for (int cf = 0; cf < CF; ++cf)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[cf][i] = 0.f;
  // (a second block past the last one reads the first block's weights again: its results are not stored)
  const h8* wa[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
    wa[nb] = p.wp + (long long)(cob0 + nb < n_cob ? cob0 + nb : cob0)*p.n_ks*4*64 + lane;
  const h8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
  // operands of k-step ks + 1 are requested before the MFMAs of k-step ks (small images run a few
  // workgroups whose reduction is a chain of load latencies otherwise)
  auto load_b = [&](int ks) -> h8 {
    const bool second = ks >= p.n_ks1;
    const _Float16* xb = second ? p.x2 : p.x1;
    const int cs = second ? p.C2s : p.C1s;
    const int c0 = (second ? ks - p.n_ks1 : ks)*16 + khalf*8;
    h8 bv = zero;
    if (pok && c0 < cs && !(PW1_ABL & 8)) bv = *reinterpret_cast<const h8*>(xb + px*cs + c0);
    return bv;
  };
  h8 bn = load_b(0), an[CF];
#pragma unroll
  for (int cf = 0; cf < CF; ++cf) an[cf] = (PW1_ABL & 1) ? zero : wa[cf >> 2][(cf & 3)*64];
  for (int ks = 0; ks < p.n_ks; ++ks) {
    const h8 bv = bn;
    h8 av[CF];
#pragma unroll
    for (int cf = 0; cf < CF; ++cf) av[cf] = an[cf];
    if (ks + 1 < p.n_ks) {
      bn = load_b(ks + 1);
#pragma unroll
      for (int cf = 0; cf < CF; ++cf) if (!(PW1_ABL & 1)) an[cf] = wa[cf >> 2][((ks + 1)*4 + (cf & 3))*64];
    }
#pragma unroll
    for (int cf = 0; cf < CF; ++cf) {
      if (PW1_ABL & 2) { acc[cf][0] += (float)av[cf][0]*(float)bv[cf & 7]; continue; }
      acc[cf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[cf], bv, acc[cf], 0, 0, 0);
    }
  }
  // D[co][pixel] -> LDS [pixel][co] -> 16-byte pieces of whole pixel rows, one 128-channel block at a time
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int cob = cob0 + nb;
    if (nb > 0) __syncthreads();                   // (the staging buffer's readers of the previous block)
#pragma unroll
    for (int cq = 0; cq < 4; ++cq)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        h4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (_Float16)acc[nb*4 + cq][g*4 + j];
        *reinterpret_cast<h4*>(stage + (wave*32 + n32)*ESTRIDE + (cq*32 + g*8 + khalf*4)*2) = o;
      }
    __syncthreads();
    if (cob >= n_cob) continue;
    const int c8 = tid & 15, co = cob*128 + c8*8;
    float bv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bv[j] = p.bias && co + j < p.Cout ? p.bias[co + j] : 0.f;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int r = (it*256 + tid) >> 4;
      const long long q = px0 + r;
      if (q >= p.npx || co >= p.Cout) continue;
      const f32x8 v = __builtin_convertvector(*reinterpret_cast<const h8*>(stage + r*ESTRIDE + c8*16), f32x8);
      f32x8 w;
#pragma unroll
      for (int j = 0; j < 8; ++j) w[j] = (v[j] + bv[j])*p.out_scale;
      const h8 o = __builtin_convertvector(w, h8);
      if (PW1_ABL & 4) { if (w[0] == 123.456f) p.y[q] = o[0]; continue; }
      if (co + 8 <= p.Cout) *reinterpret_cast<h8*>(p.y + q*p.Cys + co) = o;
      else {
#pragma unroll
        for (int j = 0; j < 8; ++j) if (co + j < p.Cout) p.y[q*p.Cys + co + j] = o[j];
      }
    }
  }
}
// The same product with the WEIGHTS staged through LDS (round 5). In the kernel above every wave fetches its own A
// fragments: 4 KB per k-step and wave against 1 KB of input, all through the vector memory path -- the ablations of
// tools/pw1_bench.py price them at 155 of the 433 us of the 512 -> 256 channel launch (no weight loads: 278 us; no
// input loads: 198; neither, no MFMAs: 60 = the stores alone). Here the 256 threads of a workgroup copy the weights of
// four k-steps (16 KB, contiguous in the packed layout) into one of two LDS buffers while the previous four are
// multiplied; the four waves read their A fragments from LDS (lane-linear 16-byte reads), one barrier per four
// k-steps. The epilogue's staging area reuses the weight buffers: 34 KB of LDS as before, same occupancy.
__global__ __launch_bounds__(256, 2) void nhwc_conv1x1_lds_kernel(const Pw1Params p) {
  constexpr int ESTRIDE = 272, KC = 4, WB = KC*4096;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2*WB > 128*ESTRIDE ? 2*WB : 128*ESTRIDE];
  unsigned char* stage = lds;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n32 = lane & 31, khalf = lane >> 5;
  const int n_cob = (p.Cout + 127)/128;
  long long ptile = blockIdx.x;
  int cob = blockIdx.y;
  if (gridDim.y == 1 && n_cob > 1) {               // XCD-aware 1-D grid, as above
    const unsigned int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    cob = (int)(slot % n_cob);
    ptile = (long long)(slot / n_cob)*8 + xcd;
    if (ptile*128 >= p.npx) return;
  }
  const long long px0 = ptile*128;
  const long long px = px0 + wave*32 + n32;
  const bool pok = px < p.npx;
  f32x16 acc[4];
#pragma unroll
  for (int cf = 0; cf < 4; ++cf)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[cf][i] = 0.f;
  const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(p.wp) + (long long)cob*p.n_ks*4096 + tid*16;
  const h8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
  auto load_b = [&](int ks) -> h8 {
    const bool second = ks >= p.n_ks1;
    const _Float16* xb = second ? p.x2 : p.x1;
    const int cs = second ? p.C2s : p.C1s;
    const int c0 = (second ? ks - p.n_ks1 : ks)*16 + khalf*8;
    h8 bv = zero;
    if (pok && ks < p.n_ks && c0 < cs) bv = *reinterpret_cast<const h8*>(xb + px*cs + c0);
    return bv;
  };
  const int n_chunks = (p.n_ks + KC - 1)/KC;
  uint4 wr[KC];
  auto load_w = [&](int c) {                       // the chunk's four k-steps: piece i = k-step KC c + i
#pragma unroll
    for (int i = 0; i < KC; ++i)
      wr[i] = KC*c + i < p.n_ks ? *reinterpret_cast<const uint4*>(wsrc + (long long)(KC*c + i)*4096) : make_uint4(0, 0, 0, 0);
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int i = 0; i < KC; ++i) *reinterpret_cast<uint4*>(lds + buf*WB + i*4096 + tid*16) = wr[i];
  };
  load_w(0);
  h8 bn = load_b(0), bn2 = load_b(1);              // input fragments two k-steps ahead
  store_w(0);
  __syncthreads();
  for (int c = 0; c < n_chunks; ++c) {
    if (c + 1 < n_chunks) load_w(c + 1);
    const unsigned char* wb = lds + (c & 1)*WB + lane*16;
#pragma unroll
    for (int i = 0; i < KC; ++i) {
      const int ks = KC*c + i;
      const h8 bv = bn;
      bn = bn2; bn2 = load_b(ks + 2);
      if (ks < p.n_ks) {
#pragma unroll
        for (int cf = 0; cf < 4; ++cf) {
          const h8 av = *reinterpret_cast<const h8*>(wb + i*4096 + cf*1024);
          acc[cf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc[cf], 0, 0, 0);
        }
      }
    }
    if (c + 1 < n_chunks) store_w((c + 1) & 1);
    __syncthreads();
  }
  // D[co][pixel] -> LDS [pixel][co] -> 16-byte pieces of whole pixel rows (the weight buffers are free now)
#pragma unroll
  for (int cf = 0; cf < 4; ++cf)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      typedef _Float16 h4 __attribute__((ext_vector_type(4)));
      h4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (_Float16)acc[cf][g*4 + j];
      *reinterpret_cast<h4*>(stage + (wave*32 + n32)*ESTRIDE + (cf*32 + g*8 + khalf*4)*2) = o;
    }
  __syncthreads();
  const int c8 = tid & 15, co = cob*128 + c8*8;
  float bv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bv[j] = p.bias && co + j < p.Cout ? p.bias[co + j] : 0.f;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int r = (it*256 + tid) >> 4;
    const long long q = px0 + r;
    if (q >= p.npx || co >= p.Cout) continue;
    const f32x8 v = __builtin_convertvector(*reinterpret_cast<const h8*>(stage + r*ESTRIDE + c8*16), f32x8);
    f32x8 w;
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = (v[j] + bv[j])*p.out_scale;
    const h8 o = __builtin_convertvector(w, h8);
    if (co + 8 <= p.Cout) *reinterpret_cast<h8*>(p.y + q*p.Cys + co) = o;
    else {
#pragma unroll
      for (int j = 0; j < 8; ++j) if (co + j < p.Cout) p.y[q*p.Cys + co + j] = o[j];
    }
  }
}
// ... and the INPUT through LDS as well: per chunk of four k-steps (64 channels) a pixel row contributes 128
// contiguous bytes, fetched by eight threads (a wave-load = eight whole cache lines instead of 32-byte pieces of 32
// rows); rows sit 144 bytes apart in LDS (the 32 lanes of a B-fragment read fall on distinct bank groups).
__global__ __launch_bounds__(256, 2) void nhwc_conv1x1_lds2_kernel(const Pw1Params p) {
  constexpr int ESTRIDE = 272, KC = 4, WB = KC*4096, XS = 144, XB = 128*XS;
  constexpr int LDS = 2*WB + 2*XB > 128*ESTRIDE ? 2*WB + 2*XB : 128*ESTRIDE;
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS];
  unsigned char* stage = lds;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n32 = lane & 31, khalf = lane >> 5;
  const int n_cob = (p.Cout + 127)/128;
  long long ptile = blockIdx.x;
  int cob = blockIdx.y;
  if (gridDim.y == 1 && n_cob > 1) {
    const unsigned int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    cob = (int)(slot % n_cob);
    ptile = (long long)(slot / n_cob)*8 + xcd;
    if (ptile*128 >= p.npx) return;
  }
  const long long px0 = ptile*128;
  f32x16 acc[4];
#pragma unroll
  for (int cf = 0; cf < 4; ++cf)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[cf][i] = 0.f;
  const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(p.wp) + (long long)cob*p.n_ks*4096 + tid*16;
  const int n_chunks = (p.n_ks + KC - 1)/KC;
  uint4 wr[KC], xr[4];
  auto load_w = [&](int c) {
#pragma unroll
    for (int i = 0; i < KC; ++i)
      wr[i] = KC*c + i < p.n_ks ? *reinterpret_cast<const uint4*>(wsrc + (long long)(KC*c + i)*4096) : make_uint4(0, 0, 0, 0);
  };
  // piece j of this thread: pixel row (tid >> 3) + 32 j, bytes 16 (tid & 7) of the chunk's 128 (k-steps KC c ..: the
  // k-step of a piece is (tid & 7) >> 1 -- the two sources never share a chunk's k-step, but may share a chunk)
  auto load_x = [&](int c) {
    const int ks = KC*c + ((tid & 7) >> 1);
    const bool second = ks >= p.n_ks1;
    const _Float16* xb = second ? p.x2 : p.x1;
    const int cs = second ? p.C2s : p.C1s;
    const int c0 = (second ? ks - p.n_ks1 : ks)*16 + (tid & 1)*8;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long long q = px0 + (tid >> 3) + 32*j;
      xr[j] = (q < p.npx && ks < p.n_ks && c0 < cs) ? *reinterpret_cast<const uint4*>(xb + q*cs + c0) : make_uint4(0, 0, 0, 0);
    }
  };
  auto store_wx = [&](int buf) {
#pragma unroll
    for (int i = 0; i < KC; ++i) *reinterpret_cast<uint4*>(lds + buf*WB + i*4096 + tid*16) = wr[i];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<uint4*>(lds + 2*WB + buf*XB + ((tid >> 3) + 32*j)*XS + (tid & 7)*16) = xr[j];
  };
  load_w(0); load_x(0);
  store_wx(0);
  __syncthreads();
  for (int c = 0; c < n_chunks; ++c) {
    if (c + 1 < n_chunks) { load_w(c + 1); load_x(c + 1); }
    const unsigned char* wb = lds + (c & 1)*WB + lane*16;
    const unsigned char* xb = lds + 2*WB + (c & 1)*XB + (wave*32 + n32)*XS + khalf*16;
#pragma unroll
    for (int i = 0; i < KC; ++i) {
      if (KC*c + i < p.n_ks) {
        const h8 bv = *reinterpret_cast<const h8*>(xb + i*32);
#pragma unroll
        for (int cf = 0; cf < 4; ++cf) {
          const h8 av = *reinterpret_cast<const h8*>(wb + i*4096 + cf*1024);
          acc[cf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc[cf], 0, 0, 0);
        }
      }
    }
    if (c + 1 < n_chunks) store_wx((c + 1) & 1);
    __syncthreads();
  }
#pragma unroll
  for (int cf = 0; cf < 4; ++cf)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      typedef _Float16 h4 __attribute__((ext_vector_type(4)));
      h4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (_Float16)acc[cf][g*4 + j];
      *reinterpret_cast<h4*>(stage + (wave*32 + n32)*ESTRIDE + (cf*32 + g*8 + khalf*4)*2) = o;
    }
  __syncthreads();
  const int c8 = tid & 15, co = cob*128 + c8*8;
  float bv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bv[j] = p.bias && co + j < p.Cout ? p.bias[co + j] : 0.f;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int r = (it*256 + tid) >> 4;
    const long long q = px0 + r;
    if (q >= p.npx || co >= p.Cout) continue;
    const f32x8 v = __builtin_convertvector(*reinterpret_cast<const h8*>(stage + r*ESTRIDE + c8*16), f32x8);
    f32x8 w;
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = (v[j] + bv[j])*p.out_scale;
    const h8 o = __builtin_convertvector(w, h8);
    if (co + 8 <= p.Cout) *reinterpret_cast<h8*>(p.y + q*p.Cys + co) = o;
    else {
#pragma unroll
      for (int j = 0; j < 8; ++j) if (co + j < p.Cout) p.y[q*p.Cys + co + j] = o[j];
    }
  }
}
// wp[co block of 128][k-step][co fragment 4][lane][8] <- w[co][ci]; the k-steps of the second
// source start at a multiple of 16 channels (C1 padded up)
__global__ __launch_bounds__(256) void conv1x1_pack_kernel(const float* w, _Float16* wp, int Cout,
                                                           int C1, int C2, int n_ks1, int n_ks,
                                                           long long total) {
  for (long long idx = (long long)blockIdx.x*256 + threadIdx.x; idx < total;
       idx += (long long)gridDim.x*256) {
    long long r = idx;
    const int j = (int)(r % 8); r /= 8;
    const int lane = (int)(r % 64); r /= 64;
    const int cf = (int)(r % 4); r /= 4;
    const int ks = (int)(r % n_ks); r /= n_ks;
    const int co = (int)r*128 + cf*32 + (lane & 31);
    const int kk = (lane >> 5)*8 + j;
    int ci = -1;
    if (ks < n_ks1) { const int c = ks*16 + kk; if (c < C1) ci = c; }
    else { const int c = (ks - n_ks1)*16 + kk; if (c < C2) ci = C1 + c; }
    float v = 0.f;
    if (co < Cout && ci >= 0) v = w[(long long)co*(C1 + C2) + ci];
    wp[idx] = (_Float16)v;
  }
}

// ---- 3x3 convolution to a FEW output channels (<= 8): the 4-channel progressive output branch
// (AuxiliaryUp.conv) and the final output convolution. x channels-last fp16 with an optional
// folded GroupNorm (+SiLU); y (B, Cout, H, W) fp32, y = [y +] conv + bias. One thread per pixel
// would re-read every input 9 times from L1; instead a workgroup stages the (8+2) x (32+2) pixel
// patch of 64-channel slabs in LDS, already activated, and its 256 threads each own one pixel.
struct SmallCoutParams {
  const _Float16* x; const _Float16* w16; const float* bias; const float* scale; const float* shift;
  const float* yin; float* y;
  int C, Cs, Cout, H, W, silu;
};
// w16: weights as fp16 [tap 9][Cout][C] (brv_nhwc_conv3x3_small_pack). The 256 threads of a
// workgroup each own one pixel of an 8 x 32 tile; per 64-channel slab the activated (10 x 34)-
// pixel patch sits in LDS as fp16, the weights of a (tap, output channel, 8 input channels) are
// wave-uniform -> scalar loads, and the products are v_dot2_f32_f16 (two MACs per lane and
// instruction, fp32 accumulation).
template <int COUT>
__global__ __launch_bounds__(256) void nhwc_conv3x3_small_kernel(const SmallCoutParams p) {
  constexpr int TR = 8, TC = 32, PR = TR + 2, PC = TC + 2, SLAB = 64;
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  __shared__ _Float16 patch[PR*PC][SLAB + 8];          // +8: rows 144 B apart (bank spread)
  __shared__ _Float16 wl[9][COUT][SLAB];               // the slab's weights (read as broadcasts)
  const int tid = threadIdx.x;
  const int b = blockIdx.z, h0 = blockIdx.y*TR, w0 = blockIdx.x*TC;
  const int r = tid / TC, c = tid % TC;
  float acc[COUT];
#pragma unroll
  for (int o = 0; o < COUT; ++o) acc[o] = 0.f;
  const long long HW = (long long)p.H*p.W;
  for (int cb = 0; cb < p.C; cb += SLAB) {
    __syncthreads();
    // all of the thread's patch pieces are requested before the first is transformed (one piece at a time the staging
    // loop was a chain of eleven memory round trips per slab: 304 us per launch at full resolution)
    constexpr int NPC = (PR*PC*(SLAB/8) + 255)/256;
    h8 raw[NPC];
    bool rok[NPC];
#pragma unroll
    for (int q = 0; q < NPC; ++q) {
      const int i = tid + 256*q;
      const int px = i / (SLAB/8), o8 = i % (SLAB/8);
      const int h = h0 + px / PC - 1, w = w0 + px % PC - 1;
      const int ch = cb + o8*8;
      rok[q] = i < PR*PC*(SLAB/8) && h >= 0 && h < p.H && w >= 0 && w < p.W && ch < p.C;
      const h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      raw[q] = z;
      if (rok[q]) raw[q] = *reinterpret_cast<const h8*>(p.x + (b*HW + (long long)h*p.W + w)*p.Cs + ch);
    }
#pragma unroll
    for (int q = 0; q < NPC; ++q) {
      const int i = tid + 256*q;
      const int px = i / (SLAB/8), o8 = i % (SLAB/8);
      const int ch = cb + o8*8;
      f32x8 v = __builtin_convertvector(raw[q], f32x8);        // (zeros where the piece lies outside)
      if (rok[q] && p.scale) {
        const float* scp = p.scale + (long long)b*p.C + ch;        // (ch, C: multiples of 8)
        const float* shp = p.shift + (long long)b*p.C + ch;
        float sc[8], sh[8];
        if (((reinterpret_cast<unsigned long long>(p.scale) | reinterpret_cast<unsigned long long>(p.shift)) & 15) == 0) {
          const float4 s0 = reinterpret_cast<const float4*>(scp)[0], s1 = reinterpret_cast<const float4*>(scp)[1];
          const float4 t0 = reinterpret_cast<const float4*>(shp)[0], t1 = reinterpret_cast<const float4*>(shp)[1];
          sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
          sh[0] = t0.x; sh[1] = t0.y; sh[2] = t0.z; sh[3] = t0.w; sh[4] = t1.x; sh[5] = t1.y; sh[6] = t1.z; sh[7] = t1.w;
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) { sc[j] = scp[j]; sh[j] = shp[j]; }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float t = sc[j]*v[j] + sh[j];
          v[j] = p.silu ? nh_silu(t) : t;
        }
      }
      if (i < PR*PC*(SLAB/8)) *reinterpret_cast<h8*>(&patch[px][o8*8]) = __builtin_convertvector(v, h8);
    }
    for (int i = tid; i < 9*COUT*(SLAB/8); i += 256) {
      const int k8 = i % (SLAB/8), o = (i / (SLAB/8)) % COUT, t = i / ((SLAB/8)*COUT);
      h8 wv = {0, 0, 0, 0, 0, 0, 0, 0};
      if (o < p.Cout && cb + k8*8 < p.C)
        wv = *reinterpret_cast<const h8*>(p.w16 + ((long long)(t*p.Cout + o))*p.C + cb + k8*8);
      *reinterpret_cast<h8*>(&wl[t][o][k8*8]) = wv;
    }
    __syncthreads();
#ifndef SMALL_W_LDS
#define SMALL_W_LDS 0          // diagnostic builds: 1 = the weights as per-thread LDS reads (round 4)
#endif
    // the weights of a (tap, output channel, octet) are the same for every thread: wave-uniform global addresses
    // (scalar loads, the products take them as scalar operands) instead of 288 16-byte LDS reads per thread and slab
    if (!SMALL_W_LDS && cb + SLAB <= p.C && COUT <= p.Cout) {
#pragma unroll 1
      for (int t = 0; t < 9; ++t) {
        const _Float16* pp = patch[(r + t/3)*PC + c + t%3];
        const _Float16* wg = p.w16 + (long long)t*p.Cout*p.C + cb;
#pragma unroll
        for (int k = 0; k < SLAB; k += 8) {
          const h8 v = *reinterpret_cast<const h8*>(pp + k);
#pragma unroll
          for (int o = 0; o < COUT; ++o) {
            const h8 wv = *reinterpret_cast<const h8*>(wg + (long long)o*p.C + k);
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc[o] = __builtin_amdgcn_fdot2(h2{v[2*j], v[2*j + 1]}, h2{wv[2*j], wv[2*j + 1]}, acc[o], false);
          }
        }
      }
    } else {
#pragma unroll 1
      for (int t = 0; t < 9; ++t) {
        const _Float16* pp = patch[(r + t/3)*PC + c + t%3];
#pragma unroll
        for (int k = 0; k < SLAB; k += 8) {
          const h8 v = *reinterpret_cast<const h8*>(pp + k);
#pragma unroll
          for (int o = 0; o < COUT; ++o) {
            const h8 wv = *reinterpret_cast<const h8*>(&wl[t][o][k]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc[o] = __builtin_amdgcn_fdot2(h2{v[2*j], v[2*j + 1]}, h2{wv[2*j], wv[2*j + 1]}, acc[o], false);
          }
        }
      }
    }
  }
  const int h = h0 + r, w = w0 + c;
  if (h >= p.H || w >= p.W) return;
#pragma unroll
  for (int o = 0; o < COUT; ++o) {
    if (o >= p.Cout) break;
    const long long idx = ((long long)b*p.Cout + o)*HW + (long long)h*p.W + w;
    float v = acc[o] + (p.bias ? p.bias[o] : 0.f);
    if (p.yin) v += p.yin[idx];
    p.y[idx] = v;
  }
}
// w16[tap][co][ci] <- w[co][ci][tap]
__global__ __launch_bounds__(256) void small_pack_kernel(const float* w, _Float16* w16, int Cout, int C) {
  const int n = 9*Cout*C;
  for (int i = blockIdx.x*256 + threadIdx.x; i < n; i += gridDim.x*256) {
    const int ci = i % C, co = (i / C) % Cout, t = i / (C*Cout);
    w16[i] = (_Float16)w[((long long)co*C + ci)*9 + t];
  }
}

// ---- x[b][px][c] = out_scale*(x + bias[c] + sum_k W[c][k]*aux[b][k][px]): the 1x1 convolution of
// the 4-channel side branch added into the trunk (AuxiliaryDown, encoder type "skip")
__global__ __launch_bounds__(256) void nhwc_add_pointwise_kernel(const _Float16* x, const float* aux,
                                                                 const float* w, const float* bias,
                                                                 _Float16* y, int C, int Cs, int K,
                                                                 long long HW, float out_scale) {
  const int oct = Cs >> 3;
  const long long b = blockIdx.y;
  const long long n = HW*oct;
  const long long step = (long long)gridDim.x*256;
  if (step % oct == 0 && K <= 4) {
    // A thread keeps its channel octet over the whole loop: its 8 x K weights and 8 biases are loaded ONCE and the
    // pixel index advances by a constant. (Per element the loop below issues ~45 loads -- weights, biases, side-branch
    // values -- and two 64-bit divisions: 320 us for 140 MB at full resolution, 0.44 TB/s.)
    const long long i0 = (long long)blockIdx.x*256 + threadIdx.x;
    const int o = (int)(i0 % oct);
    const long long dpx = step/oct;
    float wr[8][4], br[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = o*8 + j;
      br[j] = (c < C && bias) ? bias[c] : 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) wr[j][k] = (c < C && k < K) ? w[c*K + k] : 0.f;
    }
    const _Float16* xb = x + b*HW*Cs;
    _Float16* yb = y + b*HW*Cs;
    const float* ab = aux + b*K*HW;
    long long px = i0 / oct;
    for (long long i = i0; i < n; i += step, px += dpx) {
      const f32x8 v = __builtin_convertvector(*reinterpret_cast<const h8*>(xb + i*8), f32x8);
      float a[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) a[k] = k < K ? ab[k*HW + px] : 0.f;
      f32x8 r;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float t = v[j] + br[j];
#pragma unroll
        for (int k = 0; k < 4; ++k) t = fmaf(wr[j][k], a[k], t);       // (k >= K: weight and value are zero)
        r[j] = o*8 + j < C ? t*out_scale : 0.f;
      }
      *reinterpret_cast<h8*>(yb + i*8) = __builtin_convertvector(r, h8);
    }
    return;
  }
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < n; i += step) {
    const int o = (int)(i % oct);
    const long long px = i / oct;
    const f32x8 v = __builtin_convertvector(*reinterpret_cast<const h8*>(x + b*HW*Cs + i*8), f32x8);
    float a[8];
    for (int k = 0; k < K && k < 8; ++k) a[k] = aux[(b*K + k)*HW + px];
    f32x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = o*8 + j;
      float t = 0.f;
      if (c < C) {
        t = v[j] + (bias ? bias[c] : 0.f);
        for (int k = 0; k < K && k < 8; ++k) t = fmaf(w[c*K + k], a[k], t);
        t *= out_scale;
      }
      r[j] = t;
    }
    *reinterpret_cast<h8*>(y + b*HW*Cs + i*8) = __builtin_convertvector(r, h8);
  }
}

dim3 nh_grid(long long n, long long b) {
  long long g = (n + 255)/256;
  if (g < 1) g = 1;
  if (g > 4096) g = 4096;
  return dim3((unsigned)g, (unsigned)b);
}

}  // namespace

extern "C" {

int brv_nchw_to_nhwc_f16(const float* x, void* y, int64_t B, int64_t C, int64_t Cs, int64_t HW,
                         brv_stream_t stream) {
  if (B < 1 || C < 1 || Cs < C || (Cs & 7) || HW < 1) return -1;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((unsigned)((HW + 255)/256), (unsigned)(Cs/8), (unsigned)B),
                     dim3(256), 0, (hipStream_t)stream, x, (_Float16*)y, (int)C, (int)Cs, (long long)HW);
  NH_OK(hipGetLastError());
  return 0;
}

int brv_nhwc_f16_to_nchw(const void* x, float* y, int64_t B, int64_t C, int64_t Cs, int64_t HW,
                         brv_stream_t stream) {
  if (B < 1 || C < 1 || Cs < C || (Cs & 7) || HW < 1) return -1;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)((HW + 255)/256), (unsigned)((C + 7)/8), (unsigned)B),
                     dim3(256), 0, (hipStream_t)stream, (const _Float16*)x, y, (int)C, (int)Cs, (long long)HW);
  NH_OK(hipGetLastError());
  return 0;
}

int brv_nhwc_chan_stats(const void* x, double* sums, int64_t B, int64_t C, int64_t Cs, int64_t HW,
                        int64_t c_off, int64_t Ctot, brv_stream_t stream) {
  if (B < 1 || C < 1 || Cs < C || (Cs & 7) || Cs > 2048 || HW < 1 || c_off < 0 || c_off + C > Ctot) return -1;
  const int oct = (int)(Cs/8), lanes = 256/oct;
  long long slice = (long long)lanes*64;            // <= 64 pixels per thread
  const long long min_slice = (HW + 1023)/1024;     // <= 1024 workgroups per item
  if (slice < min_slice) slice = min_slice;
  const unsigned ns = (unsigned)((HW + slice - 1)/slice);
  hipLaunchKernelGGL(chan_stats_kernel, dim3(ns, (unsigned)B), dim3(256), 0, (hipStream_t)stream,
                     (const _Float16*)x, sums, (int)C, (int)Cs, (long long)HW, (int)c_off, (int)Ctot,
                     slice);
  NH_OK(hipGetLastError());
  return 0;
}

int brv_groupnorm_fold_chan(const double* sums, const float* add_bc, const float* gamma,
                            const float* beta, const float* adm_scale, const float* adm_shift,
                            float* scale_bc, float* shift_bc, int64_t B, int64_t C, int64_t HW,
                            int64_t groups, float eps, brv_stream_t stream) {
  if (B < 1 || C < 1 || groups < 1 || C % groups || HW < 1) return -1;
  hipLaunchKernelGGL(chan_fold_kernel, dim3((unsigned)(B*groups)), dim3(64), 0, (hipStream_t)stream,
                     sums, (const double*)nullptr, (int)C, add_bc, gamma, beta, adm_scale, adm_shift,
                     scale_bc, shift_bc, (int)C, (long long)HW, (int)groups, eps);
  NH_OK(hipGetLastError());
  return 0;
}

int brv_groupnorm_fold_chan2(const double* sums1, int64_t C1, const double* sums2, int64_t C2,
                             const float* add_bc, const float* gamma, const float* beta,
                             const float* adm_scale, const float* adm_shift, float* scale_bc,
                             float* shift_bc, int64_t B, int64_t HW, int64_t groups, float eps,
                             brv_stream_t stream) {
  const int64_t C = C1 + (sums2 ? C2 : 0);
  if (B < 1 || C1 < 1 || groups < 1 || C % groups || HW < 1) return -1;
  hipLaunchKernelGGL(chan_fold_kernel, dim3((unsigned)(B*groups)), dim3(64), 0, (hipStream_t)stream,
                     sums1, sums2, (int)C1, add_bc, gamma, beta, adm_scale, adm_shift, scale_bc,
                     shift_bc, (int)C, (long long)HW, (int)groups, eps);
  NH_OK(hipGetLastError());
  return 0;
}

int brv_nhwc_affine_act(const void* x, const float* scale_bc, const float* shift_bc, void* y,
                        int64_t B, int64_t C, int64_t Cs, int64_t HW, int act, brv_stream_t stream) {
  if (B < 1 || C < 1 || Cs < C || (Cs & 7) || HW < 1) return -1;
  hipLaunchKernelGGL(nhwc_affine_act_kernel, nh_grid(HW*(Cs/8), B), dim3(256), 0, (hipStream_t)stream,
                     (const _Float16*)x, scale_bc, shift_bc, (_Float16*)y, (int)C, (int)Cs,
                     (long long)HW, act);
  NH_OK(hipGetLastError());
  return 0;
}

int brv_nhwc_fir_resample2d(const void* x, const float* kernel, void* y, int64_t B, int64_t Cs,
                            int64_t H, int64_t W, int64_t Ho, int64_t Wo, int64_t K, int64_t pad_h,
                            int64_t pad_w, int up, float gain, brv_stream_t stream) {
  if (B < 1 || (Cs & 7) || Cs < 8 || H < 1 || W < 1 || Ho < 1 || Wo < 1 || K < 1) return -1;
  const dim3 grid = nh_grid(Ho*Wo*(Cs/8), B);
  if (up)
    hipLaunchKernelGGL(nhwc_fir_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream,
                       (const _Float16*)x, kernel, (_Float16*)y, (int)Cs, (int)H, (int)W, (int)Ho,
                       (int)Wo, (int)K, (int)pad_h, (int)pad_w, gain);
  else
    hipLaunchKernelGGL(nhwc_fir_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream,
                       (const _Float16*)x, kernel, (_Float16*)y, (int)Cs, (int)H, (int)W, (int)Ho,
                       (int)Wo, (int)K, (int)pad_h, (int)pad_w, gain);
  NH_OK(hipGetLastError());
  return 0;
}

int brv_nhwc_fir_resample2d_dual(const void* x, const float* scale_bc, const float* shift_bc, int act,
                                 const float* kernel, void* y_raw, void* y_act, int64_t B, int64_t C,
                                 int64_t Cs, int64_t H, int64_t W, int64_t Ho, int64_t Wo, int64_t K,
                                 int64_t pad_h, int64_t pad_w, int up, float gain, brv_stream_t stream) {
  if (B < 1 || (Cs & 7) || Cs < 8 || C < 1 || C > Cs || H < 1 || W < 1 || Ho < 1 || Wo < 1 || K < 1 ||
      K > 4 || B > 65535 || !scale_bc || !shift_bc)
    return -1;
  FirDualParams p;
  p.x = (const _Float16*)x; p.scale = scale_bc; p.shift = shift_bc; p.k = kernel;
  p.y_raw = (_Float16*)y_raw; p.y_act = (_Float16*)y_act;
  p.C = (int)C; p.Cs = (int)Cs; p.H = (int)H; p.W = (int)W; p.Ho = (int)Ho; p.Wo = (int)Wo; p.K = (int)K;
  p.ph = (int)pad_h; p.pw = (int)pad_w; p.act = act; p.gain = gain;
  const int TH = up ? 16 : 8, TW = up ? 32 : 16;
  p.tiles_w = (int)((Wo + TW - 1)/TW);
  const dim3 grid((unsigned)(p.tiles_w*((Ho + TH - 1)/TH)), (unsigned)((Cs/8 + 3)/4), (unsigned)B);
  if (up) hipLaunchKernelGGL(nhwc_fir_dual_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(nhwc_fir_dual_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, p);
  NH_OK(hipGetLastError());
  return 0;
}

int brv_nhwc_axpby(const void* a, float alpha, const void* b, float beta, void* out, int64_t n,
                   brv_stream_t stream) {
  if (n < 0 || (n & 7)) return -1;
  if (n == 0) return 0;
  hipLaunchKernelGGL(nhwc_axpby_kernel, nh_grid(n/8, 1), dim3(256), 0, (hipStream_t)stream,
                     (const _Float16*)a, alpha, (const _Float16*)b, beta, (_Float16*)out,
                     (long long)(n/8));
  NH_OK(hipGetLastError());
  return 0;
}

int64_t brv_nhwc_conv1x1_packed_size(int64_t Cout, int64_t C1, int64_t C2) {
  if (Cout < 1 || C1 < 1 || C2 < 0) return -1;
  return ((Cout + 127)/128)*128*(((C1 + 15)/16) + ((C2 + 15)/16))*16;
}

int brv_nhwc_conv1x1_pack(const float* w, void* wp, int64_t Cout, int64_t C1, int64_t C2,
                          brv_stream_t stream) {
  const int64_t total = brv_nhwc_conv1x1_packed_size(Cout, C1, C2);
  if (total < 0) return -1;
  long long g = (total + 255)/256;
  if (g > 4096) g = 4096;
  const int n_ks1 = (int)((C1 + 15)/16), n_ks = n_ks1 + (int)((C2 + 15)/16);
  hipLaunchKernelGGL(conv1x1_pack_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w,
                     (_Float16*)wp, (int)Cout, (int)C1, (int)C2, n_ks1, n_ks, (long long)total);
  NH_OK(hipGetLastError());
  return 0;
}

int brv_nhwc_conv1x1_forward(const void* x1, int64_t C1, int64_t C1s, const void* x2, int64_t C2,
                             int64_t C2s, const void* wp, const float* bias, void* y, int64_t Cys,
                             int64_t npx, int64_t Cout, float out_scale, brv_stream_t stream) {
  if (npx < 1 || C1 < 1 || Cout < 1 || (C1s & 7) || C1 > C1s || (Cys & 7) || Cout > Cys) return -1;
  if (x2 && (C2 < 1 || (C2s & 7) || C2 > C2s)) return -1;
  Pw1Params p;
  p.x1 = (const _Float16*)x1; p.x2 = (const _Float16*)x2; p.wp = (const h8*)wp; p.bias = bias;
  p.y = (_Float16*)y; p.C1 = (int)C1; p.C1s = (int)C1s; p.C2 = x2 ? (int)C2 : 0; p.C2s = (int)C2s;
  p.Cout = (int)Cout; p.Cys = (int)Cys;
  p.n_ks1 = (int)((C1 + 15)/16); p.n_ks = p.n_ks1 + (x2 ? (int)((C2 + 15)/16) : 0);
  p.npx = npx; p.out_scale = out_scale;
#ifndef BRV_PW1_NB
#define BRV_PW1_NB 1           // diagnostic builds: 2 = both 128-channel blocks of a pixel tile in one workgroup
#endif
  const unsigned n_cob = (unsigned)((Cout + 127)/128);
  const unsigned n_pt = (unsigned)((npx + 127)/128);
#ifndef BRV_PW1_XCD
#define BRV_PW1_XCD 1          // diagnostic builds: 0 = (pixel tile, output block) grid of round 4
#endif
#ifndef BRV_PW1_LDS
#define BRV_PW1_LDS 1          // diagnostic builds: 0 = every wave fetches its own weight fragments (round 4)
#endif
  // (input through LDS too from 256 input channels on: 99 / 126 / 329 against 106 / 144 / 355 us at 128 + 128 -> 128,
  // 256 + 128 -> 128, 256 + 256 -> 256 channels and 8 x 256 x 251 pixels; 143 against 136 us at 128 -> 256)
  if (BRV_PW1_LDS == 2 || (BRV_PW1_LDS == 1 && p.n_ks >= 16)) {
    if (n_cob >= 2) hipLaunchKernelGGL(nhwc_conv1x1_lds2_kernel, dim3((n_pt + 7)/8*8*n_cob), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(nhwc_conv1x1_lds2_kernel, dim3(n_pt, 1), dim3(256), 0, (hipStream_t)stream, p);
  } else if (BRV_PW1_LDS) {
    if (n_cob >= 2) hipLaunchKernelGGL(nhwc_conv1x1_lds_kernel, dim3((n_pt + 7)/8*8*n_cob), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(nhwc_conv1x1_lds_kernel, dim3(n_pt, 1), dim3(256), 0, (hipStream_t)stream, p);
  } else if (BRV_PW1_NB == 2 && n_cob >= 2)
    hipLaunchKernelGGL(nhwc_conv1x1_kernel<2>, dim3(n_pt, (n_cob + 1)/2), dim3(256), 0, (hipStream_t)stream, p);
  else if (BRV_PW1_XCD && n_cob >= 2)
    hipLaunchKernelGGL(nhwc_conv1x1_kernel<1>, dim3((n_pt + 7)/8*8*n_cob), dim3(256), 0, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL(nhwc_conv1x1_kernel<1>, dim3(n_pt, n_cob), dim3(256), 0, (hipStream_t)stream, p);
  NH_OK(hipGetLastError());
  return 0;
}

int brv_nhwc_conv3x3_small_pack(const float* w, void* w16, int64_t Cout, int64_t C, brv_stream_t stream) {
  if (Cout < 1 || Cout > 8 || C < 1) return -1;
  hipLaunchKernelGGL(small_pack_kernel, dim3((unsigned)((9*Cout*C + 255)/256)), dim3(256), 0,
                     (hipStream_t)stream, w, (_Float16*)w16, (int)Cout, (int)C);
  NH_OK(hipGetLastError());
  return 0;
}

int brv_nhwc_conv3x3_small(const void* x, const void* w16, const float* bias, const float* scale_bc,
                           const float* shift_bc, int silu, const float* y_in, float* y, int64_t B,
                           int64_t C, int64_t Cs, int64_t H, int64_t W, int64_t Cout,
                           brv_stream_t stream) {
  if (B < 1 || C < 1 || (C & 7) || Cs < C || (Cs & 7) || H < 1 || W < 1 || Cout < 1 || Cout > 8) return -1;
  SmallCoutParams p;
  p.x = (const _Float16*)x; p.w16 = (const _Float16*)w16; p.bias = bias; p.scale = scale_bc;
  p.shift = shift_bc; p.yin = y_in; p.y = y; p.C = (int)C; p.Cs = (int)Cs; p.Cout = (int)Cout;
  p.H = (int)H; p.W = (int)W; p.silu = silu;
  const dim3 grid((unsigned)((W + 31)/32), (unsigned)((H + 7)/8), (unsigned)B);
  if (Cout <= 4)
    hipLaunchKernelGGL(nhwc_conv3x3_small_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL(nhwc_conv3x3_small_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, p);
  NH_OK(hipGetLastError());
  return 0;
}

int brv_nhwc_add_pointwise(const void* x, const float* aux, const float* w, const float* bias,
                           void* y, int64_t B, int64_t C, int64_t Cs, int64_t K, int64_t HW,
                           float out_scale, brv_stream_t stream) {
  if (B < 1 || C < 1 || Cs < C || (Cs & 7) || K < 1 || K > 8 || HW < 1) return -1;
  // (eight elements per thread: the per-thread weights and biases are loaded once for them; with one element per
  // thread -- nh_grid -- they were loaded per element)
  long long gx = (HW*(Cs/8) + 2047)/2048;
  if (gx < 1) gx = 1;
  if (gx > 4096) gx = 4096;
  hipLaunchKernelGGL(nhwc_add_pointwise_kernel, dim3((unsigned)gx, (unsigned)B), dim3(256), 0, (hipStream_t)stream,
                     (const _Float16*)x, aux, w, bias, (_Float16*)y, (int)C, (int)Cs, (int)K,
                     (long long)HW, out_scale);
  NH_OK(hipGetLastError());
  return 0;
}

}  // extern "C"
