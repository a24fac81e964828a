// DCCRN building blocks, forward values (inference / validation path), fp32.
// Reference: brever/models/dccrn/dccrn.py:28-358 (DCCRN, DCCRNMaskNet, ComplexWrapper,
// EncoderBlock, DecoderBlock, LSTMBlock, ComplexLSTM, SingleLayerComplexLSTM) on top of
// torch's Conv2d / ConvTranspose2d / BatchNorm2d / PReLU / LSTM / Linear.
// Correctness-first direct kernels (one thread per output element); the dense pieces that
// matter for speed (STFT / iSTFT, Linear) are the DFT / fp32 GEMM kernels of stft.hip.
// Layout: NCHW as in the reference, (batch, channels, freqs, frames).
#include <hip/hip_runtime.h>
#include <math.h>

#include "../../include/brever_hip.h"
#include "common.cuh"

using namespace brv;

namespace {

#define DC_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return (int)e_; } while (0)

dim3 flat_grid(long long n) {
  long long g = (n + 255)/256;
  if (g < 1) g = 1;
  if (g > 8192) g = 8192;
  return dim3((unsigned)g);
}
#define GRID_STRIDE(i, n) \
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < (n); i += (long long)gridDim.x*256)

struct ConvGeom {
  int B, Cin, H, W, Cout, Ho, Wo, kh, kw, sh, sw, ph, pw;
  long long x_bs, y_bs;      // batch strides in elements (views of wider tensors)
};

// y[b][co][ho][wo] (+)= sign*(bias[co] + sum x[b][ci][ho*sh - ph + i][wo*sw - pw + j] w[co][ci][i][j])
__global__ __launch_bounds__(256) void conv2d_fwd_kernel(const float* x, const float* w,
                                                         const float* bias, float* y, ConvGeom g,
                                                         int accumulate, float sign) {
  const long long total = (long long)g.B*g.Cout*g.Ho*g.Wo;
  GRID_STRIDE(idx, total) {
    const int wo = (int)(idx % g.Wo);
    long long r = idx / g.Wo;
    const int ho = (int)(r % g.Ho); r /= g.Ho;
    const int co = (int)(r % g.Cout);
    const int b = (int)(r / g.Cout);
    float acc = bias ? bias[co] : 0.f;
    const float* xb = x + (long long)b*g.x_bs;
    for (int ci = 0; ci < g.Cin; ++ci) {
      const float* xc = xb + (long long)ci*g.H*g.W;
      const float* wc = w + ((long long)co*g.Cin + ci)*g.kh*g.kw;
      for (int i = 0; i < g.kh; ++i) {
        const int hi = ho*g.sh - g.ph + i;
        if (hi < 0 || hi >= g.H) continue;
        for (int j = 0; j < g.kw; ++j) {
          const int wi = wo*g.sw - g.pw + j;
          if (wi < 0 || wi >= g.W) continue;
          acc += xc[(long long)hi*g.W + wi]*wc[i*g.kw + j];
        }
      }
    }
    float* dst = y + (long long)b*g.y_bs + ((long long)co*g.Ho + ho)*g.Wo + wo;
    *dst = accumulate ? *dst + sign*acc : sign*acc;
  }
}

// ConvTranspose2d (weight [Cin][Cout][kh][kw]): gather form
// y[b][co][ho][wo] (+)= sign*(bias[co] + sum_{ci,i,j : ho = hi*sh - ph + i, wo = wi*sw - pw + j} x[b][ci][hi][wi] w[ci][co][i][j])
__global__ __launch_bounds__(256) void conv_transpose2d_fwd_kernel(const float* x, const float* w,
                                                                   const float* bias, float* y,
                                                                   ConvGeom g, int accumulate,
                                                                   float sign) {
  const long long total = (long long)g.B*g.Cout*g.Ho*g.Wo;
  GRID_STRIDE(idx, total) {
    const int wo = (int)(idx % g.Wo);
    long long r = idx / g.Wo;
    const int ho = (int)(r % g.Ho); r /= g.Ho;
    const int co = (int)(r % g.Cout);
    const int b = (int)(r / g.Cout);
    float acc = bias ? bias[co] : 0.f;
    const float* xb = x + (long long)b*g.x_bs;
    for (int i = 0; i < g.kh; ++i) {
      const int hn = ho + g.ph - i;
      if (hn < 0 || hn % g.sh) continue;
      const int hi = hn/g.sh;
      if (hi >= g.H) continue;
      for (int j = 0; j < g.kw; ++j) {
        const int wn = wo + g.pw - j;
        if (wn < 0 || wn % g.sw) continue;
        const int wi = wn/g.sw;
        if (wi >= g.W) continue;
        for (int ci = 0; ci < g.Cin; ++ci)
          acc += xb[((long long)ci*g.H + hi)*g.W + wi]
                 *w[(((long long)ci*g.Cout + co)*g.kh + i)*g.kw + j];
      }
    }
    float* dst = y + (long long)b*g.y_bs + ((long long)co*g.Ho + ho)*g.Wo + wo;
    *dst = accumulate ? *dst + sign*acc : sign*acc;
  }
}

// BatchNorm2d statistics: one workgroup per channel (biased variance for the normalisation,
// unbiased for the running estimate, as torch)
// Per-channel reductions over (B, HW) run as gridDim.y slices per channel whose fp64 partial
// sums land in a scratch (C, slices, K) and are combined in a fixed order by a finishing
// kernel: deterministic, and C = 16..512 channels alone would leave most of the chip idle.
// (TO = bf16_t: the output as the bf16 values the use_amp row convolutions round it to anyway -- its only readers on
// that path, csrc/cconv.hip -- at half the bytes)
__device__ __forceinline__ float4 load4(const float* base, long long i) { return reinterpret_cast<const float4*>(base)[i]; }
__device__ __forceinline__ float4 load4(const bf16_t* base, long long i) {
  const uint2 q = reinterpret_cast<const uint2*>(base)[i];
  return make_float4(__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xffff0000u), __uint_as_float(q.y << 16),
                     __uint_as_float(q.y & 0xffff0000u));
}
__device__ __forceinline__ void store4(float* base, long long i, float4 v) { reinterpret_cast<float4*>(base)[i] = v; }
__device__ __forceinline__ void store4(bf16_t* base, long long i, float4 v) {
  reinterpret_cast<uint2*>(base)[i] = make_uint2(pack2(v.x, v.y), pack2(v.z, v.w));
}
constexpr int kRedSlices = 64;
__device__ __forceinline__ void slice_range(long long n, long long& lo, long long& hi) {
  const long long per = (n + gridDim.y - 1)/gridDim.y;
  lo = (long long)blockIdx.y*per;
  hi = lo + per < n ? lo + per : n;
}
__global__ __launch_bounds__(256) void bn_stats_part_kernel(const float* x, int B, int C,
                                                            long long HW, double* part) {
  __shared__ double scr[8];
  const int c = blockIdx.x;
  long long lo, hi;
  slice_range((long long)B*HW, lo, hi);
  double s = 0.0, q = 0.0;
  for (long long i = lo + threadIdx.x; i < hi; i += 256) {
    const float v = x[((long long)(i / HW)*C + c)*HW + i % HW];
    s += v; q += (double)v*v;
  }
  s = block_sum(s, scr);
  __syncthreads();
  q = block_sum(q, scr);
  if (threadIdx.x == 0) {
    part[((long long)c*gridDim.y + blockIdx.y)*2] = s;
    part[((long long)c*gridDim.y + blockIdx.y)*2 + 1] = q;
  }
}
// the same sums when HW % 4 == 0 (every DCCRN shape): blockIdx.y = (batch item, piece of the plane) so that a block
// reads one contiguous run with 16-byte loads -- no 64-bit division per element; four elements meet in fp32, then
// join the fp64 sum (44 -> 27 us on a 131 MB tensor)
template <typename TI>
__global__ __launch_bounds__(256) void bn_stats_part4_kernel(const TI* x, int C, long long HW, int pieces,
                                                             double* part) {
  __shared__ double scr[8];
  const int c = blockIdx.x, b = blockIdx.y / pieces, pc = blockIdx.y % pieces;
  const long long n4 = HW >> 2, per = (n4 + pieces - 1)/pieces;
  const long long lo = pc*per, hi = lo + per < n4 ? lo + per : n4;
  const TI* src = x + ((long long)b*C + c)*HW;
  double s = 0.0, q = 0.0;
  for (long long i = lo + threadIdx.x; i < hi; i += 256) {
    const float4 v = load4(src, i);
    s += (double)((v.x + v.y) + (v.z + v.w));
    q += (double)((v.x*v.x + v.y*v.y) + (v.z*v.z + v.w*v.w));
  }
  s = block_sum(s, scr);
  __syncthreads();
  q = block_sum(q, scr);
  if (threadIdx.x == 0) {
    part[((long long)c*gridDim.y + blockIdx.y)*2] = s;
    part[((long long)c*gridDim.y + blockIdx.y)*2 + 1] = q;
  }
}
__global__ __launch_bounds__(256) void bn_stats_kernel(const double* part, int slices, int B, int C,
                                                       long long HW, float* mean_out,
                                                       float* invstd_out, float* running_mean,
                                                       float* running_var, float eps,
                                                       float momentum) {
  const int c = blockIdx.x*256 + threadIdx.x;
  if (c >= C) return;
  const long long n = (long long)B*HW;
  double s = 0.0, q = 0.0;
  for (int i = 0; i < slices; ++i) { s += part[((long long)c*slices + i)*2]; q += part[((long long)c*slices + i)*2 + 1]; }
  {
    const double mean = s/n;
    double var = q/n - mean*mean;
    if (var < 0) var = 0;
    mean_out[c] = (float)mean;
    invstd_out[c] = (float)(1.0/sqrt(var + eps));
    if (running_mean) {
      const double unbiased = n > 1 ? var*n/(n - 1) : var;
      running_mean[c] = (1.f - momentum)*running_mean[c] + momentum*(float)mean;
      running_var[c] = (1.f - momentum)*running_var[c] + momentum*(float)unbiased;
    }
  }
}
// y = (x - mean[c])*invstd[c]*gamma[c] + beta[c], then optional PReLU (scalar slope)
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* x, const float* mean,
                                                       const float* invstd, const float* gamma,
                                                       const float* beta, const float* slope,
                                                       float* y, int C, long long HW, long long total) {
  const float a = slope ? *slope : 1.f;
  GRID_STRIDE(i, total) {
    const int c = (int)((i / HW) % C);
    float v = (x[i] - mean[c])*invstd[c]*gamma[c] + beta[c];
    if (slope) v = v > 0.f ? v : a*v;
    y[i] = v;
  }
}
// the two element-wise passes when HW % 4 == 0: blockIdx = (piece of the plane, channel, batch item): the channel
// scalars are loaded once per block, the plane is walked with 16-byte accesses, no division per element
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void bn_apply4_kernel(const TI* x, const float* mean, const float* invstd,
                                                        const float* gamma, const float* beta, const float* slope,
                                                        TO* y, int C, long long HW) {
  const int c = blockIdx.y, b = blockIdx.z;
  const float a = slope ? *slope : 1.f;
  const long long n4 = HW >> 2, base = ((long long)b*C + c)*HW;
  const TI* src = x + base;
  TO* dst = y + base;
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < n4; i += (long long)gridDim.x*256) {
    float4 v = load4(src, i);
    // (x - mean)*invstd*gamma + beta with the reference's rounding order kept: two multiplies, then the add
    v.x = (v.x - mean[c])*invstd[c]*gamma[c] + beta[c]; v.y = (v.y - mean[c])*invstd[c]*gamma[c] + beta[c];
    v.z = (v.z - mean[c])*invstd[c]*gamma[c] + beta[c]; v.w = (v.w - mean[c])*invstd[c]*gamma[c] + beta[c];
    if (slope) {
      v.x = v.x > 0.f ? v.x : a*v.x; v.y = v.y > 0.f ? v.y : a*v.y;
      v.z = v.z > 0.f ? v.z : a*v.z; v.w = v.w > 0.f ? v.w : a*v.w;
    }
    store4(dst, i, v);
  }
}
__global__ __launch_bounds__(256) void invstd_from_var_kernel(const float* var, float* invstd,
                                                              int C, float eps) {
  const int c = blockIdx.x*256 + threadIdx.x;
  if (c < C) invstd[c] = 1.f/sqrtf(var[c] + eps);
}

// ---- complex batch norm (models/dccrn/complex_batchnorm.py) as three pieces ------------------
// x (B, 2C, HW), real half first. (1) per complex channel the five means E[xr], E[xi], E[xr^2],
// E[xi^2], E[xr xi] over (B, HW) (sliced fp64 partial sums); the 2x2 whitening and the affine
// map are a handful of per-channel scalars computed by the caller; (2) y = A x + o per channel
// with A (4, C) = [a_rr, a_ri, a_ir, a_ii] and o (2, C), then the optional scalar PReLU;
// (3) the adjoints of both.
__global__ __launch_bounds__(256) void cplx_moments_part_kernel(const float* x, int B, int C,
                                                                long long HW, double* part) {
  __shared__ double scr[8];
  const int c = blockIdx.x;
  long long lo, hi;
  slice_range((long long)B*HW, lo, hi);
  double s[5] = {0, 0, 0, 0, 0};
  for (long long e = lo + threadIdx.x; e < hi; e += 256) {
    const long long b = e / HW, i = e % HW;
    const float xr = x[((b*2*C) + c)*HW + i], xi = x[((b*2*C) + C + c)*HW + i];
    s[0] += xr; s[1] += xi; s[2] += (double)xr*xr; s[3] += (double)xi*xi; s[4] += (double)xr*xi;
  }
  for (int k = 0; k < 5; ++k) {
    const double v = block_sum(s[k], scr);
    __syncthreads();
    if (threadIdx.x == 0) part[((long long)c*gridDim.y + blockIdx.y)*5 + k] = v;
  }
}
__global__ __launch_bounds__(256) void slice_final_kernel(const double* part, int slices, int K,
                                                          int C, double inv_n, float* out) {
  const int idx = blockIdx.x*256 + threadIdx.x;          // (k, c) -> out[k][c]
  if (idx >= K*C) return;
  const int k = idx / C, c = idx % C;
  double s = 0.0;
  for (int i = 0; i < slices; ++i) s += part[((long long)c*slices + i)*K + k];
  out[idx] = (float)(s*inv_n);
}
__global__ __launch_bounds__(256) void cplx_affine_fwd_kernel(const float* x, const float* A,
                                                              const float* o, const float* slope,
                                                              float* y, int C, long long HW,
                                                              long long total) {
  const float a = slope ? *slope : 1.f;
  GRID_STRIDE(idx, total) {                       // idx over (b, c, i) of the REAL half
    const long long i = idx % HW, bc = idx / HW;
    const int c = (int)(bc % C); const long long b = bc / C;
    const long long pr = ((b*2*C) + c)*HW + i, pi = pr + (long long)C*HW;
    const float xr = x[pr], xi = x[pi];
    float yr = A[c]*xr + A[C + c]*xi + o[c];
    float yi = A[2*C + c]*xr + A[3*C + c]*xi + o[C + c];
    if (slope) { yr = yr > 0.f ? yr : a*yr; yi = yi > 0.f ? yi : a*yi; }
    y[pr] = yr; y[pi] = yi;
  }
}
// partial sums per channel: [u_r xr, u_r xi, u_i xr, u_i xi, u_r, u_i, slope grad], u = dy*prelu'
__global__ __launch_bounds__(256) void cplx_affine_bwd_part_kernel(const float* x, const float* dy,
                                                                   const float* A, const float* o,
                                                                   const float* slope, int B, int C,
                                                                   long long HW, double* part) {
  __shared__ double scr[8];
  const int c = blockIdx.x;
  const float a = slope ? *slope : 1.f;
  long long lo, hi;
  slice_range((long long)B*HW, lo, hi);
  double s[7] = {0, 0, 0, 0, 0, 0, 0};
  for (long long e = lo + threadIdx.x; e < hi; e += 256) {
    const long long b = e / HW, i = e % HW;
    const long long pr = ((b*2*C) + c)*HW + i, pi = pr + (long long)C*HW;
    const float xr = x[pr], xi = x[pi];
    float ur = dy[pr], ui = dy[pi];
    if (slope) {
      const float yr = A[c]*xr + A[C + c]*xi + o[c];
      const float yi = A[2*C + c]*xr + A[3*C + c]*xi + o[C + c];
      if (yr <= 0.f) { s[6] += (double)ur*yr; ur *= a; }
      if (yi <= 0.f) { s[6] += (double)ui*yi; ui *= a; }
    }
    s[0] += (double)ur*xr; s[1] += (double)ur*xi; s[2] += (double)ui*xr; s[3] += (double)ui*xi;
    s[4] += ur; s[5] += ui;
  }
  for (int k = 0; k < 7; ++k) {
    const double v = block_sum(s[k], scr);
    __syncthreads();
    if (threadIdx.x == 0) part[((long long)c*gridDim.y + blockIdx.y)*7 + k] = v;
  }
}
// dx = A^T u (+ the gradient that flows through the five means, gm (5, C), already divided by
// the number of elements per channel)
__global__ __launch_bounds__(256) void cplx_affine_bwd_apply_kernel(const float* x, const float* dy,
                                                                    const float* A, const float* o,
                                                                    const float* slope,
                                                                    const float* gm, float* dx,
                                                                    int C, long long HW,
                                                                    long long total) {
  const float a = slope ? *slope : 1.f;
  GRID_STRIDE(idx, total) {
    const long long i = idx % HW, bc = idx / HW;
    const int c = (int)(bc % C); const long long b = bc / C;
    const long long pr = ((b*2*C) + c)*HW + i, pi = pr + (long long)C*HW;
    const float xr = x[pr], xi = x[pi];
    float dr = 0.f, di = 0.f;
    if (dy) {
      float ur = dy[pr], ui = dy[pi];
      if (slope) {
        const float yr = A[c]*xr + A[C + c]*xi + o[c];
        const float yi = A[2*C + c]*xr + A[3*C + c]*xi + o[C + c];
        if (yr <= 0.f) ur *= a;
        if (yi <= 0.f) ui *= a;
      }
      dr = A[c]*ur + A[2*C + c]*ui;
      di = A[C + c]*ur + A[3*C + c]*ui;
    }
    if (gm) {
      dr += gm[c] + 2.f*xr*gm[2*C + c] + xi*gm[4*C + c];
      di += gm[C + c] + 2.f*xi*gm[3*C + c] + xr*gm[4*C + c];
    }
    dx[pr] = dr; dx[pi] = di;
  }
}

// LSTM recurrence for one layer (torch gate order i, f, g, o): gates_in = W_ih x precomputed
// for all steps, bias = b_ih + b_hh; one workgroup per batch item, h and c in LDS.
__global__ __launch_bounds__(256) void lstm_recurrent_kernel(const float* gates_in, const float* w_hh,
                                                             const float* bias, float* y, float* act,
                                                             float* cs, int T, int H, int per_group) {
  extern __shared__ float sm[];                   // h[H] | c[H] | gates[4H]
  float* h = sm; float* c = sm + H; float* gt = sm + 2*H;
  const int b = blockIdx.x;
  w_hh += (long long)(b / per_group)*4*H*H;        // the item's parameter group
  if (bias) bias += (long long)(b / per_group)*4*H;
  for (int i = threadIdx.x; i < H; i += 256) { h[i] = 0.f; c[i] = 0.f; }
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    const float* gi = gates_in + ((long long)b*T + t)*4*H;
    for (int r = threadIdx.x; r < 4*H; r += 256) {
      float acc = gi[r] + (bias ? bias[r] : 0.f);
      const float* wr = w_hh + (long long)r*H;
      for (int k = 0; k < H; ++k) acc += wr[k]*h[k];
      gt[r] = acc;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < H; i += 256) {
      const float ig = 1.f/(1.f + expf(-gt[i]));
      const float fg = 1.f/(1.f + expf(-gt[H + i]));
      const float gg = tanhf(gt[2*H + i]);
      const float og = 1.f/(1.f + expf(-gt[3*H + i]));
      const float cn = fg*c[i] + ig*gg;
      c[i] = cn;
      const float hn = og*tanhf(cn);
      h[i] = hn;
      y[((long long)b*T + t)*H + i] = hn;
      if (act) {                                  // saved for the backward pass
        float* a = act + ((long long)b*T + t)*4*H;
        a[i] = ig; a[H + i] = fg; a[2*H + i] = gg; a[3*H + i] = og;
        cs[((long long)b*T + t)*H + i] = cn;
      }
    }
    __syncthreads();
  }
}

// The same recurrence with the recurrent weights resident in REGISTERS (H <= 128, H % 16 == 0):
// 4H threads, thread r keeps row r of W_hh (H floats) for all T steps, h lives in LDS and is
// read as broadcast float4; the cell state stays in the register of the thread that owns the
// unit. One step = one H-long dot product per thread + the gate math: the chain of T dependent
// steps, not bandwidth, is what this kernel is bound by.
constexpr int kLstmRegH = 128;
// gate non-linearities on the hardware exp / rcp (v_exp_f32, v_rcp_f32, ~1 ulp each): the libm
// forms are a large share of a recurrence step, which is latency- not throughput-bound
__device__ __forceinline__ float fast_sigm(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) {
  return 2.f*__builtin_amdgcn_rcpf(1.f + __expf(-2.f*x)) - 1.f;
}
// Workgroup barrier that orders LDS traffic only. `__syncthreads()` also waits for every outstanding
// global store and load of the wave (s_waitcnt vmcnt(0)): inside a recurrence that puts one HBM round trip
// (the step's y / activation stores, the prefetch of the next step's input) into EVERY time step. The
// outputs of a step are not read by this kernel again, so only the LDS hand-over needs the barrier.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__global__ __launch_bounds__(512) void lstm_fwd_reg_kernel(const float* gates_in, const float* w_hh,
                                                           const float* bias, float* y, float* act,
                                                           float* cs, int T, int H, int per_group) {
  __shared__ __attribute__((aligned(16))) float h[kLstmRegH];
  __shared__ float gt[4*kLstmRegH];
  const int b = blockIdx.x, r = threadIdx.x;           // blockDim.x == 4H
  w_hh += (long long)(b / per_group)*4*H*H;            // the item's parameter group
  if (bias) bias += (long long)(b / per_group)*4*H;
  float w[kLstmRegH];
#pragma unroll
  for (int k = 0; k < kLstmRegH; ++k) w[k] = k < H ? w_hh[(long long)r*H + k] : 0.f;
  const float br = bias ? bias[r] : 0.f;
  if (r < kLstmRegH) h[r] = 0.f;
  float c = 0.f;
  const float* gi = gates_in + (long long)b*T*4*H + r;
  float g_next = gi[0];
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    float acc = g_next + br;
    if (t + 1 < T) g_next = gi[(long long)(t + 1)*4*H];
#pragma unroll
    for (int k = 0; k < kLstmRegH; k += 4) {
      if (k < H) {
        const float4 hv = *reinterpret_cast<const float4*>(h + k);
        acc += w[k]*hv.x + w[k + 1]*hv.y + w[k + 2]*hv.z + w[k + 3]*hv.w;
      }
    }
    gt[r] = acc;
    lds_barrier();
    if (r < H) {
      const float ig = fast_sigm(gt[r]);
      const float fg = fast_sigm(gt[H + r]);
      const float gg = fast_tanh(gt[2*H + r]);
      const float og = fast_sigm(gt[3*H + r]);
      c = fg*c + ig*gg;
      const float hn = og*fast_tanh(c);
      h[r] = hn;
      y[((long long)b*T + t)*H + r] = hn;
      if (act) {
        float* a = act + ((long long)b*T + t)*4*H;
        a[r] = ig; a[H + r] = fg; a[2*H + r] = gg; a[3*H + r] = og;
        cs[((long long)b*T + t)*H + r] = c;
      }
    }
    lds_barrier();
  }
}

// Backward through time with W_hh^T resident in registers: thread j owns hidden unit k = j % H
// and the quarter q = j / H of the gate rows, i.e. W_hh[qH + i][k] for i < H; the four partial
// sums of dh_{t-1}[k] meet in LDS.
__global__ __launch_bounds__(512) void lstm_bwd_reg_kernel(const float* act, const float* cs,
                                                           const float* w_hh, const float* dy,
                                                           float* dgates, int T, int H,
                                                           int per_group) {
  __shared__ __attribute__((aligned(16))) float dg[4*kLstmRegH];
  __shared__ float part[4][kLstmRegH];
  const int b = blockIdx.x, j = threadIdx.x;
  w_hh += (long long)(b / per_group)*4*H*H;
  const int k = j % H, q = j / H;
  float w[kLstmRegH];
#pragma unroll
  for (int i = 0; i < kLstmRegH; ++i) w[i] = i < H ? w_hh[(long long)(q*H + i)*H + k] : 0.f;
  part[q][k] = 0.f;
  float dc = 0.f;
  // the saved activations are requested TWO steps ahead (step t needs c of step t - 1 as well): they used
  // to open every step with a dependent HBM round trip
  struct Saved { float ig, fg, gg, og, c, dy; };
  auto fetch = [&](int t) {
    Saved v = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (j < H && t >= 0) {
      const float* a = act + ((long long)b*T + t)*4*H;
      v.ig = a[j]; v.fg = a[H + j]; v.gg = a[2*H + j]; v.og = a[3*H + j];
      v.c = cs[((long long)b*T + t)*H + j];
      v.dy = dy[((long long)b*T + t)*H + j];
    }
    return v;
  };
  Saved cur = fetch(T - 1), nxt = fetch(T - 2);
  __syncthreads();
  for (int t = T - 1; t >= 0; --t) {
    const Saved far = fetch(t - 2);
    if (j < H) {
      const float ig = cur.ig, fg = cur.fg, gg = cur.gg, og = cur.og, c = cur.c, dyt = cur.dy;
      const float cprev = t > 0 ? nxt.c : 0.f;
      const float tc = fast_tanh(c);
      const float dht = part[0][j] + part[1][j] + part[2][j] + part[3][j] + dyt;
      const float dct = dc + dht*og*(1.f - tc*tc);
      const float d0 = dct*gg*ig*(1.f - ig), d1 = dct*cprev*fg*(1.f - fg);
      const float d2 = dct*ig*(1.f - gg*gg), d3 = dht*tc*og*(1.f - og);
      dg[j] = d0; dg[H + j] = d1; dg[2*H + j] = d2; dg[3*H + j] = d3;
      float* out = dgates + ((long long)b*T + t)*4*H;
      out[j] = d0; out[H + j] = d1; out[2*H + j] = d2; out[3*H + j] = d3;
      dc = dct*fg;
    }
    cur = nxt; nxt = far;
    lds_barrier();
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < kLstmRegH; i += 4) {
      if (i < H) {
        const float4 v = *reinterpret_cast<const float4*>(dg + q*H + i);
        acc += w[i]*v.x + w[i + 1]*v.y + w[i + 2]*v.z + w[i + 3]*v.w;
      }
    }
    part[q][k] = acc;
    lds_barrier();
  }
}

// ---- quad layout (H == 128): ONE barrier per time step ---------------------------------------------------
// The kernels above give every gate row its own thread, so the four gates of a unit meet through LDS (two
// barriers per step, 1.35 / 1.75 us per step of a 641-step chain: 14 % of a DCCRN training step). Here the
// four threads of a QUAD (lanes 4u .. 4u + 3) own hidden unit u together:
//   forward : lane q holds W_hh[g H + u][32 q .. 32 q + 31] for all four gates g (128 VGPRs), multiplies its
//             quarter of h, and the four partial sums of each gate meet in two DPP steps -- every lane of the
//             quad then holds the four gate pre-activations of its unit, the cell state stays in registers;
//   backward: lane q holds W_hh[q H + i][u], i < 128 (gate q's rows, column u): its dot product with gate q's
//             gradients is one of the four partial sums of dh_{t-1}[u], again reduced by DPP; lane q then
//             computes gate q's gradient of the NEXT step.
// h / the gate gradients are double-buffered in LDS: a step ends with ONE LDS-only barrier.
// Loads whose completion the compiler does not track: hipcc's s_waitcnt insertion is conservative at the head
// of these loops (it waited for the load it had JUST issued: vmcnt(0) / vmcnt(1) in every step, i.e. one
// memory round trip per time step). The loads below are invisible to it; the kernel waits by hand with a
// counted vmcnt (loads, stores and their order are fixed per step), tied to the registers it releases.
// Round 6: those loads are plain (compiler-tracked) loads again, and the two kernels DCCRN's use_amp step runs
// (lstm_fwd_mv_kernel / lstm_bwd_mv_kernel) prefetch through LDS instead (dma_dword below). What went wrong with
// register destinations the compiler does not know to be in flight: it may COPY such a register (the phi copies of
// the rotating register sets at the loop's back edge: `v_mov_b32 v135, v136` one step after `global_load_dword
// v136` was issued, three steps before the counted wait) or park a temporary in it -- correct as long as the
// load happens to have landed, garbage when HBM is busy (the side stream's weight gradients): one run in three of
// tests/test_gpu_sizes.py::test_dccrn_default_size_gradients_fp32_and_use_amp had LSTM gradients 10^3 off.
__device__ __forceinline__ float load_untracked(const float* p) { return *p; }
// Asynchronous global -> LDS copy, one dword per lane (LDS-DMA): lane l of the wave writes LDS byte address
// `lds_wave_base` (wave-uniform) + 4 l. There is NO register destination, so nothing the compiler does with
// registers can touch a load in flight; the issuing wave waits with a counted vmcnt, other waves read behind a
// barrier. Issued as assembly: a DMA the compiler can see makes it drain vmcnt(0) in front of every LDS read.
// (m0 is reserved for exactly this use; the kernels below contain no other user of it.)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void dma_dword(const float* lane_src, unsigned int lds_wave_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off"
               :: "v"(lane_src), "s"(lds_wave_base) : "memory", "m0");
}
#pragma clang diagnostic pop
__device__ __forceinline__ unsigned int lds_u32(const void* p) { return (unsigned int)(unsigned long long)p; }
__device__ __forceinline__ float quad_sum(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  return v;
}
template <bool HAS_ACT>
__global__ __launch_bounds__(512) void lstm_fwd_quad_kernel(const float* gates_in, const float* w_hh,
                                                            const float* bias, float* y, float* act,
                                                            float* cs, int T, int per_group) {
  constexpr int H = 128;
  // h in quarters of 32 at a stride of 36 floats: the four lanes of a quad read four different 16-byte pieces
  // per instruction, which must fall on different banks (at 32 floats quarters 0 / 2 and 1 / 3 collided)
  constexpr int QS = 36;
  __shared__ __attribute__((aligned(16))) float h[2][4*QS];
  const int b = blockIdx.x, r = threadIdx.x;
  const int u = r >> 2, q = r & 3;
  w_hh += (long long)(b / per_group)*4*H*H;
  if (bias) bias += (long long)(b / per_group)*4*H;
  float w[4][32];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int i = 0; i < 32; ++i) w[g][i] = w_hh[(long long)(g*H + u)*H + 32*q + i];
  // lane q adds gate q's input projection (+ bias) to ITS partial sum of that gate
  const float bq = bias ? bias[q*H + u] : 0.f;
  if (r < 4*QS) { h[0][r] = 0.f; h[1][r] = 0.f; }
  float c = 0.f;
  const float* gi = gates_in + (long long)b*T*4*H + q*H + u;
  float g_next = gi[0];
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    // (a hand-counted wait on an untracked load, as in the backward kernel, measured SLOWER here: 531 against
    // 409 us per launch -- the asm statements pin the schedule of the whole step)
    const float gin = g_next + bq;
    if (t + 1 < T) g_next = gi[(long long)(t + 1)*4*H];
    const float* hb = h[t & 1] + QS*q;
    float p[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 32; i += 4) {
      const float4 hv = *reinterpret_cast<const float4*>(hb + i);
#pragma unroll
      for (int g = 0; g < 4; ++g)
        p[g] += w[g][i]*hv.x + w[g][i + 1]*hv.y + w[g][i + 2]*hv.z + w[g][i + 3]*hv.w;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) p[g] = quad_sum(p[g] + (q == g ? gin : 0.f));
    const float ig = fast_sigm(p[0]), fg = fast_sigm(p[1]), gg = fast_tanh(p[2]), og = fast_sigm(p[3]);
    c = fg*c + ig*gg;
    const float hn = og*fast_tanh(c);
    const long long row = (long long)b*T + t;
    if (q == 0) h[(t + 1) & 1][(u >> 5)*QS + (u & 31)] = hn;
    // exactly 1 + 2 HAS_ACT store instructions per step (the counted wait above relies on it)
    if (q == 0) y[row*H + u] = hn;
    if (HAS_ACT) {
      if (q == 0) cs[row*H + u] = c;
      act[row*4*H + q*H + u] = q == 0 ? ig : q == 1 ? fg : q == 2 ? gg : og;
    }
    lds_barrier();
  }
}

__global__ __launch_bounds__(512) void lstm_bwd_quad_kernel(const float* act, const float* cs,
                                                            const float* w_hh, const float* dy,
                                                            float* dgates, int T, int per_group) {
  constexpr int H = 128;
  // gate q's gradients at q (H + 4) floats: the four lanes of a quad read four DIFFERENT 16-byte pieces per
  // instruction; at a stride of H floats (512 B) they fell on the same banks (4-way conflict on all 32 reads)
  constexpr int GS = H + 4;
  __shared__ __attribute__((aligned(16))) float dg[2][4*GS];
  const int b = blockIdx.x, j = threadIdx.x;
  const int u = j >> 2, q = j & 3;
  w_hh += (long long)(b / per_group)*4*H*H;
  float w[H];
#pragma unroll
  for (int i = 0; i < H; ++i) w[i] = w_hh[(long long)(q*H + i)*H + u];
  dg[0][q*GS + u] = 0.f; dg[1][q*GS + u] = 0.f;
  float dc = 0.f;
  // What step t needs of the forward pass -- ig, fg, gg, og, c_t, c_{t-1}, dy_t -- through an LDS ring (round 6: see
  // dma_dword; the layout and the counts are those of lstm_bwd_mv_kernel). Slot t & 7 = [activations 4 H | c_t | c_{t-1}
  // | dy_t]; per wave and step two LDS-DMAs kPre = 4 steps ahead, then ONE store. This kernel has ONE barrier per step:
  // the wave's DMAs of step t - 1 (issued at step t + 3; 10 younger operations) land before the barrier ending step t.
  constexpr int kRing = 8, kPre = 4, kSlot = 7*H;
  __shared__ float ring[kRing][kSlot];
  const int wu = __builtin_amdgcn_readfirstlane(j >> 6), lane = j & 63;
  const unsigned int ring_a = lds_u32(&ring[0][0]) + 256u*(unsigned int)wu;
  const int part2 = wu < 6 ? wu >> 1 : 2;                         // 0: c_t, 1: c_{t-1}, 2: dy_t
  const unsigned int ring_b = lds_u32(&ring[0][0]) + (unsigned int)((4 + part2)*H*4 + 256*(wu & 1));
  const float* src_a = act + (long long)b*T*4*H + 64*wu + lane;
  const float* src_b = (part2 == 2 ? dy : cs) + (long long)b*T*H + 64*(wu & 1) + lane;
  auto issue = [&](int t) {
    const int tt = t > 0 ? t : 0;
    const int tb = part2 == 1 ? (tt > 0 ? tt - 1 : 0) : tt;
    const unsigned int slot = (unsigned int)((t & (kRing - 1))*kSlot*4);
    dma_dword(src_a + (long long)tt*4*H, ring_a + slot);
    dma_dword(src_b + (long long)tb*H, ring_b + slot);
  };
  struct Saved { float ig, fg, gg, og, c, cp, dy; };
  auto step = [&](int t) {
    issue(t - kPre);
    const float* sv = ring[t & (kRing - 1)] + u;
    Saved cur;
    cur.ig = sv[0]; cur.fg = sv[H]; cur.gg = sv[2*H]; cur.og = sv[3*H];
    cur.c = sv[4*H]; cur.cp = sv[5*H]; cur.dy = sv[6*H];
    // dh_t[u] from the gate gradients of step t + 1 (zeros for the last step): this lane's gate rows
    const float* gb = dg[(t + 1) & 1] + q*GS;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int i = 0; i < H; i += 4) {
      const float4 v = *reinterpret_cast<const float4*>(gb + i);
      a0 = __builtin_fmaf(w[i], v.x, a0); a1 = __builtin_fmaf(w[i + 1], v.y, a1);
      a2 = __builtin_fmaf(w[i + 2], v.z, a2); a3 = __builtin_fmaf(w[i + 3], v.w, a3);
    }
    const float dht = quad_sum((a0 + a1) + (a2 + a3)) + cur.dy;
    const float ig = cur.ig, fg = cur.fg, gg = cur.gg, og = cur.og;
    const float cprev = t > 0 ? cur.cp : 0.f;
    const float tc = fast_tanh(cur.c);
    const float dct = dc + dht*og*(1.f - tc*tc);
    const float dq = q == 0 ? dct*gg*ig*(1.f - ig) : q == 1 ? dct*cprev*fg*(1.f - fg)
                   : q == 2 ? dct*ig*(1.f - gg*gg) : dht*tc*og*(1.f - og);
    dg[t & 1][q*GS + u] = dq;
    dgates[((long long)b*T + t)*4*H + q*H + u] = dq;
    dc = dct*fg;
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");            // this wave's pieces of step t - 1 are in LDS
    lds_barrier();
  };
  for (int t = T - 1; t > T - 1 - kPre; --t) issue(t);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = T - 1; t >= 0; --t) step(t);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- matrix-pipe layout (H == 128, use_amp): the step's matrix-vector product on the bf16 MFMA -------------------
// The quad kernels above spend a step on 128 fp32 FMAs and 8 (forward) / 32 (backward) 16-byte LDS reads PER THREAD:
// 0.94 us per step either way, 4 x 0.47 ms of a DCCRN training step. Under use_amp the recurrent product may run on
// bf16 operands like every other product of that path (the reference's autocast hands nn.LSTM bf16 weights and a
// bf16 hidden state); accumulation, gate math, cell state and every tensor in HBM stay fp32. One workgroup per
// chain as before; lane (j = lane >> 4, n = lane & 15) of wave w owns hidden unit u = 16 w + n.
//   forward : D (16 x 16) = A (16 x 32) B (32 x 16), v_mfma_f32_16x16x32_bf16, with A = h (every row the same
//             vector: a lane reads the 8 values k = 32 kq + 8 j .. from LDS), B[k][n] = W_hh[g H + u][k] kept in
//             registers (4 gates x 4 k-steps = 64 VGPRs): 16 instructions per wave and step; every lane then holds
//             the four gate pre-activations of its unit in register 0 of the four accumulators;
//   backward: A = the gate gradients of step t + 1 (K = 4 H: 16 k-steps), B[k][n] = W_hh[k][u] (64 VGPRs);
//             lane j computes gate j's gradient of its unit.
// Only the A rows m = 0, 4, 8, 12 are loaded (lanes with n % 4 == 0; the other rows are zero): row 4 j is the one
// whose result register 0 of lane group j holds, and a 16-byte LDS read costs by the lanes that take part.
// The per-step global loads are invisible to hipcc's wait insertion (load_untracked above) and requested THREE steps
// ahead into four register sets in rotation; the kernels wait with a counted vmcnt -- loads and stores per step are
// fixed in number and order.
__device__ __forceinline__ bf16x8 lstm_frag8(const float* p, long long stride) {
  uint4 q;
  q.x = pack2(p[0], p[stride]); q.y = pack2(p[2*stride], p[3*stride]);
  q.z = pack2(p[4*stride], p[5*stride]); q.w = pack2(p[6*stride], p[7*stride]);
  return __builtin_bit_cast(bf16x8, q);
}
#ifndef LSTM_MV_MASK
#define LSTM_MV_MASK 0    // 1: only the lanes of rows 0, 4, 8, 12 read the A fragments (measured no faster: a 16-byte LDS
#endif                    // read costs the same with 16 or 64 lanes taking part; the unmasked form needs no assembly)
#ifndef LSTM_MV_ABL
#define LSTM_MV_ABL 0     // diagnostic builds of the forward kernel (results wrong): 1 no products, 2 no exponentials,
#endif                    // 4 no global stores, 8 no fragment reads, 16 no barrier
#ifndef LSTM_MV_PRIO
#define LSTM_MV_PRIO 1
#endif
// Waves w and w + 4 share a SIMD and run the same step in lockstep (one barrier per step): both issue their 16
// products, then both their gate math -- matrix pipe and vector ALUs take turns. With the first four waves at a higher
// priority their products go first and their gate math runs beside the other wave's products.
__device__ __forceinline__ void lstm_mv_prio(int w) {
  if (LSTM_MV_PRIO) { if (w < 4) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0); }
}
// A fragments straight from LDS: NR 16-byte reads 64 bytes apart and one wait, as inline assembly under the lanes'
// own branch (written as a select, hipcc turned `take ? *p : 0` into a FLAT load through a pointer that is either
// the LDS address or a zeroed scratch slot, with a full vmcnt(0) lgkmcnt(0) wait per fragment)
__device__ __forceinline__ unsigned int lstm_lds_addr(const void* p) { return (unsigned int)(unsigned long long)p; }
__device__ __forceinline__ void lstm_read4(unsigned int a, u32x4 (&q)[4]) {
  asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:64\n\tds_read_b128 %2, %4 offset:128\n\t"
               "ds_read_b128 %3, %4 offset:192\n\ts_waitcnt lgkmcnt(0)"
               : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]) : "v"(a) : "memory");
}
template <bool HAS_ACT>
__global__ __launch_bounds__(512) void lstm_fwd_mv_kernel(const float* gates_in, const float* w_hh,
                                                          const float* bias, float* y, float* act,
                                                          float* cs, int T, int per_group) {
  constexpr int H = 128;
  __shared__ __attribute__((aligned(16))) uint16_t hb[2][H];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int n = lane & 15, j = lane >> 4, u = 16*w + n;
  const bool take = (n & 3) == 0;
  lstm_mv_prio(__builtin_amdgcn_readfirstlane(w));
  w_hh += (long long)(b / per_group)*4*H*H;
  if (bias) bias += (long long)(b / per_group)*4*H;
  bf16x8 wf[4][4];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) wf[g][kq] = lstm_frag8(w_hh + (long long)(g*H + u)*H + 32*kq + 8*j, 1);
  float bq[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) bq[g] = bias ? bias[g*H + u] : 0.f;
  if (tid < 2*H) (&hb[0][0])[tid] = 0;
  float c = 0.f;
  // Input rows through an LDS ring (round 6: see dma_dword). Slot t & 7 holds the 4 H gate inputs of step t; wave
  // w copies dwords 64 w .. 64 w + 63 of a row (one LDS-DMA per wave and step), kPre = 4 steps ahead. Per wave and
  // step, in order: 1 DMA, then 1 + 2 HAS_ACT stores. The wave's DMA of step t + 1 (issued at step t - 3) must have
  // landed before the barrier that ends step t: younger than it are the stores of step t - 3 and everything of
  // steps t - 2 .. t = S + 3 (1 + S) operations (vector-memory operations of a wave retire in issue order).
  constexpr int kRing = 8, kPre = 4, S = HAS_ACT ? 3 : 1;
  __shared__ float ring[kRing][4*H];
  const int wu = __builtin_amdgcn_readfirstlane(w);
  const unsigned int ring0 = lds_u32(&ring[0][0]) + 256u*(unsigned int)wu;
  const float* gi = gates_in + (long long)b*T*4*H + 64*w + lane;
  auto issue = [&](int t) {
    dma_dword(gi + (long long)(t < T ? t : T - 1)*4*H, ring0 + (unsigned int)((t & (kRing - 1))*4*H*4));
  };
  struct In { float g0, g1, g2, g3; };
  auto step = [&](int t) {
    issue(t + kPre);
    const float* in = ring[t & (kRing - 1)] + u;
    In cur;
    cur.g0 = in[0]; cur.g1 = in[H]; cur.g2 = in[2*H]; cur.g3 = in[3*H];
    u32x4 aq[4];
    if (LSTM_MV_MASK) {
#pragma unroll
      for (int kq = 0; kq < 4; ++kq) aq[kq] = u32x4{0u, 0u, 0u, 0u};
      if (take && !(LSTM_MV_ABL & 8)) lstm_read4(lstm_lds_addr(hb[t & 1] + 8*j), aq);
    } else {
      // every lane reads (rows of A all equal): plain loads, the compiler waits fragment by fragment
#pragma unroll
      for (int kq = 0; kq < 4; ++kq)
        aq[kq] = (LSTM_MV_ABL & 8) ? u32x4{0u, 0u, 0u, 0u} : *reinterpret_cast<const u32x4*>(hb[t & 1] + 8*j + 32*kq);
    }
    f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    // two gates at a time: the first pair's exponentials run beside the second pair's products
#pragma unroll
    for (int gp = 0; gp < 4; gp += 2)
#pragma unroll
      for (int kq = 0; kq < 4; ++kq) {
        const bf16x8 a = __builtin_bit_cast(bf16x8, aq[kq]);
        if (LSTM_MV_ABL & 1) { acc[gp][0] += __uint_as_float(aq[kq].x); continue; }
        acc[gp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, wf[gp][kq], acc[gp], 0, 0, 0);
        acc[gp + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, wf[gp + 1][kq], acc[gp + 1], 0, 0, 0);
      }
    float ig, fg, gg, og, hn;
    if (LSTM_MV_ABL & 2) {
      ig = acc[0][0] + cur.g0 + bq[0]; fg = acc[1][0] + cur.g1 + bq[1]; gg = acc[2][0] + cur.g2 + bq[2];
      og = acc[3][0] + cur.g3 + bq[3];
      c = fg*c + ig*gg; hn = og*c;
    } else {
      ig = fast_sigm(acc[0][0] + cur.g0 + bq[0]); fg = fast_sigm(acc[1][0] + cur.g1 + bq[1]);
      gg = fast_tanh(acc[2][0] + cur.g2 + bq[2]); og = fast_sigm(acc[3][0] + cur.g3 + bq[3]);
      c = fg*c + ig*gg;
      hn = og*fast_tanh(c);
    }
    if (j == 0) hb[(t + 1) & 1][u] = f2bf(hn);
    const long long row = (long long)b*T + t;
    // exactly 1 + 2 HAS_ACT store instructions per step (the counted wait relies on it): lane group 0 stores y,
    // group 1 the cell state, group j gate j's activation
    if (!(LSTM_MV_ABL & 4)) {
      if (j == 0) y[row*H + u] = hn;
      if (HAS_ACT) {
        if (j == 1) cs[row*H + u] = c;
        act[row*4*H + j*H + u] = j == 0 ? ig : j == 1 ? fg : j == 2 ? gg : og;
      }
    } else if (hn == 123.456f) y[row*H + u] = ig + fg + gg + og;
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(S + 3*(1 + S)) : "memory");     // this wave's piece of row t + 1 is in LDS
    if (!(LSTM_MV_ABL & 16)) lds_barrier();
  };
  for (int t = 0; t < kPre; ++t) issue(t);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = 0; t < T; ++t) step(t);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__global__ __launch_bounds__(512) void lstm_bwd_mv_kernel(const float* act, const float* cs, const float* w_hh,
                                                          const float* dy, float* dgates, int T, int per_group) {
  constexpr int H = 128, K = 4*H;
  // The reduction (K = 4 H) is SPLIT over the eight waves: wave w multiplies the 64 gradients k = 64 w .. against all
  // 128 units (8 column blocks x 2 k-steps = 16 instructions, two 16-byte LDS reads) and the eight partial sums of
  // a unit meet through LDS. (Every wave reducing the whole K for its own 16 units read the complete gradient
  // vector per wave -- sixteen 16-byte reads per lane and step, 1024 LDS clocks per step whether or not lanes were
  // masked off: 0.88 us per step, as slow as the fp32 kernel.)
  __shared__ __attribute__((aligned(16))) uint16_t dgb[2][K];      // bf16 gate gradients, k = q H + i
  __shared__ float part[8][H];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int n = lane & 15, q = lane >> 4, u = 16*w + n;            // this lane: gate q of unit u
  lstm_mv_prio(__builtin_amdgcn_readfirstlane(w));
  w_hh += (long long)(b / per_group)*4*H*H;
  bf16x8 wf[8][2];
#pragma unroll
  for (int nb = 0; nb < 8; ++nb)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      wf[nb][ks] = lstm_frag8(w_hh + (long long)(64*w + 32*ks + 8*q)*H + 16*nb + n, H);
  (&dgb[0][0])[tid] = 0; (&dgb[0][0])[tid + 512] = 0;
  float dc = 0.f;
  // Saved tensors through an LDS ring (round 6: see dma_dword). Slot t & 7 = [activations of step t: 4 H | c_t: H |
  // c_{t-1}: H | dy_t: H]; per wave and step TWO LDS-DMAs, kPre = 4 steps ahead: wave w copies dwords 64 w .. of the
  // activation row, and waves (0, 1) / (2, 3) / (4, 5) the halves of c_t / c_{t-1} / dy_t (waves 6, 7 repeat 4, 5:
  // the same bytes to the same place, so that every wave issues the same number of operations). Per wave and
  // step, in order: 2 DMAs, then ONE store. The wave's DMAs of step t - 1 (issued at step t + 3) must have landed
  // before the barrier that ends step t: younger than them are the store of step t + 3 and the three operations of
  // each of the steps t + 2 .. t = 10 (vector-memory operations of a wave retire in issue order).
  constexpr int kRing = 8, kPre = 4, kSlot = 7*H;
  __shared__ float ring[kRing][kSlot];
  const int wu = __builtin_amdgcn_readfirstlane(w);
  const unsigned int ring_a = lds_u32(&ring[0][0]) + 256u*(unsigned int)wu;
  const int part2 = wu < 6 ? wu >> 1 : 2;                         // 0: c_t, 1: c_{t-1}, 2: dy_t
  const unsigned int ring_b = lds_u32(&ring[0][0]) + (unsigned int)((4 + part2)*H*4 + 256*(wu & 1));
  const float* src_a = act + (long long)b*T*4*H + 64*w + lane;
  const float* src_b = (part2 == 2 ? dy : cs) + (long long)b*T*H + 64*(w & 1) + lane;
  auto issue = [&](int t) {
    const int tt = t > 0 ? t : 0;
    const int tb = part2 == 1 ? (tt > 0 ? tt - 1 : 0) : tt;
    const unsigned int slot = (unsigned int)((t & (kRing - 1))*kSlot*4);
    dma_dword(src_a + (long long)tt*4*H, ring_a + slot);
    dma_dword(src_b + (long long)tb*H, ring_b + slot);
  };
  struct Saved { float ig, fg, gg, og, c, cp, dy; };
  auto step = [&](int t) {
    issue(t - kPre);
    // this wave's slice of the gate gradients of step t + 1 (zeros for the last step)
    const uint16_t* gp = dgb[(t + 1) & 1] + 64*w + 8*q;
    const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(gp), a1 = *reinterpret_cast<const bf16x8*>(gp + 32);
    f32x4 p[8];
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
      p[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, wf[nb][0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) p[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, wf[nb][1], p[nb], 0, 0, 0);
    // every lane group holds all eight blocks (the rows of D are equal): group q hands over blocks 2 q, 2 q + 1
    const float e0 = q == 0 ? p[0][0] : q == 1 ? p[2][0] : q == 2 ? p[4][0] : p[6][0];
    const float e1 = q == 0 ? p[1][0] : q == 1 ? p[3][0] : q == 2 ? p[5][0] : p[7][0];
    part[w][32*q + n] = e0; part[w][32*q + 16 + n] = e1;
    const float* sv = ring[t & (kRing - 1)] + u;
    Saved cur;
    cur.ig = sv[0]; cur.fg = sv[H]; cur.gg = sv[2*H]; cur.og = sv[3*H];
    cur.c = sv[4*H]; cur.cp = sv[5*H]; cur.dy = sv[6*H];
    lds_barrier();
    float dht = cur.dy;
#pragma unroll
    for (int i = 0; i < 8; ++i) dht += part[i][u];
    const float ig = cur.ig, fg = cur.fg, gg = cur.gg, og = cur.og;
    const float cprev = t > 0 ? cur.cp : 0.f;
    const float tc = fast_tanh(cur.c);
    const float dct = dc + dht*og*(1.f - tc*tc);
    const float dq = q == 0 ? dct*gg*ig*(1.f - ig) : q == 1 ? dct*cprev*fg*(1.f - fg)
                   : q == 2 ? dct*ig*(1.f - gg*gg) : dht*tc*og*(1.f - og);
    dgb[t & 1][q*H + u] = f2bf(dq);
    dgates[((long long)b*T + t)*4*H + q*H + u] = dq;
    dc = dct*fg;
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");            // this wave's pieces of step t - 1 are in LDS
    lds_barrier();
  };
  for (int t = T - 1; t > T - 1 - kPre; --t) issue(t);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = T - 1; t >= 0; --t) step(t);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// out = a - b  /  a + b  (the real / imaginary recombination of ComplexWrapper)
__global__ __launch_bounds__(256) void combine_kernel(const float* a, const float* b, float* out,
                                                      long long n, float sign) {
  GRID_STRIDE(i, n) out[i] = a[i] + sign*b[i];
}

// The complex combination behind a ComplexLSTM layer (dccrn.py:330-358): o = (2 modules, 2 halves, n) holds module m on
// input half h at o[m][h]; real = real(real) - imag(imag) = o[0][0] - o[1][1], imag = real(imag) + imag(real) =
// o[0][1] + o[1][0]. One launch each way instead of two selects, two chunks and two combines (and, backwards, their
// zero-filled select gradients, two concatenations, two scalings and an accumulation).
__global__ __launch_bounds__(256) void complex_mix_fwd_kernel(const float4* o, float4* real, float4* imag, long long n4) {
  GRID_STRIDE(i, n4) {
    const float4 rr = o[i], ri = o[n4 + i], ir = o[2*n4 + i], ii = o[3*n4 + i];
    real[i] = make_float4(rr.x - ii.x, rr.y - ii.y, rr.z - ii.z, rr.w - ii.w);
    imag[i] = make_float4(ri.x + ir.x, ri.y + ir.y, ri.z + ir.z, ri.w + ir.w);
  }
}
__global__ __launch_bounds__(256) void complex_mix_bwd_kernel(const float4* greal, const float4* gimag, float4* dout,
                                                              long long n4) {
  GRID_STRIDE(i, n4) {
    const float4 a = greal[i], b = gimag[i];
    dout[i] = a; dout[n4 + i] = b; dout[2*n4 + i] = b;
    dout[3*n4 + i] = make_float4(-a.x, -a.y, -a.z, -a.w);
  }
}

// DCCRN.apply_mask (dccrn.py:96-109): polar product with a tanh-bounded mask magnitude
__global__ __launch_bounds__(256) void dccrn_mask_kernel(const float* xr, const float* xi,
                                                         const float* mr, const float* mi,
                                                         float2* out, long long n, long long bs) {
  // blockIdx.y = batch item: (real | imaginary) planes of x and of the mask are bs floats apart per item
  xr += blockIdx.y*bs; xi += blockIdx.y*bs; mr += blockIdx.y*bs; mi += blockIdx.y*bs; out += blockIdx.y*n;
  for (long long i = (long long)blockIdx.x*blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x*blockDim.x) {
    const float a = xr[i], b = xi[i];
    const float in_mag = sqrtf(a*a + b*b), in_phase = atan2f(b, a);
    float pr = mr[i];
    const float pi = mi[i];
    const float mag = tanhf(sqrtf(pr*pr + pi*pi + 1e-7f));
    if (pr == 0.f) pr = 1e-7f;
    const float ph = in_phase + atan2f(pi, pr);
    const float om = in_mag*mag;
    out[i] = make_float2(om*cosf(ph), om*sinf(ph));
  }
}


// Conv2d weight gradient: dw[co][ci][i][j] (+)= sign * sum_{b,ho,wo} dy[b][co][ho][wo] *
// x[b][ci][ho*sh - ph + i][wo*sw - pw + j]; one workgroup per weight element.
// (ConvTranspose2d: call with x := the layer's output gradient, dy := the layer's input.)
__global__ __launch_bounds__(256) void conv2d_wgrad_kernel(const float* x, const float* dy,
                                                           float* dw, ConvGeom g, int accumulate,
                                                           float sign) {
  __shared__ double scr[8];
  long long id = blockIdx.x;
  const int j = (int)(id % g.kw); id /= g.kw;
  const int i = (int)(id % g.kh); id /= g.kh;
  const int ci = (int)(id % g.Cin);
  const int co = (int)(id / g.Cin);
  const long long n = (long long)g.B*g.Ho*g.Wo;
  double acc = 0.0;
  for (long long e = threadIdx.x; e < n; e += 256) {
    const int wo = (int)(e % g.Wo);
    const int ho = (int)((e / g.Wo) % g.Ho);
    const int b = (int)(e / ((long long)g.Wo*g.Ho));
    const int hi = ho*g.sh - g.ph + i, wi = wo*g.sw - g.pw + j;
    if (hi < 0 || hi >= g.H || wi < 0 || wi >= g.W) continue;
    acc += (double)dy[(long long)b*g.y_bs + ((long long)co*g.Ho + ho)*g.Wo + wo]
           *x[(long long)b*g.x_bs + ((long long)ci*g.H + hi)*g.W + wi];
  }
  acc = block_sum(acc, scr);
  if (threadIdx.x == 0) {
    float* d = dw + blockIdx.x;
    *d = accumulate ? *d + sign*(float)acc : sign*(float)acc;
  }
}
// db[c] (+)= sign * sum_{b,hw} dy[b][c][hw]
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* dy, float* db, int B,
                                                          long long HW, long long bs,
                                                          int accumulate, float sign) {
  __shared__ double scr[8];
  const int c = blockIdx.x;
  double acc = 0.0;
  for (long long e = threadIdx.x; e < (long long)B*HW; e += 256)
    acc += dy[(e / HW)*bs + (long long)c*HW + e % HW];
  acc = block_sum(acc, scr);
  if (threadIdx.x == 0) db[c] = accumulate ? db[c] + sign*(float)acc : sign*(float)acc;
}

// ---- convolution as GEMM: column matrices --------------------------------------------------
// col[b][(c*kh + i)*kw + j][ho*Wo + wo] = x[b][c][ho*sh - ph + i][wo*sw - pw + j] (0 outside)
// grid (pixel blocks, C*kh*kw, B): one column-matrix row per blockIdx.y, 32-bit index math,
// 4 pixels per thread
// ColT = float, or bf16_t for the use_amp path (the column matrix is the largest tensor of a
// convolution: half the bytes on its way to and from the MFMA product)
template <typename ColT> __device__ __forceinline__ ColT to_col(float v);
template <> __device__ __forceinline__ float to_col<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t to_col<bf16_t>(float v) { return f2bf(v); }
__device__ __forceinline__ float from_col(float v) { return v; }
__device__ __forceinline__ float from_col(bf16_t v) { return bf2f(v); }

__device__ __forceinline__ void store_col4(float* out, const float4& v) {
  *reinterpret_cast<float4*>(out) = v;
}
__device__ __forceinline__ void store_col4(bf16_t* out, const float4& v) {
  *reinterpret_cast<uint2*>(out) = make_uint2(pack2(v.x, v.y),
                                              pack2(v.z, v.w));
}

template <typename ColT>
__global__ __launch_bounds__(256) void im2col_kernel(const float* x, ColT* col, ConvGeom g) {
  const int HoWo = g.Ho*g.Wo;
  const int r = blockIdx.y, b = blockIdx.z;
  const int j = r % g.kw, i = (r / g.kw) % g.kh, c = r / (g.kw*g.kh);
  const float* xc = x + (long long)b*g.x_bs + (long long)c*g.H*g.W;
  ColT* out = col + ((long long)b*gridDim.y + r)*HoWo;
  const int p0 = (blockIdx.x*256 + threadIdx.x)*4;
  if (p0 >= HoWo) return;
  // fast path: unit stride along W, the 4 pixels in one output row and their sources inside the
  // image (or the whole source row outside it): one 16-byte load (4-byte aligned: legal on
  // gfx950), one aligned vector store
  if (g.sw == 1 && (HoWo & 3) == 0) {
    const int ho = p0 / g.Wo, wo = p0 - ho*g.Wo;
    const int hi = ho*g.sh - g.ph + i, wi = wo - g.pw + j;
    if (wo + 3 < g.Wo) {
      if (hi < 0 || hi >= g.H) { store_col4(out + p0, make_float4(0.f, 0.f, 0.f, 0.f)); return; }
      if (wi >= 0 && wi + 3 < g.W) {
        float4 v;
        __builtin_memcpy(&v, xc + (long long)hi*g.W + wi, 16);
        store_col4(out + p0, v);
        return;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int pix = p0 + q;
    if (pix >= HoWo) break;
    const int ho = pix / g.Wo, wo = pix - ho*g.Wo;
    const int hi = ho*g.sh - g.ph + i, wi = wo*g.sw - g.pw + j;
    out[pix] = to_col<ColT>((hi >= 0 && hi < g.H && wi >= 0 && wi < g.W) ? xc[hi*g.W + wi] : 0.f);
  }
}
// the adjoint: y[b][c][h][w] = bias[c] + sum over the column entries that im2col filled from it
// (g.H x g.W is the image, g.Ho x g.Wo the column grid); grid (pixel blocks, C, B)
template <typename ColT>
__global__ __launch_bounds__(256) void col2im_kernel(const ColT* col, const float* bias, float* y,
                                                     ConvGeom g) {
  const int HoWo = g.Ho*g.Wo, HW = g.H*g.W;
  const int c = blockIdx.y, b = blockIdx.z;
  const int pix = blockIdx.x*256 + threadIdx.x;
  if (pix >= HW) return;
  const int h = pix / g.W, w = pix - h*g.W;
  float acc = bias ? bias[c] : 0.f;
  const ColT* cb = col + ((long long)b*gridDim.y + c)*g.kh*g.kw*HoWo;
  for (int i = 0; i < g.kh; ++i) {
    const int hn = h + g.ph - i;
    if (hn < 0 || hn % g.sh) continue;
    const int ho = hn/g.sh;
    if (ho >= g.Ho) continue;
    for (int j = 0; j < g.kw; ++j) {
      const int wn = w + g.pw - j;
      if (wn < 0 || wn % g.sw) continue;
      const int wo = wn/g.sw;
      if (wo >= g.Wo) continue;
      acc += from_col(cb[(long long)(i*g.kw + j)*HoWo + ho*g.Wo + wo]);
    }
  }
  y[(long long)b*g.y_bs + (long long)c*HW + pix] = acc;
}
// complex weight as one real matrix: wc (2R x 2C) = [[wr, -s*wi], [s*wi, wr]], and its adjoint
__global__ __launch_bounds__(256) void cweight_pack_kernel(const float* wr, const float* wi, float* wc,
                                                           int R, int C, float s) {
  GRID_STRIDE(idx, (long long)4*R*C) {
    const int col = (int)(idx % (2*C)), row = (int)(idx / (2*C));
    const int r = row % R, c = col % C;
    const bool lower = row >= R, right = col >= C;
    float v;
    if (lower == right) v = wr[(long long)r*C + c];
    else v = (right ? -s : s)*wi[(long long)r*C + c];
    wc[idx] = v;
  }
}
__global__ __launch_bounds__(256) void cweight_unpack_kernel(const float* dwc, float* dwr, float* dwi,
                                                             int R, int C, float s) {
  GRID_STRIDE(idx, (long long)R*C) {
    const int c = (int)(idx % C), r = (int)(idx / C);
    const long long ld = 2*C;
    const float d00 = dwc[(long long)r*ld + c], d01 = dwc[(long long)r*ld + C + c];
    const float d10 = dwc[(long long)(R + r)*ld + c], d11 = dwc[(long long)(R + r)*ld + C + c];
    dwr[idx] = d00 + d11;
    dwi[idx] = s*(d10 - d01);
  }
}

// BatchNorm2d (+ optional scalar PReLU after it) backward, training mode.
// pass 1 per channel: dpre = dy*prelu'(u), u = xh*gamma + beta; sums of dpre and dpre*xh;
//                     slope gradient sum of dy*u over u < 0
__global__ __launch_bounds__(256) void bn_bwd_stats_kernel(const float* x, const float* dy,
                                                           const float* mean, const float* invstd,
                                                           const float* gamma, const float* beta,
                                                           const float* slope, int B, int C,
                                                           long long HW, double* part) {
  __shared__ double scr[8];
  const int c = blockIdx.x;
  const float a = slope ? *slope : 1.f;
  double s1 = 0.0, s2 = 0.0, sa = 0.0;
  long long lo, hi;
  slice_range((long long)B*HW, lo, hi);
  for (long long e = lo + threadIdx.x; e < hi; e += 256) {
    const long long idx = ((e / HW)*C + c)*HW + e % HW;
    const float xh = (x[idx] - mean[c])*invstd[c];
    const float u = xh*gamma[c] + beta[c];
    float d = dy[idx];
    if (slope && u <= 0.f) { sa += (double)d*u; d *= a; }
    s1 += d; s2 += (double)d*xh;
  }
  s1 = block_sum(s1, scr); __syncthreads();
  s2 = block_sum(s2, scr); __syncthreads();
  sa = block_sum(sa, scr);
  if (threadIdx.x == 0) {
    double* o = part + ((long long)c*gridDim.y + blockIdx.y)*3;
    o[0] = s1; o[1] = s2; o[2] = sa;
  }
}
// (dy2: a second gradient with respect to the output, added on the fly -- an encoder block's output feeds the next block
// AND the decoder's skip input: no pass that sums the two)
template <typename TI, typename TG>
__global__ __launch_bounds__(256) void bn_bwd_stats4_kernel(const TI* x, const TG* dy, const TG* dy2,
                                                            const float* mean, const float* invstd,
                                                            const float* gamma, const float* beta,
                                                            const float* slope, int C, long long HW,
                                                            int pieces, double* part) {
  __shared__ double scr[8];
  const int c = blockIdx.x, b = blockIdx.y / pieces, pc = blockIdx.y % pieces;
  const float a = slope ? *slope : 1.f;
  const float mu = mean[c], is = invstd[c], gm = gamma[c], bt = beta[c];
  const long long n4 = HW >> 2, per = (n4 + pieces - 1)/pieces;
  const long long lo = pc*per, hi = lo + per < n4 ? lo + per : n4;
  const TI* xs = x + ((long long)b*C + c)*HW;
  const TG* ds = dy + ((long long)b*C + c)*HW;
  const TG* ds2 = dy2 ? dy2 + ((long long)b*C + c)*HW : nullptr;
  double s1 = 0.0, s2 = 0.0, sa = 0.0;
  for (long long i = lo + threadIdx.x; i < hi; i += 256) {
    const float4 xv = load4(xs, i);
    float4 dv = load4(ds, i);
    if (ds2) { const float4 e = load4(ds2, i); dv.x += e.x; dv.y += e.y; dv.z += e.z; dv.w += e.w; }
    const float xe[4] = {xv.x, xv.y, xv.z, xv.w};
    float de[4] = {dv.x, dv.y, dv.z, dv.w};
    float t1 = 0.f, t2 = 0.f, ta = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float xh = (xe[k] - mu)*is;
      const float u = xh*gm + bt;
      float d = de[k];
      if (slope && u <= 0.f) { ta += d*u; d *= a; }
      t1 += d; t2 += d*xh;
    }
    s1 += (double)t1; s2 += (double)t2; sa += (double)ta;
  }
  s1 = block_sum(s1, scr); __syncthreads();
  s2 = block_sum(s2, scr); __syncthreads();
  sa = block_sum(sa, scr);
  if (threadIdx.x == 0) {
    double* o = part + ((long long)c*gridDim.y + blockIdx.y)*3;
    o[0] = s1; o[1] = s2; o[2] = sa;
  }
}
__global__ __launch_bounds__(256) void bn_bwd_final_kernel(const double* part, int slices, int C,
                                                           float* dgamma, float* dbeta,
                                                           float* dslope_part) {
  const int c = blockIdx.x*256 + threadIdx.x;
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0, sa = 0.0;
  for (int i = 0; i < slices; ++i) {
    const double* o = part + ((long long)c*slices + i)*3;
    s1 += o[0]; s2 += o[1]; sa += o[2];
  }
  dbeta[c] = (float)s1; dgamma[c] = (float)s2; dslope_part[c] = (float)sa;
}
// pass 2: dx = gamma*invstd*(dpre - dbeta/n - xh*dgamma/n)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* x, const float* dy,
                                                           const float* mean, const float* invstd,
                                                           const float* gamma, const float* beta,
                                                           const float* slope, const float* dgamma,
                                                           const float* dbeta, float* dx, int C,
                                                           long long HW, long long total,
                                                           float inv_n) {
  const float a = slope ? *slope : 1.f;
  GRID_STRIDE(i, total) {
    const int c = (int)((i / HW) % C);
    const float xh = (x[i] - mean[c])*invstd[c];
    const float u = xh*gamma[c] + beta[c];
    float d = dy[i];
    if (slope && u <= 0.f) d *= a;
    dx[i] = gamma[c]*invstd[c]*(d - dbeta[c]*inv_n - xh*dgamma[c]*inv_n);
  }
}

// (TO = bf16_t: dx as bf16 -- the gradient with respect to a row convolution's output, which its data and weight
// gradient kernels round to bf16 anyway; `sums`: per-block sums of the UNROUNDED dx, what the convolution's bias
// gradient is made of: [channel][batch item x gridDim.x] partials, added up by row_sum_final-style bn_dxsum_kernel)
template <typename TI, typename TG, typename TO>
__global__ __launch_bounds__(256) void bn_bwd_apply4_kernel(const TI* x, const TG* dy, const TG* dy2, const float* mean,
                                                            const float* invstd, const float* gamma,
                                                            const float* beta, const float* slope,
                                                            const float* dgamma, const float* dbeta, TO* dx,
                                                            int C, long long HW, float inv_n, double* sums) {
  __shared__ double scr[8];
  const int c = blockIdx.y, b = blockIdx.z;
  const float a = slope ? *slope : 1.f;
  const float mu = mean[c], is = invstd[c], gm = gamma[c], bt = beta[c], db = dbeta[c]*inv_n, dg = dgamma[c]*inv_n;
  const long long n4 = HW >> 2, base = ((long long)b*C + c)*HW;
  const TI* xs = x + base;
  const TG* ds = dy + base;
  const TG* ds2 = dy2 ? dy2 + base : nullptr;
  TO* dst = dx + base;
  double tot = 0.0;
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < n4; i += (long long)gridDim.x*256) {
    const float4 xv = load4(xs, i);
    float4 dv = load4(ds, i);
    if (ds2) { const float4 e = load4(ds2, i); dv.x += e.x; dv.y += e.y; dv.z += e.z; dv.w += e.w; }
    const float xe[4] = {xv.x, xv.y, xv.z, xv.w}, de[4] = {dv.x, dv.y, dv.z, dv.w};
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float xh = (xe[k] - mu)*is;
      const float u = xh*gm + bt;
      float d = de[k];
      if (slope && u <= 0.f) d *= a;
      o[k] = gm*is*(d - db - xh*dg);
    }
    store4(dst, i, make_float4(o[0], o[1], o[2], o[3]));
    tot += (double)((o[0] + o[1]) + (o[2] + o[3]));
  }
  if (sums) {
    tot = block_sum(tot, scr);
    if (threadIdx.x == 0) sums[((long long)c*gridDim.z + b)*gridDim.x + blockIdx.x] = tot;
  }
}
__global__ __launch_bounds__(256) void bn_dxsum_kernel(const double* part, int slices, int C, float* out) {
  const int c = blockIdx.x*256 + threadIdx.x;
  if (c >= C) return;
  double s = 0.0;
  for (int i = 0; i < slices; ++i) s += part[(long long)c*slices + i];
  out[c] = (float)s;
}
// LSTM backward through time for one layer and one batch item per workgroup.
// Saved by the forward: act (B, T, 4H) gate activations (i, f, g, o), cs (B, T, H) cell states,
// y (B, T, H) hidden states. In: dy (B, T, H). Out: dgates (B, T, 4H) = gradient wrt the gate
// pre-activations (the weight / input gradients are GEMMs over it).
__global__ __launch_bounds__(256) void lstm_bwd_kernel(const float* act, const float* cs,
                                                       const float* w_hh, const float* dy,
                                                       float* dgates, int T, int H, int per_group) {
  extern __shared__ float sm[];                   // dh[H] | dc[H] | dg[4H]
  float* dh = sm; float* dc = sm + H; float* dg = sm + 2*H;
  const int b = blockIdx.x;
  w_hh += (long long)(b / per_group)*4*H*H;
  for (int i = threadIdx.x; i < H; i += 256) { dh[i] = 0.f; dc[i] = 0.f; }
  __syncthreads();
  for (int t = T - 1; t >= 0; --t) {
    const float* a = act + ((long long)b*T + t)*4*H;
    for (int i = threadIdx.x; i < H; i += 256) {
      const float ig = a[i], fg = a[H + i], gg = a[2*H + i], og = a[3*H + i];
      const float c = cs[((long long)b*T + t)*H + i];
      const float cprev = t > 0 ? cs[((long long)b*T + t - 1)*H + i] : 0.f;
      const float tc = tanhf(c);
      const float dht = dh[i] + dy[((long long)b*T + t)*H + i];
      const float dct = dc[i] + dht*og*(1.f - tc*tc);
      dg[i] = dct*gg*ig*(1.f - ig);
      dg[H + i] = dct*cprev*fg*(1.f - fg);
      dg[2*H + i] = dct*ig*(1.f - gg*gg);
      dg[3*H + i] = dht*tc*og*(1.f - og);
      dc[i] = dct*fg;
    }
    __syncthreads();
    float* out = dgates + ((long long)b*T + t)*4*H;
    for (int r = threadIdx.x; r < 4*H; r += 256) out[r] = dg[r];
    // dh_{t-1} = W_hh^T dg
    for (int k = threadIdx.x; k < H; k += 256) {
      float acc = 0.f;
      for (int r = 0; r < 4*H; ++r) acc += w_hh[(long long)r*H + k]*dg[r];
      dh[k] = acc;
    }
    __syncthreads();
  }
}

// gradient of DCCRN.apply_mask with respect to the mask (the input spectrum needs none)
__global__ __launch_bounds__(256) void dccrn_mask_bwd_kernel(const float* xr, const float* xi,
                                                             const float* mr, const float* mi,
                                                             const float2* gout, float* dmr,
                                                             float* dmi, long long n, long long bs) {
  xr += blockIdx.y*bs; xi += blockIdx.y*bs; mr += blockIdx.y*bs; mi += blockIdx.y*bs; gout += blockIdx.y*n;
  dmr += blockIdx.y*bs; dmi += blockIdx.y*bs;
  for (long long i = (long long)blockIdx.x*blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x*blockDim.x) {
    const float a = xr[i], b = xi[i];
    const float in_mag = sqrtf(a*a + b*b), in_phase = atan2f(b, a);
    const float pr0 = mr[i], pi = mi[i];
    const float r1 = sqrtf(pr0*pr0 + pi*pi + 1e-7f);
    const float M = tanhf(r1);
    const float pr = pr0 == 0.f ? 1e-7f : pr0;
    const float ph = in_phase + atan2f(pi, pr);
    const float cs = cosf(ph), sn = sinf(ph);
    const float2 g = gout[i];
    const float dM = in_mag*(g.x*cs + g.y*sn);
    const float dP = in_mag*M*(-g.x*sn + g.y*cs);
    const float dMr = (1.f - M*M)/r1;
    const float r2 = pr*pr + pi*pi;
    dmr[i] = dM*dMr*pr0 + dP*(-pi/r2);
    dmi[i] = dM*dMr*pi + dP*(pr/r2);
  }
}

// dy[q] /= window-square envelope of torch.istft at sample q (centre padding n/2)
__global__ __launch_bounds__(256) void env_divide_kernel(const float* dy, const float* win,
                                                         float* out, int rows, int len, int n,
                                                         int hop, int F) {
  const long long total = (long long)rows*len;
  GRID_STRIDE(idx, total) {
    const int q = (int)(idx % len);
    const int pos = q + n/2;
    int t_hi = pos/hop; if (t_hi > F - 1) t_hi = F - 1;
    int t_lo = (pos - n + hop)/hop; if (pos - n + 1 <= 0) t_lo = 0;
    if (t_lo < 0) t_lo = 0;
    float env = 0.f;
    for (int t = t_lo; t <= t_hi; ++t) {
      const int m = pos - t*hop;
      if (m >= 0 && m < n) env += win[m]*win[m];
    }
    out[idx] = dy[idx]/env;
  }
}

// bias of the packed complex layer: [br - bi | br + bi] (dccrn.py:231-235: real = M_r(x_r) - M_i(x_i), imag = M_r(x_i) +
// M_i(x_r), each module adds its own bias), and its adjoint on the channel sums s = [s_r | s_i] of the output gradient:
// d br = s_r + s_i, d bi = s_i - s_r
__global__ __launch_bounds__(256) void cbias_pack_kernel(const float* br, const float* bi, float* out, int C) {
  const int c = blockIdx.x*256 + threadIdx.x;
  if (c < C) { out[c] = br[c] - bi[c]; out[C + c] = br[c] + bi[c]; }
}
__global__ __launch_bounds__(256) void cbias_unpack_kernel(const float* s, float* dbr, float* dbi, int C) {
  const int c = blockIdx.x*256 + threadIdx.x;
  if (c < C) { dbr[c] = s[c] + s[C + c]; dbi[c] = s[C + c] - s[c]; }
}
int red_slices(long long n) {
  long long s = (n + 16383)/16384;
  return (int)(s < 1 ? 1 : (s > kRedSlices ? kRedSlices : s));
}

}  // namespace

extern "C" {

int brv_conv2d_forward(const float* x, const float* w, const float* bias, float* y, int64_t B,
                       int64_t Cin, int64_t H, int64_t W, int64_t Cout, int64_t kh, int64_t kw,
                       int64_t sh, int64_t sw, int64_t ph, int64_t pw, int64_t x_batch_stride,
                       int64_t y_batch_stride, int accumulate, float sign, brv_stream_t stream) {
  ConvGeom g;
  g.B = (int)B; g.Cin = (int)Cin; g.H = (int)H; g.W = (int)W; g.Cout = (int)Cout;
  g.kh = (int)kh; g.kw = (int)kw; g.sh = (int)sh; g.sw = (int)sw; g.ph = (int)ph; g.pw = (int)pw;
  g.Ho = (int)((H + 2*ph - kh)/sh + 1); g.Wo = (int)((W + 2*pw - kw)/sw + 1);
  g.x_bs = x_batch_stride; g.y_bs = y_batch_stride;
  if (B < 1 || g.Ho < 1 || g.Wo < 1) return -1;
  hipLaunchKernelGGL(conv2d_fwd_kernel, flat_grid((long long)B*Cout*g.Ho*g.Wo), dim3(256), 0,
                     (hipStream_t)stream, x, w, bias, y, g, accumulate, sign);
  DC_OK(hipGetLastError());
  return 0;
}

int brv_conv_transpose2d_forward(const float* x, const float* w, const float* bias, float* y,
                                 int64_t B, int64_t Cin, int64_t H, int64_t W, int64_t Cout,
                                 int64_t kh, int64_t kw, int64_t sh, int64_t sw, int64_t ph,
                                 int64_t pw, int64_t oph, int64_t opw, int64_t x_batch_stride,
                                 int64_t y_batch_stride, int accumulate, float sign,
                                 brv_stream_t stream) {
  ConvGeom g;
  g.B = (int)B; g.Cin = (int)Cin; g.H = (int)H; g.W = (int)W; g.Cout = (int)Cout;
  g.kh = (int)kh; g.kw = (int)kw; g.sh = (int)sh; g.sw = (int)sw; g.ph = (int)ph; g.pw = (int)pw;
  g.Ho = (int)((H - 1)*sh - 2*ph + kh + oph); g.Wo = (int)((W - 1)*sw - 2*pw + kw + opw);
  g.x_bs = x_batch_stride; g.y_bs = y_batch_stride;
  if (B < 1 || g.Ho < 1 || g.Wo < 1) return -1;
  hipLaunchKernelGGL(conv_transpose2d_fwd_kernel, flat_grid((long long)B*Cout*g.Ho*g.Wo),
                     dim3(256), 0, (hipStream_t)stream, x, w, bias, y, g, accumulate, sign);
  DC_OK(hipGetLastError());
  return 0;
}

}  // extern "C"

template <typename TI, typename TO>
static int bn_forward_any(const TI* x, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, const float* prelu_slope,
                            TO* y, float* save_mean, float* save_invstd, int64_t B, int64_t C,
                            int64_t HW, float eps, float momentum, int training,
                            brv_stream_t stream) {
  if (B < 1 || C < 1 || HW < 1) return -1;
  constexpr bool kF32 = sizeof(TO) == 4 && sizeof(TI) == 4;
  if (!kF32 && ((HW & 3) || C > 65535 || B > 1024 || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15)))
    return -1;                                  // bf16 input / output exist in the four-elements-per-access form only
  hipStream_t st = (hipStream_t)stream;
  if (training) {
    const bool vec = (HW & 3) == 0 && B <= 1024 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    const int pieces = vec ? (int)(red_slices(HW) > 4 ? 4 : red_slices(HW)) : 0;
    const int slices = vec ? (int)B*pieces : red_slices(B*HW);
    double* part = nullptr;
    DC_OK(hipMallocAsync((void**)&part, (size_t)C*slices*2*sizeof(double), st));
    if (vec) hipLaunchKernelGGL(bn_stats_part4_kernel<TI>, dim3((unsigned)C, (unsigned)slices), dim3(256), 0, st,
                                x, (int)C, (long long)HW, pieces, part);
    else if constexpr (kF32) hipLaunchKernelGGL(bn_stats_part_kernel, dim3((unsigned)C, (unsigned)slices), dim3(256), 0, st,
                       x, (int)B, (int)C, (long long)HW, part);
    hipLaunchKernelGGL(bn_stats_kernel, dim3((unsigned)((C + 255)/256)), dim3(256), 0, st, part,
                       slices, (int)B, (int)C, (long long)HW, save_mean, save_invstd, running_mean,
                       running_var, eps, momentum);
    DC_OK(hipFreeAsync(part, st));
  } else {
    DC_OK(hipMemcpyAsync(save_mean, running_mean, (size_t)C*4, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(invstd_from_var_kernel, dim3((unsigned)((C + 255)/256)), dim3(256), 0, st,
                       running_var, save_invstd, (int)C, eps);
  }
  const long long total = B*C*HW;
  if ((HW & 3) == 0 && C <= 65535 && B <= 65535 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0) {
    long long gx = (HW/4 + 1023)/1024;                 // four 16-byte accesses per thread
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL((bn_apply4_kernel<TI, TO>), dim3((unsigned)gx, (unsigned)C, (unsigned)B), dim3(256), 0, st, x, save_mean,
                       save_invstd, gamma, beta, prelu_slope, y, (int)C, (long long)HW);
  } else if constexpr (kF32)
  hipLaunchKernelGGL(bn_apply_kernel, flat_grid(total), dim3(256), 0, st, x, save_mean,
                     save_invstd, gamma, beta, prelu_slope, y, (int)C, (long long)HW, total);
  DC_OK(hipGetLastError());
  return 0;
}

extern "C" {

int brv_batchnorm2d_forward(const float* x, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, const float* prelu_slope,
                            float* y, float* save_mean, float* save_invstd, int64_t B, int64_t C,
                            int64_t HW, float eps, float momentum, int training,
                            brv_stream_t stream) {
  return bn_forward_any<float, float>(x, gamma, beta, running_mean, running_var, prelu_slope, y, save_mean, save_invstd, B, C,
                                      HW, eps, momentum, training, stream);
}

int brv_batchnorm2d_forward_bf16(const float* x, const float* gamma, const float* beta,
                                 float* running_mean, float* running_var, const float* prelu_slope,
                                 void* y16, float* save_mean, float* save_invstd, int64_t B, int64_t C,
                                 int64_t HW, float eps, float momentum, int training,
                                 brv_stream_t stream) {
  return bn_forward_any<float, bf16_t>(x, gamma, beta, running_mean, running_var, prelu_slope, (bf16_t*)y16, save_mean,
                                       save_invstd, B, C, HW, eps, momentum, training, stream);
}

int brv_batchnorm2d_forward_bf16io(const void* x16, const float* gamma, const float* beta,
                                   float* running_mean, float* running_var, const float* prelu_slope,
                                   void* y16, float* save_mean, float* save_invstd, int64_t B, int64_t C,
                                   int64_t HW, float eps, float momentum, int training,
                                   brv_stream_t stream) {
  return bn_forward_any<bf16_t, bf16_t>((const bf16_t*)x16, gamma, beta, running_mean, running_var, prelu_slope,
                                        (bf16_t*)y16, save_mean, save_invstd, B, C, HW, eps, momentum, training, stream);
}

int brv_lstm_recurrent_forward(const float* gates_in, const float* w_hh, const float* bias,
                               float* y, float* act, float* cs, int64_t B, int64_t T, int64_t H,
                               int64_t groups, brv_stream_t stream) {
  if (B < 1 || T < 1 || H < 1 || groups < 1 || B % groups) return -1;
  const int per_group = (int)(B/groups);
  if (H == 128) {
    if (act && cs)
      hipLaunchKernelGGL(lstm_fwd_quad_kernel<true>, dim3((unsigned)B), dim3(512), 0, (hipStream_t)stream,
                         gates_in, w_hh, bias, y, act, cs, (int)T, per_group);
    else
      hipLaunchKernelGGL(lstm_fwd_quad_kernel<false>, dim3((unsigned)B), dim3(512), 0, (hipStream_t)stream,
                         gates_in, w_hh, bias, y, act, cs, (int)T, per_group);
  } else if (H <= kLstmRegH && H % 16 == 0)
    hipLaunchKernelGGL(lstm_fwd_reg_kernel, dim3((unsigned)B), dim3((unsigned)(4*H)), 0,
                       (hipStream_t)stream, gates_in, w_hh, bias, y, act, cs, (int)T, (int)H,
                       per_group);
  else
  hipLaunchKernelGGL(lstm_recurrent_kernel, dim3((unsigned)B), dim3(256), (size_t)6*H*4,
                     (hipStream_t)stream, gates_in, w_hh, bias, y, act, cs, (int)T, (int)H,
                     per_group);
  DC_OK(hipGetLastError());
  return 0;
}

int brv_lstm_recurrent_forward_bf16(const float* gates_in, const float* w_hh, const float* bias,
                                    float* y, float* act, float* cs, int64_t B, int64_t T, int64_t H,
                                    int64_t groups, brv_stream_t stream) {
  if (B < 1 || T < 1 || groups < 1 || B % groups) return -1;
  if (H != 128) return -1;                               // (brv_lstm_recurrent_bf16_supported)
  const int per_group = (int)(B/groups);
  if (act && cs)
    hipLaunchKernelGGL(lstm_fwd_mv_kernel<true>, dim3((unsigned)B), dim3(512), 0, (hipStream_t)stream,
                       gates_in, w_hh, bias, y, act, cs, (int)T, per_group);
  else
    hipLaunchKernelGGL(lstm_fwd_mv_kernel<false>, dim3((unsigned)B), dim3(512), 0, (hipStream_t)stream,
                       gates_in, w_hh, bias, y, act, cs, (int)T, per_group);
  DC_OK(hipGetLastError());
  return 0;
}

int brv_lstm_recurrent_backward_bf16(const float* act, const float* cs, const float* w_hh,
                                     const float* dy, float* dgates, int64_t B, int64_t T, int64_t H,
                                     int64_t groups, brv_stream_t stream) {
  if (B < 1 || T < 1 || groups < 1 || B % groups) return -1;
  if (H != 128) return -1;
  hipLaunchKernelGGL(lstm_bwd_mv_kernel, dim3((unsigned)B), dim3(512), 0, (hipStream_t)stream, act, cs,
                     w_hh, dy, dgates, (int)T, (int)(B/groups));
  DC_OK(hipGetLastError());
  return 0;
}

int brv_lstm_recurrent_bf16_supported(int64_t H) { return H == 128; }

int brv_combine(const float* a, const float* b, float* out, int64_t n, float sign,
                brv_stream_t stream) {
  if (n < 1) return -1;
  hipLaunchKernelGGL(combine_kernel, flat_grid(n), dim3(256), 0, (hipStream_t)stream, a, b, out,
                     (long long)n, sign);
  DC_OK(hipGetLastError());
  return 0;
}

int brv_complex_mix_forward(const float* o, float* real, float* imag, int64_t n, brv_stream_t stream) {
  if (!o || !real || !imag || n < 4 || (n & 3) ||
      ((reinterpret_cast<uintptr_t>(o) | reinterpret_cast<uintptr_t>(real) | reinterpret_cast<uintptr_t>(imag)) & 15)) return -1;
  hipLaunchKernelGGL(complex_mix_fwd_kernel, flat_grid(n/4), dim3(256), 0, (hipStream_t)stream, (const float4*)o,
                     (float4*)real, (float4*)imag, (long long)(n/4));
  DC_OK(hipGetLastError());
  return 0;
}

int brv_complex_mix_backward(const float* greal, const float* gimag, float* dout, int64_t n, brv_stream_t stream) {
  if (!greal || !gimag || !dout || n < 4 || (n & 3) ||
      ((reinterpret_cast<uintptr_t>(greal) | reinterpret_cast<uintptr_t>(gimag) | reinterpret_cast<uintptr_t>(dout)) & 15)) return -1;
  hipLaunchKernelGGL(complex_mix_bwd_kernel, flat_grid(n/4), dim3(256), 0, (hipStream_t)stream, (const float4*)greal,
                     (const float4*)gimag, (float4*)dout, (long long)(n/4));
  DC_OK(hipGetLastError());
  return 0;
}

int brv_dccrn_apply_mask(const float* xr, const float* xi, const float* mr, const float* mi,
                         float* out, int64_t n, brv_stream_t stream) {
  if (n < 1) return -1;
  hipLaunchKernelGGL(dccrn_mask_kernel, flat_grid(n), dim3(256), 0, (hipStream_t)stream, xr, xi,
                     mr, mi, (float2*)out, (long long)n, 0LL);
  DC_OK(hipGetLastError());
  return 0;
}

int brv_dccrn_apply_mask_batched(const float* x, const float* mask, float* out, int64_t B, int64_t n,
                                 brv_stream_t stream) {
  if (!x || !mask || !out || B < 1 || B > 65535 || n < 1) return -1;
  dim3 grid = flat_grid(n);
  if (grid.x > 4096) grid.x = 4096;
  grid.y = (unsigned)B;
  hipLaunchKernelGGL(dccrn_mask_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, x + n, mask, mask + n,
                     (float2*)out, (long long)n, (long long)(2*n));
  DC_OK(hipGetLastError());
  return 0;
}

int brv_conv2d_wgrad(const float* x, const float* dy, float* dw, float* dbias, int64_t B,
                     int64_t Cin, int64_t H, int64_t W, int64_t Cout, int64_t Ho, int64_t Wo,
                     int64_t kh, int64_t kw, int64_t sh, int64_t sw, int64_t ph, int64_t pw,
                     int64_t x_batch_stride, int64_t dy_batch_stride, int accumulate, float sign,
                     brv_stream_t stream) {
  ConvGeom g;
  g.B = (int)B; g.Cin = (int)Cin; g.H = (int)H; g.W = (int)W; g.Cout = (int)Cout;
  g.kh = (int)kh; g.kw = (int)kw; g.sh = (int)sh; g.sw = (int)sw; g.ph = (int)ph; g.pw = (int)pw;
  g.Ho = (int)Ho; g.Wo = (int)Wo; g.x_bs = x_batch_stride; g.y_bs = dy_batch_stride;
  if (B < 1 || Ho < 1 || Wo < 1) return -1;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(conv2d_wgrad_kernel, dim3((unsigned)(Cout*Cin*kh*kw)), dim3(256), 0, st, x,
                     dy, dw, g, accumulate, sign);
  if (dbias)
    hipLaunchKernelGGL(channel_sum_kernel, dim3((unsigned)Cout), dim3(256), 0, st, dy, dbias,
                       (int)B, (long long)Ho*Wo, (long long)dy_batch_stride, accumulate, sign);
  DC_OK(hipGetLastError());
  return 0;
}

}  // extern "C"

template <typename TI, typename TO, typename TG = float>
static int bn_backward_any(const TI* x, const TG* dy, const TG* dy2, const float* save_mean,
                             const float* save_invstd, const float* gamma, const float* beta,
                             const float* prelu_slope, TO* dx, float* dgamma, float* dbeta,
                             float* dslope_partial, float* dx_sums, int64_t B, int64_t C, int64_t HW,
                             brv_stream_t stream) {
  if (B < 1 || C < 1 || HW < 1) return -1;
  constexpr bool kF32 = sizeof(TO) == 4 && sizeof(TI) == 4 && sizeof(TG) == 4;
  if (dy2 && ((HW & 3) || B > 1024 || (reinterpret_cast<uintptr_t>(dy2) & 15))) return -1;     // (the vector kernels only)
  if ((!kF32 || dx_sums || dy2) && ((HW & 3) || C > 65535 || B > (kF32 ? 65535 : 1024) ||
      ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15)))
    return -1;                                  // bf16 output / dx sums: the 16-byte form only
  hipStream_t st = (hipStream_t)stream;
  const bool vec = (HW & 3) == 0 && B <= 1024 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0;
  const int pieces = vec ? (int)(red_slices(HW) > 4 ? 4 : red_slices(HW)) : 0;
  const int slices = vec ? (int)B*pieces : red_slices(B*HW);
  double* part = nullptr;
  DC_OK(hipMallocAsync((void**)&part, (size_t)C*slices*3*sizeof(double), st));
  if (vec) hipLaunchKernelGGL((bn_bwd_stats4_kernel<TI, TG>), dim3((unsigned)C, (unsigned)slices), dim3(256), 0, st, x,
                              dy, dy2, save_mean, save_invstd, gamma, beta, prelu_slope, (int)C, (long long)HW,
                              pieces, part);
  else if constexpr (kF32) hipLaunchKernelGGL(bn_bwd_stats_kernel, dim3((unsigned)C, (unsigned)slices), dim3(256), 0, st, x,
                     dy, save_mean, save_invstd, gamma, beta, prelu_slope, (int)B, (int)C,
                     (long long)HW, part);
  hipLaunchKernelGGL(bn_bwd_final_kernel, dim3((unsigned)((C + 255)/256)), dim3(256), 0, st, part,
                     slices, (int)C, dgamma, dbeta, dslope_partial);
  DC_OK(hipFreeAsync(part, st));
  const long long total = B*C*HW;
  if ((HW & 3) == 0 && C <= 65535 && B <= 65535 &&
      ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0) {
    long long gx = (HW/4 + 1023)/1024;
    if (gx < 1) gx = 1;
    double* sums = nullptr;
    if (dx_sums) DC_OK(hipMallocAsync((void**)&sums, (size_t)C*B*gx*sizeof(double), st));
    hipLaunchKernelGGL((bn_bwd_apply4_kernel<TI, TG, TO>), dim3((unsigned)gx, (unsigned)C, (unsigned)B), dim3(256), 0, st, x, dy, dy2,
                       save_mean, save_invstd, gamma, beta, prelu_slope, dgamma, dbeta, dx, (int)C, (long long)HW,
                       1.f/(float)(B*HW), sums);
    if (dx_sums) {
      hipLaunchKernelGGL(bn_dxsum_kernel, dim3((unsigned)((C + 255)/256)), dim3(256), 0, st, sums, (int)(B*gx), (int)C,
                         dx_sums);
      DC_OK(hipFreeAsync(sums, st));
    }
  } else if constexpr (kF32)
  hipLaunchKernelGGL(bn_bwd_apply_kernel, flat_grid(total), dim3(256), 0, st, x, dy, save_mean,
                     save_invstd, gamma, beta, prelu_slope, dgamma, dbeta, dx, (int)C,
                     (long long)HW, total, 1.f/(float)(B*HW));
  DC_OK(hipGetLastError());
  return 0;
}

extern "C" {

int brv_batchnorm2d_backward(const float* x, const float* dy, const float* save_mean,
                             const float* save_invstd, const float* gamma, const float* beta,
                             const float* prelu_slope, float* dx, float* dgamma, float* dbeta,
                             float* dslope_partial, int64_t B, int64_t C, int64_t HW,
                             brv_stream_t stream) {
  return bn_backward_any<float, float, float>(x, dy, nullptr, save_mean, save_invstd, gamma, beta, prelu_slope, dx, dgamma, dbeta,
                                dslope_partial, nullptr, B, C, HW, stream);
}

int brv_batchnorm2d_backward_bf16(const float* x, const float* dy, const float* save_mean,
                                  const float* save_invstd, const float* gamma, const float* beta,
                                  const float* prelu_slope, void* dx16, float* dgamma, float* dbeta,
                                  float* dslope_partial, float* dx_sums, int64_t B, int64_t C, int64_t HW,
                                  brv_stream_t stream) {
  return bn_backward_any<float, bf16_t, float>(x, dy, nullptr, save_mean, save_invstd, gamma, beta, prelu_slope, (bf16_t*)dx16, dgamma, dbeta,
                                        dslope_partial, dx_sums, B, C, HW, stream);
}

int brv_batchnorm2d_backward_bf16io(const void* x16, const float* dy, const float* save_mean,
                                    const float* save_invstd, const float* gamma, const float* beta,
                                    const float* prelu_slope, void* dx16, float* dgamma, float* dbeta,
                                    float* dslope_partial, float* dx_sums, int64_t B, int64_t C, int64_t HW,
                                    brv_stream_t stream) {
  return bn_backward_any<bf16_t, bf16_t, float>((const bf16_t*)x16, dy, nullptr, save_mean, save_invstd, gamma, beta, prelu_slope,
                                         (bf16_t*)dx16, dgamma, dbeta, dslope_partial, dx_sums, B, C, HW, stream);
}

int brv_batchnorm2d_backward_ex(const void* x, int32_t x_bf16, const void* dy_, const void* dy2_, int32_t dy_bf16,
                                const float* save_mean,
                                const float* save_invstd, const float* gamma, const float* beta,
                                const float* prelu_slope, void* dx, int32_t dx_bf16, float* dgamma, float* dbeta,
                                float* dslope_partial, float* dx_sums, int64_t B, int64_t C, int64_t HW,
                                brv_stream_t stream) {
  if (x_bf16 && !dx_bf16) return -1;
  if (dy_bf16) {           // the gradients with respect to the output as bf16 (what autocast hands a bf16 activation)
    const bf16_t* g = (const bf16_t*)dy_; const bf16_t* g2 = (const bf16_t*)dy2_;
    if (x_bf16) return bn_backward_any<bf16_t, bf16_t, bf16_t>((const bf16_t*)x, g, g2, save_mean, save_invstd, gamma, beta,
                                                               prelu_slope, (bf16_t*)dx, dgamma, dbeta, dslope_partial,
                                                               dx_sums, B, C, HW, stream);
    if (dx_bf16) return bn_backward_any<float, bf16_t, bf16_t>((const float*)x, g, g2, save_mean, save_invstd, gamma, beta,
                                                               prelu_slope, (bf16_t*)dx, dgamma, dbeta, dslope_partial,
                                                               dx_sums, B, C, HW, stream);
    return bn_backward_any<float, float, bf16_t>((const float*)x, g, g2, save_mean, save_invstd, gamma, beta, prelu_slope,
                                                 (float*)dx, dgamma, dbeta, dslope_partial, dx_sums, B, C, HW, stream);
  }
  const float* dy = (const float*)dy_; const float* dy2 = (const float*)dy2_;
  if (x_bf16) return bn_backward_any<bf16_t, bf16_t>((const bf16_t*)x, dy, dy2, save_mean, save_invstd, gamma, beta,
                                                     prelu_slope, (bf16_t*)dx, dgamma, dbeta, dslope_partial, dx_sums, B, C,
                                                     HW, stream);
  if (dx_bf16) return bn_backward_any<float, bf16_t>((const float*)x, dy, dy2, save_mean, save_invstd, gamma, beta,
                                                     prelu_slope, (bf16_t*)dx, dgamma, dbeta, dslope_partial, dx_sums, B, C,
                                                     HW, stream);
  return bn_backward_any<float, float>((const float*)x, dy, dy2, save_mean, save_invstd, gamma, beta, prelu_slope,
                                       (float*)dx, dgamma, dbeta, dslope_partial, dx_sums, B, C, HW, stream);
}

int brv_lstm_recurrent_backward(const float* act, const float* cs, const float* w_hh,
                                const float* dy, float* dgates, int64_t B, int64_t T, int64_t H,
                                int64_t groups, brv_stream_t stream) {
  if (B < 1 || T < 1 || H < 1 || groups < 1 || B % groups) return -1;
  const int per_group = (int)(B/groups);
  if (H == 128)
    hipLaunchKernelGGL(lstm_bwd_quad_kernel, dim3((unsigned)B), dim3(512), 0, (hipStream_t)stream, act, cs,
                       w_hh, dy, dgates, (int)T, per_group);
  else if (H <= kLstmRegH && H % 16 == 0)
    hipLaunchKernelGGL(lstm_bwd_reg_kernel, dim3((unsigned)B), dim3((unsigned)(4*H)), 0,
                       (hipStream_t)stream, act, cs, w_hh, dy, dgates, (int)T, (int)H, per_group);
  else
  hipLaunchKernelGGL(lstm_bwd_kernel, dim3((unsigned)B), dim3(256), (size_t)6*H*4,
                     (hipStream_t)stream, act, cs, w_hh, dy, dgates, (int)T, (int)H, per_group);
  DC_OK(hipGetLastError());
  return 0;
}

int brv_dccrn_apply_mask_backward(const float* xr, const float* xi, const float* mr,
                                  const float* mi, const float* gout, float* dmr, float* dmi,
                                  int64_t n, brv_stream_t stream) {
  if (n < 1) return -1;
  hipLaunchKernelGGL(dccrn_mask_bwd_kernel, flat_grid(n), dim3(256), 0, (hipStream_t)stream, xr,
                     xi, mr, mi, (const float2*)gout, dmr, dmi, (long long)n, 0LL);
  DC_OK(hipGetLastError());
  return 0;
}

int brv_dccrn_apply_mask_backward_batched(const float* x, const float* mask, const float* gout, float* dmask,
                                          int64_t B, int64_t n, brv_stream_t stream) {
  if (!x || !mask || !gout || !dmask || B < 1 || B > 65535 || n < 1) return -1;
  dim3 grid = flat_grid(n);
  if (grid.x > 4096) grid.x = 4096;
  grid.y = (unsigned)B;
  hipLaunchKernelGGL(dccrn_mask_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, x + n, mask, mask + n,
                     (const float2*)gout, dmask, dmask + n, (long long)n, (long long)(2*n));
  DC_OK(hipGetLastError());
  return 0;
}

int brv_istft_env_divide(const float* dy, const float* window, float* out, int64_t rows,
                         int64_t length, int64_t frame_length, int64_t hop_length, int64_t frames,
                         brv_stream_t stream) {
  if (rows < 1 || length < 1) return -1;
  hipLaunchKernelGGL(env_divide_kernel, flat_grid(rows*length), dim3(256), 0, (hipStream_t)stream,
                     dy, window, out, (int)rows, (int)length, (int)frame_length, (int)hop_length,
                     (int)frames);
  DC_OK(hipGetLastError());
  return 0;
}

extern "C++" {
template <typename ColT>
static int im2col_any(const float* x, ColT* col, int64_t B, int64_t C, int64_t H, int64_t W, int64_t kh,
                      int64_t kw, int64_t sh, int64_t sw, int64_t ph, int64_t pw, int64_t Ho,
                      int64_t Wo, brv_stream_t stream) {
  if (B < 1 || C < 1 || Ho < 1 || Wo < 1) return -1;
  ConvGeom g;
  g.B = (int)B; g.Cin = (int)C; g.H = (int)H; g.W = (int)W; g.Cout = 0;
  g.kh = (int)kh; g.kw = (int)kw; g.sh = (int)sh; g.sw = (int)sw; g.ph = (int)ph; g.pw = (int)pw;
  g.Ho = (int)Ho; g.Wo = (int)Wo; g.x_bs = C*H*W; g.y_bs = 0;
  if (C*kh*kw > 65535 || B > 65535 || H*W >= (1LL << 31) || Ho*Wo >= (1LL << 31)) return -2;
  hipLaunchKernelGGL(im2col_kernel<ColT>, dim3((unsigned)((Ho*Wo + 1023)/1024), (unsigned)(C*kh*kw),
                                               (unsigned)B), dim3(256), 0, (hipStream_t)stream, x, col, g);
  DC_OK(hipGetLastError());
  return 0;
}
template <typename ColT>
static int col2im_any(const ColT* col, const float* bias, float* y, int64_t B, int64_t C, int64_t H,
                      int64_t W, int64_t kh, int64_t kw, int64_t sh, int64_t sw, int64_t ph, int64_t pw,
                      int64_t Ho, int64_t Wo, brv_stream_t stream) {
  if (B < 1 || C < 1 || H < 1 || W < 1) return -1;
  ConvGeom g;
  g.B = (int)B; g.Cin = (int)C; g.H = (int)H; g.W = (int)W; g.Cout = 0;
  g.kh = (int)kh; g.kw = (int)kw; g.sh = (int)sh; g.sw = (int)sw; g.ph = (int)ph; g.pw = (int)pw;
  g.Ho = (int)Ho; g.Wo = (int)Wo; g.x_bs = 0; g.y_bs = C*H*W;
  if (C > 65535 || B > 65535 || H*W >= (1LL << 31) || Ho*Wo >= (1LL << 31)) return -2;
  hipLaunchKernelGGL(col2im_kernel<ColT>, dim3((unsigned)((H*W + 255)/256), (unsigned)C, (unsigned)B),
                     dim3(256), 0, (hipStream_t)stream, col, bias, y, g);
  DC_OK(hipGetLastError());
  return 0;
}
}  // extern "C++"
int brv_im2col(const float* x, float* col, int64_t B, int64_t C, int64_t H, int64_t W, int64_t kh,
               int64_t kw, int64_t sh, int64_t sw, int64_t ph, int64_t pw, int64_t Ho, int64_t Wo,
               brv_stream_t stream) {
  return im2col_any<float>(x, col, B, C, H, W, kh, kw, sh, sw, ph, pw, Ho, Wo, stream);
}
int brv_col2im(const float* col, const float* bias, float* y, int64_t B, int64_t C, int64_t H,
               int64_t W, int64_t kh, int64_t kw, int64_t sh, int64_t sw, int64_t ph, int64_t pw,
               int64_t Ho, int64_t Wo, brv_stream_t stream) {
  return col2im_any<float>(col, bias, y, B, C, H, W, kh, kw, sh, sw, ph, pw, Ho, Wo, stream);
}
int brv_im2col_bf16(const float* x, void* col, int64_t B, int64_t C, int64_t H, int64_t W, int64_t kh,
                    int64_t kw, int64_t sh, int64_t sw, int64_t ph, int64_t pw, int64_t Ho,
                    int64_t Wo, brv_stream_t stream) {
  return im2col_any<bf16_t>(x, (bf16_t*)col, B, C, H, W, kh, kw, sh, sw, ph, pw, Ho, Wo, stream);
}
int brv_col2im_bf16(const void* col, const float* bias, float* y, int64_t B, int64_t C, int64_t H,
                    int64_t W, int64_t kh, int64_t kw, int64_t sh, int64_t sw, int64_t ph, int64_t pw,
                    int64_t Ho, int64_t Wo, brv_stream_t stream) {
  return col2im_any<bf16_t>((const bf16_t*)col, bias, y, B, C, H, W, kh, kw, sh, sw, ph, pw, Ho, Wo,
                            stream);
}
int brv_complex_weight_pack(const float* wr, const float* wi, float* wc, int64_t R, int64_t C,
                            float sign, brv_stream_t stream) {
  if (R < 1 || C < 1) return -1;
  hipLaunchKernelGGL(cweight_pack_kernel, flat_grid(4*R*C), dim3(256), 0, (hipStream_t)stream, wr,
                     wi, wc, (int)R, (int)C, sign);
  DC_OK(hipGetLastError());
  return 0;
}
int brv_complex_weight_unpack(const float* dwc, float* dwr, float* dwi, int64_t R, int64_t C,
                              float sign, brv_stream_t stream) {
  if (R < 1 || C < 1) return -1;
  hipLaunchKernelGGL(cweight_unpack_kernel, flat_grid(R*C), dim3(256), 0, (hipStream_t)stream, dwc,
                     dwr, dwi, (int)R, (int)C, sign);
  DC_OK(hipGetLastError());
  return 0;
}

int brv_complex_bias_pack(const float* br, const float* bi, float* out, int64_t C, brv_stream_t stream) {
  if (!br || !bi || !out || C < 1) return -1;
  hipLaunchKernelGGL(cbias_pack_kernel, dim3((unsigned)((C + 255)/256)), dim3(256), 0, (hipStream_t)stream, br, bi,
                     out, (int)C);
  DC_OK(hipGetLastError());
  return 0;
}
int brv_complex_bias_unpack(const float* sums, float* dbr, float* dbi, int64_t C, brv_stream_t stream) {
  if (!sums || !dbr || !dbi || C < 1) return -1;
  hipLaunchKernelGGL(cbias_unpack_kernel, dim3((unsigned)((C + 255)/256)), dim3(256), 0, (hipStream_t)stream, sums,
                     dbr, dbi, (int)C);
  DC_OK(hipGetLastError());
  return 0;
}

int brv_cplx_moments(const float* x, float* moments, int64_t B, int64_t C, int64_t HW,
                     brv_stream_t stream) {
  if (B < 1 || C < 1 || HW < 1) return -1;
  hipStream_t st = (hipStream_t)stream;
  const int slices = red_slices(B*HW);
  double* part = nullptr;
  DC_OK(hipMallocAsync((void**)&part, (size_t)C*slices*5*sizeof(double), st));
  hipLaunchKernelGGL(cplx_moments_part_kernel, dim3((unsigned)C, (unsigned)slices), dim3(256), 0, st,
                     x, (int)B, (int)C, (long long)HW, part);
  hipLaunchKernelGGL(slice_final_kernel, dim3((unsigned)((5*C + 255)/256)), dim3(256), 0, st, part,
                     slices, 5, (int)C, 1.0/((double)B*(double)HW), moments);
  DC_OK(hipFreeAsync(part, st));
  DC_OK(hipGetLastError());
  return 0;
}
int brv_cplx_affine_forward(const float* x, const float* A, const float* o, const float* prelu_slope,
                            float* y, int64_t B, int64_t C, int64_t HW, brv_stream_t stream) {
  if (B < 1 || C < 1 || HW < 1) return -1;
  const long long total = B*C*HW;
  hipLaunchKernelGGL(cplx_affine_fwd_kernel, flat_grid(total), dim3(256), 0, (hipStream_t)stream, x,
                     A, o, prelu_slope, y, (int)C, (long long)HW, total);
  DC_OK(hipGetLastError());
  return 0;
}
int brv_cplx_affine_backward(const float* x, const float* dy, const float* A, const float* o,
                             const float* prelu_slope, float* dx, float* dA, float* d_o,
                             float* dslope_partial, int64_t B, int64_t C, int64_t HW,
                             brv_stream_t stream) {
  if (B < 1 || C < 1 || HW < 1) return -1;
  hipStream_t st = (hipStream_t)stream;
  const int slices = red_slices(B*HW);
  double* part = nullptr;
  float* sums = nullptr;
  DC_OK(hipMallocAsync((void**)&part, (size_t)C*slices*7*sizeof(double), st));
  DC_OK(hipMallocAsync((void**)&sums, (size_t)7*C*sizeof(float), st));
  hipLaunchKernelGGL(cplx_affine_bwd_part_kernel, dim3((unsigned)C, (unsigned)slices), dim3(256), 0,
                     st, x, dy, A, o, prelu_slope, (int)B, (int)C, (long long)HW, part);
  hipLaunchKernelGGL(slice_final_kernel, dim3((unsigned)((7*C + 255)/256)), dim3(256), 0, st, part,
                     slices, 7, (int)C, 1.0, sums);
  DC_OK(hipMemcpyAsync(dA, sums, (size_t)4*C*4, hipMemcpyDeviceToDevice, st));
  DC_OK(hipMemcpyAsync(d_o, sums + 4*C, (size_t)2*C*4, hipMemcpyDeviceToDevice, st));
  if (dslope_partial)
    DC_OK(hipMemcpyAsync(dslope_partial, sums + 6*C, (size_t)C*4, hipMemcpyDeviceToDevice, st));
  DC_OK(hipFreeAsync(part, st));
  DC_OK(hipFreeAsync(sums, st));
  const long long total = B*C*HW;
  hipLaunchKernelGGL(cplx_affine_bwd_apply_kernel, flat_grid(total), dim3(256), 0, st, x, dy, A, o,
                     prelu_slope, (const float*)nullptr, dx, (int)C, (long long)HW, total);
  DC_OK(hipGetLastError());
  return 0;
}
int brv_cplx_moments_backward(const float* x, const float* gm, float* dx, int64_t B, int64_t C,
                              int64_t HW, brv_stream_t stream) {
  if (B < 1 || C < 1 || HW < 1) return -1;
  const long long total = B*C*HW;
  hipLaunchKernelGGL(cplx_affine_bwd_apply_kernel, flat_grid(total), dim3(256), 0,
                     (hipStream_t)stream, x, (const float*)nullptr, (const float*)nullptr,
                     (const float*)nullptr, (const float*)nullptr, gm, dx, (int)C, (long long)HW,
                     total);
  DC_OK(hipGetLastError());
  return 0;
}

}  // extern "C"
