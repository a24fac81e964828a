// Conv-TasNet with fp32 activations: the `use_amp=False` path (C ABI brv_ctn_f32_*).
//
// Reference being replaced: brever/models/convtasnet/convtasnet.py:66-97 run WITHOUT autocast
// (`enhance(x, use_amp=False)` of scripts/test_model.py, `BreverTrainer(use_amp=False)`), i.e.
// every tensor and every product in fp32. The bf16 path of convtasnet.hip is the throughput
// path; this one is the precision path: channels-last fp32 tensors [item][frame][channel] with
// no channel padding. Two forms of the same arithmetic:
//  * fused (channel counts multiples of 4, <= 1024; `fused_ok`): every product on gemm_f32_big.hip
//    (fp32 MFMA or split-bf16 MFMAs of fp32 accuracy, split reductions added in a fixed order with
//    scratch from this workspace), the work between the products as the streaming kernels of
//    ctn_f32_fused.cuh -- a layer norm's normalised tensor is never stored, its consumers rebuild
//    it from z; every frame / channel sum is taken where the operands are in registers;
//  * plain (any channel counts): one brv_gemm_f32 product per 1x1 convolution, one kernel per
//    mathematical step in between (the round-2 implementation, kept as the general fallback).
// Global (non-causal) and cumulative (causal) layer norms share one formulation: a per-frame
// table (mean_t, rstd_t) in the forward pass and (U_t, V_t) in the backward pass with
//   d prelu_out[t][c] = e[t][c] gain[c] rstd_t + U_t + p[t][c] V_t,
// where the tables are totals over the item's frames (gLN: nn.GroupNorm(1, C, eps=1e-8),
// convtasnet.py:267) or prefix / suffix sums (cLN: modules/normalization.py:5-62). Frame sums
// are fp32 over the channels of one frame, the scans over frames fp64.
// Per-channel parameter gradients are reduced in two deterministic stages (row slices, then
// a fold in slice order); the weight gradients are products accumulated into the flat gradient.
#include <hip/hip_runtime.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/brever_hip.h"
#include "common.cuh"
#include "gemm_f32_big.h"

using namespace brv;

namespace {

}  // namespace
extern "C" __attribute__((visibility("hidden"))) void brv_internal_set_error(const char* msg);   // convtasnet.hip: feeds brv_last_error()
namespace {
int fail32(int code, const std::string& msg) { brv_internal_set_error(msg.c_str()); return code; }

#define HIP_OK32(expr)                                                          \
  do {                                                                          \
    hipError_t e_ = (expr);                                                     \
    if (e_ != hipSuccess)                                                       \
      return fail32((int)e_, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)
#define OK32(expr) do { if (int r_ = (expr)) return r_; } while (0)

inline long long up(long long x, long long a) { return (x + a - 1)/a*a; }

struct Blk32 {
  long long conv_w, conv_b, dconv_w, dconv_b, res_w, res_b, skip_w, skip_b,
      n1_g, n1_b, n2_g, n2_b, prelu1, prelu2;
};
// parameter offsets in ConvTasNet.parameters() order (identical to the bf16 path)
struct Lay32 {
  int N, K, Bn, H, Sc, P, nb, S, hop, causal, layers;
  long long enc_w, dec_w, ln_g, ln_b, bott_w, bott_b, tcn_prelu, out_w, out_b, n_params;
  std::vector<Blk32> blk;
  int init(const brv_ctn_config* c) {
    if (!c) return fail32(-1, "null config");
    if (c->filters < 1 || c->filter_length < 2 || c->bottleneck_channels < 1 ||
        c->hidden_channels < 1 || c->skip_channels < 1 || c->layers < 1 || c->repeats < 1 ||
        c->output_sources < 1 || c->kernel_size < 1)
      return fail32(-1, "invalid Conv-TasNet hyper-parameters");
    if (c->kernel_size > 7) return fail32(-2, "kernel_size must be <= 7 in the fp32 HIP path");
    N = c->filters; K = c->filter_length; Bn = c->bottleneck_channels; H = c->hidden_channels;
    Sc = c->skip_channels; P = c->kernel_size; layers = c->layers; nb = c->layers*c->repeats;
    S = c->output_sources; hop = K/2; causal = c->causal != 0;
    long long o = 0;
    auto take = [&](long long n) { long long r = o; o += n; return r; };
    enc_w = take((long long)N*K); dec_w = take((long long)N*K);
    ln_g = take(N); ln_b = take(N);
    bott_w = take((long long)Bn*N); bott_b = take(Bn);
    blk.resize(nb);
    for (int i = 0; i < nb; ++i) {
      Blk32& b = blk[i];
      b.conv_w = take((long long)H*Bn); b.conv_b = take(H);
      b.dconv_w = take((long long)H*P); b.dconv_b = take(H);
      if (i < nb - 1) { b.res_w = take((long long)Bn*H); b.res_b = take(Bn); }
      else { b.res_w = -1; b.res_b = -1; }
      b.skip_w = take((long long)Sc*H); b.skip_b = take(Sc);
      b.n1_g = take(H); b.n1_b = take(H); b.n2_g = take(H); b.n2_b = take(H);
      b.prelu1 = take(1); b.prelu2 = take(1);
    }
    tcn_prelu = take(1);
    out_w = take((long long)S*N*Sc); out_b = take((long long)S*N);
    n_params = o;
    return 0;
  }
  long long frames(long long L) const {
    const long long pad = ((K - L) % hop + hop) % hop;     // Python modulo (convtasnet.py:115-120)
    const long long Lp = L + pad;
    return Lp < K ? 0 : (Lp - K)/hop + 1;
  }
};

constexpr int kSliceRows = 256;         // rows per workgroup of the per-channel reductions (plain path)
#include "ctn_f32_fused.cuh"

// workspace offsets in floats
struct Ws32 {
  long long Lp, wavep, w, wn, x, x_stride, z1, z2, z_stride, tab, tab_stride, skip, h, pre, dpre,
      y, fr, dop, dy, dwm, act, G, e, dz, fsum, btab, part, part_floats, scalars, gscratch, gscratch_floats, skipb, wt, total;
  void init(const Lay32& l, long long B, long long T, long long L) {
    long long o = 0;
    auto take = [&](long long n) { long long r = o; o += up(n, 64); return r; };
    const long long BT = B*T;
    Lp = (T - 1)*l.hop + l.K;
    if (Lp < L) Lp = L;
    const long long BS = B*l.S;
    wavep = take(B*Lp);
    w = take(BT*l.N); wn = take(BT*l.N);
    x_stride = up(BT*l.Bn, 64); x = take(x_stride*l.nb);
    z_stride = up(BT*l.H, 64); z1 = take(z_stride*l.nb); z2 = take(z_stride*l.nb);
    tab_stride = up(BT*2, 64); tab = take(tab_stride*(1 + 2*l.nb));
    skip = take(BT*l.Sc);
    h = take(BT*l.H);
    pre = take(BT*l.S*l.N);
    y = take(BS*T*l.N);
    fr = take(BS*T*l.K);
    // backward
    dop = take(BS*Lp);
    dy = take(BS*T*l.N);
    dpre = take(BT*l.S*l.N);
    dwm = take(BT*l.N);
    act = take(BT*l.Sc);
    G = take(BT*(l.Bn + l.Sc));
    const long long cmax = l.H > l.N ? l.H : l.N;
    e = take(BT*cmax); dz = take(BT*cmax);
    fsum = take(BT*2); btab = take(BT*2);
    // partial sums of the per-channel reductions: slices x quantities (<= 2 + P + 1) x channels
    // (comb slices of the stencil kernels, ctn_f32_fused.cuh: at most ~4x the contiguous count)
    const long long slices = 5*((BT + kFusedRows - 1)/kFusedRows);   // >= the 256-row slices of the plain path
    long long cq = (long long)(l.P + 2)*l.H;
    if (2*cmax > cq) cq = 2*cmax;
    if ((long long)l.S*l.N > cq) cq = (long long)l.S*l.N;
    part_floats = slices*cq;
    part = take(part_floats);
    scalars = take(4096 > slices ? 4096 : slices);
    // partial tiles of split reductions (weight gradients): at most one 256 x 128 tile per compute unit
    gscratch_floats = 32768LL*320;
    gscratch = take(gscratch_floats);
    skipb = take(l.Sc);
    {
      // transposed copies of the weights a data-gradient product reads ([k][n] -> [n][k]: both operands of
      // the split-bf16 product are then contiguous in the reduction index)
      long long m = 2LL*l.H*(l.Bn > l.Sc ? l.Bn : l.Sc);
      if ((long long)l.N*l.Bn > m) m = (long long)l.N*l.Bn;
      if ((long long)l.S*l.N*l.Sc > m) m = (long long)l.S*l.N*l.Sc;
      wt = take(m);
    }          // column sums of the skip gradient: the same for every block
    total = o;
  }
};

// ---- small kernels ----------------------------------------------------------------------------
__global__ void pad_rows_kernel(const float* src, float* dst, long long rows, long long L, long long Lp) {
  const long long n = rows*Lp;
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < n; i += (long long)gridDim.x*256) {
    const long long r = i / Lp, l = i % Lp;
    dst[i] = l < L ? src[r*L + l] : 0.f;
  }
}

// per-frame sums of p = prelu(z) and p^2 over the C channels of a row; one wavefront per row
__global__ __launch_bounds__(256) void f32_frame_sums_kernel(const float* z, const float* slope,
                                                             float* fsum, long long rows, int C) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x*4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float a = slope ? *slope : 1.f;
  float s1 = 0.f, s2 = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float v = z[row*C + c];
    const float p = (v > 0.f || !slope) ? v : a*v;
    s1 += p; s2 = __builtin_fmaf(p, p, s2);
  }
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  if (lane == 0) { fsum[2*row] = s1; fsum[2*row + 1] = s2; }
}

__device__ __forceinline__ void scan2(double& a, double& b, double* scr, bool reverse) {
  const int tid = threadIdx.x;
  scr[tid] = a; scr[256 + tid] = b;
  __syncthreads();
  double sa = 0.0, sb = 0.0;
  if (!reverse) { for (int i = 0; i < tid; ++i) { sa += scr[i]; sb += scr[256 + i]; } }
  else { for (int i = tid + 1; i < 256; ++i) { sa += scr[i]; sb += scr[256 + i]; } }
  __syncthreads();
  a = sa; b = sb;
}

// forward table (mean_t, rstd_t): causal = statistics of frames <= t, else of the whole item
__global__ __launch_bounds__(256) void f32_fwd_table_kernel(const float* fsum, float* table, int T,
                                                            int C, float eps, int causal) {
  __shared__ double scr[512];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int per = (T + 255)/256;
  const int lo = min(T, tid*per), hi = min(T, lo + per);
  const float* fs = fsum + (long long)b*T*2;
  float* tb = table + (long long)b*T*2;
  double s1 = 0.0, s2 = 0.0;
  for (int t = lo; t < hi; ++t) { s1 += fs[2*t]; s2 += fs[2*t + 1]; }
  if (causal) {
    scan2(s1, s2, scr, false);
    for (int t = lo; t < hi; ++t) {
      s1 += fs[2*t]; s2 += fs[2*t + 1];
      const double n = (double)C*(double)(t + 1);
      const double mean = s1/n, var = s2/n - mean*mean;
      tb[2*t] = (float)mean; tb[2*t + 1] = (float)(1.0/sqrt(var + (double)eps));
    }
  } else {
    scr[tid] = s1; scr[256 + tid] = s2;
    __syncthreads();
    double t1 = 0.0, t2 = 0.0;
    for (int i = 0; i < 256; ++i) { t1 += scr[i]; t2 += scr[256 + i]; }
    const double n = (double)C*(double)T;
    const double mean = t1/n;
    double var = t2/n - mean*mean;
    if (var < 0.0) var = 0.0;
    const float m = (float)mean, r = (float)(1.0/sqrt(var + (double)eps));
    for (int t = lo; t < hi; ++t) { tb[2*t] = m; tb[2*t + 1] = r; }
  }
}

// y = (prelu(z) - mean_t) rstd_t gain[c] + bias[c]
__global__ void f32_norm_apply_kernel(const float* z, const float* slope, const float* table,
                                      const float* gain, const float* bias, float* y,
                                      long long rows, int C) {
  const long long n = rows*C;
  const float a = slope ? *slope : 1.f;
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < n; i += (long long)gridDim.x*256) {
    const long long row = i / C; const int c = (int)(i % C);
    const float v = z[i];
    const float p = (v > 0.f || !slope) ? v : a*v;
    y[i] = (p - table[2*row])*table[2*row + 1]*gain[c] + bias[c];
  }
}

// the same, 4 channels per thread (C % 4 == 0): 16-byte loads and stores
__device__ __forceinline__ float4 ld4(const float* p, int c4) {      // 4 floats, any alignment
  return make_float4(p[4*c4], p[4*c4 + 1], p[4*c4 + 2], p[4*c4 + 3]);
}
__global__ void f32_norm_apply4_kernel(const float4* z, const float* slope, const float* table,
                                       const float* gain, const float* bias, float4* y,
                                       long long rows, int C4) {
  const long long n = rows*C4;
  const float a = slope ? *slope : 1.f;
  const bool act = slope != nullptr;
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < n; i += (long long)gridDim.x*256) {
    const long long row = i / C4; const int c = (int)(i % C4);
    const float m = table[2*row], r = table[2*row + 1];
    const float4 v = z[i], g = ld4(gain, c), b = ld4(bias, c);
    float4 o;
    o.x = (((v.x > 0.f || !act) ? v.x : a*v.x) - m)*r*g.x + b.x;
    o.y = (((v.y > 0.f || !act) ? v.y : a*v.y) - m)*r*g.y + b.y;
    o.z = (((v.z > 0.f || !act) ? v.z : a*v.z) - m)*r*g.z + b.z;
    o.w = (((v.w > 0.f || !act) ? v.w : a*v.w) - m)*r*g.w + b.w;
    y[i] = o;
  }
}

// backward frame sums: A_t = sum_c e gain, B_t = sum_c e gain xhat
__global__ __launch_bounds__(256) void f32_bwd_frame_sums_kernel(const float* e, const float* z,
                                                                 const float* slope, const float* table,
                                                                 const float* gain, float* fsum,
                                                                 long long rows, int C) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x*4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float a = slope ? *slope : 1.f;
  const float mean = table[2*row], rstd = table[2*row + 1];
  float A = 0.f, Bq = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float v = z[row*C + c];
    const float p = (v > 0.f || !slope) ? v : a*v;
    const float xh = (p - mean)*rstd;
    const float g = e[row*C + c]*gain[c];
    A += g; Bq = __builtin_fmaf(g, xh, Bq);
  }
  A = wave_sum(A); Bq = wave_sum(Bq);
  if (lane == 0) { fsum[2*row] = A; fsum[2*row + 1] = Bq; }
}

// backward table (U_t, V_t): dp = e gain rstd_t + U_t + p V_t, with per-frame terms
//   u_t = (-rstd_t A_t + mean_t rstd_t^2 B_t)/n_t,  v_t = -rstd_t^2 B_t/n_t
// summed over frames >= t (causal, n_t = C (t+1)) or over the whole item (n_t = C T).
__global__ __launch_bounds__(256) void f32_bwd_table_kernel(const float* fsum, const float* ftab,
                                                            float* btab, int T, int C, int causal) {
  __shared__ double scr[512];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int per = (T + 255)/256;
  const int lo = min(T, tid*per), hi = min(T, lo + per);
  const float* fs = fsum + (long long)b*T*2;
  const float* ft = ftab + (long long)b*T*2;
  float* tb = btab + (long long)b*T*2;
  auto terms = [&](int t, double& u, double& v) {
    const double n = causal ? (double)C*(double)(t + 1) : (double)C*(double)T;
    const double mean = ft[2*t], r = ft[2*t + 1];
    const double A = fs[2*t], Bq = fs[2*t + 1];
    u = (-r*A + mean*r*r*Bq)/n;
    v = -r*r*Bq/n;
  };
  double s1 = 0.0, s2 = 0.0;
  for (int t = lo; t < hi; ++t) { double u, v; terms(t, u, v); s1 += u; s2 += v; }
  if (causal) {
    scan2(s1, s2, scr, true);
    for (int t = hi - 1; t >= lo; --t) {
      double u, v; terms(t, u, v);
      s1 += u; s2 += v;
      tb[2*t] = (float)s1; tb[2*t + 1] = (float)s2;
    }
  } else {
    scr[tid] = s1; scr[256 + tid] = s2;
    __syncthreads();
    double t1 = 0.0, t2 = 0.0;
    for (int i = 0; i < 256; ++i) { t1 += scr[i]; t2 += scr[256 + i]; }
    for (int t = lo; t < hi; ++t) { tb[2*t] = (float)t1; tb[2*t + 1] = (float)t2; }
  }
}

// dz = prelu'(z) (e gain rstd_t + U_t + p V_t) (+ add); slope gradient partials per workgroup
__global__ __launch_bounds__(256) void f32_norm_bwd_apply_kernel(
    const float* e, const float* z, const float* slope, const float* ftab, const float* btab,
    const float* gain, const float* add, float* dz, float* dslope_part, long long rows, int C) {
  __shared__ float scr[8];
  const long long n = rows*C;
  const float a = slope ? *slope : 1.f;
  float da = 0.f;
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < n; i += (long long)gridDim.x*256) {
    const long long row = i / C; const int c = (int)(i % C);
    const float v = z[i];
    const bool pos = v > 0.f || !slope;
    const float p = pos ? v : a*v;
    const float dp = e[i]*gain[c]*ftab[2*row + 1] + btab[2*row] + p*btab[2*row + 1];
    float o = pos ? dp : a*dp;
    if (!pos) da = __builtin_fmaf(dp, v, da);
    if (add) o += add[i];
    dz[i] = o;
  }
  if (dslope_part) {
    const float s = block_sum(da, scr);
    if (threadIdx.x == 0) dslope_part[blockIdx.x] = s;
  }
}

// 4 channels per thread (C % 4 == 0)
__global__ __launch_bounds__(256) void f32_norm_bwd_apply4_kernel(
    const float4* e, const float4* z, const float* slope, const float* ftab, const float* btab,
    const float* gain, const float4* add, float4* dz, float* dslope_part, long long rows, int C4) {
  __shared__ float scr[8];
  const long long n = rows*C4;
  const float a = slope ? *slope : 1.f;
  const bool act = slope != nullptr;
  float da = 0.f;
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < n; i += (long long)gridDim.x*256) {
    const long long row = i / C4; const int c = (int)(i % C4);
    const float r = ftab[2*row + 1], U = btab[2*row], V = btab[2*row + 1];
    const float4 v = z[i], ev = e[i], g = ld4(gain, c);
    float4 o;
#define BRV_ONE(f)                                                   \
    { const bool pos = v.f > 0.f || !act; const float p = pos ? v.f : a*v.f;   \
      const float dp = ev.f*g.f*r + U + p*V; o.f = pos ? dp : a*dp;  \
      if (!pos) da = __builtin_fmaf(dp, v.f, da); }
    BRV_ONE(x) BRV_ONE(y) BRV_ONE(z) BRV_ONE(w)
#undef BRV_ONE
    if (add) { const float4 ad = add[i]; o.x += ad.x; o.y += ad.y; o.z += ad.z; o.w += ad.w; }
    dz[i] = o;
  }
  if (dslope_part) {
    const float sres = block_sum(da, scr);
    if (threadIdx.x == 0) dslope_part[blockIdx.x] = sres;
  }
}

// ---- per-channel reductions: thread = channel, workgroup = slice of kSliceRows rows ----------
// mode 0: q0 = sum src                                   (bias gradients)
// mode 1: q0 = sum e xhat, q1 = sum e                    (norm gain / bias gradients)
// mode 2: q_k = sum_t dz2[t] h1[t + k dil - left] (k < P), q_P = sum dz2   (depthwise taps / bias)
struct Red32 {
  int mode; const float* a; const float* z; const float* slope; const float* table;
  long long rows; int C, ld, T, P, dil, left; float* part;
};
__global__ __launch_bounds__(256) void f32_chan_reduce_kernel(const Red32 p) {
  // workgroup = (row slice, 64 channels): 64 columns x 4 row lanes; a lane walks rows lane, lane + 4,
  // ... of the slice, the 4 lane sums are added in lane order (deterministic)
  __shared__ float red[4][8][65];
  const long long r0 = (long long)blockIdx.x*kSliceRows;
  const long long r1 = r0 + kSliceRows < p.rows ? r0 + kSliceRows : p.rows;
  const int nq = p.mode == 0 ? 1 : (p.mode == 1 ? 2 : p.P + 1);
  const float sl = p.slope ? *p.slope : 1.f;
  const int col = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.y*64 + col;
  float q[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) q[k] = 0.f;
  if (c < p.C) {
    for (long long r = r0 + rl; r < r1; r += 4) {
      const float av = p.a[r*p.ld + c];
      if (p.mode == 0) {
        q[0] += av;
      } else if (p.mode == 1) {
        const float v = p.z[r*p.C + c];
        const float pv = (v > 0.f || !p.slope) ? v : sl*v;
        q[0] = __builtin_fmaf(av, (pv - p.table[2*r])*p.table[2*r + 1], q[0]);
        q[1] += av;
      } else {
        const long long b = r / p.T; const int t = (int)(r % p.T);
        for (int k = 0; k < p.P; ++k) {
          const int ti = t + k*p.dil - p.left;
          if (ti >= 0 && ti < p.T) q[k] = __builtin_fmaf(av, p.z[(b*p.T + ti)*p.C + c], q[k]);
        }
        q[p.P] += av;
      }
    }
  }
  for (int k = 0; k < nq; ++k) red[rl][k][col] = q[k];
  __syncthreads();
  if (rl == 0 && c < p.C)
    for (int k = 0; k < nq; ++k)
      p.part[((long long)blockIdx.x*nq + k)*p.C + c] =
          ((red[0][k][col] + red[1][k][col]) + red[2][k][col]) + red[3][k][col];
}
// dst[c*stride + off_k] += sum over slices: 16 columns x 16 slice lanes per workgroup, each lane
// adds its slices in order, the 16 lane sums are added in lane order (deterministic)
__global__ __launch_bounds__(256) void f32_chan_fold_kernel(const float* part, int slices, int nq,
                                                            int C, float* d0, float* d1, int stride0,
                                                            int P) {
  // quantity k of channel c goes to (d0, stride0) for k < P and to d1 for k == P
  __shared__ float acc[16][17];
  const int col = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int i = blockIdx.x*16 + col;
  float s = 0.f;
  if (i < nq*C) {
    const int k = i / C, c = i % C;
    for (int sidx = sl; sidx < slices; sidx += 16) s += part[((long long)sidx*nq + k)*C + c];
  }
  acc[sl][col] = s;
  __syncthreads();
  if (sl == 0 && i < nq*C) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) t += acc[j][col];
    const int k = i / C, c = i % C;
    if (k < P) d0[(long long)c*stride0 + k] += t; else d1[c] += t;
  }
}
__global__ void f32_fold_scalar_kernel(const float* part, int n, float* dst) {
  __shared__ float scr[8];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += part[i];
  const float r = block_sum(s, scr);
  if (threadIdx.x == 0) *dst += r;
}

// depthwise dilated convolution on the normalised tensor h: z2[t][c] = bias[c] + sum_k w[c][k] h[t + k dil - left][c]
__global__ void f32_dw_fwd_kernel(const float* h, const float* taps, const float* bias, float* z2,
                                  long long B, int T, int C, int P, int dil, int left) {
  const long long n = B*T*C;
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < n; i += (long long)gridDim.x*256) {
    const int c = (int)(i % C); const long long row = i / C;
    const int t = (int)(row % T); const long long b = row / T;
    float acc = bias[c];
    for (int k = 0; k < P; ++k) {
      const int ti = t + k*dil - left;
      if (ti >= 0 && ti < T) acc = __builtin_fmaf(taps[c*P + k], h[(b*T + ti)*C + c], acc);
    }
    z2[i] = acc;
  }
}
// 4 channels per thread (C % 4 == 0): `sign` +1 reads h[t + k dil - left] (forward), -1 reads
// dz2[t - k dil + left] (transposed stencil); bias may be null
__global__ void f32_dw4_kernel(const float4* src, const float* taps, const float* bias, float4* dst,
                               long long B, int T, int C4, int P, int dil, int left, int sign) {
  const long long n = B*T*C4;
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < n; i += (long long)gridDim.x*256) {
    const int c = (int)(i % C4); const long long row = i / C4;
    const int t = (int)(row % T); const long long b = row / T;
    float4 acc = bias ? ld4(bias, c) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < P; ++k) {
      const int ti = t + sign*(k*dil - left);
      if (ti < 0 || ti >= T) continue;
      const float4 v = src[(b*T + ti)*C4 + c];
      const float* w = taps + (long long)4*c*P + k;
      acc.x = __builtin_fmaf(w[0], v.x, acc.x); acc.y = __builtin_fmaf(w[P], v.y, acc.y);
      acc.z = __builtin_fmaf(w[2*P], v.z, acc.z); acc.w = __builtin_fmaf(w[3*P], v.w, acc.w);
    }
    dst[i] = acc;
  }
}

// transposed stencil: e1[t][c] = sum_k w[c][k] dz2[t - k dil + left][c]
__global__ void f32_dw_bwd_kernel(const float* dz2, const float* taps, float* e1, long long B, int T,
                                  int C, int P, int dil, int left) {
  const long long n = B*T*C;
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < n; i += (long long)gridDim.x*256) {
    const int c = (int)(i % C); const long long row = i / C;
    const int t = (int)(row % T); const long long b = row / T;
    float acc = 0.f;
    for (int k = 0; k < P; ++k) {
      const int to = t - k*dil + left;
      if (to >= 0 && to < T) acc = __builtin_fmaf(taps[c*P + k], dz2[(b*T + to)*C + c], acc);
    }
    e1[i] = acc;
  }
}

// dst[r][c] = (src ? src[r*lds + c] : (init ? 0 : dst)) + bias[c]
__global__ void f32_bias_rows_kernel(float* dst, int ldd, const float* src, int lds, const float* bias,
                                     long long rows, int C, int init) {
  const long long n = rows*C;
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < n; i += (long long)gridDim.x*256) {
    const long long r = i / C; const int c = (int)(i % C);
    const float base = src ? src[r*lds + c] : (init ? 0.f : dst[r*ldd + c]);
    dst[r*ldd + c] = base + bias[c];
  }
}
__global__ void f32_prelu_kernel(const float* x, const float* slope, float* y, long long n) {
  const float a = *slope;
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < n; i += (long long)gridDim.x*256) {
    const float v = x[i];
    y[i] = v > 0.f ? v : a*v;
  }
}
// gskip = da * prelu'(skip) written with row stride ldg; slope-gradient partial per workgroup
__global__ __launch_bounds__(256) void f32_prelu_bwd_kernel(const float* da, const float* x,
                                                            const float* slope, float* g, int ldg,
                                                            float* part, long long rows, int C) {
  __shared__ float scr[8];
  const float a = *slope;
  const long long n = rows*C;
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < n; i += (long long)gridDim.x*256) {
    const long long r = i / C; const int c = (int)(i % C);
    const float v = x[i], d = da[i];
    const bool pos = v > 0.f;
    g[r*ldg + c] = pos ? d : a*d;
    if (!pos) acc = __builtin_fmaf(d, v, acc);
  }
  const float s = block_sum(acc, scr);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
// m = sigmoid(pre) in place; y[(b S + s)][t][n] = m w[b][t][n]      (convtasnet.py:144-151,199)
__global__ void f32_mask_kernel(float* pre, const float* w, float* y, long long B, int T, int S, int N) {
  const long long n = B*T*S*N;
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < n; i += (long long)gridDim.x*256) {
    const int c = (int)(i % N); long long q = i / N;
    const int s = (int)(q % S); q /= S;
    const int t = (int)(q % T); const long long b = q / T;
    const float m = 1.f/(1.f + __expf(-pre[i]));
    pre[i] = m;
    y[((b*S + s)*T + t)*N + c] = m*w[(b*T + t)*N + c];
  }
}
// dpre = dy w m (1 - m) (over pre's layout), dwm[b][t][n] = sum_s dy m
__global__ void f32_mask_bwd_kernel(const float* dy, const float* m, const float* w, float* dpre,
                                    float* dwm, long long B, int T, int S, int N) {
  const long long n = B*T*N;
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < n; i += (long long)gridDim.x*256) {
    const int c = (int)(i % N); const long long row = i / N;
    const int t = (int)(row % T); const long long b = row / T;
    const float wv = w[i];
    float acc = 0.f;
    for (int s = 0; s < S; ++s) {
      const float d = dy[((b*S + s)*T + t)*N + c];
      const long long j = (row*S + s)*N + c;
      const float mv = m[j];
      dpre[j] = d*wv*mv*(1.f - mv);
      acc = __builtin_fmaf(d, mv, acc);
    }
    dwm[i] = acc;
  }
}
// overlap-add of the decoder frames fr[(bs)][t][K] with hop K/2, cropped to L (convtasnet.py:71,144-151)
__global__ void f32_ola_kernel(const float* fr, float* out, long long BS, int T, int K, int hop, long long L) {
  const long long n = BS*L;
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < n; i += (long long)gridDim.x*256) {
    const long long bs = i / L, l = i % L;
    float acc = 0.f;
    long long t_hi = l / hop;
    for (long long t = t_hi; t >= 0 && l - t*hop < K; --t)
      if (t < T) acc += fr[(bs*T + t)*K + (l - t*hop)];
    out[i] = acc;
  }
}
// dst[c][r] = src[r][c] for `pairs` matrices (R x C each): small weight matrices only
__global__ __launch_bounds__(256) void f32_transpose_kernel(const float* src, float* dst, int R, int C,
                                                            long long src_pair_stride, int pairs) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c0 = blockIdx.x*32, r0 = blockIdx.y*32, z = blockIdx.z;
  const float* s = src + (long long)z*src_pair_stride; float* d = dst + (long long)z*R*C;
  for (int j = ty; j < 32; j += 8)
    tile[j][tx] = (r0 + j < R && c0 + tx < C) ? s[(long long)(r0 + j)*C + c0 + tx] : 0.f;
  __syncthreads();
  for (int j = ty; j < 32; j += 8)
    if (c0 + j < C && r0 + tx < R) d[(long long)(c0 + j)*R + r0 + tx] = tile[tx][j];
}
__global__ void f32_add_kernel(float* dst, const float* a, long long n) {
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < n; i += (long long)gridDim.x*256)
    dst[i] += a[i];
}

inline bool aligned16(const void* a, const void* b = nullptr, const void* c = nullptr,
                      const void* d = nullptr, const void* e = nullptr) {
  return ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c) | ((uintptr_t)d) | ((uintptr_t)e)) & 15) == 0;
}
inline int grid_for(long long n) {
  long long g = (n + 255)/256;
  if (g > 65535*4) g = 65535*4;
  if (g < 1) g = 1;
  return (int)g;
}
constexpr int kSlopeBlocks = 1024;

struct Ctx32 {
  const Lay32& l; const Ws32& ws; float* base; const float* params; float* grads;
  long long B, T, L, BT; hipStream_t st;
  bool fused;               // every product on gemm_f32_big.hip with this workspace's scratch (fixed-order sums)
  float* f(long long off) const { return base + off; }
  float* tab(int i) const { return base + ws.tab + ws.tab_stride*i; }
  float* xb(int i) const { return base + ws.x + ws.x_stride*i; }
  float* z1b(int i) const { return base + ws.z1 + ws.z_stride*i; }
  float* z2b(int i) const { return base + ws.z2 + ws.z_stride*i; }
};

int gemm32(const Ctx32& c, const float* a, const float* b, float* d, long long batch, long long M,
           long long N, long long K, long long lda, long long ldb, long long ldd, long long abs_,
           long long bbs, long long dbs, int ta, int tb, long long kbatch, long long akbs,
           long long bkbs, const float* bias, int acc) {
  if (c.fused) {
    // the fused path keeps every product on gemm_f32_big.hip: its split reductions take their scratch from
    // this workspace and add in a fixed order (brv_gemm_f32 has no scratch: long reductions over few tiles
    // would run on the 128 x 128 kernel, whose split adds with atomics)
    BigGemm g; memset(&g, 0, sizeof(g));
    g.M = (int)M; g.N = (int)N; g.K = (int)K; g.kbatch = kbatch > 1 ? (int)kbatch : 1; g.batch = (int)batch;
    g.A = a; g.a_bs = abs_; g.a_kbs = akbs; g.lda = (int)lda; g.ta = ta != 0;
    g.B = b; g.b_bs = bbs; g.b_kbs = bkbs; g.ldb = (int)ldb; g.tb = tb != 0;
    g.D = d; g.d_bs = dbs; g.ldd = (int)ldd;
    g.bias = bias; g.col_bias = acc == 2;
    if (acc == 1) { g.add = d; g.add_bs = dbs; g.ldadd = (int)ldd; }
    g.x3 = 1;
    g.scratch = c.f(c.ws.gscratch); g.scratch_floats = c.ws.gscratch_floats;
    if (gemm_f32_big_ok(g)) {
      const int rb = gemm_f32_big(g, c.st);
      if (rb) return fail32(rb, "fp32 Conv-TasNet path: gemm_f32_big failed");
      return 0;
    }
  }
  const int r = brv_gemm_f32(a, b, d, batch, M, N, K, lda, ldb, ldd, abs_, bbs, dbs, ta, tb, kbatch,
                             akbs, bkbs, bias, acc, (brv_stream_t)c.st);
  if (r) return fail32(r, "brv_gemm_f32 failed inside the fp32 Conv-TasNet path");
  return 0;
}
// rows-major 1x1 convolution: d[rows][n] = a[rows][k] W[n][k]^T (+ col bias | accumulate)
int conv1x1(const Ctx32& c, const float* a, int lda, const float* W, int N, int K, float* d, int ldd,
            const float* bias, int acc) {
  return gemm32(c, a, W, d, 1, c.BT, N, K, lda, K, ldd, 0, 0, 0, 0, 1, 1, 0, 0, bias, acc);
}
// data gradient: d[rows][k] (+)= g[rows][n] W[n][k]
int conv1x1_dgrad(const Ctx32& c, const float* g, int ldg, const float* W, int N, int K, float* d,
                  int ldd, int acc) {
  return gemm32(c, g, W, d, 1, c.BT, K, N, ldg, K, ldd, 0, 0, 0, 0, 0, 1, 0, 0, nullptr, acc);
}
// weight gradient: dW[n][k] += sum_rows g[rows][n] a[rows][k]
int conv1x1_wgrad(const Ctx32& c, const float* g, int ldg, const float* a, int lda, int N, int K,
                  float* dW) {
  return gemm32(c, g, a, dW, 1, N, K, c.BT, ldg, lda, K, 0, 0, 0, 1, 0, 1, 0, 0, nullptr, 1);
}

int norm_forward(const Ctx32& c, const float* z, const float* slope, float* table, int C) {
  const long long rows = c.BT;
  float* fsum = c.f(c.ws.fsum);
  hipLaunchKernelGGL(f32_frame_sums_kernel, dim3((unsigned)((rows + 3)/4)), dim3(256), 0, c.st, z,
                     slope, fsum, rows, C);
  hipLaunchKernelGGL(f32_fwd_table_kernel, dim3((unsigned)c.B), dim3(256), 0, c.st, fsum, table,
                     (int)c.T, C, 1e-8f, c.l.causal);
  HIP_OK32(hipGetLastError());
  return 0;
}
int norm_apply(const Ctx32& c, const float* z, const float* slope, const float* table,
               const float* gain, const float* bias, float* y, int C) {
  if (C % 4 == 0 && aligned16(z, y))
    hipLaunchKernelGGL(f32_norm_apply4_kernel, dim3(grid_for(c.BT*C/4)), dim3(256), 0, c.st,
                       (const float4*)z, slope, table, gain, bias, (float4*)y, c.BT, C/4);
  else
  hipLaunchKernelGGL(f32_norm_apply_kernel, dim3(grid_for(c.BT*C)), dim3(256), 0, c.st, z, slope,
                     table, gain, bias, y, c.BT, C);
  HIP_OK32(hipGetLastError());
  return 0;
}
int chan_reduce(const Ctx32& c, Red32 r, float* d0, int stride0, int nq_main, float* d1) {
  const int slices = (int)((r.rows + kSliceRows - 1)/kSliceRows);
  const int nq = r.mode == 0 ? 1 : (r.mode == 1 ? 2 : r.P + 1);
  r.part = c.f(c.ws.part);
  if ((long long)slices*nq*r.C > c.ws.part_floats) return fail32(-1, "fp32 path: reduction scratch too small");
  hipLaunchKernelGGL(f32_chan_reduce_kernel, dim3(slices, (r.C + 63)/64), dim3(256), 0, c.st, r);
  hipLaunchKernelGGL(f32_chan_fold_kernel, dim3((nq*r.C + 15)/16), dim3(256), 0, c.st, r.part,
                     slices, nq, r.C, d0, d1, stride0, nq_main);
  HIP_OK32(hipGetLastError());
  return 0;
}
int col_sum(const Ctx32& c, const float* src, int ld, int C, float* dst) {
  Red32 r; memset(&r, 0, sizeof(r));
  r.mode = 0; r.a = src; r.rows = c.BT; r.C = C; r.ld = ld;
  return chan_reduce(c, r, dst, 1, 0, dst);       // k = 0 >= P = 0 -> d1
}
// gradient through y = norm(prelu(z)): e (wrt y) -> dz (wrt z), gain / bias / slope gradients
int norm_backward(const Ctx32& c, const float* e, const float* z, const float* slope_p,
                  const float* table, const float* gain, int C, const float* add, float* dz,
                  float* dgain, float* dbias, float* dslope) {
  const long long rows = c.BT;
  Red32 r; memset(&r, 0, sizeof(r));
  r.mode = 1; r.a = e; r.z = z; r.slope = slope_p; r.table = table; r.rows = rows; r.C = C; r.ld = C;
  // quantity 0 -> dgain (stride 1, "P" = 1), quantity 1 -> dbias
  OK32(chan_reduce(c, r, dgain, 1, 1, dbias));
  float* fsum = c.f(c.ws.fsum); float* btab = c.f(c.ws.btab);
  hipLaunchKernelGGL(f32_bwd_frame_sums_kernel, dim3((unsigned)((rows + 3)/4)), dim3(256), 0, c.st, e,
                     z, slope_p, table, gain, fsum, rows, C);
  hipLaunchKernelGGL(f32_bwd_table_kernel, dim3((unsigned)c.B), dim3(256), 0, c.st, fsum, table, btab,
                     (int)c.T, C, c.l.causal);
  int g = grid_for(rows*C);
  if (g > kSlopeBlocks) g = kSlopeBlocks;
  float* part = c.f(c.ws.scalars);
  if (C % 4 == 0 && aligned16(e, z, add, dz)) {
    int g4 = grid_for(rows*C/4);
    if (g4 < g) g = g4;
    hipLaunchKernelGGL(f32_norm_bwd_apply4_kernel, dim3(g), dim3(256), 0, c.st, (const float4*)e,
                       (const float4*)z, slope_p, table, btab, gain,
                       (const float4*)add, (float4*)dz, dslope ? part : nullptr, rows, C/4);
  } else
  hipLaunchKernelGGL(f32_norm_bwd_apply_kernel, dim3(g), dim3(256), 0, c.st, e, z, slope_p, table, btab,
                     gain, add, dz, dslope ? part : nullptr, rows, C);
  if (dslope) hipLaunchKernelGGL(f32_fold_scalar_kernel, dim3(1), dim3(256), 0, c.st, part, g, dslope);
  HIP_OK32(hipGetLastError());
  return 0;
}

// ---- fused path (ctn_f32_fused.cuh + the operand transforms of gemm_f32_big.hip) -------------------
// Taken when every channel count is a multiple of 4 (16-byte accesses) and <= 1024; other
// configurations run the plain kernels above.
bool fused_ok(const Lay32& l, const void* workspace) {
  auto q = [](int v) { return v % 4 == 0; };
  return q(l.N) && q(l.Bn) && q(l.H) && q(l.Sc) && l.H <= 256*kMaxNJ && l.N <= 256*kMaxNJ &&
         (((uintptr_t)workspace) & 15) == 0;
}
int nj_of(int C) { return (C/4 + 63)/64; }

#define BRV_NJ_LAUNCH(KERNEL, C_, grid, st, arg)                                                        \
  do {                                                                                                  \
    switch (nj_of(C_)) {                                                                                \
      case 1: hipLaunchKernelGGL((KERNEL<1>), grid, dim3(256), 0, st, arg); break;                      \
      case 2: hipLaunchKernelGGL((KERNEL<2>), grid, dim3(256), 0, st, arg); break;                      \
      case 3: hipLaunchKernelGGL((KERNEL<3>), grid, dim3(256), 0, st, arg); break;                      \
      default: hipLaunchKernelGGL((KERNEL<4>), grid, dim3(256), 0, st, arg); break;                     \
    }                                                                                                   \
  } while (0)
#define BRV_NJP_LAUNCH(KERNEL, C_, P_, grid, st, arg)                                                   \
  do {                                                                                                  \
    const int nj_ = nj_of(C_);                                                                          \
    if ((P_) <= 3) {                                                                                    \
      if (nj_ == 1) hipLaunchKernelGGL((KERNEL<1, 3>), grid, dim3(256), 0, st, arg);                    \
      else if (nj_ == 2) hipLaunchKernelGGL((KERNEL<2, 3>), grid, dim3(256), 0, st, arg);               \
      else if (nj_ == 3) hipLaunchKernelGGL((KERNEL<3, 3>), grid, dim3(256), 0, st, arg);               \
      else hipLaunchKernelGGL((KERNEL<4, 3>), grid, dim3(256), 0, st, arg);                             \
    } else {                                                                                            \
      if (nj_ == 1) hipLaunchKernelGGL((KERNEL<1, 7>), grid, dim3(256), 0, st, arg);                    \
      else if (nj_ == 2) hipLaunchKernelGGL((KERNEL<2, 7>), grid, dim3(256), 0, st, arg);               \
      else if (nj_ == 3) hipLaunchKernelGGL((KERNEL<3, 7>), grid, dim3(256), 0, st, arg);               \
      else hipLaunchKernelGGL((KERNEL<4, 7>), grid, dim3(256), 0, st, arg);                             \
    }                                                                                                   \
  } while (0)

inline unsigned fused_slices(const Ctx32& c) { return (unsigned)((c.BT + kFusedRows - 1)/kFusedRows); }

// one product on the big-tile kernel with this workspace's scratch for split reductions
int big(const Ctx32& c, BigGemm g) {
  g.batch = 1; if (g.kbatch < 1) g.kbatch = 1;
  g.scratch = c.f(c.ws.gscratch); g.scratch_floats = c.ws.gscratch_floats;
  const int r = gemm_f32_big(g, c.st);
  if (r) return fail32(r, "fp32 Conv-TasNet path: gemm_f32_big refused a product");
  return 0;
}
// d[rows][n] = op(a)[rows][k] W[n][k]^T + bias[n] (+ add), op = identity or the norm transform `pro`
int conv1x1_f(const Ctx32& c, const float* a, int lda, const NormPro* pro, const float* W, int N, int K,
              float* d, int ldd, const float* bias, const float* add, int ldadd) {
  BigGemm g; memset(&g, 0, sizeof(g));
  g.M = (int)c.BT; g.N = N; g.K = K;
  g.A = a; g.lda = lda; g.B = W; g.ldb = K; g.tb = 1;
  g.D = d; g.ldd = ldd; g.bias = bias; g.col_bias = 1;
  g.add = add; g.ldadd = ldadd;
  if (pro) g.pa = *pro;
  g.x3 = 1;
  return big(c, g);
}
// dW[n][k] += sum_rows g[rows][n] op(a)[rows][k]; rows >= m_split of dW go to dW2
int wgrad_f(const Ctx32& c, const float* gr, int ldg, int Nrows, const float* a, int lda, const NormPro* pro,
            int K, float* dW, float* dW2, int m_split) {
  BigGemm g; memset(&g, 0, sizeof(g));
  g.M = Nrows; g.N = K; g.K = (int)c.BT;
  g.A = gr; g.lda = ldg; g.ta = 1; g.B = a; g.ldb = lda;
  g.D = dW; g.ldd = K; g.add = dW; g.ldadd = K;
  if (dW2) { g.D2 = dW2; g.add2 = dW2; g.m_split = m_split; }
  if (pro) g.pb = *pro;
  g.x3 = 1;
  return big(c, g);
}
// dW[m][n] += sum over `pairs` items and their T frames of a[t][m] b[t][n] (filterbank gradients: b = frames
// of a waveform, row stride hop); split reduction in fixed order like every weight gradient here
int wgrad_pairs_f(const Ctx32& c, const float* a, int lda, int M, const float* b, int ldb, int N, int T,
                  long long pairs, long long a_pair_stride, long long b_pair_stride, float* dW) {
  BigGemm g; memset(&g, 0, sizeof(g));
  g.M = M; g.N = N; g.K = T; g.kbatch = (int)pairs; g.a_kbs = a_pair_stride; g.b_kbs = b_pair_stride;
  g.A = a; g.lda = lda; g.ta = 1; g.B = b; g.ldb = ldb;
  g.D = dW; g.ldd = N; g.add = dW; g.ldadd = N;
  g.x3 = 1;
  return big(c, g);
}
// d[rows][k] = sum over `pairs` operand pairs of g[rows][n] W[n][k] (+ add)
int dgrad_f(const Ctx32& c, const float* gr, int ldg, const float* W, int N, int K, float* d, int ldd,
            const float* add, int ldadd, int pairs, long long g_pair_stride, long long w_pair_stride) {
  BigGemm g; memset(&g, 0, sizeof(g));
  g.M = (int)c.BT; g.N = K; g.K = N; g.kbatch = pairs; g.a_kbs = g_pair_stride;
  g.A = gr; g.lda = ldg;
  g.D = d; g.ldd = ldd; g.add = add; g.ldadd = ldadd;
#ifndef BRV_GEMM_NO_X3
  if (N % 4 == 0) {
    // W (N x K) -> Wt (K x N): the product then has the layout of a forward 1x1 convolution
    float* wt = c.f(c.ws.wt);
    hipLaunchKernelGGL(f32_transpose_kernel, dim3((K + 31)/32, (N + 31)/32, pairs), dim3(256), 0, c.st, W, wt, N, K,
                       w_pair_stride, pairs);
    g.B = wt; g.ldb = N; g.tb = 1; g.b_kbs = (long long)N*K; g.x3 = 1;
    return big(c, g);
  }
#endif
  g.B = W; g.ldb = K; g.b_kbs = w_pair_stride;
  return big(c, g);
}
int fold(const Ctx32& c, int nq, int C, float* const* dst, const int* stride, long long slices = -1) {
  FoldJob j; memset(&j, 0, sizeof(j));
  j.part = c.f(c.ws.part); j.slices = (int)(slices < 0 ? fused_slices(c) : slices); j.nq = nq; j.C = C;
  for (int k = 0; k < nq; ++k) { j.dst[k] = dst[k]; j.stride[k] = stride[k]; }
  hipLaunchKernelGGL(f32_fold_kernel, dim3((nq*C + 15)/16), dim3(1024), 0, c.st, j);
  HIP_OK32(hipGetLastError());
  return 0;
}
int fwd_table(const Ctx32& c, float* table, int C) {
  hipLaunchKernelGGL(f32_fwd_table_kernel, dim3((unsigned)c.B), dim3(256), 0, c.st, c.f(c.ws.fsum), table,
                     (int)c.T, C, 1e-8f, c.l.causal);
  HIP_OK32(hipGetLastError());
  return 0;
}
// gradient through y = norm(prelu(z)) in two passes (+ tables); `presummed`: pass 1 (frame sums in
// ws.fsum, dgain / dbias already folded) was done by the producer of e
int norm_backward_f(const Ctx32& c, const float* e, const float* z, const float* slope_p, const float* table,
                    const float* gain, int C, const float* add, float* dz, float* dgain, float* dbias,
                    float* dslope, float* dchan, bool presummed) {
  const FusedCommon fc{c.BT, (int)c.T, C};
  const unsigned slices = fused_slices(c);
  if ((long long)slices*2*C > c.ws.part_floats) return fail32(-1, "fp32 path: reduction scratch too small");
  float* fsum = c.f(c.ws.fsum); float* btab = c.f(c.ws.btab);
  if (!presummed) {
    BwdSums q{fc, e, z, table, slope_p, gain, fsum, c.f(c.ws.part)};
    BRV_NJ_LAUNCH(f32_bwd_sums_kernel, C, dim3(slices), c.st, q);
    float* dst[2] = {dgain, dbias}; const int stride[2] = {1, 1};
    OK32(fold(c, 2, C, dst, stride));
  }
  hipLaunchKernelGGL(f32_bwd_table_kernel, dim3((unsigned)c.B), dim3(256), 0, c.st, fsum, table, btab,
                     (int)c.T, C, c.l.causal);
  BwdApply a{fc, e, z, slope_p, table, btab, gain, add, dz, dslope ? c.f(c.ws.scalars) : nullptr,
             dchan ? c.f(c.ws.part) : nullptr};
  BRV_NJ_LAUNCH(f32_bwd_apply_fused_kernel, C, dim3(slices), c.st, a);
  if (dslope) hipLaunchKernelGGL(f32_fold_scalar_kernel, dim3(1), dim3(256), 0, c.st, c.f(c.ws.scalars), (int)slices, dslope);
  if (dchan) { float* dst[1] = {dchan}; const int stride[1] = {1}; OK32(fold(c, 1, C, dst, stride)); }
  HIP_OK32(hipGetLastError());
  return 0;
}

}  // namespace

extern "C" {

int64_t brv_ctn_f32_workspace_bytes(const brv_ctn_config* cfg, int64_t batch, int64_t length) {
  Lay32 l; if (l.init(cfg)) return -1;
  Ws32 ws; ws.init(l, batch, l.frames(length), length);
  return ws.total*4;
}

int brv_ctn_f32_forward(const brv_ctn_config* cfg, const float* params, void* workspace,
                        const float* wave, float* out, int64_t batch, int64_t length,
                        brv_stream_t stream) {
  Lay32 l; OK32(l.init(cfg));
  const long long B = batch, L = length, T = l.frames(L);
  if (B < 1 || T < 1) return fail32(-1, "empty batch or input shorter than one frame");
  Ws32 ws; ws.init(l, B, T, L);
  Ctx32 c{l, ws, (float*)workspace, params, nullptr, B, T, L, B*T, (hipStream_t)stream, fused_ok(l, workspace)};
  const long long BT = c.BT;
  hipStream_t st = c.st;
  float* w = c.f(ws.w); float* wn = c.f(ws.wn); float* h = c.f(ws.h); float* skip = c.f(ws.skip);
  // encoder: right-padded frames (row stride hop) x filterbank
  hipLaunchKernelGGL(pad_rows_kernel, dim3(grid_for(B*ws.Lp)), dim3(256), 0, st, wave, c.f(ws.wavep),
                     B, L, ws.Lp);
  OK32(gemm32(c, c.f(ws.wavep), params + l.enc_w, w, B, T, l.N, l.K, l.hop, l.K, l.N, ws.Lp, 0,
              T*l.N, 0, 1, 1, 0, 0, nullptr, 0));
  const bool fused = fused_ok(l, workspace);
  OK32(norm_forward(c, w, nullptr, c.tab(0), l.N));
  if (fused) {
    const NormPro n0{c.tab(0), params + l.ln_g, params + l.ln_b, nullptr};
    OK32(conv1x1_f(c, w, l.N, &n0, params + l.bott_w, l.Bn, l.N, c.xb(0), l.Bn, params + l.bott_b, nullptr, 0));
  } else {
    OK32(norm_apply(c, w, nullptr, c.tab(0), params + l.ln_g, params + l.ln_b, wn, l.N));
    OK32(conv1x1(c, wn, l.N, params + l.bott_w, l.Bn, l.N, c.xb(0), l.Bn, params + l.bott_b, 2));
  }
  for (int i = 0; i < l.nb; ++i) {
    const Blk32& b = l.blk[i];
    const bool has_res = i < l.nb - 1;
    const int dil = 1 << (i % l.layers);
    const int total = (l.P - 1)*dil;
    const int left = l.causal ? total : total/2;
    if (fused) {
      // z1 = conv(x); statistics of prelu(z1); z2 = dconv(norm(prelu(z1))) with the statistics of
      // prelu(z2) from the same kernel; the second norm is applied inside the [res | skip] products
      OK32(conv1x1_f(c, c.xb(i), l.Bn, nullptr, params + b.conv_w, l.H, l.Bn, c.z1b(i), l.H, params + b.conv_b, nullptr, 0));
      OK32(norm_forward(c, c.z1b(i), params + b.prelu1, c.tab(1 + 2*i), l.H));
      DwFwd d{{BT, (int)T, l.H}, c.z1b(i), c.tab(1 + 2*i), params + b.prelu1, params + b.n1_g, params + b.n1_b,
              params + b.dconv_w, params + b.dconv_b, c.z2b(i), params + b.prelu2, c.f(ws.fsum), l.P, dil, left};
      BRV_NJP_LAUNCH(f32_dw_fwd_fused_kernel, l.H, l.P, dim3((unsigned)slice_count(slice_map((int)T, dil), B, (int)T)), st, d);
      OK32(fwd_table(c, c.tab(2 + 2*i), l.H));
      const NormPro n2{c.tab(2 + 2*i), params + b.n2_g, params + b.n2_b, params + b.prelu2};
      if (has_res && l.Bn == l.Sc && l.Bn % 128 == 0 && i > 0) {
        // [res | skip] as one product: the normalised input is rebuilt once for both
        BigGemm g; memset(&g, 0, sizeof(g));
        g.M = (int)BT; g.N = l.Bn + l.Sc; g.K = l.H;
        g.A = c.z2b(i); g.lda = l.H; g.pa = n2;
        g.B = params + b.res_w; g.B2 = params + b.skip_w; g.ldb = l.H; g.tb = 1; g.n_split = l.Bn;
        g.D = c.xb(i + 1); g.D2 = skip; g.ldd = l.Bn;
        g.add = c.xb(i); g.add2 = skip; g.ldadd = l.Bn;
        g.bias = params + b.res_b; g.bias2 = params + b.skip_b; g.col_bias = 1; g.x3 = 1;
        OK32(big(c, g));
        continue;
      }
      if (has_res)
        OK32(conv1x1_f(c, c.z2b(i), l.H, &n2, params + b.res_w, l.Bn, l.H, c.xb(i + 1), l.Bn, params + b.res_b,
                       c.xb(i), l.Bn));
      OK32(conv1x1_f(c, c.z2b(i), l.H, &n2, params + b.skip_w, l.Sc, l.H, skip, l.Sc, params + b.skip_b,
                     i == 0 ? nullptr : skip, l.Sc));
      continue;
    }
    OK32(conv1x1(c, c.xb(i), l.Bn, params + b.conv_w, l.H, l.Bn, c.z1b(i), l.H, params + b.conv_b, 2));
    OK32(norm_forward(c, c.z1b(i), params + b.prelu1, c.tab(1 + 2*i), l.H));
    OK32(norm_apply(c, c.z1b(i), params + b.prelu1, c.tab(1 + 2*i), params + b.n1_g, params + b.n1_b, h, l.H));
    if (l.H % 4 == 0 && aligned16(h, c.z2b(i)))
      hipLaunchKernelGGL(f32_dw4_kernel, dim3(grid_for(BT*l.H/4)), dim3(256), 0, st, (const float4*)h,
                         params + b.dconv_w, params + b.dconv_b, (float4*)c.z2b(i), B,
                         (int)T, l.H/4, l.P, dil, left, 1);
    else
    hipLaunchKernelGGL(f32_dw_fwd_kernel, dim3(grid_for(BT*l.H)), dim3(256), 0, st, h,
                       params + b.dconv_w, params + b.dconv_b, c.z2b(i), B, (int)T, l.H, l.P, dil, left);
    OK32(norm_forward(c, c.z2b(i), params + b.prelu2, c.tab(2 + 2*i), l.H));
    OK32(norm_apply(c, c.z2b(i), params + b.prelu2, c.tab(2 + 2*i), params + b.n2_g, params + b.n2_b, h, l.H));
    if (has_res) {
      hipLaunchKernelGGL(f32_bias_rows_kernel, dim3(grid_for(BT*l.Bn)), dim3(256), 0, st, c.xb(i + 1),
                         l.Bn, c.xb(i), l.Bn, params + b.res_b, BT, l.Bn, 0);
      OK32(conv1x1(c, h, l.H, params + b.res_w, l.Bn, l.H, c.xb(i + 1), l.Bn, nullptr, 1));
    }
    hipLaunchKernelGGL(f32_bias_rows_kernel, dim3(grid_for(BT*l.Sc)), dim3(256), 0, st, skip, l.Sc,
                       (const float*)nullptr, 0, params + b.skip_b, BT, l.Sc, i == 0 ? 1 : 0);
    OK32(conv1x1(c, h, l.H, params + b.skip_w, l.Sc, l.H, skip, l.Sc, nullptr, 1));
  }
  float* act = c.f(ws.act); float* pre = c.f(ws.pre); float* y = c.f(ws.y); float* fr = c.f(ws.fr);
  hipLaunchKernelGGL(f32_prelu_kernel, dim3(grid_for(BT*l.Sc)), dim3(256), 0, st, skip,
                     params + l.tcn_prelu, act, BT*l.Sc);
  OK32(conv1x1(c, act, l.Sc, params + l.out_w, l.S*l.N, l.Sc, pre, l.S*l.N, params + l.out_b, 2));
  hipLaunchKernelGGL(f32_mask_kernel, dim3(grid_for(BT*l.S*l.N)), dim3(256), 0, st, pre, w, y, B,
                     (int)T, l.S, l.N);
  // decoder: frames = y dec_w (N x K), overlap-add
  OK32(gemm32(c, y, params + l.dec_w, fr, 1, B*l.S*T, l.K, l.N, l.N, l.K, l.K, 0, 0, 0, 0, 0, 1, 0, 0,
              nullptr, 0));
  hipLaunchKernelGGL(f32_ola_kernel, dim3(grid_for(B*l.S*L)), dim3(256), 0, st, fr, out, B*l.S,
                     (int)T, l.K, l.hop, L);
  HIP_OK32(hipGetLastError());
  return 0;
}

int brv_ctn_f32_backward(const brv_ctn_config* cfg, const float* params, void* workspace,
                         const float* wave, const float* d_out, float* grads, int64_t batch,
                         int64_t length, brv_stream_t stream) {
  return brv_ctn_f32_backward_part(cfg, params, workspace, wave, d_out, grads, batch, length, 0, 1,
                                   stream);
}

// Part `part` of `nparts` of the backward pass (same block ranges and gradient buckets as
// brv_ctn_backward_part / brv_ctn_grad_bucket; the causal model runs whole in its last part).
int brv_ctn_f32_backward_part(const brv_ctn_config* cfg, const float* params, void* workspace,
                              const float* wave, const float* d_out, float* grads, int64_t batch,
                              int64_t length, int32_t part, int32_t nparts, brv_stream_t stream) {
  Lay32 l; OK32(l.init(cfg));
  const long long B = batch, L = length, T = l.frames(L);
  if (B < 1 || T < 1) return fail32(-1, "empty batch or input shorter than one frame");
  if (nparts < 1 || part < 0 || part >= nparts) return fail32(-1, "bad part");
  if (l.causal && nparts > 1) {
    if (part != nparts - 1) return 0;
    part = 0; nparts = 1;
  }
  const int blk_lo = (int)((long long)l.nb*(nparts - 1 - part)/nparts);
  const int blk_hi = (int)((long long)l.nb*(nparts - part)/nparts) - 1;
  const bool head = part == 0, tail = part == nparts - 1;
  Ws32 ws; ws.init(l, B, T, L);
  Ctx32 c{l, ws, (float*)workspace, params, grads, B, T, L, B*T, (hipStream_t)stream, fused_ok(l, workspace)};
  const long long BT = c.BT, BS = B*l.S;
  hipStream_t st = c.st;
  float* w = c.f(ws.w); float* wn = c.f(ws.wn); float* h = c.f(ws.h); float* skip = c.f(ws.skip);
  float* act = c.f(ws.act); float* m = c.f(ws.pre); float* y = c.f(ws.y);
  float* dop = c.f(ws.dop); float* dy = c.f(ws.dy); float* dwm = c.f(ws.dwm);
  float* dpre = c.f(ws.dpre);
  float* G = c.f(ws.G); float* e = c.f(ws.e); float* dz = c.f(ws.dz);
  const int ldg = l.Bn + l.Sc;
  float* spart = c.f(ws.scalars);
  const bool fused = fused_ok(l, workspace);
  if (head) {
  // decoder: d frames = framing of the padded d_out; dy = d frames x dec_w^T; dec_w gradient
  hipLaunchKernelGGL(pad_rows_kernel, dim3(grid_for(BS*ws.Lp)), dim3(256), 0, st, d_out, dop, BS, L, ws.Lp);
  OK32(gemm32(c, dop, params + l.dec_w, dy, BS, T, l.N, l.K, l.hop, l.K, l.N, ws.Lp, 0, T*l.N, 0, 1,
              1, 0, 0, nullptr, 0));
  if (fused)
    OK32(wgrad_pairs_f(c, y, l.N, l.N, dop, l.hop, l.K, (int)T, BS, T*l.N, ws.Lp, grads + l.dec_w));
  else
  OK32(gemm32(c, y, dop, grads + l.dec_w, 1, l.N, l.K, T, l.N, l.hop, l.K, 0, 0, 0, 1, 0, BS, T*l.N,
              ws.Lp, nullptr, 1));
  // mask backward: gradient wrt the mask logits and the mask-path term of the gradient wrt w
  hipLaunchKernelGGL(f32_mask_bwd_kernel, dim3(grid_for(BT*l.N)), dim3(256), 0, st, dy, m, w, dpre, dwm,
                     B, (int)T, l.S, l.N);
  // output conv: weight / bias gradients against prelu(skip), data gradient, PReLU backward
  if (fused) OK32(wgrad_f(c, dpre, l.S*l.N, l.S*l.N, act, l.Sc, nullptr, l.Sc, grads + l.out_w, nullptr, 0));
  else
  OK32(conv1x1_wgrad(c, dpre, l.S*l.N, act, l.Sc, l.S*l.N, l.Sc, grads + l.out_w));
  OK32(col_sum(c, dpre, l.S*l.N, l.S*l.N, grads + l.out_b));
  OK32(conv1x1_dgrad(c, dpre, l.S*l.N, params + l.out_w, l.S*l.N, l.Sc, e, l.Sc, 0));
  {
    int g = grid_for(BT*l.Sc); if (g > kSlopeBlocks) g = kSlopeBlocks;
    hipLaunchKernelGGL(f32_prelu_bwd_kernel, dim3(g), dim3(256), 0, st, e, skip, params + l.tcn_prelu,
                       G + l.Bn, ldg, spart, BT, l.Sc);
    hipLaunchKernelGGL(f32_fold_scalar_kernel, dim3(1), dim3(256), 0, st, spart, g, grads + l.tcn_prelu);
  }
  // every block's skip convolution receives this same gradient (the skip outputs are summed,
  // convtasnet.py:199-203): its column sums = the gradient of every skip bias, taken once
  HIP_OK32(hipMemsetAsync(c.f(ws.skipb), 0, (size_t)l.Sc*4, st));
  OK32(col_sum(c, G + l.Bn, ldg, l.Sc, c.f(ws.skipb)));
  }   // head
  for (int i = blk_hi; i >= blk_lo; --i) {
    const Blk32& b = l.blk[i];
    const bool has_res = i < l.nb - 1;
    const int dil = 1 << (i % l.layers);
    const int total = (l.P - 1)*dil;
    const int left = l.causal ? total : total/2;
    if (fused) {
      // [res | skip] weight gradients in one product against norm2(prelu2(z2)) rebuilt on load
      const NormPro n2{c.tab(2 + 2*i), params + b.n2_g, params + b.n2_b, params + b.prelu2};
      if (has_res)
        OK32(wgrad_f(c, G, ldg, l.Bn + l.Sc, c.z2b(i), l.H, &n2, l.H, grads + b.res_w, grads + b.skip_w, l.Bn));
      else
        OK32(wgrad_f(c, G + l.Bn, ldg, l.Sc, c.z2b(i), l.H, &n2, l.H, grads + b.skip_w, nullptr, 0));
      hipLaunchKernelGGL(f32_add_kernel, dim3(grid_for(l.Sc)), dim3(256), 0, st, grads + b.skip_b, c.f(ws.skipb), (long long)l.Sc);
      if (has_res) OK32(col_sum(c, G, ldg, l.Bn, grads + b.res_b));
      // data gradient wrt h2: one product over the (res, skip) operand pairs when they have one shape
      if (has_res && l.Bn == l.Sc) {
        OK32(dgrad_f(c, G, ldg, params + b.res_w, l.Bn, l.H, e, l.H, nullptr, 0, 2, l.Bn, b.skip_w - b.res_w));
      } else {
        OK32(dgrad_f(c, G + l.Bn, ldg, params + b.skip_w, l.Sc, l.H, e, l.H, nullptr, 0, 1, 0, 0));
        if (has_res) OK32(dgrad_f(c, G, ldg, params + b.res_w, l.Bn, l.H, e, l.H, e, l.H, 1, 0, 0));
      }
      OK32(norm_backward_f(c, e, c.z2b(i), params + b.prelu2, c.tab(2 + 2*i), params + b.n2_g, l.H, nullptr, dz,
                           grads + b.n2_g, grads + b.n2_b, grads + b.prelu2, grads + b.dconv_b, false));
      // depthwise conv: tap gradients against h1 rebuilt from z1, transposed stencil -> e (wrt h1), and
      // pass 1 of the first norm's backward on that e
      {
        const long long dw_slices = slice_count(slice_map((int)T, dil), B, (int)T);
        if (dw_slices*(l.P + 2)*l.H > ws.part_floats) return fail32(-1, "fp32 path: reduction scratch too small");
        DwBwd d{{BT, (int)T, l.H}, dz, c.z1b(i), c.tab(1 + 2*i), params + b.prelu1, params + b.n1_g, params + b.n1_b,
                params + b.dconv_w, e, c.f(ws.fsum), c.f(ws.part), l.P, dil, left};
        BRV_NJP_LAUNCH(f32_dw_bwd_fused_kernel, l.H, l.P, dim3((unsigned)dw_slices), st, d);
        float* dst[9]; int stride[9];
        for (int k = 0; k < l.P; ++k) { dst[k] = grads + b.dconv_w + k; stride[k] = l.P; }
        dst[l.P] = grads + b.n1_g; stride[l.P] = 1; dst[l.P + 1] = grads + b.n1_b; stride[l.P + 1] = 1;
        OK32(fold(c, l.P + 2, l.H, dst, stride, dw_slices));
      }
      OK32(norm_backward_f(c, e, c.z1b(i), params + b.prelu1, c.tab(1 + 2*i), params + b.n1_g, l.H, nullptr, dz,
                           nullptr, nullptr, grads + b.prelu1, grads + b.conv_b, true));
      OK32(wgrad_f(c, dz, l.H, l.H, c.xb(i), l.Bn, nullptr, l.Bn, grads + b.conv_w, nullptr, 0));
      OK32(dgrad_f(c, dz, l.H, params + b.conv_w, l.H, l.Bn, G, ldg, has_res ? G : nullptr, ldg, 1, 0, 0));
      continue;
    }
    // h2 again; [res | skip] weight / bias gradients; data gradient -> e (wrt h2)
    OK32(norm_apply(c, c.z2b(i), params + b.prelu2, c.tab(2 + 2*i), params + b.n2_g, params + b.n2_b, h, l.H));
    OK32(conv1x1_wgrad(c, G + l.Bn, ldg, h, l.H, l.Sc, l.H, grads + b.skip_w));
    hipLaunchKernelGGL(f32_add_kernel, dim3(grid_for(l.Sc)), dim3(256), 0, st, grads + b.skip_b, c.f(ws.skipb), (long long)l.Sc);
    OK32(conv1x1_dgrad(c, G + l.Bn, ldg, params + b.skip_w, l.Sc, l.H, e, l.H, 0));
    if (has_res) {
      OK32(conv1x1_wgrad(c, G, ldg, h, l.H, l.Bn, l.H, grads + b.res_w));
      OK32(col_sum(c, G, ldg, l.Bn, grads + b.res_b));
      OK32(conv1x1_dgrad(c, G, ldg, params + b.res_w, l.Bn, l.H, e, l.H, 1));
    }
    OK32(norm_backward(c, e, c.z2b(i), params + b.prelu2, c.tab(2 + 2*i), params + b.n2_g, l.H, nullptr,
                       dz, grads + b.n2_g, grads + b.n2_b, grads + b.prelu2));
    // depthwise conv: taps / bias gradients against h1, transposed stencil -> e (wrt h1)
    OK32(norm_apply(c, c.z1b(i), params + b.prelu1, c.tab(1 + 2*i), params + b.n1_g, params + b.n1_b, h, l.H));
    {
      Red32 r; memset(&r, 0, sizeof(r));
      r.mode = 2; r.a = dz; r.z = h; r.rows = BT; r.C = l.H; r.ld = l.H; r.T = (int)T; r.P = l.P;
      r.dil = dil; r.left = left;
      OK32(chan_reduce(c, r, grads + b.dconv_w, l.P, l.P, grads + b.dconv_b));
    }
    if (l.H % 4 == 0 && aligned16(dz, e))
      hipLaunchKernelGGL(f32_dw4_kernel, dim3(grid_for(BT*l.H/4)), dim3(256), 0, st, (const float4*)dz,
                         params + b.dconv_w, (const float*)nullptr, (float4*)e, B, (int)T, l.H/4, l.P,
                         dil, left, -1);
    else
    hipLaunchKernelGGL(f32_dw_bwd_kernel, dim3(grid_for(BT*l.H)), dim3(256), 0, st, dz,
                       params + b.dconv_w, e, B, (int)T, l.H, l.P, dil, left);
    OK32(norm_backward(c, e, c.z1b(i), params + b.prelu1, c.tab(1 + 2*i), params + b.n1_g, l.H, nullptr,
                       dz, grads + b.n1_g, grads + b.n1_b, grads + b.prelu1));
    // first 1x1 conv: weight / bias gradients, data gradient (+ residual path) -> G[:, :Bn]
    OK32(conv1x1_wgrad(c, dz, l.H, c.xb(i), l.Bn, l.H, l.Bn, grads + b.conv_w));
    OK32(col_sum(c, dz, l.H, l.H, grads + b.conv_b));
    OK32(conv1x1_dgrad(c, dz, l.H, params + b.conv_w, l.H, l.Bn, G, ldg, has_res ? 1 : 0));
  }
  if (tail) {
  // bottleneck conv, first layer norm, encoder
  if (fused) {
    const NormPro n0{c.tab(0), params + l.ln_g, params + l.ln_b, nullptr};
    OK32(wgrad_f(c, G, ldg, l.Bn, w, l.N, &n0, l.N, grads + l.bott_w, nullptr, 0));
    OK32(col_sum(c, G, ldg, l.Bn, grads + l.bott_b));
    OK32(dgrad_f(c, G, ldg, params + l.bott_w, l.Bn, l.N, e, l.N, nullptr, 0, 1, 0, 0));
    OK32(norm_backward_f(c, e, w, nullptr, c.tab(0), params + l.ln_g, l.N, dwm, dz, grads + l.ln_g,
                         grads + l.ln_b, nullptr, nullptr, false));
  } else {
  OK32(conv1x1_wgrad(c, G, ldg, wn, l.N, l.Bn, l.N, grads + l.bott_w));
  OK32(col_sum(c, G, ldg, l.Bn, grads + l.bott_b));
  OK32(conv1x1_dgrad(c, G, ldg, params + l.bott_w, l.Bn, l.N, e, l.N, 0));
  OK32(norm_backward(c, e, w, nullptr, c.tab(0), params + l.ln_g, l.N, dwm, dz, grads + l.ln_g,
                     grads + l.ln_b, nullptr));
  }
  if (fused)
    OK32(wgrad_pairs_f(c, dz, l.N, l.N, c.f(ws.wavep), l.hop, l.K, (int)T, B, T*l.N, ws.Lp, grads + l.enc_w));
  else
  OK32(gemm32(c, dz, c.f(ws.wavep), grads + l.enc_w, 1, l.N, l.K, T, l.N, l.hop, l.K, 0, 0, 0, 1, 0, B,
              T*l.N, ws.Lp, nullptr, 1));
  }   // tail
  HIP_OK32(hipGetLastError());
  (void)wave; (void)wn;
  return 0;
}

}  // extern "C"
