// Causal (cumulative) group normalisation as a standalone operator, fp32:
// brever/modules/normalization.py:5-62 (CausalGroupNorm and its LayerNorm / InstanceNorm
// specialisations). x is viewed as (BG = batch x groups, R = channels-of-the-group x inner
// dims, T frames, time last); frame t is normalised with the mean and variance of everything
// up to and including frame t:
//   S1_t = sum_{tau<=t} sum_r x, S2_t likewise of x^2, n_t = R (t+1),
//   mean_t = S1_t/n_t, var_t = S2_t/n_t - mean_t^2, y = (x - mean_t) rstd_t gain_c + bias_c.
// Forward: per-frame sums (coalesced along t) -> blocked prefix scan per (item, group) in fp64
// -> apply. Backward: per-frame sums of dxhat and dxhat*xhat -> SUFFIX scan of the gradients of
// S1 / S2 (a frame's statistics feed every later frame) -> dx = dxhat rstd + U + 2 x V; the
// gain / bias gradients are per-channel reductions.
#include <hip/hip_runtime.h>
#include <math.h>

#include "../../include/brever_hip.h"
#include "common.cuh"

using namespace brv;

namespace {

#define CN_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return (int)e_; } while (0)

// out[bg][t] = (sum_r a, sum_r b) with (a, b) = (x, x^2) or (dxhat, dxhat*xhat)
template <bool BWD>
__global__ __launch_bounds__(256) void cgn_frame_sums_kernel(const float* x, const float* dy,
                                                             const float* gain, const float2* stats,
                                                             double2* out, int R, int T, int G,
                                                             int inner) {
  const long long bg = blockIdx.y;
  const int t = blockIdx.x*256 + threadIdx.x;
  if (t >= T) return;
  const float* xp = x + bg*R*(long long)T + t;
  const float* dp = BWD ? dy + bg*R*(long long)T + t : nullptr;
  const int c0 = (int)(bg % G)*(R/inner);
  float2 st = make_float2(0.f, 1.f);
  if (BWD) st = stats[bg*T + t];
  double a = 0.0, b = 0.0;
  for (int r = 0; r < R; ++r) {
    const float v = xp[(long long)r*T];
    if (BWD) {
      const float dxh = dp[(long long)r*T]*gain[c0 + r/inner];
      a += dxh; b += (double)dxh*((v - st.x)*st.y);
    } else {
      a += v; b += (double)v*v;
    }
  }
  out[bg*T + t] = make_double2(a, b);
}

// inclusive prefix (REV false) or suffix (REV true) scan over the T frames of one (item,
// group): each of the 256 threads owns a contiguous segment
template <bool REV, typename F>
__device__ __forceinline__ void blocked_scan(double2* vals, int T, F finish) {
  __shared__ double2 tot[256];
  const int seg = (T + 255)/256;
  const int lo = threadIdx.x*seg, hi = min(T, lo + seg);
  double2 s = make_double2(0.0, 0.0);
  for (int k = lo; k < hi; ++k) {
    const int t = REV ? T - 1 - k : k;
    s.x += vals[t].x; s.y += vals[t].y;
  }
  tot[threadIdx.x] = s;
  __syncthreads();
  double2 base = make_double2(0.0, 0.0);
  for (int j = 0; j < (int)threadIdx.x; ++j) { base.x += tot[j].x; base.y += tot[j].y; }
  for (int k = lo; k < hi; ++k) {
    const int t = REV ? T - 1 - k : k;
    base.x += vals[t].x; base.y += vals[t].y;
    finish(t, base);
  }
}

__global__ __launch_bounds__(256) void cgn_scan_kernel(double2* fs, float2* stats, int R, int T,
                                                       float eps) {
  double2* v = fs + (long long)blockIdx.x*T;
  float2* st = stats + (long long)blockIdx.x*T;
  blocked_scan<false>(v, T, [&](int t, const double2& c) {
    const double n = (double)R*(t + 1);
    const double mean = c.x/n;
    const double var = c.y/n - mean*mean;
    st[t] = make_float2((float)mean, (float)(1.0/sqrt(var + (double)eps)));
  });
}
// in: (A_t, B_t) -> (dS1_t, dS2_t) in place, then suffix sums (U, V)
__global__ __launch_bounds__(256) void cgn_bwd_scan_kernel(double2* ab, const float2* stats,
                                                           float2* uv, int R, int T) {
  double2* v = ab + (long long)blockIdx.x*T;
  const float2* st = stats + (long long)blockIdx.x*T;
  for (int t = threadIdx.x; t < T; t += 256) {
    const double n = (double)R*(t + 1), mean = st[t].x, r = st[t].y;
    const double A = v[t].x, Bq = v[t].y;
    // d mean = -r A, d var = -B r^2/2 (B = sum dxhat*xhat); S1 feeds the mean and, through
    // mean^2, the variance
    v[t] = make_double2((-r*A + Bq*r*r*mean)/n, -0.5*Bq*r*r/n);
  }
  __syncthreads();
  float2* o = uv + (long long)blockIdx.x*T;
  blocked_scan<true>(v, T, [&](int t, const double2& c) { o[t] = make_float2((float)c.x, (float)c.y); });
}
__global__ __launch_bounds__(256) void cgn_apply_kernel(const float* x, const float2* stats,
                                                        const float* gain, const float* bias,
                                                        float* y, int R, int T, int G, int inner,
                                                        long long total) {
  for (long long idx = (long long)blockIdx.x*256 + threadIdx.x; idx < total;
       idx += (long long)gridDim.x*256) {
    const int t = (int)(idx % T);
    const long long row = idx / T, bg = row / R;
    const int c = (int)(bg % G)*(R/inner) + (int)(row % R)/inner;
    const float2 st = stats[bg*T + t];
    y[idx] = (x[idx] - st.x)*st.y*gain[c] + bias[c];
  }
}
__global__ __launch_bounds__(256) void cgn_bwd_apply_kernel(const float* x, const float* dy,
                                                            const float2* stats, const float2* uv,
                                                            const float* gain, float* dx, int R,
                                                            int T, int G, int inner,
                                                            long long total) {
  for (long long idx = (long long)blockIdx.x*256 + threadIdx.x; idx < total;
       idx += (long long)gridDim.x*256) {
    const int t = (int)(idx % T);
    const long long row = idx / T, bg = row / R;
    const int c = (int)(bg % G)*(R/inner) + (int)(row % R)/inner;
    const float2 st = stats[bg*T + t], w = uv[bg*T + t];
    dx[idx] = dy[idx]*gain[c]*st.y + w.x + 2.f*x[idx]*w.y;
  }
}
// dgain[c] = sum dy*xhat, dbias[c] = sum dy over (batch, inner, frames): one workgroup per channel
__global__ __launch_bounds__(256) void cgn_param_grads_kernel(const float* x, const float* dy,
                                                              const float2* stats, float* dgain,
                                                              float* dbias, int B, int C, int G,
                                                              int inner, int T) {
  __shared__ double scr[8];
  const int c = blockIdx.x, cpg = C/G, g = c/cpg;
  double sg = 0.0, sb = 0.0;
  const long long per = (long long)inner*T;
  for (long long e = threadIdx.x; e < (long long)B*per; e += 256) {
    const long long b = e / per, rem = e % per;
    const int t = (int)(rem % T);
    const long long idx = ((b*C + c)*inner)*(long long)T + rem;
    const float2 st = stats[(b*G + g)*(long long)T + t];
    const float d = dy[idx];
    sg += (double)d*((x[idx] - st.x)*st.y); sb += d;
  }
  sg = block_sum(sg, scr); __syncthreads();
  sb = block_sum(sb, scr);
  if (threadIdx.x == 0) { dgain[c] = (float)sg; dbias[c] = (float)sb; }
}

dim3 flat(long long n) { long long g = (n + 255)/256; return dim3((unsigned)(g < 1 ? 1 : (g > 8192 ? 8192 : g))); }

}  // namespace

extern "C" {

int64_t brv_causal_groupnorm_scratch_bytes(int64_t B, int64_t groups, int64_t T) {
  return B*groups*T*(int64_t)sizeof(double2);
}
int brv_causal_groupnorm_forward(const float* x, const float* gain, const float* bias, float* y,
                                 float* stats, void* scratch, int64_t B, int64_t C, int64_t inner,
                                 int64_t T, int64_t groups, float eps, brv_stream_t stream) {
  if (B < 1 || C < 1 || inner < 1 || T < 1 || groups < 1 || C % groups) return -1;
  hipStream_t st = (hipStream_t)stream;
  const int R = (int)(C/groups*inner);
  const dim3 fgrid((unsigned)((T + 255)/256), (unsigned)(B*groups));
  hipLaunchKernelGGL((cgn_frame_sums_kernel<false>), fgrid, dim3(256), 0, st, x, (const float*)nullptr,
                     gain, (const float2*)nullptr, (double2*)scratch, R, (int)T, (int)groups, (int)inner);
  hipLaunchKernelGGL(cgn_scan_kernel, dim3((unsigned)(B*groups)), dim3(256), 0, st, (double2*)scratch,
                     (float2*)stats, R, (int)T, eps);
  const long long total = B*C*inner*T;
  hipLaunchKernelGGL(cgn_apply_kernel, flat(total), dim3(256), 0, st, x, (const float2*)stats, gain,
                     bias, y, R, (int)T, (int)groups, (int)inner, total);
  CN_OK(hipGetLastError());
  return 0;
}
int brv_causal_groupnorm_backward(const float* x, const float* dy, const float* gain,
                                  const float* stats, float* dx, float* dgain, float* dbias,
                                  void* scratch, float* uv_scratch, int64_t B, int64_t C,
                                  int64_t inner, int64_t T, int64_t groups, brv_stream_t stream) {
  if (B < 1 || C < 1 || inner < 1 || T < 1 || groups < 1 || C % groups) return -1;
  hipStream_t st = (hipStream_t)stream;
  const int R = (int)(C/groups*inner);
  const dim3 fgrid((unsigned)((T + 255)/256), (unsigned)(B*groups));
  hipLaunchKernelGGL((cgn_frame_sums_kernel<true>), fgrid, dim3(256), 0, st, x, dy, gain,
                     (const float2*)stats, (double2*)scratch, R, (int)T, (int)groups, (int)inner);
  hipLaunchKernelGGL(cgn_bwd_scan_kernel, dim3((unsigned)(B*groups)), dim3(256), 0, st,
                     (double2*)scratch, (const float2*)stats, (float2*)uv_scratch, R, (int)T);
  const long long total = B*C*inner*T;
  hipLaunchKernelGGL(cgn_bwd_apply_kernel, flat(total), dim3(256), 0, st, x, dy, (const float2*)stats,
                     (const float2*)uv_scratch, gain, dx, R, (int)T, (int)groups, (int)inner, total);
  hipLaunchKernelGGL(cgn_param_grads_kernel, dim3((unsigned)C), dim3(256), 0, st, x, dy,
                     (const float2*)stats, dgain, dbias, (int)B, (int)C, (int)groups, (int)inner,
                     (int)T);
  CN_OK(hipGetLastError());
  return 0;
}

}  // extern "C"
