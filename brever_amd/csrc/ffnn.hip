// Elementwise / small-reduction kernels of the FFNN mask model and its log-mel features.
// Reference: brever/models/ffnn/ffnn.py:72-203 (transform, irm, stack, _FFNN,
// StaticNormalizer, CumulativeNormalizer) and brever/modules/features.py:142-205 (fbe).
// The dense parts (mel filterbank, Linear layers and their gradients) are brv_gemm_f32 /
// brv_matmul_f32 in stft.hip. All tensors fp32, layouts as in the reference:
// features/labels (B, rows, frames), spectra complex64 (B, channels, bins, frames).
#include <hip/hip_runtime.h>
#include <math.h>

#include "../../include/brever_hip.h"
#include "common.cuh"

using namespace brv;

namespace {

#define FF_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return (int)e_; } while (0)

dim3 flat_grid(long long n) {
  long long g = (n + 255)/256;
  if (g < 1) g = 1;
  if (g > 4096) g = 4096;
  return dim3((unsigned)g);
}
#define GRID_STRIDE(i, n) \
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < (n); i += (long long)gridDim.x*256)

// out[b][i] = mean_c |spec[b][c][i]|^2                      (features.py:186-188)
__global__ __launch_bounds__(256) void fbe_power_kernel(const float2* spec, float* out, int C,
                                                        long long n, long long total) {
  GRID_STRIDE(idx, total) {
    const long long b = idx / n, i = idx % n;
    float s = 0.f;
    for (int c = 0; c < C; ++c) {
      const float2 v = spec[((long long)b*C + c)*n + i];
      s += v.x*v.x + v.y*v.y;
    }
    out[idx] = s/(float)C;
  }
}
// mode 1: log(x + eps); mode 2: x^(1/3)                      (features.py:194-198)
__global__ __launch_bounds__(256) void compress_kernel(const float* x, float* out, long long n,
                                                       int mode, float eps) {
  GRID_STRIDE(i, n)
    out[i] = mode == 1 ? logf(x[i] + eps) : (mode == 2 ? cbrtf(x[i]) : (mode == 3 ? sqrtf(x[i]) : x[i]));
}
// binaural cues per time-frequency unit of a 2-channel spectrum (features.py:222-262):
// mode 0: ILD = 20 log10((|X_R| + eps)/(|X_L| + eps)); mode 1: IPD = angle(X_R) - angle(X_L)
__global__ __launch_bounds__(256) void binaural_kernel(const float2* spec, float* out, long long n,
                                                       long long total, int mode, float eps) {
  GRID_STRIDE(idx, total) {
    const long long b = idx / n, i = idx % n;
    const float2 l = spec[(b*2)*n + i], r = spec[(b*2 + 1)*n + i];
    if (mode == 0)
      out[idx] = 20.f*log10f((sqrtf(r.x*r.x + r.y*r.y) + eps)/(sqrtf(l.x*l.x + l.y*l.y) + eps));
    else
      out[idx] = atan2f(r.y, r.x) - atan2f(l.y, l.x);
  }
}
// Interaural coherence per time-frequency unit (features.py:263-293): exponentially weighted
// auto- and cross-power spectra along the frames, phi[t] = (1 - alpha) x[t] + alpha phi[t-1]
// (torchaudio.functional.lfilter with a = [1, -alpha], b = [1 - alpha, 0], whose default
// clamp=True limits every OUTPUT sample to [-1, 1] while the recursion runs on the unclamped
// state), then |phi_lr|^2 / (phi_ll phi_rr). One thread per (item, bin) walks the frames.
__global__ __launch_bounds__(256) void ic_kernel(const float2* spec, float* out, int bins, int F,
                                                 long long total, float alpha) {
  GRID_STRIDE(idx, total) {                       // idx over (b, bin)
    const long long b = idx / bins; const int k = (int)(idx % bins);
    const float2* l = spec + ((b*2)*bins + k)*(long long)F;
    const float2* r = spec + ((b*2 + 1)*bins + k)*(long long)F;
    float* o = out + idx*F;
    float pll = 0.f, prr = 0.f, pre = 0.f, pim = 0.f;
    const float beta = 1.f - alpha;
    for (int t = 0; t < F; ++t) {
      const float2 a = l[t], c = r[t];
      // x_lr = |L||R| exp(j(angle L - angle R)) = L conj(R)
      pll = beta*(a.x*a.x + a.y*a.y) + alpha*pll;
      prr = beta*(c.x*c.x + c.y*c.y) + alpha*prr;
      pre = beta*(a.x*c.x + a.y*c.y) + alpha*pre;
      pim = beta*(a.y*c.x - a.x*c.y) + alpha*pim;
      const float cl = fminf(fmaxf(pll, -1.f), 1.f), cr = fminf(fmaxf(prr, -1.f), 1.f);
      const float ce = fminf(fmaxf(pre, -1.f), 1.f), ci = fminf(fmaxf(pim, -1.f), 1.f);
      // the reference takes sqrt(re^2 + im^2) and squares it again
      const float mag = sqrtf(ce*ce + ci*ci);
      o[t] = mag*mag/(cl*cr);
    }
  }
}
// x[b][m][t] /= sum_m x[b][m][t] + eps                      (features.py:190-191, 'pdf')
__global__ __launch_bounds__(256) void col_normalize_kernel(float* x, int M, int T, long long total,
                                                            float eps) {
  GRID_STRIDE(idx, total) {                       // idx over (b, t)
    const long long b = idx / T; const int t = (int)(idx % T);
    float* col = x + b*M*T + t;
    float s = 0.f;
    for (int m = 0; m < M; ++m) s += col[(long long)m*T];
    const float inv = 1.f/(s + eps);
    for (int m = 0; m < M; ++m) col[(long long)m*T] *= inv;
  }
}
// out (B, 3M, T) = [x | first difference | second difference] along the frames, zero-padded
// on the left                                               (features.py:207-218)
__global__ __launch_bounds__(256) void deltas_kernel(const float* x, float* out, int M, int T,
                                                     long long total) {
  GRID_STRIDE(idx, total) {
    const int t = (int)(idx % T);
    const long long bm = idx / T, b = bm / M, m = bm % M;
    const float v0 = x[idx], v1 = t >= 1 ? x[idx - 1] : 0.f, v2 = t >= 2 ? x[idx - 2] : 0.f;
    float* o = out + (b*3*M + m)*T + t;
    o[0] = v0;
    o[(long long)M*T] = t >= 1 ? v0 - v1 : 0.f;
    o[(long long)2*M*T] = t >= 2 ? v0 - 2.f*v1 + v2 : 0.f;
  }
}
// (1 + bg/(fg + eps))^(-1/2)                                 (ffnn.py:121-128)
__global__ __launch_bounds__(256) void irm_kernel(const float* fg, const float* bg, float* out,
                                                  long long n, float eps) {
  GRID_STRIDE(i, n) out[i] = 1.f/sqrtf(1.f + bg[i]/(fg[i] + eps));
}
// out[b][k*nf + f][t] = x[b][f][max(t - k, 0)], k = 0..stacks        (ffnn.py:130-140)
__global__ __launch_bounds__(256) void stack_kernel(const float* x, float* out, int nf, int T,
                                                    int stacks, long long total) {
  GRID_STRIDE(idx, total) {
    const int t = (int)(idx % T);
    const long long r = idx / T;
    const int row = (int)(r % ((long long)(stacks + 1)*nf));
    const long long b = r / ((long long)(stacks + 1)*nf);
    const int k = row / nf, f = row % nf;
    const int ts = t - k < 0 ? 0 : t - k;
    out[idx] = x[((long long)b*nf + f)*T + ts];
  }
}
// (x - mean[row])/std[row]                                    (ffnn.py:186-187)
__global__ __launch_bounds__(256) void static_norm_kernel(const float* x, const float* mean,
                                                          const float* stdv, float* out, int rows,
                                                          int T, long long total) {
  GRID_STRIDE(idx, total) {
    const int row = (int)((idx / T) % rows);
    out[idx] = (x[idx] - mean[row])/stdv[row];
  }
}
// running mean / variance along time, one thread per (b, row)   (ffnn.py:195-203)
__global__ __launch_bounds__(256) void cumulative_norm_kernel(const float* x, float* out, int T,
                                                              long long nrows, float eps) {
  GRID_STRIDE(r, nrows) {
    const float* xi = x + r*T;
    float* oi = out + r*T;
    float s = 0.f, q = 0.f;
    for (int t = 0; t < T; ++t) {
      s += xi[t]; q += xi[t]*xi[t];
      const float n = (float)(t + 1);
      const float mean = s/n;
      const float var = q/n - mean*mean;
      oi[t] = (xi[t] - mean)/sqrtf(var + eps);
    }
  }
}
// y = relu(x)*mask*scale (mask: the dropout keep mask, or null)         (ffnn.py:160-162)
__global__ __launch_bounds__(256) void relu_dropout_fwd_kernel(const float* x, const float* mask,
                                                               float* out, long long n, float scale) {
  GRID_STRIDE(i, n) {
    const float v = x[i] > 0.f ? x[i] : 0.f;
    out[i] = mask ? v*mask[i]*scale : v;
  }
}
__global__ __launch_bounds__(256) void relu_dropout_bwd_kernel(const float* x, const float* mask,
                                                               const float* dy, float* dx,
                                                               long long n, float scale) {
  GRID_STRIDE(i, n) {
    const float g = x[i] > 0.f ? dy[i] : 0.f;
    dx[i] = mask ? g*mask[i]*scale : g;
  }
}
// y = x*mask*scale: nn.Dropout with a caller-drawn keep mask (forward, and backward on dy)
__global__ __launch_bounds__(256) void dropout_apply_kernel(const float* x, const float* mask,
                                                            float* out, long long n, float scale) {
  GRID_STRIDE(i, n) out[i] = x[i]*mask[i]*scale;
}
__global__ __launch_bounds__(256) void sigmoid_fwd_kernel(const float* x, float* out, long long n) {
  GRID_STRIDE(i, n) out[i] = 1.f/(1.f + expf(-x[i]));
}
__global__ __launch_bounds__(256) void sigmoid_bwd_kernel(const float* y, const float* dy,
                                                          float* dx, long long n) {
  GRID_STRIDE(i, n) dx[i] = dy[i]*y[i]*(1.f - y[i]);
}
// out[m] = sum_{b,t} x[b][m][t]  (bias gradient): one workgroup per row, fixed order
// (long rows: gridDim.y slices per row write fp64 partials that a second kernel adds in order)
__global__ __launch_bounds__(256) void row_sum_kernel(const float* x, float* out, double* part, int B,
                                                      int M, int T) {
  __shared__ double scr[8];
  const int m = blockIdx.x;
  const long long n = (long long)B*T;
  const long long per = (n + gridDim.y - 1)/gridDim.y;
  const long long lo = (long long)blockIdx.y*per, hi = lo + per < n ? lo + per : n;
  double s = 0.0;
  for (long long e = lo + threadIdx.x; e < hi; e += 256)
    s += (double)x[((e / T)*M + m)*T + e % T];
  s = block_sum(s, scr);
  if (threadIdx.x == 0) {
    if (part) part[(long long)m*gridDim.y + blockIdx.y] = s; else out[m] = (float)s;
  }
}
// the same sums when T % 4 == 0: blockIdx.y = (batch item, piece of the row) -- contiguous 16-byte loads, no
// 64-bit division per element (the DCCRN bias gradients: 131 MB tensors)
__global__ __launch_bounds__(256) void row_sum4_kernel(const float* x, double* part, int M, long long T, int pieces) {
  __shared__ double scr[8];
  const int m = blockIdx.x, b = blockIdx.y / pieces, pc = blockIdx.y % pieces;
  const long long n4 = T >> 2, per = (n4 + pieces - 1)/pieces;
  const long long lo = pc*per, hi = lo + per < n4 ? lo + per : n4;
  const float4* src = reinterpret_cast<const float4*>(x + ((long long)b*M + m)*T);
  double s = 0.0;
  for (long long i = lo + threadIdx.x; i < hi; i += 256) {
    const float4 v = src[i];
    s += (double)((v.x + v.y) + (v.z + v.w));
  }
  s = block_sum(s, scr);
  if (threadIdx.x == 0) part[(long long)m*gridDim.y + blockIdx.y] = s;
}
__global__ __launch_bounds__(256) void row_sum_final_kernel(const double* part, float* out, int M,
                                                            int slices) {
  const int m = blockIdx.x*256 + threadIdx.x;
  if (m >= M) return;
  double s = 0.0;
  for (int i = 0; i < slices; ++i) s += part[(long long)m*slices + i];
  out[m] = (float)s;
}
// out[b][i] = mask[b][i] * mean_c spec[b][c][i]   (complex)           (ffnn.py:113-115)
__global__ __launch_bounds__(256) void masked_mean_spec_kernel(const float2* spec, const float* mask,
                                                               float2* out, int C, long long n,
                                                               long long total) {
  GRID_STRIDE(idx, total) {
    const long long b = idx / n, i = idx % n;
    float re = 0.f, im = 0.f;
    for (int c = 0; c < C; ++c) {
      const float2 v = spec[((long long)b*C + c)*n + i];
      re += v.x; im += v.y;
    }
    const float m = mask[idx]/(float)C;
    out[idx] = make_float2(re*m, im*m);
  }
}

}  // namespace

extern "C" {

int brv_fbe_power(const float* spec, float* out, int64_t B, int64_t C, int64_t n, brv_stream_t stream) {
  if (B < 1 || C < 1 || n < 1) return -1;
  hipLaunchKernelGGL(fbe_power_kernel, flat_grid(B*n), dim3(256), 0, (hipStream_t)stream,
                     (const float2*)spec, out, (int)C, (long long)n, (long long)(B*n));
  FF_OK(hipGetLastError());
  return 0;
}
int brv_compress(const float* x, float* out, int64_t n, int mode, float eps, brv_stream_t stream) {
  if (n < 1 || mode < 0 || mode > 3) return -1;
  hipLaunchKernelGGL(compress_kernel, flat_grid(n), dim3(256), 0, (hipStream_t)stream, x, out,
                     (long long)n, mode, eps);
  FF_OK(hipGetLastError());
  return 0;
}
int brv_binaural(const float* spec, float* out, int64_t B, int64_t n, int mode, float eps,
                 brv_stream_t stream) {
  if (B < 1 || n < 1 || mode < 0 || mode > 1) return -1;
  hipLaunchKernelGGL(binaural_kernel, flat_grid(B*n), dim3(256), 0, (hipStream_t)stream,
                     (const float2*)spec, out, (long long)n, (long long)(B*n), mode, eps);
  FF_OK(hipGetLastError());
  return 0;
}
int brv_interaural_coherence(const float* spec, float* out, int64_t B, int64_t bins, int64_t F,
                             float alpha, brv_stream_t stream) {
  if (B < 1 || bins < 1 || F < 1) return -1;
  hipLaunchKernelGGL(ic_kernel, flat_grid(B*bins), dim3(256), 0, (hipStream_t)stream,
                     (const float2*)spec, out, (int)bins, (int)F, (long long)(B*bins), alpha);
  FF_OK(hipGetLastError());
  return 0;
}
int brv_col_normalize(float* x, int64_t B, int64_t M, int64_t T, float eps, brv_stream_t stream) {
  if (B < 1 || M < 1 || T < 1) return -1;
  hipLaunchKernelGGL(col_normalize_kernel, flat_grid(B*T), dim3(256), 0, (hipStream_t)stream, x,
                     (int)M, (int)T, (long long)(B*T), eps);
  FF_OK(hipGetLastError());
  return 0;
}
int brv_deltas(const float* x, float* out, int64_t B, int64_t M, int64_t T, brv_stream_t stream) {
  if (B < 1 || M < 1 || T < 1) return -1;
  hipLaunchKernelGGL(deltas_kernel, flat_grid(B*M*T), dim3(256), 0, (hipStream_t)stream, x, out,
                     (int)M, (int)T, (long long)(B*M*T));
  FF_OK(hipGetLastError());
  return 0;
}
int brv_irm(const float* fg, const float* bg, float* out, int64_t n, float eps, brv_stream_t stream) {
  if (n < 1) return -1;
  hipLaunchKernelGGL(irm_kernel, flat_grid(n), dim3(256), 0, (hipStream_t)stream, fg, bg, out,
                     (long long)n, eps);
  FF_OK(hipGetLastError());
  return 0;
}
int brv_stack_frames(const float* x, float* out, int64_t B, int64_t nf, int64_t T, int64_t stacks,
                     brv_stream_t stream) {
  if (B < 1 || nf < 1 || T < 1 || stacks < 0) return -1;
  const long long total = B*(stacks + 1)*nf*T;
  hipLaunchKernelGGL(stack_kernel, flat_grid(total), dim3(256), 0, (hipStream_t)stream, x, out,
                     (int)nf, (int)T, (int)stacks, total);
  FF_OK(hipGetLastError());
  return 0;
}
int brv_static_norm(const float* x, const float* mean, const float* stdv, float* out, int64_t B,
                    int64_t rows, int64_t T, brv_stream_t stream) {
  if (B < 1 || rows < 1 || T < 1) return -1;
  const long long total = B*rows*T;
  hipLaunchKernelGGL(static_norm_kernel, flat_grid(total), dim3(256), 0, (hipStream_t)stream, x,
                     mean, stdv, out, (int)rows, (int)T, total);
  FF_OK(hipGetLastError());
  return 0;
}
int brv_cumulative_norm(const float* x, float* out, int64_t nrows, int64_t T, float eps,
                        brv_stream_t stream) {
  if (nrows < 1 || T < 1) return -1;
  hipLaunchKernelGGL(cumulative_norm_kernel, flat_grid(nrows), dim3(256), 0, (hipStream_t)stream,
                     x, out, (int)T, (long long)nrows, eps);
  FF_OK(hipGetLastError());
  return 0;
}
int brv_relu_dropout_forward(const float* x, const float* mask, float* out, int64_t n, float scale,
                             brv_stream_t stream) {
  if (n < 1) return -1;
  hipLaunchKernelGGL(relu_dropout_fwd_kernel, flat_grid(n), dim3(256), 0, (hipStream_t)stream, x,
                     mask, out, (long long)n, scale);
  FF_OK(hipGetLastError());
  return 0;
}
int brv_relu_dropout_backward(const float* x, const float* mask, const float* dy, float* dx,
                              int64_t n, float scale, brv_stream_t stream) {
  if (n < 1) return -1;
  hipLaunchKernelGGL(relu_dropout_bwd_kernel, flat_grid(n), dim3(256), 0, (hipStream_t)stream, x,
                     mask, dy, dx, (long long)n, scale);
  FF_OK(hipGetLastError());
  return 0;
}
int brv_dropout_apply(const float* x, const float* mask, float* out, int64_t n, float scale,
                      brv_stream_t stream) {
  if (n < 1 || !mask) return -1;
  hipLaunchKernelGGL(dropout_apply_kernel, flat_grid(n), dim3(256), 0, (hipStream_t)stream, x, mask,
                     out, (long long)n, scale);
  FF_OK(hipGetLastError());
  return 0;
}
int brv_sigmoid_forward(const float* x, float* out, int64_t n, brv_stream_t stream) {
  if (n < 1) return -1;
  hipLaunchKernelGGL(sigmoid_fwd_kernel, flat_grid(n), dim3(256), 0, (hipStream_t)stream, x, out,
                     (long long)n);
  FF_OK(hipGetLastError());
  return 0;
}
int brv_sigmoid_backward(const float* y, const float* dy, float* dx, int64_t n, brv_stream_t stream) {
  if (n < 1) return -1;
  hipLaunchKernelGGL(sigmoid_bwd_kernel, flat_grid(n), dim3(256), 0, (hipStream_t)stream, y, dy,
                     dx, (long long)n);
  FF_OK(hipGetLastError());
  return 0;
}
int brv_row_sum(const float* x, float* out, int64_t B, int64_t M, int64_t T, brv_stream_t stream) {
  if (B < 1 || M < 1 || T < 1) return -1;
  hipStream_t st = (hipStream_t)stream;
  long long slices = (B*T + 16383)/16384;
  if (slices > 64) slices = 64;
  if (slices > 1 && (T & 3) == 0 && B <= 1024 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    long long pieces = (T + 16383)/16384;
    if (pieces > 4) pieces = 4;
    const long long sl = B*pieces;
    double* part = nullptr;
    FF_OK(hipMallocAsync((void**)&part, (size_t)M*sl*sizeof(double), st));
    hipLaunchKernelGGL(row_sum4_kernel, dim3((unsigned)M, (unsigned)sl), dim3(256), 0, st, x, part, (int)M,
                       (long long)T, (int)pieces);
    hipLaunchKernelGGL(row_sum_final_kernel, dim3((unsigned)((M + 255)/256)), dim3(256), 0, st, part,
                       out, (int)M, (int)sl);
    FF_OK(hipFreeAsync(part, st));
    FF_OK(hipGetLastError());
    return 0;
  }
  if (slices <= 1) {
    hipLaunchKernelGGL(row_sum_kernel, dim3((unsigned)M, 1), dim3(256), 0, st, x, out,
                       (double*)nullptr, (int)B, (int)M, (int)T);
  } else {
    double* part = nullptr;
    FF_OK(hipMallocAsync((void**)&part, (size_t)M*slices*sizeof(double), st));
    hipLaunchKernelGGL(row_sum_kernel, dim3((unsigned)M, (unsigned)slices), dim3(256), 0, st, x, out,
                       part, (int)B, (int)M, (int)T);
    hipLaunchKernelGGL(row_sum_final_kernel, dim3((unsigned)((M + 255)/256)), dim3(256), 0, st, part,
                       out, (int)M, (int)slices);
    FF_OK(hipFreeAsync(part, st));
  }
  FF_OK(hipGetLastError());
  return 0;
}
int brv_masked_mean_spec(const float* spec, const float* mask, float* out, int64_t B, int64_t C,
                         int64_t n, brv_stream_t stream) {
  if (B < 1 || C < 1 || n < 1) return -1;
  hipLaunchKernelGGL(masked_mean_spec_kernel, flat_grid(B*n), dim3(256), 0, (hipStream_t)stream,
                     (const float2*)spec, mask, (float2*)out, (int)C, (long long)n, (long long)(B*n));
  FF_OK(hipGetLastError());
  return 0;
}

}  // extern "C"
