// Building blocks of the SGMSE+ score network (NCSN++ / ADM style U-Net), forward values.
// Reference: brever/models/sgmse/net.py:12-477 (DiffusionUNet, UNetBlock, AttentionBlock,
// NoiseEmbedding, GaussianFourierProjection, GroupNorm) and brever/modules/resampling.py:8-61
// (Resample). Convolutions are brv_conv2d_forward (dccrn.hip), matrix products brv_gemm_f32.
// fp32, NCHW, correctness-first kernels.
#include <hip/hip_runtime.h>
#include <math.h>

#include "../../include/brever_hip.h"
#include "common.cuh"

using namespace brv;

namespace {

#define SG_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return (int)e_; } while (0)

dim3 flat_grid(long long n) {
  long long g = (n + 255)/256;
  if (g < 1) g = 1;
  if (g > 8192) g = 8192;
  return dim3((unsigned)g);
}
#define GRID_STRIDE(i, n) \
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < (n); i += (long long)gridDim.x*256)

__device__ __forceinline__ float silu(float v) { return v/(1.f + expf(-v)); }

// GroupNorm over (channels of the group, H, W) per item, input x + add[b][c] (nullable), then
// optional SiLU. One workgroup per (item, group).
__global__ __launch_bounds__(256) void groupnorm_kernel(const float* x, const float* add,
                                                        const float* gamma, const float* beta,
                                                        float* y, int C, long long HW, int groups,
                                                        float eps, int act) {
  __shared__ double scr[8];
  __shared__ float stat[2];
  const int b = blockIdx.x / groups, g = blockIdx.x % groups;
  const int cpg = C/groups;
  const long long n = (long long)cpg*HW;
  const float* xg = x + ((long long)b*C + (long long)g*cpg)*HW;
  float* yg = y + ((long long)b*C + (long long)g*cpg)*HW;
  const float* ag = add ? add + (long long)b*C + (long long)g*cpg : nullptr;
  double s = 0.0, q = 0.0;
  for (long long i = threadIdx.x; i < n; i += 256) {
    const float v = xg[i] + (ag ? ag[i / HW] : 0.f);
    s += v; q += (double)v*v;
  }
  s = block_sum(s, scr); __syncthreads();
  q = block_sum(q, scr);
  if (threadIdx.x == 0) {
    const double mean = s/n;
    double var = q/n - mean*mean;
    if (var < 0) var = 0;
    stat[0] = (float)mean; stat[1] = (float)(1.0/sqrt(var + eps));
  }
  __syncthreads();
  const float mean = stat[0], rstd = stat[1];
  for (long long i = threadIdx.x; i < n; i += 256) {
    const int c = g*cpg + (int)(i / HW);
    float v = (xg[i] + (ag ? ag[i / HW] : 0.f) - mean)*rstd*gamma[c] + beta[c];
    yg[i] = act ? silu(v) : v;
  }
}

__global__ __launch_bounds__(256) void silu_kernel(const float* x, float* y, long long n) {
  GRID_STRIDE(i, n) y[i] = silu(x[i]);
}

// softmax over the last dimension, one workgroup per row
__global__ __launch_bounds__(256) void softmax_kernel(const float* x, float* y, int cols) {
  __shared__ float red[8];
  __shared__ float bc[2];
  const float* xr = x + (long long)blockIdx.x*cols;
  float* yr = y + (long long)blockIdx.x*cols;
  float m = -INFINITY;
  for (int i = threadIdx.x; i < cols; i += 256) m = fmaxf(m, xr[i]);
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) bc[0] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  m = bc[0];
  float s = 0.f;
  for (int i = threadIdx.x; i < cols; i += 256) s += expf(xr[i] - m);
  s = block_sum(s, red);
  if (threadIdx.x == 0) bc[1] = s;
  __syncthreads();
  const float inv = 1.f/bc[1];
  for (int i = threadIdx.x; i < cols; i += 256) yr[i] = expf(xr[i] - m)*inv;
}

// Depthwise FIR resampling by 2 (Resample.forward): down = conv2d(stride 2, padding (ph, pw));
// up = conv_transpose2d(stride 2, padding, output_padding) with the kernel times `gain`.
__global__ __launch_bounds__(256) void fir_down_kernel(const float* x, const float* k, float* y,
                                                       long long planes, int H, int W, int Ho,
                                                       int Wo, int K, int ph, int pw) {
  const long long total = planes*Ho*Wo;
  GRID_STRIDE(idx, total) {
    const int wo = (int)(idx % Wo), ho = (int)((idx / Wo) % Ho);
    const long long pl = idx / ((long long)Wo*Ho);
    const float* xp = x + pl*H*W;
    float acc = 0.f;
    for (int i = 0; i < K; ++i) {
      const int hi = ho*2 - ph + i;
      if (hi < 0 || hi >= H) continue;
      for (int j = 0; j < K; ++j) {
        const int wi = wo*2 - pw + j;
        if (wi < 0 || wi >= W) continue;
        acc += xp[(long long)hi*W + wi]*k[i*K + j];
      }
    }
    y[idx] = acc;
  }
}
__global__ __launch_bounds__(256) void fir_up_kernel(const float* x, const float* k, float* y,
                                                     long long planes, int H, int W, int Ho, int Wo,
                                                     int K, int ph, int pw, float gain) {
  const long long total = planes*Ho*Wo;
  GRID_STRIDE(idx, total) {
    const int wo = (int)(idx % Wo), ho = (int)((idx / Wo) % Ho);
    const long long pl = idx / ((long long)Wo*Ho);
    const float* xp = x + pl*H*W;
    float acc = 0.f;
    for (int i = 0; i < K; ++i) {
      const int hn = ho + ph - i;
      if (hn < 0 || (hn & 1)) continue;
      const int hi = hn >> 1;
      if (hi >= H) continue;
      for (int j = 0; j < K; ++j) {
        const int wn = wo + pw - j;
        if (wn < 0 || (wn & 1)) continue;
        const int wi = wn >> 1;
        if (wi >= W) continue;
        acc += xp[(long long)hi*W + wi]*k[i*K + j];
      }
    }
    y[idx] = gain*acc;
  }
}

// out = alpha*a + beta*b  (b nullable)
__global__ __launch_bounds__(256) void axpby_kernel(const float* a, float alpha, const float* b,
                                                    float beta, float* out, long long n) {
  GRID_STRIDE(i, n) out[i] = alpha*a[i] + (b ? beta*b[i] : 0.f);
}

// out[i] = [sin(2 pi x_i b_j) | cos(2 pi x_i b_j)]   (GaussianFourierProjection)
__global__ __launch_bounds__(256) void fourier_kernel(const float* x, const float* b, float* out,
                                                      int n, int m) {
  GRID_STRIDE(idx, (long long)n*m) {
    const int i = (int)(idx / m), j = (int)(idx % m);
    const float v = 6.283185307179586f*(x[i]*b[j]);      // rounding order of 2*pi*outer(x, b)
    out[(long long)i*2*m + j] = sinf(v);
    out[(long long)i*2*m + m + j] = cosf(v);
  }
}

}  // namespace

extern "C" {

int brv_groupnorm_forward(const float* x, const float* add_bc, const float* gamma,
                          const float* beta, float* y, int64_t B, int64_t C, int64_t HW,
                          int64_t groups, float eps, int act_silu, brv_stream_t stream) {
  if (B < 1 || C < 1 || HW < 1 || groups < 1 || C % groups) return -1;
  hipLaunchKernelGGL(groupnorm_kernel, dim3((unsigned)(B*groups)), dim3(256), 0,
                     (hipStream_t)stream, x, add_bc, gamma, beta, y, (int)C, (long long)HW,
                     (int)groups, eps, act_silu);
  SG_OK(hipGetLastError());
  return 0;
}
int brv_silu(const float* x, float* y, int64_t n, brv_stream_t stream) {
  if (n < 1) return -1;
  hipLaunchKernelGGL(silu_kernel, flat_grid(n), dim3(256), 0, (hipStream_t)stream, x, y, (long long)n);
  SG_OK(hipGetLastError());
  return 0;
}
int brv_softmax_rows(const float* x, float* y, int64_t rows, int64_t cols, brv_stream_t stream) {
  if (rows < 1 || cols < 1) return -1;
  hipLaunchKernelGGL(softmax_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, y,
                     (int)cols);
  SG_OK(hipGetLastError());
  return 0;
}
int brv_fir_resample2d(const float* x, const float* kernel, float* y, int64_t planes, int64_t H,
                       int64_t W, int64_t Ho, int64_t Wo, int64_t K, int64_t pad_h, int64_t pad_w,
                       int up, float gain, brv_stream_t stream) {
  if (planes < 1 || Ho < 1 || Wo < 1) return -1;
  if (up)
    hipLaunchKernelGGL(fir_up_kernel, flat_grid(planes*Ho*Wo), dim3(256), 0, (hipStream_t)stream, x,
                       kernel, y, (long long)planes, (int)H, (int)W, (int)Ho, (int)Wo, (int)K,
                       (int)pad_h, (int)pad_w, gain);
  else
    hipLaunchKernelGGL(fir_down_kernel, flat_grid(planes*Ho*Wo), dim3(256), 0, (hipStream_t)stream,
                       x, kernel, y, (long long)planes, (int)H, (int)W, (int)Ho, (int)Wo, (int)K,
                       (int)pad_h, (int)pad_w);
  SG_OK(hipGetLastError());
  return 0;
}
int brv_axpby(const float* a, float alpha, const float* b, float beta, float* out, int64_t n,
              brv_stream_t stream) {
  if (n < 1) return -1;
  hipLaunchKernelGGL(axpby_kernel, flat_grid(n), dim3(256), 0, (hipStream_t)stream, a, alpha, b,
                     beta, out, (long long)n);
  SG_OK(hipGetLastError());
  return 0;
}
int brv_fourier_features(const float* x, const float* b, float* out, int64_t n, int64_t m,
                         brv_stream_t stream) {
  if (n < 1 || m < 1) return -1;
  hipLaunchKernelGGL(fourier_kernel, flat_grid(n*m), dim3(256), 0, (hipStream_t)stream, x, b, out,
                     (int)n, (int)m);
  SG_OK(hipGetLastError());
  return 0;
}

}  // extern "C"
