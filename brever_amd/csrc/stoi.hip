// STOI / ESTOI intelligibility metrics (brever/metrics.py:19-45,98-109: the `stoi` / `estoi`
// entries of MetricRegistry, which the reference computes with the pystoi / batch_pystoi wheels)
// as HIP kernels for a padded batch with per-item lengths. Algorithm and constants: Taal et al.
// 2011 (STOI), Jensen & Taal 2016 (ESTOI) in pystoi's processing order -- see oracle/stoi.py,
// the NumPy restatement these kernels are tested against (parity with the wheel: unpinned).
//
//   brv_resample_poly      polyphase FIR resampling (scipy.signal.resample_poly semantics with a
//                          caller-built, already padded and scaled filter)
//   brv_stoi_compact       drop the frames > 40 dB below the loudest clean frame, overlap-add
//                          the kept Hann frames of both signals; per-item geometry
//   (512-point DFT of the 256-sample frames: one brv_gemm_f32 with a windowed basis, by the caller)
//   brv_stoi_bands         one-third octave band magnitudes of every frame
//   brv_stoi_correlate     30-frame segments: clipped normalised correlation (STOI) or
//                          row / column normalised correlation (ESTOI), mean over segments
// Every item is independent; all kernels are HBM-bound streaming passes over a few MB.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/brever_hip.h"
#include "common.cuh"

using namespace brv;

namespace {

constexpr int kFrame = 256, kHop = 128, kBands = 15, kSeg = 30;
constexpr double kEps = 2.220446049250313e-16;

// hann(kFrame + 2)[1:-1][m]
__device__ __forceinline__ float hann_inner(int m) {
  return 0.5f - 0.5f*cospif(2.f*(float)(m + 1)/(float)(kFrame + 1));
}
__host__ __device__ inline int frame_count(long long n) {     // len(range(0, n - kFrame, kHop))
  return n > kFrame ? (int)((n - kFrame + kHop - 1)/kHop) : 0;
}

// y[r][m] = sum_i hpad[(m + n_pre_remove)*down - i*up] x[r][i], m < ceil(n_in*up/down), else 0
__global__ void resample_poly_kernel(const float* x, const float* hpad, float* y,
                                     const int64_t* lengths, long long in_stride,
                                     long long out_stride, int up, int down, int hlen,
                                     int n_pre_remove) {
  const int r = blockIdx.y;
  const long long n_in = lengths ? lengths[r] : in_stride;
  const long long n_out = (n_in*up + down - 1)/down;
  const float* xr = x + (long long)r*in_stride;
  for (long long m = (long long)blockIdx.x*256 + threadIdx.x; m < out_stride;
       m += (long long)gridDim.x*256) {
    float acc = 0.f;
    if (m < n_out) {
      const long long c = (m + n_pre_remove)*down;          // filter index = c - i*up
      long long i_lo = (c - (hlen - 1) + up - 1)/up;        // c - i*up <= hlen - 1
      if (c - (hlen - 1) < 0) i_lo = 0;
      long long i_hi = c/up;                                // c - i*up >= 0
      if (i_hi > n_in - 1) i_hi = n_in - 1;
      for (long long i = i_lo; i <= i_hi; ++i)
        acc = __builtin_fmaf(hpad[c - i*up], xr[i], acc);
    }
    y[(long long)r*out_stride + m] = acc;
  }
}

// energy (dB) of every Hann frame of the clean signal; one wavefront per frame
__global__ __launch_bounds__(256) void stoi_energy_kernel(const float* clean, const int64_t* lengths,
                                                          float* energy, long long stride, int nf_max) {
  const int r = blockIdx.y, lane = threadIdx.x & 63;
  const int f = blockIdx.x*4 + (threadIdx.x >> 6);
  if (f >= nf_max) return;
  const int nf = frame_count(lengths[r]);
  float e = -INFINITY;
  if (f < nf) {
    const float* x = clean + (long long)r*stride + (long long)f*kHop;
    float s = 0.f;
    for (int m = lane; m < kFrame; m += 64) { const float v = hann_inner(m)*x[m]; s = __builtin_fmaf(v, v, s); }
    s = wave_sum(s);
    e = 20.f*log10f(sqrtf(s) + (float)kEps);
  }
  if (lane == 0) energy[(long long)r*nf_max + f] = e;
}

// per item: loudest frame, kept-frame list, geometry {n2, nf2, nseg, kept}
__global__ __launch_bounds__(256) void stoi_select_kernel(const float* energy, const int64_t* lengths,
                                                          int* kept, int* geom, int nf_max,
                                                          float dyn_range) {
  __shared__ float smax[256];
  __shared__ int scnt[257];
  const int r = blockIdx.x, tid = threadIdx.x;
  const int nf = frame_count(lengths[r]);
  const float* e = energy + (long long)r*nf_max;
  float mx = -INFINITY;
  for (int f = tid; f < nf; f += 256) mx = fmaxf(mx, e[f]);
  smax[tid] = mx;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (tid < s) smax[tid] = fmaxf(smax[tid], smax[tid + s]); __syncthreads(); }
  mx = smax[0];
  const int per = (nf + 255)/256;
  const int lo = min(nf, tid*per), hi = min(nf, lo + per);
  int c = 0;
  for (int f = lo; f < hi; ++f) c += (mx - dyn_range - e[f]) < 0.f;
  scnt[tid + 1] = c;
  if (tid == 0) scnt[0] = 0;
  __syncthreads();
  if (tid == 0) for (int i = 1; i <= 256; ++i) scnt[i] += scnt[i - 1];
  __syncthreads();
  int pos = scnt[tid];
  for (int f = lo; f < hi; ++f)
    if ((mx - dyn_range - e[f]) < 0.f) kept[(long long)r*nf_max + pos++] = f;
  if (tid == 0) {
    const int k = scnt[256];
    const int n2 = k > 0 ? kFrame + (k - 1)*kHop : 0;
    const int nf2 = frame_count(n2);
    geom[4*r] = n2; geom[4*r + 1] = nf2; geom[4*r + 2] = nf2 >= kSeg ? nf2 - kSeg + 1 : 0;
    geom[4*r + 3] = k;
  }
}

// overlap-add of the kept Hann frames: out[n] = sum over the <= 2 kept frames covering n
__global__ void stoi_ola_kernel(const float* src, const int* kept, const int* geom, float* dst,
                                long long stride, long long out_stride, int nf_max) {
  const int r = blockIdx.y;
  const int n2 = geom[4*r], k = geom[4*r + 3];
  const float* x = src + (long long)r*stride;
  const int* kp = kept + (long long)r*nf_max;
  for (long long n = (long long)blockIdx.x*256 + threadIdx.x; n < out_stride;
       n += (long long)gridDim.x*256) {
    float acc = 0.f;
    if (n < n2) {
      const int j = (int)(n / kHop);
      for (int jj = j - 1; jj <= j; ++jj) {
        if (jj < 0 || jj >= k) continue;
        const int m = (int)(n - (long long)jj*kHop);
        if (m >= kFrame) continue;
        acc += hann_inner(m)*x[(long long)kp[jj]*kHop + m];
      }
    }
    dst[(long long)r*out_stride + n] = acc;
  }
}

// tob[r][band][f] = sqrt(sum over the band's bins of re^2 + im^2); spec rows are frames with
// (re, im) interleaved for bins [bin0, bin0 + nbins)
__global__ void stoi_bands_kernel(const float* spec, const int* edges, float* tob, int rows,
                                  int nf_max, int ncols, int bin0) {
  const long long total = (long long)rows*kBands*nf_max;
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < total; i += (long long)gridDim.x*256) {
    const int f = (int)(i % nf_max); const int band = (int)((i / nf_max) % kBands);
    const long long r = i / ((long long)nf_max*kBands);
    const float* row = spec + (r*nf_max + f)*ncols;
    float s = 0.f;
    for (int b = edges[2*band]; b < edges[2*band + 1]; ++b) {
      const float re = row[2*(b - bin0)], im = row[2*(b - bin0) + 1];
      s = __builtin_fmaf(re, re, __builtin_fmaf(im, im, s));
    }
    tob[i] = sqrtf(s);
  }
}

// one workgroup (64 lanes) per (item, segment): partial[r][s] = segment's contribution
__global__ __launch_bounds__(64) void stoi_segment_kernel(const float* tx, const float* ty,
                                                          const int* geom, float* partial, int nf_max,
                                                          int nseg_max, int extended, float clip) {
  __shared__ float xs[kBands][kSeg + 1], ys[kBands][kSeg + 1];
  __shared__ float red[64];
  const int r = blockIdx.y, s = blockIdx.x, tid = threadIdx.x;
  if (s >= geom[4*r + 2]) return;
  const float* bx = tx + (long long)r*kBands*nf_max;
  const float* by = ty + (long long)r*kBands*nf_max;
  for (int i = tid; i < kBands*kSeg; i += 64) {
    const int j = i / kSeg, n = i % kSeg;
    xs[j][n] = bx[(long long)j*nf_max + s + n];
    ys[j][n] = by[(long long)j*nf_max + s + n];
  }
  __syncthreads();
  const float eps = (float)kEps;
  float contrib = 0.f;
  if (!extended) {
    // per band: scale y to the energy of x, clip, remove means, normalise, correlate
    if (tid < kBands) {
      const int j = tid;
      float nx = 0.f, ny = 0.f;
      for (int n = 0; n < kSeg; ++n) { nx = __builtin_fmaf(xs[j][n], xs[j][n], nx); ny = __builtin_fmaf(ys[j][n], ys[j][n], ny); }
      const float sc = sqrtf(nx)/(sqrtf(ny) + eps);
      float my = 0.f, mx = 0.f;
      for (int n = 0; n < kSeg; ++n) {
        const float yp = fminf(ys[j][n]*sc, xs[j][n]*(1.f + clip));
        ys[j][n] = yp; my += yp; mx += xs[j][n];
      }
      my /= kSeg; mx /= kSeg;
      float vy = 0.f, vx = 0.f, c = 0.f;
      for (int n = 0; n < kSeg; ++n) {
        const float a = ys[j][n] - my, b = xs[j][n] - mx;
        vy = __builtin_fmaf(a, a, vy); vx = __builtin_fmaf(b, b, vx); c = __builtin_fmaf(a, b, c);
      }
      contrib = c/((sqrtf(vy) + eps)*(sqrtf(vx) + eps));
    }
  } else {
    // rows (bands): remove the mean over frames, unit norm; columns (frames): the same over bands
    if (tid < kBands) {
      for (int which = 0; which < 2; ++which) {
        float (*a)[kSeg + 1] = which ? ys : xs;
        float m = 0.f;
        for (int n = 0; n < kSeg; ++n) m += a[tid][n];
        m /= kSeg;
        float v = 0.f;
        for (int n = 0; n < kSeg; ++n) { a[tid][n] -= m; v = __builtin_fmaf(a[tid][n], a[tid][n], v); }
        const float inv = 1.f/(sqrtf(v) + eps);
        for (int n = 0; n < kSeg; ++n) a[tid][n] *= inv;
      }
    }
    __syncthreads();
    if (tid < kSeg) {
      float vals[2][kBands];
      for (int which = 0; which < 2; ++which) {
        float (*a)[kSeg + 1] = which ? ys : xs;
        float m = 0.f;
        for (int j = 0; j < kBands; ++j) m += a[j][tid];
        m /= kBands;
        float v = 0.f;
        for (int j = 0; j < kBands; ++j) { const float d = a[j][tid] - m; vals[which][j] = d; v = __builtin_fmaf(d, d, v); }
        const float inv = 1.f/(sqrtf(v) + eps);
        for (int j = 0; j < kBands; ++j) vals[which][j] *= inv;
      }
      for (int j = 0; j < kBands; ++j) contrib = __builtin_fmaf(vals[0][j], vals[1][j], contrib);
      contrib /= kSeg;
    }
  }
  red[tid] = contrib;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int i = 0; i < 64; ++i) t += red[i];
    partial[(long long)r*nseg_max + s] = t;
  }
}

__global__ __launch_bounds__(256) void stoi_mean_kernel(const float* partial, const int* geom, float* out,
                                                        int nseg_max, int extended) {
  __shared__ double scr[8];
  const int r = blockIdx.x;
  const int nseg = geom[4*r + 2];
  double s = 0.0;
  for (int i = threadIdx.x; i < nseg; i += 256) s += partial[(long long)r*nseg_max + i];
  const double t = block_sum(s, scr);
  if (threadIdx.x == 0)
    out[r] = nseg > 0 ? (float)(t/((double)nseg*(extended ? 1.0 : (double)kBands))) : 1e-5f;
}

}  // namespace

extern "C" {

int brv_resample_poly(const float* x, const float* hpad, float* y, const int64_t* lengths,
                      int64_t rows, int64_t in_stride, int64_t out_stride, int64_t up, int64_t down,
                      int64_t hpad_len, int64_t n_pre_remove, brv_stream_t stream) {
  if (rows < 1 || in_stride < 1 || out_stride < 1 || up < 1 || down < 1 || hpad_len < 1) return -1;
  int gx = (int)((out_stride + 255)/256);
  if (gx > 4096) gx = 4096;
  hipLaunchKernelGGL(resample_poly_kernel, dim3(gx, (unsigned)rows), dim3(256), 0, (hipStream_t)stream,
                     x, hpad, y, lengths, (long long)in_stride, (long long)out_stride, (int)up,
                     (int)down, (int)hpad_len, (int)n_pre_remove);
  return (int)hipGetLastError();
}

int64_t brv_stoi_frames(int64_t length) { return frame_count(length); }

int brv_stoi_compact(const float* clean, const float* proc, const int64_t* lengths, int64_t rows,
                     int64_t stride, float* clean_out, float* proc_out, int64_t out_stride,
                     int32_t* geom, float* energy_scratch, int32_t* kept_scratch, int64_t nf_max,
                     float dyn_range, brv_stream_t stream) {
  if (rows < 1 || stride < 1 || nf_max < 1 || out_stride < 1) return -1;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(stoi_energy_kernel, dim3((unsigned)((nf_max + 3)/4), (unsigned)rows), dim3(256), 0,
                     st, clean, lengths, energy_scratch, (long long)stride, (int)nf_max);
  hipLaunchKernelGGL(stoi_select_kernel, dim3((unsigned)rows), dim3(256), 0, st, energy_scratch,
                     lengths, kept_scratch, geom, (int)nf_max, dyn_range);
  int gx = (int)((out_stride + 255)/256);
  if (gx > 2048) gx = 2048;
  hipLaunchKernelGGL(stoi_ola_kernel, dim3(gx, (unsigned)rows), dim3(256), 0, st, clean, kept_scratch,
                     geom, clean_out, (long long)stride, (long long)out_stride, (int)nf_max);
  hipLaunchKernelGGL(stoi_ola_kernel, dim3(gx, (unsigned)rows), dim3(256), 0, st, proc, kept_scratch,
                     geom, proc_out, (long long)stride, (long long)out_stride, (int)nf_max);
  return (int)hipGetLastError();
}

int brv_stoi_bands(const float* spec, const int32_t* edges, float* tob, int64_t rows, int64_t nf_max,
                   int64_t ncols, int64_t bin0, brv_stream_t stream) {
  if (rows < 1 || nf_max < 1 || ncols < 2) return -1;
  const long long total = (long long)rows*kBands*nf_max;
  int gx = (int)((total + 255)/256);
  if (gx > 8192) gx = 8192;
  hipLaunchKernelGGL(stoi_bands_kernel, dim3(gx), dim3(256), 0, (hipStream_t)stream, spec, edges, tob,
                     (int)rows, (int)nf_max, (int)ncols, (int)bin0);
  return (int)hipGetLastError();
}

int brv_stoi_correlate(const float* tob_clean, const float* tob_proc, const int32_t* geom,
                       float* partial_scratch, float* out, int64_t rows, int64_t nf_max,
                       int extended, float clip, brv_stream_t stream) {
  if (rows < 1 || nf_max < 1) return -1;
  hipStream_t st = (hipStream_t)stream;
  const int nseg_max = nf_max >= kSeg ? (int)nf_max - kSeg + 1 : 1;
  hipLaunchKernelGGL(stoi_segment_kernel, dim3((unsigned)nseg_max, (unsigned)rows), dim3(64), 0, st,
                     tob_clean, tob_proc, geom, partial_scratch, (int)nf_max, nseg_max, extended, clip);
  hipLaunchKernelGGL(stoi_mean_kernel, dim3((unsigned)rows), dim3(256), 0, st, partial_scratch, geom, out,
                     nseg_max, extended);
  return (int)hipGetLastError();
}

}  // extern "C"
