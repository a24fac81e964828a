// C ABI of the standalone criteria and optimizer kernels (include/brever_hip.h).
// Reference: brever/criterion.py:21-132,229-234; brever/models/base.py:296-301.
#include <hip/hip_runtime.h>
#include <math.h>
#include <string>

#include "../../include/brever_hip.h"
#include "misc_kernels.cuh"

using namespace brv;

namespace {
int ops_fail(int code, const char* what, hipError_t e) {
  (void)what; (void)e;
  return code;
}
#define OPS_OK(expr)                                                   \
  do {                                                                 \
    hipError_t e_ = (expr);                                            \
    if (e_ != hipSuccess) return ops_fail((int)e_, #expr, e_);         \
  } while (0)

// scratch layout: mom [rows][6] f64 | cross [B][S][S] f64 | coef [rows][8] f32
struct LossScratch {
  double* mom; double* cross; float* coef; size_t zero_bytes;
  LossScratch(void* p, int64_t B, int64_t S) {
    mom = (double*)p;
    cross = mom + B*S*6;
    coef = (float*)(cross + B*S*S);
    zero_bytes = (size_t)(B*S*6 + B*S*S)*sizeof(double);
  }
};

int moments(const float* x, const float* y, const int64_t* lengths, int64_t B, int64_t S,
            int64_t L, int64_t stride, const LossScratch& sc, hipStream_t st, int64_t ybs = -1,
            int64_t yss = -1) {
  OPS_OK(hipMemsetAsync(sc.mom, 0, sc.zero_bytes, st));
  MomentsParams p;
  p.x = x; p.y = y; p.stride = stride; p.L = (int)L;
  p.ybs = ybs < 0 ? S*stride : ybs; p.yss = yss < 0 ? stride : yss;
  p.lengths = (const long long*)lengths; p.S = (int)S; p.mom = sc.mom;
  int gx = (int)((L + 256*16 - 1)/(256*16));
  if (gx < 1) gx = 1;
  if (gx > 256) gx = 256;
  hipLaunchKernelGGL(masked_moments_kernel, dim3(gx, (unsigned)(B*S)), dim3(256), 0, st, p);
  OPS_OK(hipGetLastError());
  return 0;
}
}  // namespace

extern "C" {

int64_t brv_loss_scratch_bytes(int64_t B, int64_t S) {
  return (B*S*6 + B*S*S)*8 + B*S*8*4 + 64;
}

int brv_snr_forward(const float* x, const float* y, const int64_t* lengths, int64_t B,
                    int64_t S, int64_t L, int64_t stride, void* scratch, float* loss,
                    brv_stream_t stream) {
  return brv_snr_forward_strided(x, y, S*stride, stride, lengths, B, S, L, stride, scratch, loss, stream);
}

int brv_snr_forward_strided(const float* x, const float* y, int64_t y_batch_stride,
                            int64_t y_source_stride, const int64_t* lengths, int64_t B, int64_t S,
                            int64_t L, int64_t stride, void* scratch, float* loss,
                            brv_stream_t stream) {
  if (B < 1 || S < 1 || L < 1) return -1;
  hipStream_t st = (hipStream_t)stream;
  LossScratch sc(scratch, B, S);
  if (int r = moments(x, y, lengths, B, S, L, stride, sc, st, y_batch_stride, y_source_stride)) return r;
  hipLaunchKernelGGL(snr_finalize_kernel, dim3((unsigned)((B + 63)/64)), dim3(64), 0, st,
                     sc.mom, (int)B, (int)S, loss, sc.coef);
  OPS_OK(hipGetLastError());
  return 0;
}

int brv_snr_backward(const float* x, const float* y, const int64_t* lengths, int64_t B,
                     int64_t S, int64_t L, int64_t stride, const void* scratch,
                     const float* gscale, float* dx, brv_stream_t stream) {
  return brv_snr_backward_strided(x, y, S*stride, stride, lengths, B, S, L, stride, scratch, gscale, dx,
                                  stream);
}

int brv_snr_backward_strided(const float* x, const float* y, int64_t y_batch_stride,
                             int64_t y_source_stride, const int64_t* lengths, int64_t B, int64_t S,
                             int64_t L, int64_t stride, const void* scratch, const float* gscale,
                             float* dx, brv_stream_t stream) {
  if (B < 1 || S < 1 || L < 1) return -1;
  hipStream_t st = (hipStream_t)stream;
  LossScratch sc(const_cast<void*>(scratch), B, S);
  SnrBwdParams p;
  p.x = x; p.y = y; p.dx = dx; p.stride = stride; p.L = (int)L;
  p.lengths = (const long long*)lengths; p.S = (int)S; p.coef = sc.coef; p.gscale = gscale;
  p.ybs = y_batch_stride; p.yss = y_source_stride;
  int gx = (int)((L + 256*8 - 1)/(256*8));
  if (gx < 1) gx = 1;
  if (gx > 512) gx = 512;
  hipLaunchKernelGGL(snr_bwd_kernel, dim3(gx, (unsigned)(B*S)), dim3(256), 0, st, p);
  OPS_OK(hipGetLastError());
  return 0;
}

int brv_sisnr_forward(const float* x, const float* y, const int64_t* lengths, int64_t B,
                      int64_t S, int64_t L, int64_t stride, void* scratch, float* loss,
                      brv_stream_t stream) {
  if (B < 1 || S < 1 || S > 4 || L < 1) return -1;
  hipStream_t st = (hipStream_t)stream;
  LossScratch sc(scratch, B, S);
  if (int r = moments(x, y, lengths, B, S, L, stride, sc, st)) return r;
  CrossParams c;
  c.x = x; c.y = y; c.stride = stride; c.L = (int)L;
  c.lengths = (const long long*)lengths; c.S = (int)S; c.cross = sc.cross;
  int gx = (int)((L + 256*16 - 1)/(256*16));
  if (gx < 1) gx = 1;
  if (gx > 256) gx = 256;
  hipLaunchKernelGGL(masked_cross_kernel, dim3(gx, (unsigned)B, (unsigned)(S*S)), dim3(256),
                     0, st, c);
  OPS_OK(hipGetLastError());
  hipLaunchKernelGGL(sisnr_finalize_kernel, dim3((unsigned)((B + 63)/64)), dim3(64), 0, st,
                     sc.mom, sc.cross, (int)B, (int)S, (const long long*)lengths, loss, sc.coef);
  OPS_OK(hipGetLastError());
  return 0;
}

int brv_sisnr_backward(const float* x, const float* y, const int64_t* lengths, int64_t B,
                       int64_t S, int64_t L, int64_t stride, const void* scratch,
                       const float* gscale, float* dx, brv_stream_t stream) {
  if (B < 1 || S < 1 || S > 4 || L < 1) return -1;
  hipStream_t st = (hipStream_t)stream;
  LossScratch sc(const_cast<void*>(scratch), B, S);
  SisnrBwdParams p;
  p.x = x; p.y = y; p.dx = dx; p.stride = stride; p.L = (int)L;
  p.lengths = (const long long*)lengths; p.S = (int)S; p.coef = sc.coef; p.gscale = gscale;
  int gx = (int)((L + 256*8 - 1)/(256*8));
  if (gx < 1) gx = 1;
  if (gx > 512) gx = 512;
  hipLaunchKernelGGL(sisnr_bwd_kernel, dim3(gx, (unsigned)(B*S)), dim3(256), 0, st, p);
  OPS_OK(hipGetLastError());
  return 0;
}

int brv_mse_backward(const float* x, const float* y, const int64_t* lengths,
                     const float* weight, int64_t B, int64_t S, int64_t L, int64_t stride,
                     const float* gscale, float* dx, brv_stream_t stream) {
  if (B < 1 || S < 1 || L < 1) return -1;
  hipStream_t st = (hipStream_t)stream;
  MseBwdParams p;
  p.x = x; p.y = y; p.dx = dx; p.stride = stride; p.L = (int)L;
  p.lengths = (const long long*)lengths; p.S = (int)S; p.weight = weight; p.gscale = gscale;
  int gx = (int)((L + 256*8 - 1)/(256*8));
  if (gx < 1) gx = 1;
  if (gx > 512) gx = 512;
  hipLaunchKernelGGL(mse_bwd_kernel, dim3(gx, (unsigned)(B*S)), dim3(256), 0, st, p);
  OPS_OK(hipGetLastError());
  return 0;
}

int brv_mse_forward(const float* x, const float* y, const int64_t* lengths,
                    const float* weight, int64_t B, int64_t S, int64_t L, int64_t stride,
                    void* scratch, float* loss, brv_stream_t stream) {
  if (B < 1 || S < 1 || L < 1) return -1;
  hipStream_t st = (hipStream_t)stream;
  LossScratch sc(scratch, B, S);
  if (int r = moments(x, y, lengths, B, S, L, stride, sc, st)) return r;
  hipLaunchKernelGGL(mse_finalize_kernel, dim3((unsigned)((B + 63)/64)), dim3(64), 0, st,
                     sc.mom, (int)B, (int)S, (const long long*)lengths, weight, loss);
  OPS_OK(hipGetLastError());
  return 0;
}

namespace {
dim3 row_grid(int64_t rows, int64_t n, int per_thread) {
  long long gx = (n + 256LL*per_thread - 1)/(256LL*per_thread);
  if (gx < 1) gx = 1;
  if (gx > 512) gx = 512;
  return dim3((unsigned)gx, (unsigned)rows);
}
}  // namespace

int brv_apply_mask(const float* x, const int64_t* lengths, float* out, int64_t B, int64_t S,
                   int64_t L, brv_stream_t stream) {
  if (B < 1 || S < 1 || L < 1) return -1;
  hipLaunchKernelGGL(mask_rows_kernel, row_grid(B*S, L, 8), dim3(256), 0, (hipStream_t)stream,
                     x, (const long long*)lengths, out, (int)S, (long long)L);
  OPS_OK(hipGetLastError());
  return 0;
}

int brv_l1_forward(const float* x, const float* y, double* sums, int64_t rows, int64_t n,
                   brv_stream_t stream) {
  if (rows < 1 || n < 1) return -1;
  hipStream_t st = (hipStream_t)stream;
  OPS_OK(hipMemsetAsync(sums, 0, (size_t)rows*sizeof(double), st));
  hipLaunchKernelGGL(l1_fwd_kernel, row_grid(rows, n, 16), dim3(256), 0, st, x, y, sums, (long long)n);
  OPS_OK(hipGetLastError());
  return 0;
}

int brv_l1_backward(const float* x, const float* y, const float* grow, float* dx, int64_t rows,
                    int64_t n, int accumulate, brv_stream_t stream) {
  if (rows < 1 || n < 1) return -1;
  hipLaunchKernelGGL(l1_bwd_kernel, row_grid(rows, n, 8), dim3(256), 0, (hipStream_t)stream,
                     x, y, grow, dx, (long long)n, accumulate);
  OPS_OK(hipGetLastError());
  return 0;
}

int brv_mag_l1_forward(const float* xspec, const float* yspec, double* sums, int64_t rows,
                       int64_t n, brv_stream_t stream) {
  if (rows < 1 || n < 1) return -1;
  hipStream_t st = (hipStream_t)stream;
  OPS_OK(hipMemsetAsync(sums, 0, (size_t)rows*sizeof(double), st));
  hipLaunchKernelGGL(mag_l1_fwd_kernel, row_grid(rows, n, 8), dim3(256), 0, st,
                     (const float2*)xspec, (const float2*)yspec, sums, (long long)n);
  OPS_OK(hipGetLastError());
  return 0;
}

int brv_mag_l1_backward(const float* xspec, const float* yspec, const float* grow,
                        float* dxspec, int64_t rows, int64_t n, brv_stream_t stream) {
  if (rows < 1 || n < 1) return -1;
  hipLaunchKernelGGL(mag_l1_bwd_kernel, row_grid(rows, n, 4), dim3(256), 0, (hipStream_t)stream,
                     (const float2*)xspec, (const float2*)yspec, grow, (float2*)dxspec, (long long)n);
  OPS_OK(hipGetLastError());
  return 0;
}

int brv_clip_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                       int64_t n, float grad_scale, float max_norm, float lr, float beta1,
                       float beta2, float eps, int64_t step, void* scratch,
                       float* norm_out, brv_stream_t stream) {
  if (n < 1 || step < 1) return -1;
  hipStream_t st = (hipStream_t)stream;
  double* acc = (double*)scratch;
  OPS_OK(hipMemsetAsync(acc, 0, sizeof(double), st));
  int gx = (int)((n/4 + 255)/256);
  if (gx < 1) gx = 1;
  if (gx > 1024) gx = 1024;
  hipLaunchKernelGGL(sumsq_kernel, dim3(gx), dim3(256), 0, st, grads, (long long)n, acc);
  OPS_OK(hipGetLastError());
  AdamParams a;
  a.p = params; a.g = grads; a.m = exp_avg; a.v = exp_avg_sq; a.n = n;
  a.sumsq = acc; a.grad_scale = grad_scale; a.max_norm = max_norm;
  a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
  a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  a.bc2 = (float)(1.0 - pow((double)beta2, (double)step));
  a.norm_out = norm_out; a.zero_next = nullptr;
  int ga = (int)((n + 255)/256);
  if (ga > 2048) ga = 2048;
  hipLaunchKernelGGL(clip_adam_kernel, dim3(ga), dim3(256), 0, st, a);
  OPS_OK(hipGetLastError());
  return 0;
}

// The same step in two launches instead of up to five (memset, [add], norm, clip + Adam): `scratch`
// holds TWO fp64 accumulators, both zero before the first call; call number `slot` (0 / 1, alternating)
// accumulates into its own and the Adam kernel zeroes the other one for the next call. `grads2`
// (nullable): a second gradient buffer that is added into `grads` first and left zeroed.
int brv_clip_adam_step2(float* params, float* grads, float* grads2, float* exp_avg,
                        float* exp_avg_sq, int64_t n, float grad_scale, float max_norm, float lr,
                        float beta1, float beta2, float eps, int64_t step, void* scratch,
                        int32_t slot, float* norm_out, brv_stream_t stream) {
  if (n < 1 || step < 1 || (slot != 0 && slot != 1)) return -1;
  hipStream_t st = (hipStream_t)stream;
  double* acc = (double*)scratch + slot;
  int gx = (int)((n/4 + 255)/256);
  if (gx < 1) gx = 1;
  if (gx > 1024) gx = 1024;
  if (grads2) hipLaunchKernelGGL(sum_sumsq_kernel, dim3(gx), dim3(256), 0, st, grads, grads2, (long long)n, acc);
  else hipLaunchKernelGGL(sumsq_kernel, dim3(gx), dim3(256), 0, st, grads, (long long)n, acc);
  OPS_OK(hipGetLastError());
  AdamParams a;
  a.p = params; a.g = grads; a.m = exp_avg; a.v = exp_avg_sq; a.n = n;
  a.sumsq = acc; a.grad_scale = grad_scale; a.max_norm = max_norm;
  a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
  a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  a.bc2 = (float)(1.0 - pow((double)beta2, (double)step));
  a.norm_out = norm_out; a.zero_next = (double*)scratch + (1 - slot);
  int ga = (int)((n + 255)/256);
  if (ga > 2048) ga = 2048;
  hipLaunchKernelGGL(clip_adam_kernel, dim3(ga), dim3(256), 0, st, a);
  OPS_OK(hipGetLastError());
  return 0;
}

int brv_memset_zero(void* ptr, int64_t bytes, brv_stream_t stream) {
  if (bytes < 0) return -1;
  if (bytes == 0) return 0;
  OPS_OK(hipMemsetAsync(ptr, 0, (size_t)bytes, (hipStream_t)stream));
  return 0;
}

int brv_mean_f32(const float* x, int64_t n, float* out, brv_stream_t stream) {
  if (n < 1 || n > (1 << 20)) return -1;
  hipLaunchKernelGGL(mean_small_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, x, (int)n, out);
  OPS_OK(hipGetLastError());
  return 0;
}


// ema += (1 - beta)*(param - ema), each operation rounded on its own (the reference's
// `ema_param += (1 - beta) * (param - ema_param)`, brever/modules/ema.py:36-39: no FMA
// contraction, so the averages are bit-identical to the reference's)
__global__ __launch_bounds__(256) void ema_update_kernel(float* ema, const float* param,
                                                         float one_minus_beta, long long n) {
#pragma clang fp contract(off)
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < n; i += (long long)gridDim.x*256) {
    const float d = param[i] - ema[i];
    const float s = one_minus_beta*d;
    ema[i] = ema[i] + s;
  }
}
int brv_ema_update(float* ema, const float* param, float one_minus_beta, int64_t n,
                   brv_stream_t stream) {
  if (n < 1) return -1;
  long long g = (n + 255)/256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(ema_update_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, ema,
                     param, one_minus_beta, (long long)n);
  OPS_OK(hipGetLastError());
  return 0;
}

// Scale-invariant pre-scaling of MultiResYuLoss (criterion.py:207-212): per row (item, source)
// alpha = <x, y>/(<x, x> + eps) over the samples below the item length; out = alpha*x there,
// zero beyond. One workgroup per row: reduce, then scale. stats[row] = (alpha, <x, x> + eps).
__global__ __launch_bounds__(256) void si_scale_fwd_kernel(const float* x, const float* y,
                                                           const long long* lengths, float* out,
                                                           double* stats, int S, long long L,
                                                           float eps) {
  __shared__ double scr[8];
  __shared__ double bc[2];
  const long long row = blockIdx.x;
  const long long n = lengths[row / S] < L ? lengths[row / S] : L;
  const float* xr = x + row*L; const float* yr = y + row*L; float* orow = out + row*L;
  double sxy = 0.0, sxx = 0.0;
  for (long long i = threadIdx.x; i < n; i += 256) { sxy += (double)xr[i]*yr[i]; sxx += (double)xr[i]*xr[i]; }
  sxy = block_sum(sxy, scr); __syncthreads();
  sxx = block_sum(sxx, scr);
  if (threadIdx.x == 0) {
    // fp32 arithmetic of the reference: sums, then the quotient, all in float
    const float d = (float)sxx + eps;
    bc[0] = (double)((float)sxy/d); bc[1] = (double)d;
    stats[2*row] = bc[0]; stats[2*row + 1] = bc[1];
  }
  __syncthreads();
  const float a = (float)bc[0];
  for (long long i = threadIdx.x; i < L; i += 256) orow[i] = i < n ? a*xr[i] : 0.f;
}
// dx = alpha*g + <g, x>*(y - 2 alpha x)/D below the length, 0 beyond
__global__ __launch_bounds__(256) void si_scale_bwd_kernel(const float* g, const float* x,
                                                           const float* y, const long long* lengths,
                                                           const double* stats, float* dx, int S,
                                                           long long L) {
  __shared__ double scr[8];
  __shared__ double bc;
  const long long row = blockIdx.x;
  const long long n = lengths[row / S] < L ? lengths[row / S] : L;
  const float* gr = g + row*L; const float* xr = x + row*L; const float* yr = y + row*L;
  double sgx = 0.0;
  for (long long i = threadIdx.x; i < n; i += 256) sgx += (double)gr[i]*xr[i];
  sgx = block_sum(sgx, scr);
  if (threadIdx.x == 0) bc = sgx;
  __syncthreads();
  const float a = (float)stats[2*row], c = (float)(bc/stats[2*row + 1]);
  float* drow = dx + row*L;
  for (long long i = threadIdx.x; i < L; i += 256)
    drow[i] = i < n ? a*gr[i] + c*(yr[i] - 2.f*a*xr[i]) : 0.f;
}
int brv_si_scale_forward(const float* x, const float* y, const int64_t* lengths, float* out,
                         double* stats, int64_t B, int64_t S, int64_t L, float eps,
                         brv_stream_t stream) {
  if (B < 1 || S < 1 || L < 1) return -1;
  hipLaunchKernelGGL(si_scale_fwd_kernel, dim3((unsigned)(B*S)), dim3(256), 0, (hipStream_t)stream,
                     x, y, (const long long*)lengths, out, stats, (int)S, (long long)L, eps);
  OPS_OK(hipGetLastError());
  return 0;
}
int brv_si_scale_backward(const float* g, const float* x, const float* y, const int64_t* lengths,
                          const double* stats, float* dx, int64_t B, int64_t S, int64_t L,
                          brv_stream_t stream) {
  if (B < 1 || S < 1 || L < 1) return -1;
  hipLaunchKernelGGL(si_scale_bwd_kernel, dim3((unsigned)(B*S)), dim3(256), 0, (hipStream_t)stream,
                     g, x, y, (const long long*)lengths, stats, dx, (int)S, (long long)L);
  OPS_OK(hipGetLastError());
  return 0;
}
}  // extern "C"
