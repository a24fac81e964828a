// Losses, optimizer and weight-preparation kernels (all HBM-bound).
//
//  masked_moments / snr_* / sisnr_* / mse_* : length-masked criteria of
//      brever/criterion.py:21-132 (apply_mask :229-234 folded in; the lengths stay
//      on the device, no host synchronisation)
//  sumsq / clip_adam : torch.nn.utils.clip_grad_norm_ + torch.optim.Adam.step on one
//      flat fp32 buffer (brever/models/base.py:296-301)
//  prep_weights : fp32 master weights -> padded bf16 GEMM operands (both layouts)
#pragma once
#include "common.cuh"

namespace brv {

// ---------------------------------------------------------------------------
// Raw masked moments of rows x[r][:len], y[r][:len] for r = item*S + s.
// mom[r] = {sum x, sum y, sum x^2, sum y^2, sum x*y, sum (x-y)^2}
struct MomentsParams {
  const float* x; const float* y; long long stride; int L;
  long long ybs, yss;                       // y row (item b, source s) starts at y + b*ybs + s*yss
  const long long* lengths; int S;          // rows per batch item
  double* mom;                              // [rows][6]
};

__global__ __launch_bounds__(256) void masked_moments_kernel(const MomentsParams p) {
  __shared__ double dscr[8];
  const int r = blockIdx.y;
  long long len = p.lengths[r / p.S];
  if (len > p.L) len = p.L;
  const float* x = p.x + (long long)r*p.stride;
  const float* y = p.y + (long long)(r / p.S)*p.ybs + (long long)(r % p.S)*p.yss;
  double a[6] = {0, 0, 0, 0, 0, 0};
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < len;
       i += (long long)gridDim.x*256) {
    const double xv = x[i], yv = y[i], d = xv - yv;
    a[0] += xv; a[1] += yv; a[2] += xv*xv; a[3] += yv*yv; a[4] += xv*yv; a[5] += d*d;
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const double s = block_sum(a[k], dscr);
    if (threadIdx.x == 0 && s != 0.0) atomic_add_f64(p.mom + 6*r + k, s);
  }
}

// Cross moments for PIT SI-SNR: cross[item][i][j] = sum x_j * y_i (est j, ref i)
struct CrossParams {
  const float* x; const float* y; long long stride; int L;
  const long long* lengths; int S; double* cross;
};
__global__ __launch_bounds__(256) void masked_cross_kernel(const CrossParams p) {
  __shared__ double dscr[8];
  const int item = blockIdx.y;
  const int i = blockIdx.z / p.S, j = blockIdx.z % p.S;
  long long len = p.lengths[item];
  if (len > p.L) len = p.L;
  const float* y = p.y + ((long long)item*p.S + i)*p.stride;
  const float* x = p.x + ((long long)item*p.S + j)*p.stride;
  double a = 0;
  for (long long n = (long long)blockIdx.x*256 + threadIdx.x; n < len;
       n += (long long)gridDim.x*256) a += (double)x[n]*y[n];
  const double s = block_sum(a, dscr);
  if (threadIdx.x == 0 && s != 0.0)
    atomic_add_f64(p.cross + ((long long)item*p.S + i)*p.S + j, s);
}

constexpr double kEps32 = 1.1920928955078125e-07;   // torch.finfo(float32).eps

// loss[b] = -(1/S) sum_s 10 log10( sum y^2 / (sum (y-x)^2 + eps) + eps )
// coef[r] = d loss[b] / d D_r * 2   (so that d loss / d x = coef * (x - y))
__global__ void snr_finalize_kernel(const double* mom, int B, int S, float* loss,
                                    float* coef) {
  const int b = blockIdx.x*blockDim.x + threadIdx.x;
  if (b >= B) return;
  double acc = 0;
  for (int s = 0; s < S; ++s) {
    const double* m = mom + 6*((long long)b*S + s);
    // the reference evaluates this in fp32: keep the same rounding of the ratio
    const float num = (float)m[3];
    const float den = (float)m[5] + (float)kEps32;
    const float ratio = num/den;
    const float val = 10.f*log10f(ratio + (float)kEps32);
    acc += val;
    if (coef) {
      const double R = (double)ratio;
      coef[(long long)b*S + s] =
          (float)((10.0/log(10.0))/S*R/((R + kEps32)*((double)den))*2.0);
    }
  }
  loss[b] = (float)(-acc/S);
}

struct SnrBwdParams {
  const float* x; const float* y; float* dx; long long stride; int L;
  long long ybs, yss;
  const long long* lengths; int S; const float* coef; const float* gscale;  // [B]
};
__global__ __launch_bounds__(256) void snr_bwd_kernel(const SnrBwdParams p) {
  const int r = blockIdx.y;
  const int b = r / p.S;
  long long len = p.lengths[b];
  if (len > p.L) len = p.L;
  const float c = p.coef[r]*p.gscale[b];
  const float* x = p.x + (long long)r*p.stride;
  const float* y = p.y + (long long)b*p.ybs + (long long)(r % p.S)*p.yss;
  float* dx = p.dx + (long long)r*p.stride;
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < p.L;
       i += (long long)gridDim.x*256)
    dx[i] = i < len ? c*(x[i] - y[i]) : 0.f;
}

// mse: loss[b] = mean_s ( sum (x-y)^2 / len ) * weight[b]
__global__ void mse_finalize_kernel(const double* mom, int B, int S,
                                    const long long* lengths, const float* weight,
                                    float* loss) {
  const int b = blockIdx.x*blockDim.x + threadIdx.x;
  if (b >= B) return;
  double acc = 0;
  for (int s = 0; s < S; ++s) {
    float v = (float)mom[6*((long long)b*S + s) + 5];
    v = v/(float)lengths[b];
    if (weight) v *= weight[b];
    acc += v;
  }
  loss[b] = (float)(acc/S);
}

// PIT SI-SNR from raw moments (closed form in fp64), S <= 4.
// coef (nullable): for estimate j of item b, 8 floats
//   {c1, c2, mean_x, mean_y(ref), ref index, 0, 0, 0}
// such that  d loss[b] / d x[b][j][n] = c1*(y[b][ref][n] - mean_y) + c2*(x[b][j][n] - mean_x)
// for n < length (the mean removal of criterion.py:48-49 is transparent to the
// gradient because both centred signals have zero mean over the valid samples).
__global__ void sisnr_finalize_kernel(const double* mom, const double* cross, int B,
                                      int S, const long long* lengths, float* loss,
                                      float* coef) {
  const int b = blockIdx.x*blockDim.x + threadIdx.x;
  if (b >= B) return;
  const double n = (double)lengths[b];
  double val[4][4], c1[4][4], c2[4][4];
  for (int i = 0; i < S; ++i) {            // reference source i
    const double* mi = mom + 6*((long long)b*S + i);
    const double my = mi[1]/n;
    const double syy = mi[3] - n*my*my;
    for (int j = 0; j < S; ++j) {          // estimate j
      const double* mj = mom + 6*((long long)b*S + j);
      const double mx = mj[0]/n;
      const double sxx = mj[2] - n*mx*mx;
      const double sxy = cross[((long long)b*S + i)*S + j] - n*mx*my;
      const double tgt = sxy*sxy/syy;                 // ||s_target||^2
      double noise = sxx - tgt;                       // ||e_noise||^2
      if (noise < 0) noise = 0;
      const double den = noise + kEps32;
      const float ratio = (float)tgt/((float)noise + (float)kEps32);
      val[i][j] = 10.f*log10f(ratio + (float)kEps32);
      const double R = tgt/den;
      const double dv = (10.0/log(10.0))/(R + kEps32);        // d val / d R
      c1[i][j] = dv*(2.0*sxy/syy)*(1.0/den + tgt/(den*den));  // via <x, s>
      c2[i][j] = dv*(-tgt/(den*den))*2.0;                     // via ||x||^2
    }
  }
  // enumerate permutations (estimate perm[i] is assigned to reference i)
  int perm[4] = {0, 1, 2, 3}, best_perm[4] = {0, 1, 2, 3};
  int c[4] = {0, 0, 0, 0};
  auto score = [&]() { double s = 0; for (int i = 0; i < S; ++i) s += val[i][perm[i]]; return s; };
  double best = score();
  int i = 0;
  while (i < S) {
    if (c[i] < i) {
      const int k = (i & 1) ? c[i] : 0;
      const int tmp = perm[k]; perm[k] = perm[i]; perm[i] = tmp;
      const double s = score();
      if (s > best) { best = s; for (int q = 0; q < S; ++q) best_perm[q] = perm[q]; }
      ++c[i]; i = 0;
    } else { c[i] = 0; ++i; }
  }
  loss[b] = (float)(-best/S);
  if (coef) {
    for (int r = 0; r < S; ++r) {
      const int j = best_perm[r];
      float* o = coef + 8*((long long)b*S + j);
      o[0] = (float)(-c1[r][j]/S); o[1] = (float)(-c2[r][j]/S);
      o[2] = (float)(mom[6*((long long)b*S + j)]/n);
      o[3] = (float)(mom[6*((long long)b*S + r) + 1]/n);
      o[4] = (float)r; o[5] = 0.f; o[6] = 0.f; o[7] = 0.f;
    }
  }
}

struct SisnrBwdParams {
  const float* x; const float* y; float* dx; long long stride; int L;
  const long long* lengths; int S; const float* coef; const float* gscale;
};
__global__ __launch_bounds__(256) void sisnr_bwd_kernel(const SisnrBwdParams p) {
  const int r = blockIdx.y;                       // row of the estimate: item*S + j
  const int b = r / p.S;
  long long len = p.lengths[b];
  if (len > p.L) len = p.L;
  const float* cf = p.coef + 8*(long long)r;
  const float g = p.gscale[b];
  const float c1 = cf[0]*g, c2 = cf[1]*g, mx = cf[2], my = cf[3];
  const int ref = (int)cf[4];
  const float* x = p.x + (long long)r*p.stride;
  const float* y = p.y + ((long long)b*p.S + ref)*p.stride;
  float* dx = p.dx + (long long)r*p.stride;
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < p.L;
       i += (long long)gridDim.x*256)
    dx[i] = i < len ? c1*(y[i] - my) + c2*(x[i] - mx) : 0.f;
}

// d mse[b] / d x = weight[b] * 2/(S*len) * (x - y) on the valid samples
struct MseBwdParams {
  const float* x; const float* y; float* dx; long long stride; int L;
  const long long* lengths; int S; const float* weight; const float* gscale;
};
__global__ __launch_bounds__(256) void mse_bwd_kernel(const MseBwdParams p) {
  const int r = blockIdx.y;
  const int b = r / p.S;
  long long len = p.lengths[b];
  if (len > p.L) len = p.L;
  const float c = p.gscale[b]*(p.weight ? p.weight[b] : 1.f)*2.f/((float)p.S*(float)p.lengths[b]);
  const float* x = p.x + (long long)r*p.stride;
  const float* y = p.y + (long long)r*p.stride;
  float* dx = p.dx + (long long)r*p.stride;
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < p.L;
       i += (long long)gridDim.x*256)
    dx[i] = i < len ? c*(x[i] - y[i]) : 0.f;
}

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sumsq_kernel(const float* g, long long n,
                                                    double* acc) {
  __shared__ double dscr[8];
  double a = 0;
  const long long n4 = n/4;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < n4;
       i += (long long)gridDim.x*256) {
    const float4 v = g4[i];
    a += (double)v.x*v.x + (double)v.y*v.y + (double)v.z*v.z + (double)v.w*v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < n - n4*4) {
    const float v = g[n4*4 + threadIdx.x];
    a += (double)v*v;
  }
  const double s = block_sum(a, dscr);
  if (threadIdx.x == 0) atomic_add_f64(acc, s);
}

// g += g2, g2 = 0, and the squared norm of the sum: the step end of two kernel chains that wrote their
// weight gradients to two buffers (one pass instead of add + norm passes; the second buffer is left
// zeroed for the next step, so it is never memset)
__global__ __launch_bounds__(256) void sum_sumsq_kernel(float* g, float* g2, long long n,
                                                        double* acc) {
  __shared__ double dscr[8];
  double a = 0;
  const long long n4 = n/4;
  float4* g4 = reinterpret_cast<float4*>(g);
  float4* h4 = reinterpret_cast<float4*>(g2);
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < n4;
       i += (long long)gridDim.x*256) {
    float4 v = g4[i];
    const float4 w = h4[i];
    v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    g4[i] = v;
    h4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    a += (double)v.x*v.x + (double)v.y*v.y + (double)v.z*v.z + (double)v.w*v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < n - n4*4) {
    const long long i = n4*4 + threadIdx.x;
    const float v = g[i] + g2[i];
    g[i] = v; g2[i] = 0.f;
    a += (double)v*v;
  }
  const double s = block_sum(a, dscr);
  if (threadIdx.x == 0) atomic_add_f64(acc, s);
}

// out[0] = mean of x[0 .. n) (n small: the per-item losses of a batch)
__global__ __launch_bounds__(64) void mean_small_kernel(const float* x, int n, float* out) {
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) a += x[i];
  a = wave_sum(a);
  if (threadIdx.x == 0) out[0] = a/(float)n;
}

struct AdamParams {
  float* p; float* g; float* m; float* v; long long n;
  const double* sumsq;     // squared global gradient norm (after grad_scale)
  float grad_scale;        // multiplies g before everything (1/world for DDP mean)
  float max_norm;          // <= 0: no clipping
  float lr, beta1, beta2, eps;
  float bc1, bc2;          // 1 - beta^step
  float* norm_out;         // nullable: total norm
  double* zero_next;       // nullable: accumulator of the NEXT step's squared norm, zeroed here
};
__global__ __launch_bounds__(256) void clip_adam_kernel(const AdamParams a) {
  const double total = sqrt(*a.sumsq)*(double)a.grad_scale;
  float clip = 1.f;
  if (a.max_norm > 0.f) {
    const float c = a.max_norm/((float)total + 1e-6f);   // clip_grad_norm_
    clip = c < 1.f ? c : 1.f;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (a.norm_out) *a.norm_out = (float)total;
    if (a.zero_next) *a.zero_next = 0.0;       // (nobody reads that slot in this launch)
  }
  const float gs = a.grad_scale*clip;
  const float step_size = a.lr/a.bc1;
  const float rbc2 = 1.f/sqrtf(a.bc2);
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < a.n;
       i += (long long)gridDim.x*256) {
    const float g = a.g[i]*gs;
    const float m = a.beta1*a.m[i] + (1.f - a.beta1)*g;
    const float v = a.beta2*a.v[i] + (1.f - a.beta2)*g*g;
    const float denom = sqrtf(v)*rbc2 + a.eps;
    a.p[i] -= step_size*(m/denom);
    a.m[i] = m; a.v[i] = v; a.g[i] = g;
  }
}

// ---- pieces of MultiResYuLoss (brever/criterion.py:135-226) ------------------
// out[r][i] = i < lengths[r / S] ? x[r][i] : 0           (apply_mask, criterion.py:229-234)
__global__ __launch_bounds__(256) void mask_rows_kernel(const float* x, const long long* lengths,
                                                        float* out, int S, long long L) {
  const int r = blockIdx.y;
  long long len = lengths[r / S];
  if (len > L) len = L;
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < L; i += (long long)gridDim.x*256)
    out[(long long)r*L + i] = i < len ? x[(long long)r*L + i] : 0.f;
}
// sums[r] += sum_i |x[r][i] - y[r][i]|  (fp64 accumulation, one atomic per workgroup)
__global__ __launch_bounds__(256) void l1_fwd_kernel(const float* x, const float* y, double* sums,
                                                     long long n) {
  __shared__ double scr[8];
  const int r = blockIdx.y;
  double s = 0.0;
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < n; i += (long long)gridDim.x*256)
    s += (double)fabsf(x[(long long)r*n + i] - y[(long long)r*n + i]);
  s = block_sum(s, scr);
  if (threadIdx.x == 0 && s != 0.0) atomic_add_f64(sums + r, s);
}
// dx[r][i] (+)= g[r]*sign(x - y)   (sign(0) = 0, as torch's abs backward)
__global__ __launch_bounds__(256) void l1_bwd_kernel(const float* x, const float* y, const float* g,
                                                     float* dx, long long n, int accumulate) {
  const int r = blockIdx.y;
  const float gr = g[r];
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < n; i += (long long)gridDim.x*256) {
    const float d = x[(long long)r*n + i] - y[(long long)r*n + i];
    const float v = d > 0.f ? gr : (d < 0.f ? -gr : 0.f);
    if (accumulate) dx[(long long)r*n + i] += v; else dx[(long long)r*n + i] = v;
  }
}
// sums[r] += sum_i | |X[r][i]| - |Y[r][i]| |   over n complex values per row
__global__ __launch_bounds__(256) void mag_l1_fwd_kernel(const float2* X, const float2* Y,
                                                         double* sums, long long n) {
  __shared__ double scr[8];
  const int r = blockIdx.y;
  double s = 0.0;
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < n; i += (long long)gridDim.x*256) {
    const float2 a = X[(long long)r*n + i], b = Y[(long long)r*n + i];
    s += (double)fabsf(sqrtf(a.x*a.x + a.y*a.y) - sqrtf(b.x*b.x + b.y*b.y));
  }
  s = block_sum(s, scr);
  if (threadIdx.x == 0 && s != 0.0) atomic_add_f64(sums + r, s);
}
// dX[r][i] = g[r]*sign(|X| - |Y|)*X/|X|   (0 where |X| = 0: the subgradient torch uses)
__global__ __launch_bounds__(256) void mag_l1_bwd_kernel(const float2* X, const float2* Y,
                                                         const float* g, float2* dX, long long n) {
  const int r = blockIdx.y;
  const float gr = g[r];
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < n; i += (long long)gridDim.x*256) {
    const float2 a = X[(long long)r*n + i], b = Y[(long long)r*n + i];
    const float ma = sqrtf(a.x*a.x + a.y*a.y), mb = sqrtf(b.x*b.x + b.y*b.y);
    const float sg = ma > mb ? gr : (ma < mb ? -gr : 0.f);
    const float inv = ma > 0.f ? sg/ma : 0.f;
    dX[(long long)r*n + i] = make_float2(a.x*inv, a.y*inv);
  }
}

}  // namespace brv
