// Row-wise operators of TF-GridNet, fp32 (brever/models/tfgridnet/tfgridnet.py):
//   * brv_rownorm_*: layer normalisation of contiguous rows with an optional PReLU in front and a
//     per-(group, column) gain / bias. One operator covers nn.LayerNorm(emb_dim) on the
//     channels-last grid (tfgridnet.py:199,213: rows = (item, frame, band), one group),
//     LayerNormalization4DCF (tfgridnet.py:356-380: rows = (item, frame) holding channel x band,
//     one group) and AllHeadPReLULayerNormalization4DCF (tfgridnet.py:383-415: rows = (item,
//     head, frame) holding E x band, one group and one PReLU slope per head). The group of row r
//     is (r / inner) % groups.
//   * brv_row_std / brv_row_scale: the RMS normalisation of the mixture by its unbiased standard
//     deviation and its reversal on the output (tfgridnet.py:108-109,128).
// All of them are HBM-bound streaming kernels: a wavefront owns a row (rows are 32 ... 4 128
// floats: 128 B ... 16 KB contiguous), reads it twice (second pass from L2) and writes it once.
// The parameter gradients are column sums over the rows of a group: fixed slices of rows are
// reduced by one workgroup each into a partial table that a second kernel folds in a fixed
// order (deterministic, no atomics).
#include <hip/hip_runtime.h>
#include <math.h>

#include "../../include/brever_hip.h"
#include "common.cuh"

using namespace brv;

namespace {

#define TG_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return (int)e_; } while (0)

// row slices of the parameter-gradient reduction: enough workgroups to fill the chip whatever the
// row length (32 ... 4 128), at most 1 024 partial tables
__host__ __device__ inline int pgrad_slices(long long n, long long groups) {
  const long long wgs = ((n + 63)/64)*groups;
  long long s = (2048 + wgs - 1)/wgs;
  return (int)(s < 1 ? 1 : (s > 1024 ? 1024 : s));
}

__device__ __forceinline__ float prelu(float v, float a) { return v > 0.f ? v : a*v; }

// y = (prelu(x) - mean) * rstd * gain[g][j] + bias[g][j]; stats[r] = (mean, rstd)
__global__ __launch_bounds__(256) void rownorm_fwd_kernel(const float* __restrict__ x,
                                                          const float* __restrict__ slope,
                                                          const float* __restrict__ gain,
                                                          const float* __restrict__ bias,
                                                          float* __restrict__ y, float2* stats,
                                                          long long rows, int n, int inner, int G,
                                                          float eps) {
  const int lane = threadIdx.x & 63;
  const long long r = (long long)blockIdx.x*4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int g = (int)((r / inner) % G);
  const float a = slope ? slope[g] : 1.f;
  const float* xr = x + r*n;
  float s = 0.f;
  for (int j = lane; j < n; j += 64) s += prelu(xr[j], a);
  const float mean = wave_sum(s)/n;
  float q = 0.f;
  for (int j = lane; j < n; j += 64) { const float d = prelu(xr[j], a) - mean; q += d*d; }
  const float rstd = 1.f/sqrtf(wave_sum(q)/n + eps);
  const float* gg = gain + (long long)g*n;
  const float* bb = bias + (long long)g*n;
  float* yr = y + r*n;
  for (int j = lane; j < n; j += 64) yr[j] = (prelu(xr[j], a) - mean)*rstd*gg[j] + bb[j];
  if (lane == 0) stats[r] = make_float2(mean, rstd);
}

// dx = prelu'(x) * rstd * (dyg - mean(dyg) - xhat * mean(dyg * xhat)), dyg = dy * gain;
// dslope_rows[r] = sum over the row of (gradient wrt prelu(x)) * x where x <= 0
__global__ __launch_bounds__(256) void rownorm_bwd_kernel(const float* __restrict__ x,
                                                          const float* __restrict__ dy,
                                                          const float* __restrict__ slope,
                                                          const float* __restrict__ gain,
                                                          const float2* __restrict__ stats,
                                                          float* __restrict__ dx,
                                                          float* __restrict__ dslope_rows,
                                                          long long rows, int n, int inner, int G) {
  const int lane = threadIdx.x & 63;
  const long long r = (long long)blockIdx.x*4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int g = (int)((r / inner) % G);
  const float a = slope ? slope[g] : 1.f;
  const float2 st = stats[r];
  const float* xr = x + r*n;
  const float* dr = dy + r*n;
  const float* gg = gain + (long long)g*n;
  float s1 = 0.f, s2 = 0.f;
  for (int j = lane; j < n; j += 64) {
    const float d = dr[j]*gg[j];
    s1 += d; s2 += d*((prelu(xr[j], a) - st.x)*st.y);
  }
  const float m1 = wave_sum(s1)/n, m2 = wave_sum(s2)/n;
  float* out = dx + r*n;
  float da = 0.f;
  for (int j = lane; j < n; j += 64) {
    const float v = xr[j];
    const float xh = (prelu(v, a) - st.x)*st.y;
    const float dp = st.y*(dr[j]*gg[j] - m1 - xh*m2);
    out[j] = v > 0.f ? dp : a*dp;
    if (!(v > 0.f)) da += dp*v;
  }
  if (dslope_rows) {
    da = wave_sum(da);
    if (lane == 0) dslope_rows[r] = da;
  }
}

// Short rows (n <= 64, the nn.LayerNorm(emb_dim) rows of the grid: 32 floats): a wavefront
// carries 64/npad rows (npad = n rounded up to a power of two), one element per lane, read once;
// the row reductions are xor-shuffles inside the lane group.
__device__ __forceinline__ float group_sum(float v, int npad) {
  for (int off = npad >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
// (kSmallIt row sets per wavefront, their loads requested together at clamped indices: with one element per
// thread and launch a 258 516 x 32 layer norm was 32 K workgroups of a single load each, 46 us for 66 MB)
constexpr int kSmallIt = 4;
__global__ __launch_bounds__(256) void rownorm_small_fwd_kernel(const float* __restrict__ x,
                                                                const float* __restrict__ slope,
                                                                const float* __restrict__ gain,
                                                                const float* __restrict__ bias,
                                                                float* __restrict__ y, float2* stats,
                                                                long long rows, int n, int npad,
                                                                int inner, int G, float eps) {
  const int lane = threadIdx.x & 63, rpw = 64/npad;
  const int sub = lane / npad, j = lane % npad, jc = j < n ? j : n - 1;
  const long long base = ((long long)blockIdx.x*4 + (threadIdx.x >> 6))*rpw*kSmallIt + sub;
  float xv[kSmallIt], gv[kSmallIt], bv[kSmallIt], av[kSmallIt];
#pragma unroll
  for (int it = 0; it < kSmallIt; ++it) {
    const long long r = base + (long long)it*rpw, rr = r < rows ? r : rows - 1;
    const int g = G == 1 ? 0 : (int)((rr / inner) % G);      // (a 64-bit division per element otherwise)
    xv[it] = x[rr*n + jc];
    gv[it] = gain[(long long)g*n + jc];
    bv[it] = bias[(long long)g*n + jc];
    av[it] = slope ? slope[g] : 1.f;
  }
#pragma unroll
  for (int it = 0; it < kSmallIt; ++it) {
    const long long r = base + (long long)it*rpw;
    const bool live = r < rows && j < n;
    const float v = live ? prelu(xv[it], av[it]) : 0.f;
    const float mean = group_sum(v, npad)/n;
    const float d = live ? v - mean : 0.f;
    const float rstd = 1.f/sqrtf(group_sum(d*d, npad)/n + eps);
    if (live) y[r*n + j] = d*rstd*gv[it] + bv[it];
    if (j == 0 && r < rows) stats[r] = make_float2(mean, rstd);
  }
}
__global__ __launch_bounds__(256) void rownorm_small_bwd_kernel(const float* __restrict__ x,
                                                                const float* __restrict__ dy,
                                                                const float* __restrict__ slope,
                                                                const float* __restrict__ gain,
                                                                const float2* __restrict__ stats,
                                                                float* __restrict__ dx,
                                                                float* __restrict__ dslope_rows,
                                                                long long rows, int n, int npad,
                                                                int inner, int G) {
  const int lane = threadIdx.x & 63, rpw = 64/npad;
  const int sub = lane / npad, j = lane % npad, jc = j < n ? j : n - 1;
  const long long base = ((long long)blockIdx.x*4 + (threadIdx.x >> 6))*rpw*kSmallIt + sub;
  float xv[kSmallIt], dv[kSmallIt], gv[kSmallIt], av[kSmallIt];
  float2 sv[kSmallIt];
#pragma unroll
  for (int it = 0; it < kSmallIt; ++it) {
    const long long r = base + (long long)it*rpw, rr = r < rows ? r : rows - 1;
    const int g = G == 1 ? 0 : (int)((rr / inner) % G);      // (a 64-bit division per element otherwise)
    xv[it] = x[rr*n + jc];
    dv[it] = dy[rr*n + jc];
    sv[it] = stats[rr];
    gv[it] = gain[(long long)g*n + jc];
    av[it] = slope ? slope[g] : 1.f;
  }
#pragma unroll
  for (int it = 0; it < kSmallIt; ++it) {
    const long long r = base + (long long)it*rpw;
    const bool live = r < rows && j < n;
    const float a = av[it];
    const float2 st = sv[it];
    const float v = live ? xv[it] : 0.f;
    const float xh = live ? (prelu(v, a) - st.x)*st.y : 0.f;
    const float dg = live ? dv[it]*gv[it] : 0.f;
    const float m1 = group_sum(dg, npad)/n, m2 = group_sum(dg*xh, npad)/n;
    const float dp = st.y*(dg - m1 - xh*m2);
    if (live) dx[r*n + j] = v > 0.f ? dp : a*dp;
    if (dslope_rows) {
      const float da = group_sum(live && !(v > 0.f) ? dp*v : 0.f, npad);
      if (j == 0 && r < rows) dslope_rows[r] = da;
    }
  }
}

// part[slice][g][j] = sum over the slice's rows of group g of (dy * xhat, dy)
__global__ __launch_bounds__(256) void rownorm_pgrad_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ dy,
                                                            const float* __restrict__ slope,
                                                            const float2* __restrict__ stats,
                                                            float2* __restrict__ part,
                                                            long long rows, int n, int inner, int G) {
  __shared__ float2 red[4][64];
  const int kSlices = gridDim.z;
  const int col = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int j = blockIdx.x*64 + col;
  const int g = blockIdx.y, slice = blockIdx.z;
  const float a = slope ? slope[g] : 1.f;
  const long long per_group = rows/G;                       // rows of one group
  const long long chunk = (per_group + kSlices - 1)/kSlices;
  const long long k0 = slice*chunk, k1 = min(per_group, k0 + chunk);
  float sg = 0.f, sb = 0.f;
  {
    // four rows per lane in flight: unconditional loads at a clamped column (lanes past n add nothing)
    const int jc = j < n ? j : n - 1;
    auto row_of = [&](long long k) { return G == 1 ? k : (k / inner)*((long long)inner*G) + (long long)g*inner + k % inner; };
    float sg1 = 0.f, sb1 = 0.f, sg2 = 0.f, sb2 = 0.f, sg3 = 0.f, sb3 = 0.f;
    long long k = k0 + rl;
    for (; k + 12 < k1; k += 16) {
      const long long r0 = row_of(k), r1 = row_of(k + 4), r2 = row_of(k + 8), r3 = row_of(k + 12);
      const float2 s0 = stats[r0], s1 = stats[r1], s2 = stats[r2], s3 = stats[r3];
      const float d0 = dy[r0*n + jc], d1 = dy[r1*n + jc], d2 = dy[r2*n + jc], d3 = dy[r3*n + jc];
      const float x0 = x[r0*n + jc], x1 = x[r1*n + jc], x2 = x[r2*n + jc], x3 = x[r3*n + jc];
      sg += d0*((prelu(x0, a) - s0.x)*s0.y); sb += d0;
      sg1 += d1*((prelu(x1, a) - s1.x)*s1.y); sb1 += d1;
      sg2 += d2*((prelu(x2, a) - s2.x)*s2.y); sb2 += d2;
      sg3 += d3*((prelu(x3, a) - s3.x)*s3.y); sb3 += d3;
    }
    for (; k < k1; k += 4) {
      const long long r = row_of(k);
      const float2 st = stats[r];
      const float d = dy[r*n + jc];
      sg += d*((prelu(x[r*n + jc], a) - st.x)*st.y);
      sb += d;
    }
    sg = (sg + sg1) + (sg2 + sg3); sb = (sb + sb1) + (sb2 + sb3);
    if (j >= n) sg = sb = 0.f;
  }
  red[rl][col] = make_float2(sg, sb);
  __syncthreads();
  if (rl == 0 && j < n) {
    float2 t = red[0][col];
#pragma unroll
    for (int k = 1; k < 4; ++k) { t.x += red[k][col].x; t.y += red[k][col].y; }
    part[((long long)slice*G + g)*n + j] = t;
  }
}

__global__ __launch_bounds__(256) void rownorm_pgrad_fold_kernel(const float2* __restrict__ part,
                                                                 float* __restrict__ dgain,
                                                                 float* __restrict__ dbias,
                                                                 long long total, int kSlices) {
  // 32 columns per workgroup, the slices dealt to 8 slice lanes with eight loads in flight each (up to 1 024
  // slices for 32 outputs: four lanes walking 256 slices one load at a time took 24 us), fixed summation order
  __shared__ float2 red[8][32];
  const int col = threadIdx.x & 31, ln = threadIdx.x >> 5;
  const long long i = (long long)blockIdx.x*32 + col, ic = i < total ? i : total - 1;
  float sg[8], sb[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) sg[u] = sb[u] = 0.f;
  int s = ln;
  for (; s + 56 < kSlices; s += 64) {
#pragma unroll
    for (int u = 0; u < 8; ++u) { const float2 t = part[(long long)(s + 8*u)*total + ic]; sg[u] += t.x; sb[u] += t.y; }
  }
  for (; s < kSlices; s += 8) { const float2 t = part[(long long)s*total + ic]; sg[0] += t.x; sb[0] += t.y; }
  red[ln][col] = make_float2(((sg[0] + sg[1]) + (sg[2] + sg[3])) + ((sg[4] + sg[5]) + (sg[6] + sg[7])),
                             ((sb[0] + sb[1]) + (sb[2] + sb[3])) + ((sb[4] + sb[5]) + (sb[6] + sb[7])));
  __syncthreads();
  if (ln == 0 && i < total) {
    float2 t = red[0][col];
#pragma unroll
    for (int u = 1; u < 8; ++u) { t.x += red[u][col].x; t.y += red[u][col].y; }
    dgain[i] = t.x;
    dbias[i] = t.y;
  }
}

// column sums of a (rows x cols) matrix (bias gradients of the linear layers / LSTM gates):
// 64 columns x a slice of rows per workgroup, then a fold over the slices in a fixed order
constexpr int kColSlices = 128;
__global__ __launch_bounds__(256) void col_sum_part_kernel(const float* __restrict__ x,
                                                           float* __restrict__ part, long long rows,
                                                           int cols) {
  __shared__ float red[4][64];
  const int col = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int j = blockIdx.x*64 + col, slice = blockIdx.y;
  const float* xb = x + (long long)blockIdx.z*rows*cols;
  const long long chunk = (rows + kColSlices - 1)/kColSlices;
  const long long r0 = slice*chunk, r1 = min(rows, r0 + chunk);
  float acc = 0.f;
  if (j < cols)
    for (long long r = r0 + rl; r < r1; r += 4) acc += xb[r*cols + j];
  red[rl][col] = acc;
  __syncthreads();
  if (rl == 0 && j < cols)
    part[((long long)blockIdx.z*kColSlices + slice)*cols + j] =
        (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
}
// the same partial sums with 16-byte loads (cols % 4 == 0, 16-byte aligned x): a thread owns 4 consecutive columns,
// the lanes of a wavefront run over consecutive column groups and then consecutive rows -- 1 KB of contiguous memory
// per load instruction whatever the width (cw = column groups per workgroup <= 64, 256 / cw row lanes), four
// independent loads in flight per thread; the row lanes are folded through LDS in a fixed order
__global__ __launch_bounds__(256) void col_sum_part4_kernel(const float* __restrict__ x,
                                                            float* __restrict__ part, long long rows,
                                                            int cols) {
  __shared__ float4 red[256];
  const int c4 = cols >> 2;
  const int g0 = blockIdx.x*64, cw = min(64, c4 - g0), RL = 256/cw;
  const int t = threadIdx.x, gl = t % cw, rl = t / cw, slice = blockIdx.y;
  const float4* xb = reinterpret_cast<const float4*>(x + (long long)blockIdx.z*rows*cols) + g0 + gl;
  const long long chunk = (rows + kColSlices - 1)/kColSlices;
  const long long r0 = slice*chunk, r1 = min(rows, r0 + chunk);
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
  auto add = [](float4& a, const float4& v) { a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; };
  if (rl < RL) {
    long long r = r0 + rl;
    for (; r + 3*RL < r1; r += 4*RL) {
      const float4 v0 = xb[r*c4], v1 = xb[(r + RL)*c4], v2 = xb[(r + 2*RL)*c4], v3 = xb[(r + 3*RL)*c4];
      add(a0, v0); add(a1, v1); add(a2, v2); add(a3, v3);
    }
    for (; r < r1; r += RL) add(a0, xb[r*c4]);
    add(a0, a1); add(a2, a3); add(a0, a2);
    red[t] = a0;
  }
  __syncthreads();
  if (t < cw) {
    float4 acc = red[t];
    for (int q = 1; q < RL; ++q) add(acc, red[q*cw + t]);
    reinterpret_cast<float4*>(part + ((long long)blockIdx.z*kColSlices + slice)*cols)[g0 + t] = acc;
  }
}
// bf16 input (the saved gate gradients of the use_amp recurrences): a thread owns 8 consecutive columns (one 16-byte
// load = 8 values), otherwise as col_sum_part4_kernel; sums in fp32
__global__ __launch_bounds__(256) void col_sum_part8h_kernel(const unsigned short* __restrict__ x,
                                                             float* __restrict__ part, long long rows, int cols) {
  __shared__ float red[256][8];
  const int c8 = cols >> 3;
  const int g0 = blockIdx.x*64, cw = min(64, c8 - g0), RL = 256/cw;
  const int t = threadIdx.x, gl = t % cw, rl = t / cw, slice = blockIdx.y;
  const uint4* xb = reinterpret_cast<const uint4*>(x + (long long)blockIdx.z*rows*cols) + g0 + gl;
  const long long chunk = (rows + kColSlices - 1)/kColSlices;
  const long long r0 = slice*chunk, r1 = min(rows, r0 + chunk);
  float a0[8], a1[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) a0[u] = a1[u] = 0.f;
  auto add = [](float* a, const uint4& v) {
    a[0] += __uint_as_float(v.x << 16); a[1] += __uint_as_float(v.x & 0xffff0000u);
    a[2] += __uint_as_float(v.y << 16); a[3] += __uint_as_float(v.y & 0xffff0000u);
    a[4] += __uint_as_float(v.z << 16); a[5] += __uint_as_float(v.z & 0xffff0000u);
    a[6] += __uint_as_float(v.w << 16); a[7] += __uint_as_float(v.w & 0xffff0000u);
  };
  if (rl < RL) {
    long long r = r0 + rl;
    for (; r + 3*RL < r1; r += 4*RL) {
      const uint4 v0 = xb[r*c8], v1 = xb[(r + RL)*c8], v2 = xb[(r + 2*RL)*c8], v3 = xb[(r + 3*RL)*c8];
      add(a0, v0); add(a1, v1); add(a0, v2); add(a1, v3);
    }
    for (; r < r1; r += RL) add(a0, xb[r*c8]);
#pragma unroll
    for (int u = 0; u < 8; ++u) red[t][u] = a0[u] + a1[u];
  }
  __syncthreads();
  if (t < cw) {
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = red[t][u];
    for (int q = 1; q < RL; ++q)
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] += red[q*cw + t][u];
    float4* o = reinterpret_cast<float4*>(part + ((long long)blockIdx.z*kColSlices + slice)*cols) + 2*(g0 + t);
    o[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    o[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
  }
}
__global__ __launch_bounds__(256) void col_sum_fold_kernel(const float* __restrict__ part,
                                                           float* __restrict__ out, int cols) {
  const int j = blockIdx.x*256 + threadIdx.x;
  if (j >= cols) return;
  const float* p = part + (long long)blockIdx.y*kColSlices*cols;
  float acc = 0.f;
  for (int s = 0; s < kColSlices; ++s) acc += p[(long long)s*cols + j];
  out[(long long)blockIdx.y*cols + j] = acc;
}

// y (M x N) (+)= x (M x K) @ op(w) (+ bias[n]) for the narrow layers of the grid blocks (M = batch x frames x
// bands ~ 2.6e5, K and N of 16 - 64: the band / frame linear layers of GridNetBlock, reference
// brever/models/tfgridnet/tfgridnet.py intra / inter linear + their data gradients): 66 MB of traffic against
// 0.5 GFLOP -- a 128 x 128 MFMA tile wastes 3/4 of its rows and columns on them. One thread per row: the row in
// 16-byte loads, the weights through the scalar cache (uniform addresses -> s_load, SGPR operands of the FMAs),
// N accumulators in registers, fp32 throughout (use_amp included: more than the reference's fp16 keeps).
template <int N, int KT, bool TB>          // KT: K when it is 16 / 32 / 64 (the row is requested at once), else 0
__global__ __launch_bounds__(256) void linear_small_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y,
                                                           long long M, int K, long long lda, int ldw, long long ldd,
                                                           int accumulate) {
  const long long row = (long long)blockIdx.x*256 + threadIdx.x;
  const long long r = row < M ? row : M - 1;                 // unconditional loads; the store is conditional
  const float4* xr = reinterpret_cast<const float4*>(x + r*lda);
  float4* yr = reinterpret_cast<float4*>(y + r*ldd);
  float acc[N];
#pragma unroll
  for (int j = 0; j < N; ++j) acc[j] = bias ? bias[j] : 0.f;
  if (accumulate) {
#pragma unroll
    for (int j = 0; j < N/4; ++j) {
      const float4 o = yr[j];
      acc[4*j] += o.x; acc[4*j + 1] += o.y; acc[4*j + 2] += o.z; acc[4*j + 3] += o.w;
    }
  }
  auto step = [&](int k4, const float4& xv) {
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const float w0 = TB ? w[j*ldw + k4] : w[k4*ldw + j];
      const float w1 = TB ? w[j*ldw + k4 + 1] : w[(k4 + 1)*ldw + j];
      const float w2 = TB ? w[j*ldw + k4 + 2] : w[(k4 + 2)*ldw + j];
      const float w3 = TB ? w[j*ldw + k4 + 3] : w[(k4 + 3)*ldw + j];
      acc[j] = fmaf(xv.w, w3, fmaf(xv.z, w2, fmaf(xv.y, w1, fmaf(xv.x, w0, acc[j]))));
    }
  };
  if constexpr (KT > 0) {
    float4 xv[KT/4];
#pragma unroll
    for (int q = 0; q < KT/4; ++q) xv[q] = xr[q];
#pragma unroll
    for (int q = 0; q < KT/4; ++q) step(4*q, xv[q]);
  } else {
    for (int k4 = 0; k4 < K; k4 += 4) step(k4, xr[k4 >> 2]);
  }
  if (row < M) {
#pragma unroll
    for (int j = 0; j < N/4; ++j) yr[j] = make_float4(acc[4*j], acc[4*j + 1], acc[4*j + 2], acc[4*j + 3]);
  }
}

// d (MI x NJ) = sum over rows r of a[r][i] b[r][j] for the same narrow layers (their weight gradients: 2.6e5 rows,
// MI, NJ of 16 / 32): the rows are cut into slices, a workgroup stages 64 rows of both operands in LDS (16-byte
// loads) and every thread owns one i and four consecutive j; partial matrices are folded in slice order
// (fixed order, no atomics). 66 MB of traffic per call; the 128 x 128 product with 512 atomic splits took ~80 us.
constexpr int kWgSlices = 512, kWgChunk = 64;
template <int MI, int NJ>
__global__ __launch_bounds__(256) void wgrad_small_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          float* __restrict__ part, long long R, long long lda,
                                                          long long ldb) {
  static_assert(MI*NJ/4 <= 256 && kWgChunk*MI/4 % 256 == 0 && kWgChunk*NJ/4 % 256 == 0, "tile");
  __shared__ __attribute__((aligned(16))) float As[kWgChunk][MI];
  __shared__ __attribute__((aligned(16))) float Bs[kWgChunk][NJ];
  const int t = threadIdx.x;
  const int i = t / (NJ/4), j4 = t % (NJ/4);
  const bool owner = t < MI*NJ/4;
  const long long per = ((R + kWgSlices - 1)/kWgSlices + kWgChunk - 1)/kWgChunk*kWgChunk;
  const long long r0 = (long long)blockIdx.x*per, r1 = min(R, r0 + per);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long long rc = r0; rc < r1; rc += kWgChunk) {
    // unconditional loads at clamped rows, zeroed by a mask (rows past the slice add nothing)
#pragma unroll
    for (int q = 0; q < kWgChunk*MI/4/256; ++q) {
      const int e = t + 256*q, rr = e / (MI/4), c4 = e % (MI/4);
      const long long r = rc + rr;
      float4 v = *reinterpret_cast<const float4*>(a + (r < r1 ? r : r1 - 1)*lda + 4*c4);
      if (r >= r1) v = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(&As[rr][4*c4]) = v;
    }
#pragma unroll
    for (int q = 0; q < kWgChunk*NJ/4/256; ++q) {
      const int e = t + 256*q, rr = e / (NJ/4), c4 = e % (NJ/4);
      const long long r = rc + rr;
      const float4 v = *reinterpret_cast<const float4*>(b + (r < r1 ? r : r1 - 1)*ldb + 4*c4);
      *reinterpret_cast<float4*>(&Bs[rr][4*c4]) = v;
    }
    __syncthreads();
    if (owner) {
#pragma unroll 16
      for (int rr = 0; rr < kWgChunk; ++rr) {
        const float x = As[rr][i];
        const float4 y = *reinterpret_cast<const float4*>(&Bs[rr][4*j4]);
        acc.x = fmaf(x, y.x, acc.x); acc.y = fmaf(x, y.y, acc.y);
        acc.z = fmaf(x, y.z, acc.z); acc.w = fmaf(x, y.w, acc.w);
      }
    }
    __syncthreads();
  }
  if (owner) *reinterpret_cast<float4*>(part + ((long long)blockIdx.x*MI + i)*NJ + 4*j4) = acc;
}
// fold: 32 outputs per workgroup, eight threads per output over the slices (eight loads in flight each), the eight
// partial sums added in lane order through LDS
__global__ __launch_bounds__(256) void wgrad_small_fold_kernel(const float* __restrict__ part, float* __restrict__ d,
                                                               int MI, int NJ, long long ldd, int slices) {
  __shared__ float red[8][32];
  const int t = threadIdx.x, el = t & 31, ln = t >> 5;
  const int e = blockIdx.x*32 + el;                     // MI*NJ is a multiple of 32
  const long long n = (long long)MI*NJ;
  float s[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) s[u] = 0.f;
  int q = ln;
  for (; q + 56 < slices; q += 64) {
#pragma unroll
    for (int u = 0; u < 8; ++u) s[u] += part[(long long)(q + 8*u)*n + e];
  }
  for (; q < slices; q += 8) s[0] += part[(long long)q*n + e];
  red[ln][el] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  __syncthreads();
  if (ln == 0) {
    float v = red[0][el];
#pragma unroll
    for (int u = 1; u < 8; ++u) v += red[u][el];
    d[(long long)(e / NJ)*ldd + e % NJ] = v;
  }
}

// unbiased standard deviation of each row (two passes, fp64 accumulators)
// (1024 threads and 16-byte loads: one workgroup per row -- four rows at the benchmark batch -- with 256 threads and
// scalar loads was two chains of 500 dependent loads per thread: 172 us per launch)
__global__ __launch_bounds__(1024) void row_std_kernel(const float* __restrict__ x,
                                                       float* __restrict__ out, long long n) {
  __shared__ double scr[16];
  __shared__ double mean_s;
  const float* xr = x + (long long)blockIdx.x*n;
  const int tid = threadIdx.x, nt = blockDim.x;
  auto block_sum16 = [&](double v) {               // sum over the 16 waves; result valid in thread 0
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) scr[tid >> 6] = v;
    __syncthreads();
    double r = 0.0;
    if (tid == 0) for (int i = 0; i < 16; ++i) r += scr[i];
    return r;
  };
  const bool vec = (n & 3) == 0 && (reinterpret_cast<unsigned long long>(xr) & 15) == 0;
  const long long n4 = vec ? n >> 2 : 0;
  const float4* x4 = reinterpret_cast<const float4*>(xr);
  double s = 0.0;
  for (long long j = tid; j < n4; j += nt) { const float4 v = x4[j]; s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w); }
  for (long long j = 4*n4 + tid; j < n; j += nt) s += xr[j];
  const double tot = block_sum16(s);
  if (tid == 0) mean_s = tot/(double)n;
  __syncthreads();
  const double mean = mean_s;
  double q = 0.0;
  for (long long j = tid; j < n4; j += nt) {
    const float4 v = x4[j];
    const double a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
    q += (a*a + b*b) + (c*c + d*d);
  }
  for (long long j = 4*n4 + tid; j < n; j += nt) { const double d = xr[j] - mean; q += d*d; }
  const double var = block_sum16(q);
  if (tid == 0) out[blockIdx.x] = (float)sqrt(var/(double)(n - 1));
}

template <bool DIV>
__global__ __launch_bounds__(256) void row_scale_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ s,
                                                        float* __restrict__ y, long long n) {
  const long long r = blockIdx.y;
  const float f = s[r];
  for (long long j = (long long)blockIdx.x*256 + threadIdx.x; j < n; j += (long long)gridDim.x*256)
    y[r*n + j] = DIV ? x[r*n + j]/f : x[r*n + j]*f;
}

}  // namespace

// (B, T, F, H, E) <-> (B, H, T, E, F): the head split in front of the attention's row norms and the head merge behind the
// attention products (tfgridnet.py:315-353: channels-last activations, heads as the outer batch of the products). One
// workgroup per (item, frame): the (F, H E) plane goes through LDS, both sides in whole contiguous runs. (As
// torch permute + copy -- a generic five-dimensional strided copy -- these ran at 1.3 TB/s: 1.7 ms of a 27.9 ms step.)
template <bool MERGE>
__global__ __launch_bounds__(256) void head_permute_kernel(const float* in, float* out, int T, int Fq, int H, int E) {
  extern __shared__ float plane[];                 // [H E][Fp]
  const int HE = H*E, Fp = Fq | 1;                 // odd stride: the transposed accesses touch 32 different banks
  const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const long long cl = ((long long)b*T + t)*Fq*HE;                 // channels-last plane of this frame
  const int n = Fq*HE;
  // H E a power of two >= 4 (the network's 16 and 32): 16-byte accesses on the channels-last side, shifts for the
  // (band, channel) split; otherwise element by element with a division
  const bool fast = (HE & (HE - 1)) == 0 && HE >= 4 && ((reinterpret_cast<unsigned long long>(in) |
                    reinterpret_cast<unsigned long long>(out)) & 15) == 0;
  const int lg = 31 - __builtin_clz(HE);
  auto rows = [&](auto&& body) {                   // wave w takes the rows c = w, w + 4, ...; lanes run along the bands
    for (int c = w; c < HE; c += 4) {
      const int h = c / E, e = c - h*E;
      const long long ro = ((((long long)b*H + h)*T + t)*E + e)*Fq;
      for (int f = lane; f < Fq; f += 64) body(ro + f, c*Fp + f);
    }
  };
  if (!MERGE) {
    if (fast) {
      for (int i = tid*4; i < n; i += 1024) {
        const float4 v = *reinterpret_cast<const float4*>(in + cl + i);
        const int f = i >> lg, c = i & (HE - 1);
        plane[c*Fp + f] = v.x; plane[(c + 1)*Fp + f] = v.y; plane[(c + 2)*Fp + f] = v.z; plane[(c + 3)*Fp + f] = v.w;
      }
    } else {
      for (int i = tid; i < n; i += 256) { const int f = i / HE, c = i - f*HE; plane[c*Fp + f] = in[cl + i]; }
    }
    __syncthreads();
    rows([&](long long g, int l) { out[g] = plane[l]; });
  } else {
    rows([&](long long g, int l) { plane[l] = in[g]; });
    __syncthreads();
    if (fast) {
      for (int i = tid*4; i < n; i += 1024) {
        const int f = i >> lg, c = i & (HE - 1);
        float4 v;
        v.x = plane[c*Fp + f]; v.y = plane[(c + 1)*Fp + f]; v.z = plane[(c + 2)*Fp + f]; v.w = plane[(c + 3)*Fp + f];
        *reinterpret_cast<float4*>(out + cl + i) = v;
      }
    } else {
      for (int i = tid; i < n; i += 256) { const int f = i / HE, c = i - f*HE; out[cl + i] = plane[c*Fp + f]; }
    }
  }
}

extern "C" {

int brv_head_permute(const float* in, float* out, int64_t B, int64_t T, int64_t F, int64_t H, int64_t E, int merge,
                     brv_stream_t stream) {
  if (!in || !out || B < 1 || T < 1 || F < 1 || H < 1 || E < 1 || B > 65535) return -1;
  const size_t lds = (size_t)H*E*(F | 1)*sizeof(float);
  if (lds > 64*1024) return -1;                    // (brv_head_permute_supported)
  if (merge) hipLaunchKernelGGL(head_permute_kernel<true>, dim3((unsigned)T, (unsigned)B), dim3(256), lds,
                                (hipStream_t)stream, in, out, (int)T, (int)F, (int)H, (int)E);
  else hipLaunchKernelGGL(head_permute_kernel<false>, dim3((unsigned)T, (unsigned)B), dim3(256), lds,
                          (hipStream_t)stream, in, out, (int)T, (int)F, (int)H, (int)E);
  TG_OK(hipGetLastError());
  return 0;
}
int brv_head_permute_supported(int64_t F, int64_t H, int64_t E) {
  return F >= 1 && H >= 1 && E >= 1 && (size_t)H*E*(F | 1)*sizeof(float) <= 64*1024;
}

int brv_rownorm_forward(const float* x, const float* slope, const float* gain, const float* bias,
                        float* y, float* stats, int64_t rows, int64_t n, int64_t inner,
                        int64_t groups, float eps, brv_stream_t stream) {
  if (rows < 1 || n < 1 || inner < 1 || groups < 1 || rows % (inner*groups)) return -1;
  hipStream_t st = (hipStream_t)stream;
  if (n <= 64) {
    int npad = 1;
    while (npad < n) npad <<= 1;
    const long long rows_per_wg = 4*(64/npad)*kSmallIt;
    hipLaunchKernelGGL(rownorm_small_fwd_kernel, dim3((unsigned)((rows + rows_per_wg - 1)/rows_per_wg)),
                       dim3(256), 0, st, x, slope, gain, bias, y, reinterpret_cast<float2*>(stats),
                       (long long)rows, (int)n, npad, (int)inner, (int)groups, eps);
  } else
  hipLaunchKernelGGL(rownorm_fwd_kernel, dim3((unsigned)((rows + 3)/4)), dim3(256), 0, st, x, slope,
                     gain, bias, y, reinterpret_cast<float2*>(stats), (long long)rows, (int)n,
                     (int)inner, (int)groups, eps);
  TG_OK(hipGetLastError());
  return 0;
}

int64_t brv_rownorm_scratch_bytes(int64_t n, int64_t groups) {
  return (int64_t)pgrad_slices(n, groups)*groups*n*(int64_t)sizeof(float2);
}

int brv_rownorm_backward(const float* x, const float* dy, const float* slope, const float* gain,
                         const float* stats, float* dx, float* dgain, float* dbias,
                         float* dslope_rows, void* scratch, int64_t rows, int64_t n, int64_t inner,
                         int64_t groups, brv_stream_t stream) {
  if (rows < 1 || n < 1 || inner < 1 || groups < 1 || rows % (inner*groups)) return -1;
  hipStream_t st = (hipStream_t)stream;
  const float2* stp = reinterpret_cast<const float2*>(stats);
  if (n <= 64) {
    int npad = 1;
    while (npad < n) npad <<= 1;
    const long long rows_per_wg = 4*(64/npad)*kSmallIt;
    hipLaunchKernelGGL(rownorm_small_bwd_kernel, dim3((unsigned)((rows + rows_per_wg - 1)/rows_per_wg)),
                       dim3(256), 0, st, x, dy, slope, gain, stp, dx, slope ? dslope_rows : nullptr,
                       (long long)rows, (int)n, npad, (int)inner, (int)groups);
  } else
  hipLaunchKernelGGL(rownorm_bwd_kernel, dim3((unsigned)((rows + 3)/4)), dim3(256), 0, st, x, dy,
                     slope, gain, stp, dx, slope ? dslope_rows : nullptr, (long long)rows, (int)n,
                     (int)inner, (int)groups);
  float2* part = reinterpret_cast<float2*>(scratch);
  const int kSlices = pgrad_slices(n, groups);
  hipLaunchKernelGGL(rownorm_pgrad_kernel, dim3((unsigned)((n + 63)/64), (unsigned)groups, kSlices),
                     dim3(256), 0, st, x, dy, slope, stp, part, (long long)rows, (int)n, (int)inner,
                     (int)groups);
  const long long total = (long long)groups*n;
  hipLaunchKernelGGL(rownorm_pgrad_fold_kernel, dim3((unsigned)((total + 31)/32)), dim3(256), 0,
                     st, part, dgain, dbias, total, kSlices);
  TG_OK(hipGetLastError());
  return 0;
}

int64_t brv_col_sum_scratch_bytes(int64_t batch, int64_t cols) {
  return batch*kColSlices*cols*(int64_t)sizeof(float);
}
int brv_col_sum(const float* x, float* out, void* scratch, int64_t batch, int64_t rows, int64_t cols,
                brv_stream_t stream) {
  if (batch < 1 || rows < 1 || cols < 1 || batch > 65535) return -1;
  hipStream_t st = (hipStream_t)stream;
  if (cols % 4 == 0 && (((uintptr_t)x | (uintptr_t)scratch) & 15) == 0)
    hipLaunchKernelGGL(col_sum_part4_kernel, dim3((unsigned)((cols/4 + 63)/64), kColSlices, (unsigned)batch),
                       dim3(256), 0, st, x, (float*)scratch, (long long)rows, (int)cols);
  else
    hipLaunchKernelGGL(col_sum_part_kernel, dim3((unsigned)((cols + 63)/64), kColSlices, (unsigned)batch),
                       dim3(256), 0, st, x, (float*)scratch, (long long)rows, (int)cols);
  hipLaunchKernelGGL(col_sum_fold_kernel, dim3((unsigned)((cols + 255)/256), (unsigned)batch), dim3(256),
                     0, st, (const float*)scratch, out, (int)cols);
  TG_OK(hipGetLastError());
  return 0;
}
int brv_linear_small_supported(int64_t M, int64_t N, int64_t K) {
  return (N == 16 || N == 32 || N == 64) && K >= 4 && K <= 64 && K % 4 == 0 && M >= 4096;
}
int brv_linear_small(const float* x, const float* w, const float* bias, float* y, int64_t M, int64_t N, int64_t K,
                     int64_t lda, int64_t ldw, int64_t ldd, int trans_b, int accumulate, brv_stream_t stream) {
  if (!brv_linear_small_supported(M, N, K) || (lda & 3) || (ldd & 3) || lda < K || ldd < N ||
      (((uintptr_t)x | (uintptr_t)y) & 15) || ldw < (trans_b ? K : N) || M > (1LL << 38))
    return -1;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)((M + 255)/256));
#define BRV_LS2(N_, K_) do { \
    if (trans_b) hipLaunchKernelGGL((linear_small_kernel<N_, K_, true>), grid, dim3(256), 0, st, x, w, bias, y, \
                                    (long long)M, (int)K, (long long)lda, (int)ldw, (long long)ldd, accumulate); \
    else hipLaunchKernelGGL((linear_small_kernel<N_, K_, false>), grid, dim3(256), 0, st, x, w, bias, y, \
                            (long long)M, (int)K, (long long)lda, (int)ldw, (long long)ldd, accumulate); } while (0)
#define BRV_LS(N_) do { if (K == 16) BRV_LS2(N_, 16); else if (K == 32) BRV_LS2(N_, 32); \
                        else if (K == 64) BRV_LS2(N_, 64); else BRV_LS2(N_, 0); } while (0)
  if (N == 16) BRV_LS(16); else if (N == 32) BRV_LS(32); else BRV_LS(64);
#undef BRV_LS2
#undef BRV_LS
  TG_OK(hipGetLastError());
  return 0;
}
int brv_linear_small_wgrad_supported(int64_t rows, int64_t MI, int64_t NJ) {
  return (MI == 16 || MI == 32) && (NJ == 16 || NJ == 32) && rows >= 4096;
}
int64_t brv_linear_small_wgrad_scratch_bytes(int64_t MI, int64_t NJ) { return (int64_t)kWgSlices*MI*NJ*4; }
int brv_linear_small_wgrad(const float* a, const float* b, float* d, void* scratch, int64_t rows, int64_t MI,
                           int64_t NJ, int64_t lda, int64_t ldb, int64_t ldd, brv_stream_t stream) {
  if (!brv_linear_small_wgrad_supported(rows, MI, NJ) || (lda & 3) || (ldb & 3) || lda < MI || ldb < NJ ||
      ldd < NJ || (((uintptr_t)a | (uintptr_t)b | (uintptr_t)scratch) & 15) || !scratch)
    return -1;
  hipStream_t st = (hipStream_t)stream;
  const long long per = ((rows + kWgSlices - 1)/kWgSlices + kWgChunk - 1)/kWgChunk*kWgChunk;
  const int slices = (int)((rows + per - 1)/per);
  float* part = (float*)scratch;
#define BRV_WS(MI_, NJ_) hipLaunchKernelGGL((wgrad_small_kernel<MI_, NJ_>), dim3(slices), dim3(256), 0, st, a, b, \
                                            part, (long long)rows, (long long)lda, (long long)ldb)
  if (MI == 16 && NJ == 16) BRV_WS(16, 16);
  else if (MI == 16) BRV_WS(16, 32);
  else if (NJ == 16) BRV_WS(32, 16);
  else BRV_WS(32, 32);
#undef BRV_WS
  hipLaunchKernelGGL(wgrad_small_fold_kernel, dim3((unsigned)(MI*NJ/32)), dim3(256), 0, st, part, d,
                     (int)MI, (int)NJ, (long long)ldd, slices);
  TG_OK(hipGetLastError());
  return 0;
}
int brv_col_sum_bf16(const void* x, float* out, void* scratch, int64_t batch, int64_t rows, int64_t cols,
                     brv_stream_t stream) {
  if (batch < 1 || rows < 1 || cols < 8 || batch > 65535 || cols % 8 != 0 ||
      (((uintptr_t)x | (uintptr_t)scratch) & 15))
    return -1;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(col_sum_part8h_kernel, dim3((unsigned)((cols/8 + 63)/64), kColSlices, (unsigned)batch),
                     dim3(256), 0, st, (const unsigned short*)x, (float*)scratch, (long long)rows, (int)cols);
  hipLaunchKernelGGL(col_sum_fold_kernel, dim3((unsigned)((cols + 255)/256), (unsigned)batch), dim3(256),
                     0, st, (const float*)scratch, out, (int)cols);
  TG_OK(hipGetLastError());
  return 0;
}
int brv_row_std(const float* x, float* out, int64_t rows, int64_t n, brv_stream_t stream) {
  if (rows < 1 || n < 2) return -1;
  hipLaunchKernelGGL(row_std_kernel, dim3((unsigned)rows), dim3(1024), 0, (hipStream_t)stream, x,
                     out, (long long)n);
  TG_OK(hipGetLastError());
  return 0;
}

int brv_row_scale(const float* x, const float* s, float* y, int64_t rows, int64_t n, int divide,
                  brv_stream_t stream) {
  if (rows < 1 || n < 1 || rows > 65535) return -1;
  const unsigned gx = (unsigned)min((long long)1024, (long long)((n + 255)/256));
  if (divide)
    hipLaunchKernelGGL(row_scale_kernel<true>, dim3(gx, (unsigned)rows), dim3(256), 0,
                       (hipStream_t)stream, x, s, y, (long long)n);
  else
    hipLaunchKernelGGL(row_scale_kernel<false>, dim3(gx, (unsigned)rows), dim3(256), 0,
                       (hipStream_t)stream, x, s, y, (long long)n);
  TG_OK(hipGetLastError());
  return 0;
}

}  // extern "C"
