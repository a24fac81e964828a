// Cumulative (causal) layer normalisation, forward and backward, fused with the PReLU in
// front of it. Channels-last bf16 tensors [item][frame][Cp].
//
// Reference: CausalLayerNorm (brever/modules/normalization.py:5-62) as used by the causal
// Conv-TasNet (brever/models/convtasnet/convtasnet.py:263-268, eps 1e-8): with
// p = PReLU(z) and n_t = C*(t+1),
//   mean_t = (sum_{tau<=t} sum_c p) / n_t,   var_t = (sum_{tau<=t} sum_c p^2) / n_t - mean_t^2,
//   y[t][c] = (p[t][c] - mean_t) * rstd_t * gain[c] + bias[c],   rstd_t = 1/sqrt(var_t + eps).
// Three launches each way, because every frame depends on a prefix (forward) or a suffix
// (backward) of per-frame channel sums:
//   forward : frame sums -> prefix scan (fp64) -> apply
//   backward: frame sums of g^ = g*gain and g^*x^  -> suffix scan (fp64) -> apply
// with  dp[tau][c] = g^[tau][c]*rstd_tau + R1_tau + 2 p[tau][c] R2_tau,
//   R1_tau = sum_{t>=tau} (dmu_t - 2 mean_t dvar_t)/n_t,   R2_tau = sum_{t>=tau} dvar_t/n_t,
//   dmu_t = -rstd_t P_t,   dvar_t = -rstd_t^2 Q_t / 2,   P_t = sum_c g^,  Q_t = sum_c g^ x^.
// The causal model keeps the normalised tensors in HBM (the non-causal path re-applies its
// per-item statistics in the consumers instead); it is the correctness-first variant.
#pragma once
#include "common.cuh"
#include "tcn_kernels.cuh"

namespace brv {

struct ClnParams {
  const bf16_t* z;          // input before the PReLU (slope null: no PReLU)
  const float* slope;
  bf16_t* y;                // fwd: normalised output
  float* fsum;              // [B][T][2] per-frame sums (fwd: p, p^2; bwd: P, Q)
  float* table;             // [B][T][2] fwd: (mean, rstd); bwd: (R1, R2)
  const float* fwd_table;   // bwd: the forward table
  const float* gain; const float* bias;
  int B, T, Cp, C;
  float eps;
  // backward
  const bf16_t* g;          // gradient wrt y
  bf16_t* dz;               // gradient wrt z
  const bf16_t* add_in; int n_add;   // dz += sum_s add_in[(b*n_add + s)][t][c] (per-source terms)
  float* dgain; float* dbias; float* dslope; long long rep_stride;
};

constexpr int CLN_FPW = 8;                 // frames per wave and workgroup pass
constexpr int CLN_FPB = 4*CLN_FPW;         // frames per workgroup

// ---- forward 1: per-frame sums of p and p^2 over the true channels ------------------------
__global__ __launch_bounds__(256) void cln_frame_sums_kernel(const ClnParams p) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int n_tt = ceil_div(p.T, CLN_FPB);
  const int b = blockIdx.x / n_tt;
  const int t0 = (blockIdx.x % n_tt)*CLN_FPB + wid*CLN_FPW;
  const float a = p.slope ? *p.slope : 1.f;
  for (int u = 0; u < CLN_FPW; ++u) {
    const int t = t0 + u;
    if (t >= p.T) break;                                   // wave-uniform
    float s1 = 0.f, s2 = 0.f;
    for (int c0 = lane*8; c0 < p.Cp; c0 += 512) {
      float f[8];
      unpack8(*reinterpret_cast<const uint4*>(p.z + ((long long)b*p.T + t)*p.Cp + c0), f);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = c0 + j < p.C ? prelu(f[j], a) : 0.f;
        s1 += v; s2 += v*v;
      }
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if (lane == 0) {
      p.fsum[((long long)b*p.T + t)*2] = s1;
      p.fsum[((long long)b*p.T + t)*2 + 1] = s2;
    }
  }
}

// Block-wide inclusive scan of per-thread segment totals (256 threads), fp64.
__device__ __forceinline__ void block_scan2(double& a, double& b, double* scr, bool reverse) {
  // scr: 2*256 doubles
  const int tid = threadIdx.x;
  scr[tid] = a; scr[256 + tid] = b;
  __syncthreads();
  double sa = 0.0, sb = 0.0;                               // exclusive prefix (or suffix)
  if (!reverse) { for (int i = 0; i < tid; ++i) { sa += scr[i]; sb += scr[256 + i]; } }
  else { for (int i = tid + 1; i < 256; ++i) { sa += scr[i]; sb += scr[256 + i]; } }
  __syncthreads();
  a = sa; b = sb;
}

// ---- forward 2: prefix sums over frames -> (mean_t, rstd_t); one workgroup per item -------
__global__ __launch_bounds__(256) void cln_scan_kernel(const ClnParams p) {
  __shared__ double scr[512];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int per = ceil_div(p.T, 256);
  const int lo = tid*per, hi = min(p.T, lo + per);
  const float* fs = p.fsum + (long long)b*p.T*2;
  double s1 = 0.0, s2 = 0.0;
  for (int t = lo; t < hi; ++t) { s1 += fs[2*t]; s2 += fs[2*t + 1]; }
  block_scan2(s1, s2, scr, false);                          // sums of the segments before
  float* tb = p.table + (long long)b*p.T*2;
  for (int t = lo; t < hi; ++t) {
    s1 += fs[2*t]; s2 += fs[2*t + 1];
    const double n = (double)p.C*(double)(t + 1);
    const double mean = s1/n;
    const double var = s2/n - mean*mean;
    tb[2*t] = (float)mean;
    tb[2*t + 1] = (float)(1.0/sqrt(var + (double)p.eps));
  }
}

// ---- forward 3: y = (prelu(z) - mean_t) rstd_t gain + bias ------------------------------------
__global__ __launch_bounds__(256) void cln_apply_kernel(const ClnParams p) {
  const int cpr = p.Cp/8;
  const long long total = (long long)p.B*p.T*cpr;
  const float a = p.slope ? *p.slope : 1.f;
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < total; i += (long long)gridDim.x*256) {
    const int c0 = (int)(i % cpr)*8;
    const long long row = i / cpr;                          // b*T + t
    const float mean = p.table[2*row], rstd = p.table[2*row + 1];
    float f[8], g8[8], b8[8], o[8];
    unpack8(*reinterpret_cast<const uint4*>(p.z + row*p.Cp + c0), f);
    load8_masked(p.gain, c0, p.C, g8);
    load8_masked(p.bias, c0, p.C, b8);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (prelu(f[j], a) - mean)*rstd*g8[j] + b8[j];
    *reinterpret_cast<uint4*>(p.y + row*p.Cp + c0) = pack8(o);
  }
}

// ---- backward 1: per-frame P = sum g^, Q = sum g^ x^; per-channel dgain, dbias -------------------
__global__ __launch_bounds__(256) void cln_bwd_sums_kernel(const ClnParams p) {
  __shared__ float red[4*512];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int n_tt = ceil_div(p.T, CLN_FPB);
  const int b = blockIdx.x / n_tt;
  const int t0 = (blockIdx.x % n_tt)*CLN_FPB + wid*CLN_FPW;
  const float a = p.slope ? *p.slope : 1.f;
  for (int cb = 0; cb < p.Cp; cb += 512) {
    const int c0 = cb + lane*8;
    const bool lane_ok = c0 < p.Cp;
    float g8[8], dga[8], dbi[8];
    load8_masked(p.gain, c0, p.C, g8);
#pragma unroll
    for (int j = 0; j < 8; ++j) { dga[j] = 0.f; dbi[j] = 0.f; }
    for (int u = 0; u < CLN_FPW; ++u) {
      const int t = t0 + u;
      if (t >= p.T) break;                                 // wave-uniform
      const long long row = (long long)b*p.T + t;
      const float mean = p.fwd_table[2*row], rstd = p.fwd_table[2*row + 1];
      float P = 0.f, Q = 0.f;
      if (lane_ok) {
        float zf[8], gf[8];
        unpack8(*reinterpret_cast<const uint4*>(p.z + row*p.Cp + c0), zf);
        unpack8(*reinterpret_cast<const uint4*>(p.g + row*p.Cp + c0), gf);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xh = (prelu(zf[j], a) - mean)*rstd;
          const float gh = gf[j]*g8[j];
          P += gh; Q += gh*xh;
          dga[j] += gf[j]*xh; dbi[j] += gf[j];
        }
      }
      P = wave_sum(P); Q = wave_sum(Q);
      if (lane == 0) {
        // several channel blocks (Cp > 512) accumulate into the same frame slot
        if (cb == 0) { p.fsum[2*row] = P; p.fsum[2*row + 1] = Q; }
        else { p.fsum[2*row] += P; p.fsum[2*row + 1] += Q; }
      }
    }
    // per-channel sums over the workgroup's frames -> one atomic per channel and quantity
    for (int which = 0; which < 2; ++which) {
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 8; ++j) red[wid*512 + lane*8 + j] = which ? dbi[j] : dga[j];
      __syncthreads();
      float* dst = (which ? p.dbias : p.dgain) + (long long)(blockIdx.x % kReplicas)*p.rep_stride;
      for (int cc = tid; cc < 512; cc += 256) {
        const float s = red[cc] + red[512 + cc] + red[1024 + cc] + red[1536 + cc];
        if (cb + cc < p.C) atomic_add_f32(dst + cb + cc, s);
      }
    }
  }
}

// ---- backward 2: suffix sums -> (R1_t, R2_t); one workgroup per item ------------------------------
__global__ __launch_bounds__(256) void cln_bwd_scan_kernel(const ClnParams p) {
  __shared__ double scr[512];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int per = ceil_div(p.T, 256);
  const int lo = tid*per, hi = min(p.T, lo + per);
  const float* fs = p.fsum + (long long)b*p.T*2;
  const float* ft = p.fwd_table + (long long)b*p.T*2;
  auto terms = [&](int t, double& d1, double& d2) {
    const double n = (double)p.C*(double)(t + 1);
    const double mean = ft[2*t], rstd = ft[2*t + 1];
    const double dmu = -rstd*(double)fs[2*t];
    const double dvar = -0.5*rstd*rstd*(double)fs[2*t + 1];
    d1 = (dmu - 2.0*mean*dvar)/n;
    d2 = dvar/n;
  };
  double s1 = 0.0, s2 = 0.0;
  for (int t = lo; t < hi; ++t) { double d1, d2; terms(t, d1, d2); s1 += d1; s2 += d2; }
  block_scan2(s1, s2, scr, true);                           // sums of the segments after
  float* tb = p.table + (long long)b*p.T*2;
  for (int t = hi - 1; t >= lo; --t) {
    double d1, d2; terms(t, d1, d2);
    s1 += d1; s2 += d2;
    tb[2*t] = (float)s1;
    tb[2*t + 1] = (float)s2;
  }
}

// ---- backward 3: dz = prelu'(z) (g^ rstd + R1 + 2 p R2) (+ add_in), slope gradient ---------------
__global__ __launch_bounds__(256) void cln_bwd_apply_kernel(const ClnParams p) {
  __shared__ float fscr[8];
  const int cpr = p.Cp/8;
  const long long total = (long long)p.B*p.T*cpr;
  const float a = p.slope ? *p.slope : 1.f;
  float da = 0.f;
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < total; i += (long long)gridDim.x*256) {
    const int c0 = (int)(i % cpr)*8;
    const long long row = i / cpr;
    const float rstd = p.fwd_table[2*row + 1];
    const float R1 = p.table[2*row], R2 = p.table[2*row + 1];
    float zf[8], gf[8], g8[8], o[8];
    unpack8(*reinterpret_cast<const uint4*>(p.z + row*p.Cp + c0), zf);
    unpack8(*reinterpret_cast<const uint4*>(p.g + row*p.Cp + c0), gf);
    load8_masked(p.gain, c0, p.C, g8);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool pos = zf[j] > 0.f || p.slope == nullptr;
      const float pv = pos ? zf[j] : a*zf[j];
      const float dp = c0 + j < p.C ? gf[j]*g8[j]*rstd + R1 + 2.f*pv*R2 : 0.f;
      o[j] = pos ? dp : a*dp;
      if (!pos) da += dp*zf[j];
    }
    for (int s = 0; s < p.n_add; ++s) {
      float d[8];
      const long long arow = ((row / p.T)*p.n_add + s)*p.T + row % p.T;
      unpack8(*reinterpret_cast<const uint4*>(p.add_in + arow*p.Cp + c0), d);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] += d[j];
    }
    *reinterpret_cast<uint4*>(p.dz + row*p.Cp + c0) = pack8(o);
  }
  if (p.dslope) {
    const float s = block_sum(da, fscr);
    if (threadIdx.x == 0) atomic_add_f32(p.dslope + (long long)(blockIdx.x % kReplicas)*p.rep_stride, s);
  }
}

}  // namespace brv
