// HBM-bound stencil / elementwise kernels of the Conv-TasNet TCN block.
//
//  dwconv_fwd : z2 = dconv( gLN1( prelu1(z1) ) ) + bias           (+ gLN2 statistics)
//  dz_kernel  : dz = prelu'(z) * gLN_backward(e, z)  in place     (+ PReLU slope grad)
//  dwconv_bwd : e1 = gamma1 * dconv^T(dz2); tap / bias / gamma1 / beta1 gradients,
//               partial sums of the gLN1 backward
//  gln0_bwd_combine : gradient wrt the encoder output (norm path + mask path)
//
// Layout: [B][T][Cp] bf16, 8 channels (16 B) per lane access; one wavefront reads a
// whole 512-channel frame (1 KiB contiguous) per instruction; a workgroup covers 128
// consecutive frames (32 per wave), so dilated taps are mostly L1/L2 hits.
//
// Reference ops: F.pad + depthwise nn.Conv1d, nn.PReLU, nn.GroupNorm(1, C) in
// brever/models/convtasnet/convtasnet.py:240-260,263-268 and their autograd.
#pragma once
#include "common.cuh"

namespace brv {

struct DwParams {
  const bf16_t* z1;        // pre-PReLU1 activations (saved by the forward)
  bf16_t* z2;              // fwd: output (pre-PReLU2)
  const bf16_t* dz2;       // bwd: gradient wrt z2
  bf16_t* e1;              // bwd: gamma1 * d(h1n)
  int B, T, Cp, C;
  const float* slope1; const double* stats1; const float* gamma1; const float* beta1;
  double inv_n; float eps;
  const float* taps;       // [C][P]
  const float* bias;       // [C]
  int dil, left;
  double* stats2; const float* slope2;                   // fwd
  float* dgamma1; float* dbeta1; float* dtaps; float* dbias; double* sums1;   // bwd
};

constexpr int DW_TT = 128;     // frames per workgroup (4 waves x 32 consecutive frames)
constexpr int DW_RPW = DW_TT/4;

// One wavefront handles whole frames: 64 lanes x 8 channels = 512 channels = 1 KiB
// contiguous per access (channel chunks beyond 512 are looped over).
template <int P>
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const DwParams p) {
  __shared__ double dscr[16];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int b = blockIdx.y;
  const int T = p.T;
  const int tw0 = blockIdx.x*DW_TT + wid*DW_RPW;
  const NormStat ns = norm_stat(p.stats1, b, p.inv_n, p.eps);
  const float a1 = *p.slope1, a2 = *p.slope2;
  double s_sum = 0.0, s_sq = 0.0;

  for (int c0 = lane*8; c0 < p.Cp; c0 += 512) {
    float sc[8], sh[8], bs[8], w[P][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = c0 + j;
      const bool ok = c < p.C;
      const float g = ok ? p.gamma1[c] : 0.f;
      sc[j] = ns.rstd*g;
      sh[j] = (ok ? p.beta1[c] : 0.f) - ns.mean*ns.rstd*g;
      bs[j] = ok ? p.bias[c] : 0.f;
#pragma unroll
      for (int k = 0; k < P; ++k) w[k][j] = ok ? p.taps[c*P + k] : 0.f;
    }
    const bf16_t* zin = p.z1 + (long long)b*T*p.Cp + c0;
    bf16_t* zout = p.z2 + (long long)b*T*p.Cp + c0;
#pragma unroll 4
    for (int i = 0; i < DW_RPW; ++i) {
      const int t = tw0 + i;
      if (t >= T) break;
      float acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = bs[j];
#pragma unroll
      for (int k = 0; k < P; ++k) {
        const int ti = t + k*p.dil - p.left;
        if (ti < 0 || ti >= T) continue;          // zero padding of the normalised input
        float f[8];
        unpack8(*reinterpret_cast<const uint4*>(zin + (long long)ti*p.Cp), f);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          acc[j] += w[k][j]*(prelu(f[j], a1)*sc[j] + sh[j]);
      }
      const uint4 q = pack8(acc);
      *reinterpret_cast<uint4*>(zout + (long long)t*p.Cp) = q;
      float r[8]; unpack8(q, r);
      float ls = 0.f, lq = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (c0 + j < p.C) {
          const float pv = prelu(r[j], a2);
          ls += pv; lq += pv*pv;
        }
      }
      s_sum += ls; s_sq += lq;
    }
  }
  const double r0 = block_sum(s_sum, dscr);
  const double r1 = block_sum(s_sq, dscr + 8);
  if (tid == 0) {
    atomic_add_f64(p.stats2 + 2*b, r0);
    atomic_add_f64(p.stats2 + 2*b + 1, r1);
  }
}

// ---------------------------------------------------------------------------
struct DzParams {
  bf16_t* e;               // in: gamma * dy ; out: dz
  const bf16_t* z;
  int B, T, Cp, C;
  const float* slope; const double* stats; const double* sums;
  double inv_n; float eps;
  float* dslope;
};

__global__ __launch_bounds__(256) void dz_kernel(const DzParams p) {
  __shared__ float fscr[8];
  const int tid = threadIdx.x;
  const int cpr = p.Cp/8;                               // chunks per frame
  const long long per_item = (long long)p.T*cpr;
  const int b = blockIdx.y;
  const NormStat ns = norm_stat(p.stats, b, p.inv_n, p.eps);
  const float m1 = (float)(p.sums[2*b]*p.inv_n);
  const float m2 = (float)(p.sums[2*b + 1]*p.inv_n);
  const float a = *p.slope;
  float da = 0.f;
#pragma unroll 4
  for (long long i = (long long)blockIdx.x*256 + tid; i < per_item;
       i += (long long)gridDim.x*256) {
    const int c0 = (int)(i % cpr)*8;
    const long long off = (long long)b*p.T*p.Cp + i*8;
    float ev[8], zv[8], o[8];
    unpack8(*reinterpret_cast<const uint4*>(p.e + off), ev);
    unpack8(*reinterpret_cast<const uint4*>(p.z + off), zv);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool pos = zv[j] > 0.f;
      const float pv = pos ? zv[j] : a*zv[j];
      const float xh = (pv - ns.mean)*ns.rstd;
      const float dh = ns.rstd*(ev[j] - m1 - xh*m2);
      const bool ok = c0 + j < p.C;
      o[j] = ok ? (pos ? dh : a*dh) : 0.f;
      if (ok && !pos) da += dh*zv[j];
    }
    *reinterpret_cast<uint4*>(p.e + off) = pack8(o);
  }
  const float s = block_sum(da, fscr);
  if (tid == 0) atomic_add_f32(p.dslope, s);
}

// ---------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(256) void dwconv_bwd_kernel(const DwParams p) {
  __shared__ float red[4*512];
  __shared__ double dscr[16];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int b = blockIdx.y;
  const int T = p.T;
  const int tw0 = blockIdx.x*DW_TT + wid*DW_RPW;
  const NormStat ns = norm_stat(p.stats1, b, p.inv_n, p.eps);
  const float a1 = *p.slope1;
  double s1 = 0.0, s2 = 0.0;

  for (int cb = 0; cb < p.Cp; cb += 512) {
    const int c0 = cb + lane*8;
    const bool lane_ok = c0 < p.Cp;
    float sc[8], sh[8], gm[8], w[P][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = c0 + j;
      const bool ok = c < p.C;
      gm[j] = ok ? p.gamma1[c] : 0.f;
      sc[j] = ns.rstd*gm[j];
      sh[j] = (ok ? p.beta1[c] : 0.f) - ns.mean*ns.rstd*gm[j];
#pragma unroll
      for (int k = 0; k < P; ++k) w[k][j] = ok ? p.taps[c*P + k] : 0.f;
    }
    float dgam[8], dbet[8], dbia[8], dtap[P][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      dgam[j] = 0.f; dbet[j] = 0.f; dbia[j] = 0.f;
#pragma unroll
      for (int k = 0; k < P; ++k) dtap[k][j] = 0.f;
    }
    const long long base = (long long)b*T*p.Cp + c0;
    if (lane_ok) {
#pragma unroll 2
      for (int i = 0; i < DW_RPW; ++i) {
        const int t = tw0 + i;
        if (t >= T) break;
        float zc[8], dzc[8], dh[8];
        unpack8(*reinterpret_cast<const uint4*>(p.z1 + base + (long long)t*p.Cp), zc);
        unpack8(*reinterpret_cast<const uint4*>(p.dz2 + base + (long long)t*p.Cp), dzc);
#pragma unroll
        for (int j = 0; j < 8; ++j) { dh[j] = 0.f; dbia[j] += dzc[j]; }
#pragma unroll
        for (int k = 0; k < P; ++k) {
          // forward: z2[t'] += w[k] * h1n[t' + k*dil - left]
          const int shift = k*p.dil - p.left;
          const int to = t - shift;               // output frame fed by h1n[t] through tap k
          if (to >= 0 && to < T) {
            float g[8];
            if (shift == 0) {
#pragma unroll
              for (int j = 0; j < 8; ++j) g[j] = dzc[j];
            } else {
              unpack8(*reinterpret_cast<const uint4*>(p.dz2 + base + (long long)to*p.Cp), g);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) dh[j] += w[k][j]*g[j];
          }
          const int ti = t + shift;               // input frame read by output t through tap k
          if (ti >= 0 && ti < T) {
            float f[8];
            if (shift == 0) {
#pragma unroll
              for (int j = 0; j < 8; ++j) f[j] = zc[j];
            } else {
              unpack8(*reinterpret_cast<const uint4*>(p.z1 + base + (long long)ti*p.Cp), f);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
              dtap[k][j] += dzc[j]*(prelu(f[j], a1)*sc[j] + sh[j]);
          }
        }
        float o[8];
        float l1 = 0.f, l2 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xh = (prelu(zc[j], a1) - ns.mean)*ns.rstd;
          const float ev = gm[j]*dh[j];
          o[j] = ev;
          l1 += ev; l2 += ev*xh;
          dgam[j] += dh[j]*xh; dbet[j] += dh[j];
        }
        s1 += l1; s2 += l2;
        *reinterpret_cast<uint4*>(p.e1 + base + (long long)t*p.Cp) = pack8(o);
      }
    }
    // per-channel reductions over the 4 waves, then one atomic per channel and quantity
    auto reduce_cols = [&](const float (&v)[8], float* dst, int stride, int offset) {
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 8; ++j) red[wid*512 + lane*8 + j] = v[j];
      __syncthreads();
      for (int cc = tid; cc < 512; cc += 256) {
        const float sum = red[cc] + red[512 + cc] + red[1024 + cc] + red[1536 + cc];
        const int c = cb + cc;
        if (c < p.C) atomic_add_f32(dst + (long long)c*stride + offset, sum);
      }
    };
    reduce_cols(dgam, p.dgamma1, 1, 0);
    reduce_cols(dbet, p.dbeta1, 1, 0);
    reduce_cols(dbia, p.dbias, 1, 0);
#pragma unroll
    for (int k = 0; k < P; ++k) reduce_cols(dtap[k], p.dtaps, P, k);
  }
  const double r0 = block_sum(s1, dscr);
  const double r1 = block_sum(s2, dscr + 8);
  if (tid == 0) {
    atomic_add_f64(p.sums1 + 2*b, r0);
    atomic_add_f64(p.sums1 + 2*b + 1, r1);
  }
}

// ---------------------------------------------------------------------------
struct CombineParams {
  const bf16_t* e0;        // gamma0 * d(norm output)            [B][T][Cp]
  const bf16_t* w;         // encoder output                     [B][T][Cp]
  const bf16_t* dw1;       // mask-path gradient                 [B*S][T][Cp]
  bf16_t* dw;              // total gradient wrt encoder output  [B][T][Cp]
  int B, T, Cp, C, S;
  const double* stats; const double* sums; double inv_n; float eps;
};

__global__ __launch_bounds__(256) void gln0_bwd_combine_kernel(const CombineParams p) {
  const int tid = threadIdx.x;
  const int cpr = p.Cp/8;
  const long long per_item = (long long)p.T*cpr;
  const int b = blockIdx.y;
  const NormStat ns = norm_stat(p.stats, b, p.inv_n, p.eps);
  const float m1 = (float)(p.sums[2*b]*p.inv_n);
  const float m2 = (float)(p.sums[2*b + 1]*p.inv_n);
  for (long long i = (long long)blockIdx.x*256 + tid; i < per_item;
       i += (long long)gridDim.x*256) {
    const int c0 = (int)(i % cpr)*8;
    const long long off = (long long)b*p.T*p.Cp + i*8;
    float ev[8], wv[8], o[8];
    unpack8(*reinterpret_cast<const uint4*>(p.e0 + off), ev);
    unpack8(*reinterpret_cast<const uint4*>(p.w + off), wv);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xh = (wv[j] - ns.mean)*ns.rstd;
      o[j] = (c0 + j < p.C) ? ns.rstd*(ev[j] - m1 - xh*m2) : 0.f;
    }
    for (int s = 0; s < p.S; ++s) {
      float d[8];
      unpack8(*reinterpret_cast<const uint4*>(
                  p.dw1 + ((long long)(b*p.S + s)*p.T)*p.Cp + i*8), d);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] += d[j];
    }
    *reinterpret_cast<uint4*>(p.dw + off) = pack8(o);
  }
}

}  // namespace brv
