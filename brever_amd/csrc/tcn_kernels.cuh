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
  long long rep_stride;    // bwd: per-channel gradients go to replica (wg % kReplicas)
  // bwd, fused gLN_2 / PReLU_2 backward (dwconv_bwd_halo_kernel): `dz2` then carries
  // e2 = gamma2*dy and dz2 is built from it and z2 in LDS; it never exists in HBM
  const bf16_t* z2in; const double* sums2; float* dslope2;
};

// Per-channel gradients (norm gains/biases, depthwise taps/biases) are summed over
// every frame of the batch. Adding them from thousands of workgroups into a few KB
// is the "all adders on one row" case of the float-atomic unit (an order of magnitude
// below its rate), so workgroups add into one of kReplicas copies of the vector-
// gradient block and vgrad_reduce_kernel folds the copies into the real gradient.
constexpr int kReplicas = 64;

#ifndef BRV_DW_NB
#define BRV_DW_NB 4
#endif
constexpr int DW_NB = BRV_DW_NB;   // rows whose tap loads are in flight together (forward)
constexpr int DW_TT_F = 32;    // forward: frames per workgroup (4 waves x 8 frames; 64 measured slower)
constexpr int DW_TT_B = 128;   // backward: 4 waves x 32 frames (64: more per-channel atomics; 256: too few workgroups)

// Workgroup id -> (item, frame tile): the work list is cut into 8 contiguous chunks,
// one per XCD (ids congruent mod 8 share an XCD), so the tiles whose dilated taps
// overlap are neighbours in time on one L2 instead of being fetched by 3 XCDs.
__device__ __forceinline__ int dw_xcd_remap(int id, int nwg) {
  const int xcd = id & 7, slot = id >> 3;
  const int q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd*(q + 1) : r*(q + 1) + (xcd - r)*q) + slot;
}

// One wavefront handles whole frames: 64 lanes x 8 channels = 512 channels = 1 KiB
// contiguous per access (channel chunks beyond 512 are looped over).
template <int P>
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const DwParams p) {
  __shared__ double dscr[16];
  __shared__ float wcs[P + 1][512];        // per-tap constant terms (+ bias) of 512 channels
  constexpr int DW_RPW = DW_TT_F/4;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int T = p.T;
  const int n_tt = ceil_div(T, DW_TT_F);
  const int vid = dw_xcd_remap(blockIdx.x, gridDim.x);
  const int b = vid / n_tt;
  const int tw0 = (vid % n_tt)*DW_TT_F + wid*DW_RPW;
  const NormStat ns = norm_stat(p.stats1, b, p.inv_n, p.eps);
  const float a1 = *p.slope1, a2 = *p.slope2;
  double s_sum = 0.0, s_sq = 0.0;

  // PReLU_1 + gLN_1 + tap weight folded per channel:
  //   w_k*(gamma*rstd*prelu(z) + beta - mean*rstd*gamma) = wa_k*z + wb_k*|z| + wc_k
  // (prelu(z) = c1*z + c2*|z|): two FMAs per tap and element, |z| is a source modifier.
  // Taps outside [0, T) read zeros through the item's buffer descriptor, which removes
  // the z terms; the constant terms of the valid taps are summed per frame (uniform).
  const float c1 = 0.5f*(1.f + a1), c2 = 0.5f*(1.f - a1);
  const float d1 = 0.5f*(1.f + a2), d2 = 0.5f*(1.f - a2);     // statistics of PReLU_2
  for (int c0 = lane*8; c0 < p.Cp; c0 += 512) {
    // constant terms: `call` = bias + all taps (interior frames, in registers); the
    // per-tap values are needed only at the item's edges and wait in LDS
    float call[8], wa[P][8], wb[P][8];
    __syncthreads();                                       // previous chunk's readers done
    {
      float bs[8];
      float g8[8], b8[8], tp[P][8];
      load8_masked(p.gamma1, c0, p.C, g8);
      load8_masked(p.beta1, c0, p.C, b8);
      load8_masked(p.bias, c0, p.C, bs);
      // taps are [C][P]: P chunks of 8 consecutive floats cover this lane's 8 channels
#pragma unroll
      for (int k = 0; k < P; ++k) load8_masked(p.taps, c0*P + 8*k, p.C*P, tp[k]);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float sc = ns.rstd*g8[j];
        const float sh = b8[j] - ns.mean*ns.rstd*g8[j];
        call[j] = bs[j];
#pragma unroll
        for (int k = 0; k < P; ++k) {
          const float w = tp[(j*P + k)/8][(j*P + k)%8];
          wa[k][j] = w*c1*sc; wb[k][j] = w*c2*sc;
          call[j] += w*sh;
          if (wid == 0) wcs[k][lane*8 + j] = w*sh;
        }
        if (wid == 0) wcs[P][lane*8 + j] = bs[j];
      }
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(p.z1 + (long long)b*T*p.Cp, (long long)T*p.Cp*2);
    const __amdgpu_buffer_rsrc_t rout = make_rsrc(p.z2 + (long long)b*T*p.Cp, (long long)T*p.Cp*2);
    const unsigned int coff = (unsigned int)(c0*2);
    const unsigned int row = (unsigned int)(p.Cp*2);
#pragma unroll 1
    for (int i0 = 0; i0 < DW_RPW; i0 += DW_NB) {
      uint4 raw[DW_NB][P];
#pragma unroll
      for (int u = 0; u < DW_NB; ++u) {
        const int t = tw0 + i0 + u;
#pragma unroll
        for (int k = 0; k < P; ++k) {
          const int ti = t + k*p.dil - p.left;            // < 0 wraps to a huge offset: zeros
          raw[u][k] = buf_load16(rin, (unsigned int)ti*row + coff);
        }
      }
      float ls = 0.f, lq = 0.f;
#pragma unroll
      for (int u = 0; u < DW_NB; ++u) {
        const int t = tw0 + i0 + u;
        float acc[8];
        if (t - p.left >= 0 && t + (P - 1)*p.dil - p.left < T) {     // wave-uniform
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] = call[j];
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] = wcs[P][lane*8 + j];
#pragma unroll
          for (int k = 0; k < P; ++k) {
            const int ti = t + k*p.dil - p.left;
            if (ti >= 0 && ti < T) {
#pragma unroll
              for (int j = 0; j < 8; ++j) acc[j] += wcs[k][lane*8 + j];
            }
          }
        }
#pragma unroll
        for (int k = 0; k < P; ++k) {
          float f[8];
          unpack8(raw[u][k], f);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            acc[j] = __builtin_fmaf(wa[k][j], f[j], acc[j]);
            acc[j] = __builtin_fmaf(wb[k][j], __builtin_fabsf(f[j]), acc[j]);
          }
        }
        const uint4 q = pack8(acc);
        buf_store16(rout, (unsigned int)t*row + coff, q);      // t >= T: dropped
        float r[8]; unpack8(q, r);
        const float live = t < T ? 1.f : 0.f;
        float fs = 0.f, fq = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          // padded channels have zero taps, bias and affine: r == 0 there, no mask needed
          const float pv = __builtin_fmaf(d2, __builtin_fabsf(r[j]), d1*r[j]);
          fs += pv; fq = __builtin_fmaf(pv, pv, fq);
        }
        ls = __builtin_fmaf(live, fs, ls); lq = __builtin_fmaf(live, fq, lq);
      }
      s_sum += ls; s_sq += lq;
    }
  }
  const double r0 = block_sum(s_sum, dscr);
  const double r1 = block_sum(s_sq, dscr + 8);
  if (tid == 0) {
    atomic_add_f64(p.stats2 + stat_sum(b), r0);
    atomic_add_f64(p.stats2 + stat_sq(b), r1);
  }
}

// ---------------------------------------------------------------------------
struct DzParams {
  bf16_t* e;               // in: gamma * dy ; out: dz
  const bf16_t* z;
  int B, T, Cp, C;
  const float* slope; const double* stats; const double* sums;
  double inv_n; float eps;
  float* dslope;           // replicated (kReplicas copies rep_stride floats apart)
  long long rep_stride;
};

__global__ __launch_bounds__(256) void dz_kernel(const DzParams p) {
  __shared__ float fscr[8];
  const int tid = threadIdx.x;
  const int cpr = p.Cp/8;                               // chunks per frame
  const long long per_item = (long long)p.T*cpr;
  const int b = blockIdx.y;
  const NormStat ns = norm_stat(p.stats, b, p.inv_n, p.eps);
  const float m1 = (float)(p.sums[stat_sum(b)]*p.inv_n);
  const float m2 = (float)(p.sums[stat_sq(b)]*p.inv_n);
  const float a = *p.slope;
  float da = 0.f;
#pragma unroll 1
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
  // one copy per 128-byte line: same-line atomics serialise (~12 ns each)
  if (tid == 0) atomic_add_f32(p.dslope + (long long)(blockIdx.x % kReplicas)*p.rep_stride, s);
}

// ---------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(256) void dwconv_bwd_kernel(const DwParams p) {
  __shared__ float red[4*512];
  __shared__ double dscr[16];
  constexpr int DW_RPW = DW_TT_B/4;
  constexpr int UB = 2;                    // frames in flight per wave
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int T = p.T;
  const int n_tt = ceil_div(T, DW_TT_B);
  const int vid = dw_xcd_remap(blockIdx.x, gridDim.x);
  const int b = vid / n_tt;
  const int tw0 = (vid % n_tt)*DW_TT_B + wid*DW_RPW;
  const NormStat ns = norm_stat(p.stats1, b, p.inv_n, p.eps);
  const float a1 = *p.slope1;
  double s1 = 0.0, s2 = 0.0;

  // gLN_1 input statistics as scalars: xh = (prelu(z) - mean)*rstd = xa*z + xb*|z| + xc
  const float xa = 0.5f*(1.f + a1)*ns.rstd, xb = 0.5f*(1.f - a1)*ns.rstd, xc = -ns.mean*ns.rstd;
  // tap whose output frame is the input frame itself (the centre tap of a "same" padding):
  // its dz2 doubles as the centre value of the bias gradient
  int kc = -1;
#pragma unroll
  for (int k = 0; k < P; ++k) if (k*p.dil == p.left) kc = k;

  for (int cb = 0; cb < p.Cp; cb += 512) {
    const int c0 = cb + lane*8;
    const bool lane_ok = c0 < p.Cp;
    f32x2 gm[4], be[4], w[P][4];
    {
      float g8[8], b8[8], tp[P][8];
      load8_masked(p.gamma1, c0, p.C, g8);
      load8_masked(p.beta1, c0, p.C, b8);
#pragma unroll
      for (int k = 0; k < P; ++k) load8_masked(p.taps, c0*P + 8*k, p.C*P, tp[k]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        gm[j] = f32x2{g8[2*j], g8[2*j + 1]};
        be[j] = f32x2{b8[2*j], b8[2*j + 1]};
#pragma unroll
        for (int k = 0; k < P; ++k)
          w[k][j] = f32x2{tp[(2*j*P + k)/8][(2*j*P + k)%8], tp[((2*j + 1)*P + k)/8][((2*j + 1)*P + k)%8]};
      }
    }
    f32x2 dgam[4], dbet[4], dbia[4], dtap[P][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      dgam[j] = f32x2{0.f, 0.f}; dbet[j] = f32x2{0.f, 0.f}; dbia[j] = f32x2{0.f, 0.f};
#pragma unroll
      for (int k = 0; k < P; ++k) dtap[k][j] = f32x2{0.f, 0.f};
    }
    // frames outside [0, T) read zeros through the item's descriptors; a frame t >= T gets
    // out-of-range offsets for ALL its loads, so every contribution of that frame vanishes
    const __amdgpu_buffer_rsrc_t rdz = make_rsrc(p.dz2 + (long long)b*T*p.Cp, lane_ok ? (long long)T*p.Cp*2 : 0);
    const __amdgpu_buffer_rsrc_t rz1 = make_rsrc(p.z1 + (long long)b*T*p.Cp, lane_ok ? (long long)T*p.Cp*2 : 0);
    const __amdgpu_buffer_rsrc_t re1 = make_rsrc(p.e1 + (long long)b*T*p.Cp, lane_ok ? (long long)T*p.Cp*2 : 0);
    const unsigned int coff = (unsigned int)(c0*2), row = (unsigned int)(p.Cp*2);
    float l1 = 0.f, l2 = 0.f;
#pragma unroll 1
    for (int i0 = 0; i0 < DW_RPW; i0 += UB) {
      // raw[u][k]: dz2 at the output frame that reads frame t through tap k
      // (forward: z2[t'] += w[k]*h1n[t' + k*dil - left], so t' = t - k*dil + left).
      // The same values serve the data gradient (sum_k w[k]*dz2[t']) and, paired with
      // h1n at frame t itself, the tap gradient: sum_t' dz2[t']*h1n[t'+shift] re-indexed
      // over t = t'+shift. No shifted read of z1 is needed.
      uint4 raw[UB][P], rawz[UB], rawc[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int t = tw0 + i0 + u;
        const bool live = t < T;
#pragma unroll
        for (int k = 0; k < P; ++k) {
          const int to = t - (k*p.dil - p.left);
          const unsigned int off = live ? (unsigned int)to*row + coff : kOob;
          raw[u][k] = buf_load16(rdz, off);
        }
        rawz[u] = buf_load16(rz1, (unsigned int)t*row + coff);
        rawc[u] = buf_load16(rdz, kc < 0 ? (unsigned int)t*row + coff : kOob);
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int t = tw0 + i0 + u;
        float zc[8];
        unpack8(rawz[u], zc);
        f32x2 xh[4], hn[4], dh[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          xh[j].x = __builtin_fmaf(xb, __builtin_fabsf(zc[2*j]), __builtin_fmaf(xa, zc[2*j], xc));
          xh[j].y = __builtin_fmaf(xb, __builtin_fabsf(zc[2*j + 1]), __builtin_fmaf(xa, zc[2*j + 1], xc));
          hn[j] = gm[j]*xh[j] + be[j];                  // gLN_1 output at frame t
          dh[j] = f32x2{0.f, 0.f};
        }
#pragma unroll
        for (int k = 0; k < P; ++k) {
          float g[8];
          unpack8(raw[u][k], g);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x2 gk = {g[2*j], g[2*j + 1]};
            dh[j] += w[k][j]*gk;
            dtap[k][j] += gk*hn[j];
            if (k == kc) dbia[j] += gk;                 // wave-uniform
          }
        }
        if (kc < 0) {
          float g[8];
          unpack8(rawc[u], g);
#pragma unroll
          for (int j = 0; j < 4; ++j) dbia[j] += f32x2{g[2*j], g[2*j + 1]};
        }
        f32x2 o[4], a1s = {0.f, 0.f}, a2s = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x2 ev = gm[j]*dh[j];
          o[j] = ev;
          a1s += ev; a2s += ev*xh[j];
          dgam[j] += dh[j]*xh[j]; dbet[j] += dh[j];
        }
        l1 += a1s.x + a1s.y; l2 += a2s.x + a2s.y;
        buf_store16(re1, (unsigned int)t*row + coff, pack8v(o));     // t >= T: dropped
      }
    }
    s1 += l1; s2 += l2;
    // per-channel reductions over the 4 waves, then one atomic per channel and quantity
    auto reduce_cols = [&](const f32x2 (&v)[4], float* dst, int stride, int offset) {
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        red[wid*512 + lane*8 + 2*j] = v[j].x; red[wid*512 + lane*8 + 2*j + 1] = v[j].y;
      }
      __syncthreads();
      float* rdst = dst + (long long)(blockIdx.x % kReplicas)*p.rep_stride;
      for (int cc = tid; cc < 512; cc += 256) {
        const float sum = red[cc] + red[512 + cc] + red[1024 + cc] + red[1536 + cc];
        const int c = cb + cc;
        if (c < p.C) atomic_add_f32(rdst + (long long)c*stride + offset, sum);
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
    atomic_add_f64(p.sums1 + stat_sum(b), r0);
    atomic_add_f64(p.sums1 + stat_sq(b), r1);
  }
}

// ---------------------------------------------------------------------------
// Depthwise-conv backward with the gLN_2 / PReLU_2 backward folded in, each dz2 element
// computed ONCE: a workgroup owns (item, HL_TT frames, HL_CG channels), first builds
// dz2 = prelu2'(z2)*rstd2*(e2 - m1 - xh2*m2) for its frames plus the halo the taps reach into
// LDS (bf16), then runs the transposed stencil out of LDS. The dz2 tensor never exists in HBM:
// the separate gln_prelu_bwd pass (read e2, z2; write dz2) and this kernel's re-read of dz2
// become one read of e2 and z2 with a halo overhead of (P-1)*R/HL_TT <= 6 % (comb tiles, below).
// Replaces dz_kernel + dwconv_bwd_kernel for gLN_2; outputs and parameter gradients are the same.
constexpr int HL_TT = 256;                 // frames per workgroup
constexpr int HL_CG = 64;                  // channels per workgroup (128-byte rows)
constexpr int HL_RMAX = 8;                 // consecutive frames per comb tooth
// Tile shape. The stencil only couples frames that are `dil` apart, so a tile is a COMB:
// R consecutive frames (residues r0 .. r0+R-1 modulo dil) x K teeth dil apart, R*K = HL_TT; its
// window adds P-1 teeth, i.e. (P-1)*R halo rows whatever the dilation. (A contiguous 256-frame
// tile needs (P-1)*dil halo rows: 2x the tile at dilation 128, which doubled phase 1 and cut the
// residency to 2 workgroups per CU for the widest layers.) Rows of 64 channels are separate
// 128-byte segments in HBM anyway, so non-contiguous frames cost nothing. For dil <= HL_RMAX the
// comb IS the contiguous tile. Needs left % dil == 0 (odd kernel sizes, or causal padding).
__host__ __device__ inline int hl_rows_per_tooth(int dil) {
  int r = 1;
  while (2*r <= dil && 2*r <= HL_RMAX) r *= 2;
  return r;
}
__host__ __device__ inline int hl_window_rows(int dil, int P) {
  const int R = hl_rows_per_tooth(dil);
  return (HL_TT/R + P - 1)*R;
}
constexpr int HL_MAXROWS = HL_TT + 4*HL_RMAX;   // window rows for kernel sizes up to 5

template <int P>
__global__ __launch_bounds__(256) void dwconv_bwd_halo_kernel(const DwParams p) {
  // window of (K + P - 1) teeth x R rows x 128 B
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
  bf16_t* win = reinterpret_cast<bf16_t*>(dyn_lds);
  __shared__ float red[32*HL_CG];
  __shared__ double dscr[16];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int T = p.T, d = p.dil;
  const int R = hl_rows_per_tooth(d), K = HL_TT/R;
  const int n_rt = ceil_div(d, R);                       // residue groups
  const int n_teeth = (T - 1)/d + 1;
  const int n_qt = ceil_div(n_teeth, K);                 // tooth groups
  const int n_tt = n_rt*n_qt, n_cg = p.Cp/HL_CG;
  int id = blockIdx.x;
  const int cg = id % n_cg; id /= n_cg;
  const int b = id / n_tt;
  const int tile = id % n_tt;
  const int r0 = (tile % n_rt)*R, q0 = (tile / n_rt)*K;
  const int cl = (tid & 7)*8;                          // channel offset inside the group
  const int c0 = cg*HL_CG + cl;
  const int rslot = tid >> 3;                          // 32 row slots per pass
  const int W = (K + P - 1)*R;                         // rows of the window
  const int qbase = q0 - (P - 1) + p.left/d;           // tooth of window row 0

  // ---- phase 1: dz2 of the window -> LDS ------------------------------------------------
  const NormStat n2 = norm_stat(p.stats2, b, p.inv_n, p.eps);
  const float a2 = *p.slope2;
  const float m1 = (float)(p.sums2[stat_sum(b)]*p.inv_n);
  const float m2 = (float)(p.sums2[stat_sq(b)]*p.inv_n);
  const float ya = 0.5f*(1.f + a2)*n2.rstd, yb = 0.5f*(1.f - a2)*n2.rstd, yc = -n2.mean*n2.rstd;
  const float R2 = n2.rstd, K0 = -m1*n2.rstd, M2R = -m2*n2.rstd;
  const __amdgpu_buffer_rsrc_t re2 = make_rsrc(p.dz2 + (long long)b*T*p.Cp, (long long)T*p.Cp*2);
  const __amdgpu_buffer_rsrc_t rz2 = make_rsrc(p.z2in + (long long)b*T*p.Cp, (long long)T*p.Cp*2);
  const unsigned int row = (unsigned int)(p.Cp*2), coff = (unsigned int)(c0*2);
  float da2 = 0.f;
  f32x2 dbia[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) dbia[j] = f32x2{0.f, 0.f};
  // window row r -> frame: tooth qbase + r / R, residue r0 + r % R
  auto frame_of = [&](int r, bool& ok) {
    const int qi = r / R, ri = r % R;
    const int q = qbase + qi;
    ok = r < W && q >= 0 && r0 + ri < d;
    return q*d + r0 + ri;
  };
  // four rows per thread in flight (8 loads) before any of them is consumed
  for (int rw = rslot; rw < W; rw += 128) {
    uint4 qe[4], qz[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      bool ok;
      const int tf = frame_of(rw + 32*u, ok);
      const bool in = ok && tf < T;
      const unsigned int off = in ? (unsigned int)tf*row + coff : kOob;
      qe[u] = buf_load16(re2, off); qz[u] = buf_load16(rz2, off);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = rw + 32*u;
      if (r >= W) break;
      bool ok;
      const int tf = frame_of(r, ok);
      const bool in = ok && tf < T;
      float e[8], z[8], g[8];
      unpack8(qe[u], e); unpack8(qz[u], z);
      const float on = in ? 1.f : 0.f;
      const float rr = on*R2, k0 = on*K0, mm = on*M2R;
      // each element is "owned" by the tile whose teeth [q0, q0 + K) contain it
      const int q = qbase + r / R;
      const bool centre = in && q >= q0 && q < q0 + K;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xh2 = __builtin_fmaf(yb, __builtin_fabsf(z[j]), __builtin_fmaf(ya, z[j], yc));
        const float uu = __builtin_fmaf(mm, xh2, __builtin_fmaf(e[j], rr, k0));
        const bool pos = z[j] > 0.f;
        g[j] = c0 + j < p.C ? (pos ? uu : a2*uu) : 0.f;
        if (centre && !pos) da2 += uu*z[j];
      }
      const uint4 q4 = pack8(g);
      *reinterpret_cast<uint4*>(win + r*HL_CG + cl) = q4;
      if (centre) {                                      // bias gradient = sum of (rounded) dz2
        float gr[8]; unpack8(q4, gr);
#pragma unroll
        for (int j = 0; j < 4; ++j) dbia[j] += f32x2{gr[2*j], gr[2*j + 1]};
      }
    }
  }
  __syncthreads();

  // ---- phase 2: transposed stencil out of LDS -------------------------------------------------
  const NormStat ns = norm_stat(p.stats1, b, p.inv_n, p.eps);
  const float a1 = *p.slope1;
  const float xa = 0.5f*(1.f + a1)*ns.rstd, xb = 0.5f*(1.f - a1)*ns.rstd, xc = -ns.mean*ns.rstd;
  f32x2 gm[4], be[4], w[P][4];
  {
    float g8[8], b8[8], tp[P][8];
    load8_masked(p.gamma1, c0, p.C, g8);
    load8_masked(p.beta1, c0, p.C, b8);
#pragma unroll
    for (int k = 0; k < P; ++k) load8_masked(p.taps, c0*P + 8*k, p.C*P, tp[k]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      gm[j] = f32x2{g8[2*j], g8[2*j + 1]};
      be[j] = f32x2{b8[2*j], b8[2*j + 1]};
#pragma unroll
      for (int k = 0; k < P; ++k)
        w[k][j] = f32x2{tp[(2*j*P + k)/8][(2*j*P + k)%8], tp[((2*j + 1)*P + k)/8][((2*j + 1)*P + k)%8]};
    }
  }
  f32x2 dgam[4], dbet[4], dtap[P][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    dgam[j] = f32x2{0.f, 0.f}; dbet[j] = f32x2{0.f, 0.f};
#pragma unroll
    for (int k = 0; k < P; ++k) dtap[k][j] = f32x2{0.f, 0.f};
  }
  const __amdgpu_buffer_rsrc_t rz1 = make_rsrc(p.z1 + (long long)b*T*p.Cp, (long long)T*p.Cp*2);
  const __amdgpu_buffer_rsrc_t re1 = make_rsrc(p.e1 + (long long)b*T*p.Cp, (long long)T*p.Cp*2);
  float l1 = 0.f, l2 = 0.f;
  // output row i -> frame: tooth q0 + i / R, residue r0 + i % R
  auto out_frame = [&](int i, bool& ok) {
    const int ro = r0 + i % R;
    ok = ro < d;
    return (q0 + i / R)*d + ro;
  };
  for (int i0 = rslot; i0 < HL_TT; i0 += 128) {
   uint4 qz4[4];
#pragma unroll
   for (int u = 0; u < 4; ++u) {
     bool ok;
     const int t = out_frame(i0 + 32*u, ok);
     qz4[u] = buf_load16(rz1, (ok && t < T) ? (unsigned int)t*row + coff : kOob);   // outside: zeros
   }
#pragma unroll
   for (int u = 0; u < 4; ++u) {
    const int i = i0 + 32*u;
    bool ok;
    const int t = out_frame(i, ok);
    const uint4 qz = qz4[u];
    const float live = (ok && t < T) ? 1.f : 0.f;
    float zc[8];
    unpack8(qz, zc);
    f32x2 xh[4], hn[4], dh[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      xh[j].x = __builtin_fmaf(xb, __builtin_fabsf(zc[2*j]), __builtin_fmaf(xa, zc[2*j], xc));
      xh[j].y = __builtin_fmaf(xb, __builtin_fabsf(zc[2*j + 1]), __builtin_fmaf(xa, zc[2*j + 1], xc));
      hn[j] = f32x2{live, live}*(gm[j]*xh[j] + be[j]);   // gLN_1 output at frame t (0 past the end)
      dh[j] = f32x2{0.f, 0.f};
    }
#pragma unroll
    for (int k = 0; k < P; ++k) {
      // output frame that reads frame t through tap k: tooth + left/dil - k = window tooth
      // (i / R) + P - 1 - k, same residue
      const int r = (i / R + P - 1 - k)*R + i % R;
      float g[8];
      unpack8(*reinterpret_cast<const uint4*>(win + r*HL_CG + cl), g);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x2 gk = {g[2*j], g[2*j + 1]};
        dh[j] += w[k][j]*gk;
        dtap[k][j] += gk*hn[j];
      }
    }
    f32x2 o[4], a1s = {0.f, 0.f}, a2s = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x2 dl = f32x2{live, live}*dh[j];          // frames past the end contribute nothing
      const f32x2 ev = gm[j]*dl;
      o[j] = ev;
      a1s += ev; a2s += ev*xh[j];
      dgam[j] += dl*xh[j]; dbet[j] += dl;
    }
    l1 += a1s.x + a1s.y; l2 += a2s.x + a2s.y;
    buf_store16(re1, live != 0.f ? (unsigned int)t*row + coff : kOob, pack8v(o));   // outside: dropped
   }
  }

  // ---- reductions: 32 row slots share each channel chunk ----------------------------------------
  auto reduce_cols = [&](const f32x2 (&v)[4], float* dst, int stride, int offset) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      red[rslot*HL_CG + cl + 2*j] = v[j].x; red[rslot*HL_CG + cl + 2*j + 1] = v[j].y;
    }
    __syncthreads();
    if (tid < HL_CG) {
      float sum = 0.f;
#pragma unroll 8
      for (int r = 0; r < 32; ++r) sum += red[r*HL_CG + tid];
      const int c = cg*HL_CG + tid;
      float* rdst = dst + (long long)(blockIdx.x % kReplicas)*p.rep_stride;
      if (c < p.C) atomic_add_f32(rdst + (long long)c*stride + offset, sum);
    }
  };
  reduce_cols(dgam, p.dgamma1, 1, 0);
  reduce_cols(dbet, p.dbeta1, 1, 0);
  reduce_cols(dbia, p.dbias, 1, 0);
#pragma unroll
  for (int k = 0; k < P; ++k) reduce_cols(dtap[k], p.dtaps, P, k);
  const double tot1 = block_sum((double)l1, dscr);
  const double tot2 = block_sum((double)l2, dscr + 8);
  if (tid == 0) {
    atomic_add_f64(p.sums1 + stat_sum(b), tot1);
    atomic_add_f64(p.sums1 + stat_sq(b), tot2);
  }
  __syncthreads();
  const float sa = block_sum(da2, red);
  if (tid == 0) atomic_add_f32(p.dslope2 + (long long)(blockIdx.x % kReplicas)*p.rep_stride, sa);
}

// ---------------------------------------------------------------------------
// Vector-gradient block layout (floats): [ln_g N][ln_b N] then per TCN block
// [n1_g H][n1_b H][n2_g H][n2_b H][dconv_w H*P][dconv_b H].
struct VgradParams {
  const float* vg; float* grads; long long rep_stride;
  int N, H, P, nb;
  long long ln_g_off;        // flat offset of tcn.layer_norm.weight
  long long blk0_off;        // flat offset of block 0's conv.weight
  long long blk_full;        // floats per (non-last) block
  long long o_dconv_w, o_dconv_b, o_n1_g_full, o_n1_g_last;   // offsets inside a block
  long long tcn_prelu_off;   // PReLU slopes: slot 0 = tcn, 1+2i / 2+2i = block i (right after n2_b)
  // partial folds (backward in parts, gradient buckets): only blocks [blk_lo, blk_hi], the first
  // layer norm iff do_ln, the TCN output PReLU slope iff do_tcn
  int blk_lo, blk_hi, do_ln, do_tcn;
};
__global__ __launch_bounds__(256) void vgrad_reduce_kernel(const VgradParams p) {
  const long long per_blk = (long long)p.H*(5 + p.P);
  const long long n_vec = 2LL*p.N + per_blk*p.nb;
  const long long total = n_vec + 1 + 2*p.nb;
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < total;
       i += (long long)gridDim.x*256) {
    long long dst;
    if (i >= n_vec) {                              // PReLU slopes
      const int k = (int)(i - n_vec);
      if (k == 0) { if (!p.do_tcn) continue; dst = p.tcn_prelu_off; }
      else {                                       // the two slopes follow n2_b
        const int blk = (k - 1) >> 1;
        if (blk < p.blk_lo || blk > p.blk_hi) continue;
        const long long n1 = blk == p.nb - 1 ? p.o_n1_g_last : p.o_n1_g_full;
        dst = p.blk0_off + blk*p.blk_full + n1 + 4LL*p.H + ((k - 1) & 1);
      }
    } else if (i < 2LL*p.N) {
      if (!p.do_ln) continue;
      dst = p.ln_g_off + i;                        // ln_g then ln_b are adjacent
    } else {
      const long long j = i - 2LL*p.N;
      const int blk = (int)(j / per_blk);
      if (blk < p.blk_lo || blk > p.blk_hi) continue;
      const long long k = j % per_blk;
      const long long base = p.blk0_off + blk*p.blk_full;
      const long long n1 = blk == p.nb - 1 ? p.o_n1_g_last : p.o_n1_g_full;
      if (k < 4LL*p.H) dst = base + n1 + k;         // n1_g n1_b n2_g n2_b adjacent
      else if (k < 4LL*p.H + (long long)p.H*p.P) dst = base + p.o_dconv_w + (k - 4LL*p.H);
      else dst = base + p.o_dconv_b + (k - 4LL*p.H - (long long)p.H*p.P);
    }
    float s = 0.f;
#pragma unroll 8
    for (int r = 0; r < kReplicas; ++r) s += p.vg[(long long)r*p.rep_stride + i];
    p.grads[dst] += s;
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
  const float m1 = (float)(p.sums[stat_sum(b)]*p.inv_n);
  const float m2 = (float)(p.sums[stat_sq(b)]*p.inv_n);
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
