// Fused streaming kernels of the fp32 Conv-TasNet path (included by ctn_f32.hip).
//
// The plain path runs one kernel per mathematical step (frame sums, normalise, stencil, column
// sums ...: 4.5 GB of traffic per TCN block at 16 x 4 s). These kernels do the same arithmetic in
// the order of the data instead (reference: brever/models/convtasnet/convtasnet.py:225-262, the
// block `conv -> prelu -> norm -> dconv -> prelu -> norm -> res / skip`):
//   * the normalised tensor of a layer norm is never stored: its consumers (the depthwise
//     stencil here, the 1x1 convolutions through the operand transform of gemm_f32_big.hip)
//     apply `(prelu(z) - mean_t) rstd_t gain + bias` to z as they load it;
//   * every per-frame sum (layer-norm statistics forward, A_t / B_t backward) and every
//     per-channel sum (norm gain / bias, depthwise taps / bias, 1x1 bias gradients) is taken by
//     the kernel that has the operands in registers anyway.
// Layout of all of them: a wavefront owns a frame at a time -- lane l holds channels
// 4 (l + 64 j) .. + 3 (j < NJ, 16-byte accesses) -- so a per-frame sum is one wavefront reduction
// and a per-channel sum stays in the lane's registers over the frames of its slice. A workgroup
// (4 wavefronts) walks kFusedRows consecutive frames; its per-channel sums are added in wavefront
// order and written to `part[slice][quantity][channel]`, the slices are added in order by
// f32_fold_kernel: no atomics anywhere, results are bitwise repeatable.
#pragma once

constexpr int kFusedRows = 64;      // frames per workgroup
#ifndef BRV_DWB_ABL                 // ablations of the fp32 stencil backward (timing experiments): 1 no tap-gradient
#define BRV_DWB_ABL 0               // arithmetic, 2 no shifted dz2 loads, 4 no shifted z1 loads, 8 no wavefront sums
#endif
constexpr int kMaxNJ = 4;           // channels <= 1024

__device__ __forceinline__ float4 ld4u(const float* p) { return make_float4(p[0], p[1], p[2], p[3]); }
__device__ __forceinline__ float4 f4(float v) { return make_float4(v, v, v, v); }
__device__ __forceinline__ float sum4(const float4& v) { return (v.x + v.y) + (v.z + v.w); }

// p = prelu(z), xhat = (p - mean) rstd
__device__ __forceinline__ float prelu1(float v, float a, bool act) { return (v > 0.f || !act) ? v : a*v; }

// Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8): the slice a workgroup takes is chosen
// so that every XCD walks ONE contiguous range of frames in order -- the frames t +- k dil a stencil
// reads besides its own slice were then read (or are about to be) by the same XCD's L2 (stencil forward
// 73.8 -> 70 us; the backward stencil is bound by its per-frame latency chain, not by traffic).
__device__ __forceinline__ int fused_slice() {
  const int nwg = gridDim.x, id = blockIdx.x;
  const int xcd = id & 7, slot = id >> 3;
  const int q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd*(q + 1) : r*(q + 1) + (xcd - r)*q) + slot;
}

// Sums of TWO per-lane values over the wavefront in 6 cross-lane steps instead of 12 (wave_sum is six
// ds_bpermute per value): v_permlane32_swap packs (a: lanes 0-31 + lanes 32-63) into the lower half and the
// same of b into the upper half, one bpermute folds the two 16-lane rows of each half, row rotations and
// quad permutes (DPP, no LDS traffic) fold the 16 lanes of a row. Lanes 0-31 return sum(a), lanes 32-63
// sum(b). The association is fixed: results stay bitwise repeatable.
__device__ __forceinline__ float wave_sum2(float a, float b) {
#ifdef BRV_F32_WAVESUM_PLAIN      // comparison builds: the two sums as 2 x 6 ds_bpermute
  const float sa = wave_sum(a), sb = wave_sum(b);
  return (threadIdx.x & 32) ? sb : sa;
#endif
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  float c = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  c += __shfl_xor(c, 16, 64);
  c += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(c), 0x128, 0xf, 0xf, false));   // row_ror:8
  c += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(c), 0x124, 0xf, 0xf, false));   // row_ror:4
  c += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(c), 0x4e, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
  c += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(c), 0xb1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
  return c;
}

struct FusedCommon {
  long long rows; int T, C;          // rows = B*T frames of C channels (C % 4 == 0, 16-byte aligned)
};

// Which frames a stencil workgroup takes. Contiguous slices (frames r0 .. r0 + 63) make every frame's taps
// at +- k dil somebody else's frames once dil exceeds the slice: PMC showed the stencil backward moving 591
// MB per launch for 393 MB of tensors. COMB slices (dil >= 4 and >= 16 teeth per item): wavefront w walks
// the frames of ONE residue class, t = (16 tg + j) dil + 4 rg + w for j = 0 .. 15 -- the taps of a frame
// are the frames the same wavefront visits next to it (PMC: 591 -> 507 MB for the backward stencil, 352 -> 294
// for the forward one; the launch times did not move -- the stencils are bound by their per-frame chain of
// loads and arithmetic, not by these bytes). The
// per-frame outputs are indexed by frame and the per-channel sums are folded over all slices, so the
// order of the frames changes nothing else.
struct SliceMap { int comb, dil, n_rg, spi; };       // spi: slices per item (comb only)
__host__ __device__ inline SliceMap slice_map(int T, int dil) {
  SliceMap m; m.dil = dil;
  const int teeth = (T + dil - 1)/dil;
  m.comb = dil >= 4 && teeth >= 16;
  m.n_rg = (dil + 3)/4;
  m.spi = m.n_rg*((teeth + 15)/16);
  return m;
}
__host__ __device__ inline long long slice_count(const SliceMap& m, long long B, int T) {
  return m.comb ? B*m.spi : (B*T + kFusedRows - 1)/kFusedRows;
}
// frame `it` (0 .. 63; wavefront w takes it = w, w + 4, ...) of slice `s` -> global row, or -1
__device__ __forceinline__ long long slice_row(const SliceMap& m, int s, int it, long long rows, int T) {
  if (!m.comb) {
    const long long row = (long long)s*kFusedRows + it;
    return row < rows ? row : -1;
  }
  const int b = s / m.spi, rem = s % m.spi;
  const int rg = rem % m.n_rg, tg = rem / m.n_rg;
  const int res = 4*rg + (it & 3), t = (16*tg + (it >> 2))*m.dil + res;
  return (res < m.dil && t < T) ? (long long)b*T + t : -1;
}

// per-channel sums of one workgroup: q[nq][4] per lane and j -> part[(slice*nq + k)*C + c], wavefront order
template <int NQ, int NJ>
__device__ __forceinline__ void write_chan_partials(float4 (&q)[NQ][NJ], int nq, int C, float* part,
                                                    float (*red)[256][4]) {
  // red: [3][256][4] floats of LDS; wavefronts 1..3 park their sums, wavefront 0 adds them in order
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NQ; ++k) {
    if (k >= nq) break;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = 4*(lane + 64*j);
      __syncthreads();
      if (w > 0) { red[w - 1][lane][0] = q[k][j].x; red[w - 1][lane][1] = q[k][j].y; red[w - 1][lane][2] = q[k][j].z; red[w - 1][lane][3] = q[k][j].w; }
      __syncthreads();
      if (w == 0 && c < C) {
        float4 t = q[k][j];
#pragma unroll
        for (int s = 0; s < 3; ++s) { t.x += red[s][lane][0]; t.y += red[s][lane][1]; t.z += red[s][lane][2]; t.w += red[s][lane][3]; }
        *reinterpret_cast<float4*>(part + ((long long)fused_slice()*nq + k)*C + c) = t;
      }
    }
  }
}

// ---- forward: z2 = dconv(norm1(prelu1(z1))) + bias, frame sums of prelu2(z2) --------------------
struct DwFwd {
  FusedCommon s;
  const float* z1; const float* tab1; const float* slope1; const float* gain1; const float* bias1;
  const float* taps; const float* dbias; float* z2; const float* slope2; float* fsum2;
  int P, dil, left;
};
template <int NJ, int PM>
__global__ __launch_bounds__(256) void f32_dw_fwd_fused_kernel(const DwFwd p) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int C = p.s.C, T = p.s.T;
  const float a1 = *p.slope1, a2 = *p.slope2;
  float4 g1[NJ], b1[NJ], db[NJ], tp[PM][NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = 4*(lane + 64*j);
    const bool ok = c < C;
    const int cc = ok ? c : 0;
    g1[j] = ld4u(p.gain1 + cc); b1[j] = ld4u(p.bias1 + cc); db[j] = ld4u(p.dbias + cc);
#pragma unroll
    for (int k = 0; k < PM; ++k)
      tp[k][j] = k < p.P ? make_float4(p.taps[(cc + 0)*p.P + k], p.taps[(cc + 1)*p.P + k], p.taps[(cc + 2)*p.P + k],
                                       p.taps[(cc + 3)*p.P + k]) : f4(0.f);
  }
  const SliceMap sm = slice_map(T, p.dil);
  const int sl = fused_slice();
  for (int it = w; it < kFusedRows; it += 4) {
    const long long row = slice_row(sm, sl, it, p.s.rows, T);
    if (row < 0) continue;
    const int t = (int)(row % T);
    float4 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = db[j];
#pragma unroll
    for (int k = 0; k < PM; ++k) {
      if (k >= p.P) break;
      const int off = k*p.dil - p.left;
      if (t + off < 0 || t + off >= T) continue;
      const long long rr = row + off;
      const float mean = p.tab1[2*rr], rstd = p.tab1[2*rr + 1];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int c = 4*(lane + 64*j);
        if (c >= C) continue;
        const float4 v = *reinterpret_cast<const float4*>(p.z1 + rr*C + c);
        acc[j].x = __builtin_fmaf(tp[k][j].x, (prelu1(v.x, a1, true) - mean)*rstd*g1[j].x + b1[j].x, acc[j].x);
        acc[j].y = __builtin_fmaf(tp[k][j].y, (prelu1(v.y, a1, true) - mean)*rstd*g1[j].y + b1[j].y, acc[j].y);
        acc[j].z = __builtin_fmaf(tp[k][j].z, (prelu1(v.z, a1, true) - mean)*rstd*g1[j].z + b1[j].z, acc[j].z);
        acc[j].w = __builtin_fmaf(tp[k][j].w, (prelu1(v.w, a1, true) - mean)*rstd*g1[j].w + b1[j].w, acc[j].w);
      }
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = 4*(lane + 64*j);
      if (c >= C) continue;
      *reinterpret_cast<float4*>(p.z2 + row*C + c) = acc[j];
      const float4 q = make_float4(prelu1(acc[j].x, a2, true), prelu1(acc[j].y, a2, true),
                                   prelu1(acc[j].z, a2, true), prelu1(acc[j].w, a2, true));
      s1 += sum4(q);
      s2 += (q.x*q.x + q.y*q.y) + (q.z*q.z + q.w*q.w);
    }
    const float s12 = wave_sum2(s1, s2);
    if ((lane & 31) == 0) p.fsum2[2*row + (lane >> 5)] = s12;
  }
}

// ---- backward of y = norm(prelu(z)), pass 1: frame sums A_t = sum_c e gain, B_t = sum_c e gain xhat
// and the per-channel sums dgain = sum_t e xhat, dbias = sum_t e ---------------------------------------
struct BwdSums {
  FusedCommon s;
  const float* e; const float* z; const float* tab; const float* slope; const float* gain;
  float* fsum; float* part;        // part[slice][2][C]: quantity 0 = dgain, 1 = dbias
};
template <int NJ>
__global__ __launch_bounds__(256) void f32_bwd_sums_kernel(const BwdSums p) {
  __shared__ float red[3][256][4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int C = p.s.C;
  const bool act = p.slope != nullptr;
  const float a = act ? *p.slope : 1.f;
  float4 gn[NJ], q[2][NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = 4*(lane + 64*j);
    gn[j] = ld4u(p.gain + (c < C ? c : 0));
    q[0][j] = f4(0.f); q[1][j] = f4(0.f);
  }
  const long long r0 = (long long)fused_slice()*kFusedRows;
  for (int it = w; it < kFusedRows; it += 4) {
    const long long row = r0 + it;
    if (row >= p.s.rows) break;
    const float mean = p.tab[2*row], rstd = p.tab[2*row + 1];
    float A = 0.f, Bq = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = 4*(lane + 64*j);
      if (c >= C) continue;
      const float4 ev = *reinterpret_cast<const float4*>(p.e + row*C + c);
      const float4 zv = *reinterpret_cast<const float4*>(p.z + row*C + c);
#define BRV_ONE(f)                                                         \
      { const float xh = (prelu1(zv.f, a, act) - mean)*rstd;                 \
        const float gg = ev.f*gn[j].f;                                       \
        A += gg; Bq = __builtin_fmaf(gg, xh, Bq);                            \
        q[0][j].f = __builtin_fmaf(ev.f, xh, q[0][j].f); q[1][j].f += ev.f; }
      BRV_ONE(x) BRV_ONE(y) BRV_ONE(z) BRV_ONE(w)
#undef BRV_ONE
    }
    const float ab = wave_sum2(A, Bq);
    if ((lane & 31) == 0) p.fsum[2*row + (lane >> 5)] = ab;
  }
  write_chan_partials<2, NJ>(q, 2, C, p.part, red);
}

// ---- pass 2: dz = prelu'(z) (e gain rstd_t + U_t + p V_t) (+ add); slope-gradient partial per
// workgroup; per-channel sums of dz (the bias gradient of the convolution that produced z) -----------
struct BwdApply {
  FusedCommon s;
  const float* e; const float* z; const float* slope; const float* ftab; const float* btab;
  const float* gain; const float* add; float* dz;
  float* slope_part;               // [slices] or null
  float* part;                     // part[slice][1][C] = sum_t dz, or null
};
template <int NJ>
__global__ __launch_bounds__(256) void f32_bwd_apply_fused_kernel(const BwdApply p) {
  __shared__ float red[3][256][4];
  __shared__ float sred[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int C = p.s.C;
  const bool act = p.slope != nullptr;
  const float a = act ? *p.slope : 1.f;
  float4 gn[NJ], q[1][NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = 4*(lane + 64*j);
    gn[j] = ld4u(p.gain + (c < C ? c : 0));
    q[0][j] = f4(0.f);
  }
  float da = 0.f;
  const long long r0 = (long long)fused_slice()*kFusedRows;
  for (int it = w; it < kFusedRows; it += 4) {
    const long long row = r0 + it;
    if (row >= p.s.rows) break;
    const float r = p.ftab[2*row + 1], U = p.btab[2*row], V = p.btab[2*row + 1];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = 4*(lane + 64*j);
      if (c >= C) continue;
      const float4 ev = *reinterpret_cast<const float4*>(p.e + row*C + c);
      const float4 zv = *reinterpret_cast<const float4*>(p.z + row*C + c);
      float4 o;
#define BRV_ONE(f)                                                              \
      { const bool pos = zv.f > 0.f || !act; const float pv = pos ? zv.f : a*zv.f; \
        const float dp = ev.f*gn[j].f*r + U + pv*V; o.f = pos ? dp : a*dp;         \
        if (!pos) da = __builtin_fmaf(dp, zv.f, da); }
      BRV_ONE(x) BRV_ONE(y) BRV_ONE(z) BRV_ONE(w)
#undef BRV_ONE
      if (p.add) {
        const float4 ad = *reinterpret_cast<const float4*>(p.add + row*C + c);
        o.x += ad.x; o.y += ad.y; o.z += ad.z; o.w += ad.w;
      }
      *reinterpret_cast<float4*>(p.dz + row*C + c) = o;
      q[0][j].x += o.x; q[0][j].y += o.y; q[0][j].z += o.z; q[0][j].w += o.w;
    }
  }
  if (p.slope_part) {
    da = wave_sum(da);
    if (lane == 0) sred[w] = da;
    __syncthreads();
    if (threadIdx.x == 0) p.slope_part[fused_slice()] = ((sred[0] + sred[1]) + sred[2]) + sred[3];
  }
  if (p.part) write_chan_partials<1, NJ>(q, 1, C, p.part, red);
}

// ---- depthwise backward, fused with pass 1 of the first norm's backward -----------------------------
//   tap gradients  q_k[c] = sum_t dz2[t][c] h1[t + k dil - left][c]   (h1 = norm1(prelu1(z1)), rebuilt)
//   e1[t][c]       = sum_k w[c][k] dz2[t - k dil + left][c]           (gradient wrt h1)
//   norm1 pass 1 on e1: A_t, B_t, dgain1, dbias1
struct DwBwd {
  FusedCommon s;
  const float* dz2; const float* z1; const float* tab1; const float* slope1; const float* gain1;
  const float* bias1; const float* taps; float* e1; float* fsum; float* part;   // part[slice][P + 2][C]
  int P, dil, left;
};
template <int NJ, int PM>
__global__ __launch_bounds__(256) void f32_dw_bwd_fused_kernel(const DwBwd p) {
  __shared__ float red[3][256][4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int C = p.s.C, T = p.s.T;
  const float a1 = *p.slope1;
  float4 g1[NJ], b1[NJ], tp[PM][NJ], q[PM + 2][NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = 4*(lane + 64*j);
    const int cc = c < C ? c : 0;
    g1[j] = ld4u(p.gain1 + cc); b1[j] = ld4u(p.bias1 + cc);
#pragma unroll
    for (int k = 0; k < PM; ++k)
      tp[k][j] = k < p.P ? make_float4(p.taps[(cc + 0)*p.P + k], p.taps[(cc + 1)*p.P + k], p.taps[(cc + 2)*p.P + k],
                                       p.taps[(cc + 3)*p.P + k]) : f4(0.f);
#pragma unroll
    for (int k = 0; k < PM + 2; ++k) q[k][j] = f4(0.f);
  }
  const SliceMap sm = slice_map(T, p.dil);
  const int sl = fused_slice();
  for (int it = w; it < kFusedRows; it += 4) {
    const long long row = slice_row(sm, sl, it, p.s.rows, T);
    if (row < 0) continue;
    const int t = (int)(row % T);
    const float mean = p.tab1[2*row], rstd = p.tab1[2*row + 1];
    // all loads of the frame first (rows outside the item are replaced by the frame itself and
    // weighted with zero: no branch between the loads, they are in flight together)
    float4 d0[NJ], zc[NJ], zt[PM][NJ], dt[PM][NJ];
    float m2[PM], r2[PM], w1[PM], w2[PM];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = 4*(lane + 64*j);
      const int cc = c < C ? c : 0;
      d0[j] = *reinterpret_cast<const float4*>(p.dz2 + row*C + cc);
      zc[j] = *reinterpret_cast<const float4*>(p.z1 + row*C + cc);
    }
#pragma unroll
    for (int k = 0; k < PM; ++k) {
      const int off = k < p.P ? k*p.dil - p.left : 0;
      const bool v1 = k < p.P && t + off >= 0 && t + off < T;
      const bool v2 = k < p.P && t - off >= 0 && t - off < T;
      const long long rr1 = v1 ? row + off : row, rr2 = v2 ? row - off : row;
      w1[k] = v1 ? 1.f : 0.f; w2[k] = v2 ? 1.f : 0.f;
      m2[k] = p.tab1[2*rr1]; r2[k] = p.tab1[2*rr1 + 1];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int c = 4*(lane + 64*j);
        const int cc = c < C ? c : 0;
        zt[k][j] = (BRV_DWB_ABL & 4) ? zc[j] : *reinterpret_cast<const float4*>(p.z1 + rr1*C + cc);
        dt[k][j] = (BRV_DWB_ABL & 2) ? d0[j] : *reinterpret_cast<const float4*>(p.dz2 + rr2*C + cc);
      }
    }
    float4 e1[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) e1[j] = f4(0.f);
#pragma unroll
    for (int k = 0; k < PM; ++k)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const float4 v = zt[k][j];
        const float sx = d0[j].x*w1[k], sy = d0[j].y*w1[k], sz = d0[j].z*w1[k], sw = d0[j].w*w1[k];
        if (BRV_DWB_ABL & 1) { q[k][j].x += sx*v.x; continue; }
        q[k][j].x = __builtin_fmaf(sx, (prelu1(v.x, a1, true) - m2[k])*r2[k]*g1[j].x + b1[j].x, q[k][j].x);
        q[k][j].y = __builtin_fmaf(sy, (prelu1(v.y, a1, true) - m2[k])*r2[k]*g1[j].y + b1[j].y, q[k][j].y);
        q[k][j].z = __builtin_fmaf(sz, (prelu1(v.z, a1, true) - m2[k])*r2[k]*g1[j].z + b1[j].z, q[k][j].z);
        q[k][j].w = __builtin_fmaf(sw, (prelu1(v.w, a1, true) - m2[k])*r2[k]*g1[j].w + b1[j].w, q[k][j].w);
        const float4 dv = dt[k][j];
        e1[j].x = __builtin_fmaf(tp[k][j].x*w2[k], dv.x, e1[j].x); e1[j].y = __builtin_fmaf(tp[k][j].y*w2[k], dv.y, e1[j].y);
        e1[j].z = __builtin_fmaf(tp[k][j].z*w2[k], dv.z, e1[j].z); e1[j].w = __builtin_fmaf(tp[k][j].w*w2[k], dv.w, e1[j].w);
      }
    float A = 0.f, Bq = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = 4*(lane + 64*j);
      if (c >= C) continue;
      *reinterpret_cast<float4*>(p.e1 + row*C + c) = e1[j];
      const float4 zv = zc[j];
#define BRV_ONE(f)                                                         \
      { const float xh = (prelu1(zv.f, a1, true) - mean)*rstd;               \
        const float gg = e1[j].f*g1[j].f;                                    \
        A += gg; Bq = __builtin_fmaf(gg, xh, Bq);                            \
        q[PM][j].f = __builtin_fmaf(e1[j].f, xh, q[PM][j].f); q[PM + 1][j].f += e1[j].f; }
      BRV_ONE(x) BRV_ONE(y) BRV_ONE(z) BRV_ONE(w)
#undef BRV_ONE
    }
    const float ab = (BRV_DWB_ABL & 8) ? A + Bq : wave_sum2(A, Bq);
    if ((lane & 31) == 0) p.fsum[2*row + (lane >> 5)] = ab;
  }
  // quantities 0 .. P-1: taps, P: dgain1, P + 1: dbias1 (accumulated at PM, PM + 1)
  if (p.P < PM) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const float4 dg = q[PM][j], dbv = q[PM + 1][j];
#pragma unroll
      for (int k = 0; k < PM; ++k)
        if (k == p.P) { q[k][j] = dg; q[k + 1][j] = dbv; }
    }
  }
  write_chan_partials<PM + 2, NJ>(q, p.P + 2, C, p.part, red);
}

// ---- fold of the per-slice partial sums: quantity k of channel c goes to dst[k][c*stride[k]] --------
struct FoldJob { const float* part; int slices, nq, C; float* dst[9]; int stride[9]; };
__global__ __launch_bounds__(1024) void f32_fold_kernel(const FoldJob p) {
  // 16 (quantity, channel) columns x 64 slice lanes per workgroup: a lane adds its slices in order
  // (4 interleaved chains: the loads of a step are independent), the 64 lane sums are added in lane
  // order by one thread per column
  __shared__ float acc[64][17];
  const int col = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int i = blockIdx.x*16 + col;
  const long long ld = (long long)p.nq*p.C;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < p.nq*p.C) {
    int sidx = sl;
    for (; sidx + 192 < p.slices; sidx += 256) {
      const float v0 = p.part[sidx*ld + i], v1 = p.part[(sidx + 64)*ld + i];
      const float v2 = p.part[(sidx + 128)*ld + i], v3 = p.part[(sidx + 192)*ld + i];
      s0 += v0; s1 += v1; s2 += v2; s3 += v3;
    }
    for (; sidx < p.slices; sidx += 64) s0 += p.part[sidx*ld + i];
  }
  acc[sl][col] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sl == 0 && i < p.nq*p.C) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 64; ++j) t += acc[j][col];
    const int k = i / p.C, c = i % p.C;
    p.dst[k][(long long)c*p.stride[k]] += t;
  }
}
