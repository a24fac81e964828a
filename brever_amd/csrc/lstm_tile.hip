// LSTM recurrence for MANY short chains (TF-GridNet: thousands of sequences of 33 ... 126
// steps, hidden size 128; brever/models/tfgridnet/tfgridnet.py:200-216,268-313 -> nn.LSTM).
//
// dccrn.hip's recurrence gives every chain a workgroup of its own, which is right for DCCRN's
// few long chains and wrong here: a step is then one 128-long dot product per thread between two
// barriers. This kernel steps a TILE of 16 chains per workgroup, so that a step is a small
// matrix product on the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32, bitwise an fmaf chain):
//
//   forward   G^T (4H x 16) = W_hh (4H x H) . h^T (H x 16)   + input projection
//   backward  dh^T (H x 16) = W_hh^T (H x 4H) . dG^T (4H x 16)
//
// Eight wavefronts; wave w keeps its 64 rows of W_hh (forward: the four gates of hidden units
// 16w .. 16w+15) or its 16 rows of W_hh^T (backward: the same units) as MFMA A fragments in 128
// VGPRs for all T steps. The B operand (h^T or dG^T) is re-read from LDS every step, stored so
// that four consecutive k-slices come back with one ds_read_b128. Rows are assigned to fragment
// slots such that a lane's accumulators are the four gates of four CONSECUTIVE units of one
// chain: all gate arithmetic is lane-local and every global access is 16 or 64 contiguous bytes.
// One barrier per step (double-buffered B operand).
//
// Gate layout: the input projection, the saved activations and the gate gradients use the
// INTERLEAVED order (column = 4*unit + gate, gates i, f, g, o) instead of torch's gate-major
// order; the host permutes the rows of W_ih once per call (models/dccrn.py _LSTMFunction).
#include <hip/hip_runtime.h>
#include <math.h>

#include "../../include/brever_hip.h"
#include "common.cuh"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int LH = 128;       // hidden size
constexpr int LC = 16;        // chains per workgroup

// gate non-linearities on the hardware exp / rcp (v_exp_f32, v_rcp_f32: ~1 ulp each; the gate
// math is as long as the MFMA phase of a step when done with the libm forms)
__device__ __forceinline__ float sigm(float x) {
  return __builtin_amdgcn_rcpf(1.f + __expf(-x));
}
__device__ __forceinline__ float tanh_fast(float x) {      // 2 sigm(2x) - 1, saturates cleanly
  return 2.f*__builtin_amdgcn_rcpf(1.f + __expf(-2.f*x)) - 1.f;
}

// B-operand storage: element (k, n) of a (K x 16) matrix, k = 4*kk + j: the four kk of a group
// q = kk/4 are adjacent, so lane (j, n) fetches its operands for kk = 4q .. 4q+3 with one b128
__device__ __forceinline__ int bslot(int k, int n) {
  const int kk = k >> 2, j = k & 3;
  return ((((kk >> 2) << 2) + j)*LC + n)*4 + (kk & 3);
}

__global__ __launch_bounds__(512) void lstm_tile_fwd_kernel(const float* __restrict__ gates_in,
                                                            const float* __restrict__ w_hh,
                                                            const float* __restrict__ bias,
                                                            float* __restrict__ y,
                                                            float* __restrict__ act,
                                                            float* __restrict__ cs, int per_group,
                                                            int T, int reverse_mask, long long y_ld,
                                                            long long y_goff) {
  __shared__ __attribute__((aligned(16))) float hbuf[2][LH*LC];
  const int tiles = (per_group + LC - 1)/LC;
  const int grp = blockIdx.x / tiles, tile = blockIdx.x % tiles;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int n = lane & 15, j = lane >> 4;
  const int cl = tile*LC + n;
  const bool live = cl < per_group;
  const long long chain = (long long)grp*per_group + (live ? cl : per_group - 1);
  w_hh += (long long)grp*4*LH*LH;
  // A fragments: block b, slice kk: A[m = lane & 15][k = 4 kk + (lane >> 4)]; row m = 4 jm + g
  // of block b is gate g of unit 16 w + 4 jm + b
  float wf[4][LH/4];
  {
    const int jm = (lane & 15) >> 2, g = lane & 3;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const float* row = w_hh + (long long)(g*LH + 16*w + 4*jm + b)*LH + (lane >> 4);
#pragma unroll
      for (int kk = 0; kk < LH/4; ++kk) wf[b][kk] = row[4*kk];
    }
  }
  const int u0 = 16*w + 4*j;                      // this lane's cells: units u0 .. u0+3 of chain n
  f32x4 bs[4];
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int g = 0; g < 4; ++g) bs[b][g] = bias ? bias[(long long)grp*4*LH + g*LH + u0 + b] : 0.f;
  float c[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i = tid; i < LH*LC; i += 512) hbuf[0][i] = 0.f;
  // a reversed group walks the frames backwards: step t of the recurrence is frame T-1-t of
  // gates_in and y (the saved act / cs stay indexed by the recurrence step)
  const bool rev = (reverse_mask >> grp) & 1;
  const int t_first = rev ? T - 1 : 0, t_inc = rev ? -1 : 1;
  const float* gin = gates_in + chain*T*4*LH + 4*u0;
  float* yrow = y + (long long)(live ? cl : per_group - 1)*T*y_ld + grp*y_goff + u0;
  f32x4 gnext[4];
#pragma unroll
  for (int b = 0; b < 4; ++b)
    gnext[b] = *reinterpret_cast<const f32x4*>(gin + (long long)t_first*4*LH + 4*b);
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    const int tt = t_first + t*t_inc;
    f32x4 acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = gnext[b] + bs[b];
    if (t + 1 < T) {
#pragma unroll
      for (int b = 0; b < 4; ++b)
        gnext[b] = *reinterpret_cast<const f32x4*>(gin + (long long)(tt + t_inc)*4*LH + 4*b);
    }
    const float* hb = hbuf[t & 1];
#pragma unroll
    for (int q = 0; q < LH/16; ++q) {
      const f32x4 hv = *reinterpret_cast<const f32x4*>(hb + ((q*4 + j)*LC + n)*4);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[b][4*q + s], hv[s], acc[b], 0, 0, 0);
    }
    f32x4 hn, cn;
    f32x4 a[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const float ig = sigm(acc[b][0]), fg = sigm(acc[b][1]);
      const float gg = tanh_fast(acc[b][2]), og = sigm(acc[b][3]);
      c[b] = fg*c[b] + ig*gg;
      cn[b] = c[b];
      hn[b] = og*tanh_fast(c[b]);
      a[b] = f32x4{ig, fg, gg, og};
    }
    float* hw = hbuf[(t + 1) & 1];
#pragma unroll
    for (int b = 0; b < 4; ++b) hw[bslot(u0 + b, n)] = hn[b];
    if (live) {
      const long long step = chain*T + t;
      *reinterpret_cast<f32x4*>(yrow + tt*y_ld) = hn;
      if (act) {
        *reinterpret_cast<f32x4*>(cs + step*LH + u0) = cn;
#pragma unroll
        for (int b = 0; b < 4; ++b)
          *reinterpret_cast<f32x4*>(act + step*4*LH + 4*(u0 + b)) = a[b];
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(512) void lstm_tile_bwd_kernel(const float* __restrict__ act,
                                                            const float* __restrict__ cs,
                                                            const float* __restrict__ w_hh,
                                                            const float* __restrict__ dy,
                                                            float* __restrict__ dgates,
                                                            int per_group, int T, int reverse_mask,
                                                            long long dy_ld, long long dy_goff) {
  __shared__ __attribute__((aligned(16))) float gbuf[2][4*LH*LC];      // dG^T, 2 x 32 KB
  const int tiles = (per_group + LC - 1)/LC;
  const int grp = blockIdx.x / tiles, tile = blockIdx.x % tiles;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int n = lane & 15, j = lane >> 4;
  const int cl = tile*LC + n;
  const bool live = cl < per_group;
  const long long chain = (long long)grp*per_group + (live ? cl : per_group - 1);
  w_hh += (long long)grp*4*LH*LH;
  // A = W_hh^T rows 16 w .. 16 w + 15: A[m][k] = W_hh[row(k)][16 w + m], k in interleaved order
  // (k = 4 unit' + gate <-> torch row gate*H + unit'); slice kk covers k = 4 kk .. 4 kk + 3 =
  // the four gates of unit' = kk, the lane supplying gate (lane >> 4)
  float wf[LH];
#pragma unroll
  for (int kk = 0; kk < LH; ++kk)
    wf[kk] = w_hh[(long long)((lane >> 4)*LH + kk)*LH + 16*w + (lane & 15)];
  const int u0 = 16*w + 4*j;
  f32x4 dh = {0.f, 0.f, 0.f, 0.f};       // gradient reaching h_t through the recurrence
  f32x4 dc = {0.f, 0.f, 0.f, 0.f};
  const bool rev = (reverse_mask >> grp) & 1;
  const float* dyrow = dy + (long long)(live ? cl : per_group - 1)*T*dy_ld + grp*dy_goff + u0;
  for (int t = T - 1; t >= 0; --t) {
    const long long step = chain*T + t;
    const int tt = rev ? T - 1 - t : t;             // frame of recurrence step t
    f32x4 a[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) a[b] = *reinterpret_cast<const f32x4*>(act + step*4*LH + 4*(u0 + b));
    const f32x4 cv = *reinterpret_cast<const f32x4*>(cs + step*LH + u0);
    f32x4 cp = {0.f, 0.f, 0.f, 0.f};
    if (t > 0) cp = *reinterpret_cast<const f32x4*>(cs + (step - 1)*LH + u0);
    const f32x4 dyv = *reinterpret_cast<const f32x4*>(dyrow + tt*dy_ld);
    float* gw = gbuf[t & 1];
    f32x4 d[4];                                   // d[gate][b]
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const float ig = a[b][0], fg = a[b][1], gg = a[b][2], og = a[b][3];
      const float tc = tanh_fast(cv[b]);
      const float dht = dh[b] + dyv[b];
      const float dct = dc[b] + dht*og*(1.f - tc*tc);
      d[0][b] = dct*gg*ig*(1.f - ig);
      d[1][b] = dct*cp[b]*fg*(1.f - fg);
      d[2][b] = dct*ig*(1.f - gg*gg);
      d[3][b] = dht*tc*og*(1.f - og);
      dc[b] = dct*fg;
    }
    // element (k = 4 (u0 + b) + g, n): kk = u0 + b -> group (u0 >> 2) = 4 w + j, kk & 3 = b
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<f32x4*>(gw + (((4*w + j)*4 + g)*LC + n)*4) = d[g];
    if (live) {
#pragma unroll
      for (int b = 0; b < 4; ++b)
        *reinterpret_cast<f32x4*>(dgates + (chain*T + tt)*4*LH + 4*(u0 + b)) =
            f32x4{d[0][b], d[1][b], d[2][b], d[3][b]};
    }
    __syncthreads();
    if (t == 0) break;
    f32x4 p[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) p[s] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < LH/4; ++q) {
      const f32x4 gv = *reinterpret_cast<const f32x4*>(gw + ((q*4 + j)*LC + n)*4);
#pragma unroll
      for (int s = 0; s < 4; ++s)
        p[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[4*q + s], gv[s], p[s], 0, 0, 0);
    }
    dh = (p[0] + p[1]) + (p[2] + p[3]);
  }
}

// ---- bf16-operand variants (use_amp): W_hh and the B operand (h^T or dG^T) are rounded to bf16
// and multiplied on v_mfma_f32_16x16x32_bf16 (K = 32 per instruction: 16 instead of 128 MFMAs per
// wave and step); accumulators, cell state, gate math and every tensor in HBM stay fp32. The B
// operand sits in LDS chain-major ([chain][k], rows padded by 16 B so that the 16-byte fragment
// reads of 16 chains fall on distinct banks); same row -> fragment assignment as above.
using brv::bf16x8;
using brv::pack2;
constexpr int HROW = LH + 8;          // bf16 elements per chain row of h^T
constexpr int GROW = 4*LH + 8;        // ... of dG^T

__device__ __forceinline__ bf16x8 load_frag(const float* p, long long stride) {
  uint4 q;
  q.x = pack2(p[0], p[stride]); q.y = pack2(p[2*stride], p[3*stride]);
  q.z = pack2(p[4*stride], p[5*stride]); q.w = pack2(p[6*stride], p[7*stride]);
  return __builtin_bit_cast(bf16x8, q);
}

// IO16: gates_in and the saved activations are bf16 in memory (lowp == 2): the two largest
// streams of this HBM-bound kernel at half the bytes
__device__ __forceinline__ f32x4 load4(const float* p, long long i, bool half) {
  if (!half) return *reinterpret_cast<const f32x4*>(p + i);
  const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(p) + i);
  return f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
               __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
}
__device__ __forceinline__ void store4(float* p, long long i, const f32x4& v, bool half) {
  if (!half) { *reinterpret_cast<f32x4*>(p + i) = v; return; }
  *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p) + i) =
      make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
}

template <bool IO16>
__global__ __launch_bounds__(512) void lstm_tile_fwd_bf16_kernel(const float* __restrict__ gates_in,
                                                                 const float* __restrict__ w_hh,
                                                                 const float* __restrict__ bias,
                                                                 float* __restrict__ y,
                                                                 float* __restrict__ act,
                                                                 float* __restrict__ cs,
                                                                 int per_group, int T,
                                                                 int reverse_mask, long long y_ld,
                                                                 long long y_goff) {
  __shared__ __attribute__((aligned(16))) uint16_t hbuf[2][LC*HROW];
  const int tiles = (per_group + LC - 1)/LC;
  const int grp = blockIdx.x / tiles, tile = blockIdx.x % tiles;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int n = lane & 15, j = lane >> 4;
  const int cl = tile*LC + n;
  const bool live = cl < per_group;
  const long long chain = (long long)grp*per_group + (live ? cl : per_group - 1);
  w_hh += (long long)grp*4*LH*LH;
  // A[m = lane & 15][k = 32 kq + 8 (lane >> 4) + i]; row m = 4 jm + g of block b = gate g of unit
  // 16 w + 4 jm + b
  bf16x8 wf[4][LH/32];
  {
    const int jm = (lane & 15) >> 2, g = lane & 3;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const float* row = w_hh + (long long)(g*LH + 16*w + 4*jm + b)*LH + 8*(lane >> 4);
#pragma unroll
      for (int kq = 0; kq < LH/32; ++kq) wf[b][kq] = load_frag(row + 32*kq, 1);
    }
  }
  const int u0 = 16*w + 4*j;
  f32x4 bs[4];
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int g = 0; g < 4; ++g) bs[b][g] = bias ? bias[(long long)grp*4*LH + g*LH + u0 + b] : 0.f;
  float c[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i = tid; i < LC*HROW; i += 512) hbuf[0][i] = 0;
  const bool rev = (reverse_mask >> grp) & 1;
  const int t_first = rev ? T - 1 : 0, t_inc = rev ? -1 : 1;
  const float* gin = gates_in;
  const long long gbase = chain*T*4*LH + 4*u0;
  float* yrow = y + (long long)(live ? cl : per_group - 1)*T*y_ld + grp*y_goff + u0;
  f32x4 gnext[4];
#pragma unroll
  for (int b = 0; b < 4; ++b)
    gnext[b] = load4(gin, gbase + (long long)t_first*4*LH + 4*b, IO16);
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    const int tt = t_first + t*t_inc;
    f32x4 acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = gnext[b] + bs[b];
    if (t + 1 < T) {
#pragma unroll
      for (int b = 0; b < 4; ++b)
        gnext[b] = load4(gin, gbase + (long long)(tt + t_inc)*4*LH + 4*b, IO16);
    }
    const uint16_t* hb = hbuf[t & 1] + n*HROW + 8*j;
#pragma unroll
    for (int kq = 0; kq < LH/32; ++kq) {
      const bf16x8 hv = *reinterpret_cast<const bf16x8*>(hb + 32*kq);
#pragma unroll
      for (int b = 0; b < 4; ++b)
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[b][kq], hv, acc[b], 0, 0, 0);
    }
    f32x4 hn, cn;
    f32x4 a[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const float ig = sigm(acc[b][0]), fg = sigm(acc[b][1]);
      const float gg = tanh_fast(acc[b][2]), og = sigm(acc[b][3]);
      c[b] = fg*c[b] + ig*gg;
      cn[b] = c[b];
      hn[b] = og*tanh_fast(c[b]);
      a[b] = f32x4{ig, fg, gg, og};
    }
    *reinterpret_cast<uint2*>(hbuf[(t + 1) & 1] + n*HROW + u0) =
        make_uint2(pack2(hn[0], hn[1]), pack2(hn[2], hn[3]));
    if (live) {
      const long long step = chain*T + t;
      *reinterpret_cast<f32x4*>(yrow + tt*y_ld) = hn;
      if (act) {
        *reinterpret_cast<f32x4*>(cs + step*LH + u0) = cn;
#pragma unroll
        for (int b = 0; b < 4; ++b)
          store4(act, step*4*LH + 4*(u0 + b), a[b], IO16);
      }
    }
    __syncthreads();
  }
}

template <bool IO16>
__global__ __launch_bounds__(512) void lstm_tile_bwd_bf16_kernel(const float* __restrict__ act,
                                                                 const float* __restrict__ cs,
                                                                 const float* __restrict__ w_hh,
                                                                 const float* __restrict__ dy,
                                                                 float* __restrict__ dgates,
                                                                 int per_group, int T,
                                                                 int reverse_mask, long long dy_ld,
                                                                 long long dy_goff) {
  __shared__ __attribute__((aligned(16))) uint16_t gbuf[2][LC*GROW];
  const int tiles = (per_group + LC - 1)/LC;
  const int grp = blockIdx.x / tiles, tile = blockIdx.x % tiles;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int n = lane & 15, j = lane >> 4;
  const int cl = tile*LC + n;
  const bool live = cl < per_group;
  const long long chain = (long long)grp*per_group + (live ? cl : per_group - 1);
  w_hh += (long long)grp*4*LH*LH;
  // A[m][k] = W_hh[torch row of k][16 w + m], k = 4 unit' + gate = 32 kq + 8 (lane >> 4) + i:
  // unit' = 8 kq + 2 (lane >> 4) + (i >> 2), gate = i & 3
  bf16x8 wf[4*LH/32];
#pragma unroll
  for (int kq = 0; kq < 4*LH/32; ++kq) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int unit = 8*kq + 2*(lane >> 4) + (i >> 2), gate = i & 3;
      v[i] = w_hh[(long long)(gate*LH + unit)*LH + 16*w + (lane & 15)];
    }
    uint4 q;
    q.x = pack2(v[0], v[1]); q.y = pack2(v[2], v[3]); q.z = pack2(v[4], v[5]); q.w = pack2(v[6], v[7]);
    wf[kq] = __builtin_bit_cast(bf16x8, q);
  }
  const int u0 = 16*w + 4*j;
  f32x4 dh = {0.f, 0.f, 0.f, 0.f};
  f32x4 dc = {0.f, 0.f, 0.f, 0.f};
  const bool rev = (reverse_mask >> grp) & 1;
  const float* dyrow = dy + (long long)(live ? cl : per_group - 1)*T*dy_ld + grp*dy_goff + u0;
  for (int t = T - 1; t >= 0; --t) {
    const long long step = chain*T + t;
    const int tt = rev ? T - 1 - t : t;
    f32x4 a[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) a[b] = load4(act, step*4*LH + 4*(u0 + b), IO16);
    const f32x4 cv = *reinterpret_cast<const f32x4*>(cs + step*LH + u0);
    f32x4 cp = {0.f, 0.f, 0.f, 0.f};
    if (t > 0) cp = *reinterpret_cast<const f32x4*>(cs + (step - 1)*LH + u0);
    const f32x4 dyv = *reinterpret_cast<const f32x4*>(dyrow + tt*dy_ld);
    f32x4 d[4];                                   // d[b] = the four gate gradients of unit u0 + b
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const float ig = a[b][0], fg = a[b][1], gg = a[b][2], og = a[b][3];
      const float tc = tanh_fast(cv[b]);
      const float dht = dh[b] + dyv[b];
      const float dct = dc[b] + dht*og*(1.f - tc*tc);
      d[b] = f32x4{dct*gg*ig*(1.f - ig), dct*cp[b]*fg*(1.f - fg), dct*ig*(1.f - gg*gg),
                   dht*tc*og*(1.f - og)};
      dc[b] = dct*fg;
    }
    // k = 4 (u0 + b) + g: 16 consecutive bf16 of this chain's row
    uint16_t* gw = gbuf[t & 1] + n*GROW + 4*u0;
    *reinterpret_cast<uint4*>(gw) = make_uint4(pack2(d[0][0], d[0][1]), pack2(d[0][2], d[0][3]),
                                               pack2(d[1][0], d[1][1]), pack2(d[1][2], d[1][3]));
    *reinterpret_cast<uint4*>(gw + 8) = make_uint4(pack2(d[2][0], d[2][1]), pack2(d[2][2], d[2][3]),
                                                   pack2(d[3][0], d[3][1]), pack2(d[3][2], d[3][3]));
    if (live) {
#pragma unroll
      for (int b = 0; b < 4; ++b)
        store4(dgates, (chain*T + tt)*4*LH + 4*(u0 + b), d[b], IO16);
    }
    __syncthreads();
    if (t == 0) break;
    f32x4 p[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const uint16_t* gr = gbuf[t & 1] + n*GROW + 8*j;
#pragma unroll
    for (int kq = 0; kq < 4*LH/32; ++kq) {
      const bf16x8 gv = *reinterpret_cast<const bf16x8*>(gr + 32*kq);
      p[kq & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kq], gv, p[kq & 1], 0, 0, 0);
    }
    dh = p[0] + p[1];
  }
}

// ---- four chains per workgroup (use_amp, round 5) ---------------------------------------------------------------
// The 16-chain kernels above are bound by their gate math: a lane owns four (chain, unit) pairs -- 40 exponentials
// and reciprocals per step and wave, 2.2 us per forward step -- and TF-GridNet's 1 056 .. 4 032 chains make only 66 ..
// 252 workgroups. Here the roles of the MFMA operands are swapped (as in dccrn.hip's lstm_*_mv_kernel): A = the hidden
// states, row m = chain m % 4 (every chain four times), B = 16 weight rows per instruction held in registers; lane
// (j = lane >> 4, n = lane & 15) of wave w then finds the pre-activations of ALL four chains of unit 16 w + n in the four
// registers of an accumulator and keeps chain j: ONE (chain, unit) pair per lane, the same 16 instructions per wave
// and step, four times the workgroups. Interleaved gate layout as above (one 16- or 8-byte load / store per lane and
// tensor); per-step loads invisible to hipcc's wait insertion, requested three steps ahead, counted vmcnt (only the
// younger LOADS may be outstanding), LDS-only step barrier.
constexpr int QC = 4;                 // chains per workgroup
typedef unsigned int q4_u32x2 __attribute__((ext_vector_type(2)));
template <bool IO16> struct Q4Raw { typedef f32x4 type; };
template <> struct Q4Raw<true> { typedef q4_u32x2 type; };
// Round 6: plain, compiler-tracked loads. Until then these were assembly loads the compiler could not see, waited for
// by a hand-counted vmcnt three steps later -- but a register the compiler does not know to be in flight may be copied
// (the rotating register sets need phi copies at the loop's back edge) or reused before the load has landed: see
// csrc/dccrn.hip `dma_dword`, where the same scheme produced wrong LSTM gradients in one run of three under HBM load.
// The hand-counted waits below stay (they are no-ops behind the compiler's own).
__device__ __forceinline__ void q4_load(f32x4& v, const float* p, long long i) {
  v = *reinterpret_cast<const f32x4*>(p + i);
}
__device__ __forceinline__ void q4_load(q4_u32x2& v, const float* p, long long i) {
  v = *reinterpret_cast<const q4_u32x2*>(reinterpret_cast<const uint16_t*>(p) + i);
}
__device__ __forceinline__ void q4_load1(float& v, const float* p, long long i) { v = p[i]; }
__device__ __forceinline__ f32x4 q4_value(const f32x4& v) { return v; }
__device__ __forceinline__ f32x4 q4_value(const q4_u32x2& u) {
  return f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
               __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
}
__device__ __forceinline__ void q4_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ float q4_pick(const f32x4& v, int j) {
  return j == 0 ? v[0] : j == 1 ? v[1] : j == 2 ? v[2] : v[3];
}

template <bool IO16>
__global__ __launch_bounds__(512) void lstm_q4_fwd_kernel(const float* __restrict__ gates_in,
                                                          const float* __restrict__ w_hh,
                                                          const float* __restrict__ bias,
                                                          float* __restrict__ y, float* __restrict__ act,
                                                          float* __restrict__ cs, int per_group, int T,
                                                          int reverse_mask, long long y_ld, long long y_goff) {
  typedef typename Q4Raw<IO16>::type Raw;
  __shared__ __attribute__((aligned(16))) uint16_t hb[2][QC*HROW];
  const int tiles = (per_group + QC - 1)/QC;
  const int grp = blockIdx.x / tiles, tile = blockIdx.x % tiles;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int n = lane & 15, j = lane >> 4, u = 16*w + n;            // this lane: chain j of the tile, unit u
  const int cl = tile*QC + j;
  const bool live = cl < per_group;
  const long long chain = (long long)grp*per_group + (live ? cl : per_group - 1);
  w_hh += (long long)grp*4*LH*LH;
  // B[k = 32 kq + 8 j + i][n] = W_hh[g H + u][k]
  bf16x8 wf[4][LH/32];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int kq = 0; kq < LH/32; ++kq) wf[g][kq] = load_frag(w_hh + (long long)(g*LH + u)*LH + 32*kq + 8*j, 1);
  f32x4 bs;
#pragma unroll
  for (int g = 0; g < 4; ++g) bs[g] = bias ? bias[(long long)grp*4*LH + g*LH + u] : 0.f;
  float c = 0.f;
  for (int i = tid; i < 2*QC*HROW; i += 512) (&hb[0][0])[i] = 0;
  const bool rev = (reverse_mask >> grp) & 1;
  const int t_first = rev ? T - 1 : 0, t_inc = rev ? -1 : 1;
  const long long gbase = chain*T*4*LH + 4*u;
  float* yrow = y + (long long)(live ? cl : per_group - 1)*T*y_ld + grp*y_goff + u;
  auto fetch = [&](int t, Raw& r) {
    const int tt = t_first + (t < T ? t : T - 1)*t_inc;
    q4_load(r, gates_in, gbase + (long long)tt*4*LH);
  };
  // per step: 1 load (step t + 3), then the stores: 3 younger loads may stay in flight
  auto wait_set = [](Raw& r) { asm volatile("s_waitcnt vmcnt(3)" : "+v"(r) :: "memory"); };
  // A fragment row m = chain m % 4: lane (m = lane & 15, j) reads 16 bytes of chain (lane & 3)
  const int arow = (lane & 3)*HROW + 8*j;
  auto step = [&](Raw& r, int t) {
    wait_set(r);
    const int tt = t_first + t*t_inc;
    const uint16_t* hp = hb[t & 1] + arow;
    f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kq = 0; kq < LH/32; ++kq) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(hp + 32*kq);
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, wf[g][kq], acc[g], 0, 0, 0);
    }
    const f32x4 gin = q4_value(r);
    const float ig = sigm(q4_pick(acc[0], j) + gin[0] + bs[0]), fg = sigm(q4_pick(acc[1], j) + gin[1] + bs[1]);
    const float gg = tanh_fast(q4_pick(acc[2], j) + gin[2] + bs[2]), og = sigm(q4_pick(acc[3], j) + gin[3] + bs[3]);
    c = fg*c + ig*gg;
    const float hn = og*tanh_fast(c);
    hb[(t + 1) & 1][j*HROW + u] = brv::f2bf(hn);
    if (live) {
      const long long st = chain*T + t;
      yrow[tt*y_ld] = hn;
      if (act) {
        cs[st*LH + u] = c;
        store4(act, st*4*LH + 4*u, f32x4{ig, fg, gg, og}, IO16);
      }
    }
    q4_barrier();
  };
  Raw s0, s1, s2, s3;
  fetch(0, s0); fetch(1, s1); fetch(2, s2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the counted wait needs full queues behind it)
  __syncthreads();
  int t = 0;
  while (true) {
    fetch(t + 3, s3); step(s0, t); if (++t >= T) break;
    fetch(t + 3, s0); step(s1, t); if (++t >= T) break;
    fetch(t + 3, s1); step(s2, t); if (++t >= T) break;
    fetch(t + 3, s2); step(s3, t); if (++t >= T) break;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <bool IO16>
__global__ __launch_bounds__(512) void lstm_q4_bwd_kernel(const float* __restrict__ act,
                                                          const float* __restrict__ cs,
                                                          const float* __restrict__ w_hh,
                                                          const float* __restrict__ dy,
                                                          float* __restrict__ dgates, int per_group, int T,
                                                          int reverse_mask, long long dy_ld, long long dy_goff) {
  typedef typename Q4Raw<IO16>::type Raw;
  constexpr int K = 4*LH, GR = K + 8;              // gate gradients of a chain, k = gate H + unit (rows 16 B apart in banks)
  __shared__ __attribute__((aligned(16))) uint16_t dgb[2][QC*GR];
  __shared__ float part[8][QC][LH];
  const int tiles = (per_group + QC - 1)/QC;
  const int grp = blockIdx.x / tiles, tile = blockIdx.x % tiles;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int n = lane & 15, j = lane >> 4, u = 16*w + n;            // this lane: chain j of the tile, unit u
  const int cl = tile*QC + j;
  const bool live = cl < per_group;
  const long long chain = (long long)grp*per_group + (live ? cl : per_group - 1);
  w_hh += (long long)grp*4*LH*LH;
  // the reduction (K = 4 H) is split over the waves: wave w multiplies k = 64 w .. 64 w + 63 against all 128 units;
  // B[k = 64 w + 32 ks + 8 j + i][n] = W_hh[k][16 nb + n]
  bf16x8 wf[8][2];
#pragma unroll
  for (int nb = 0; nb < 8; ++nb)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      wf[nb][ks] = load_frag(w_hh + (long long)(64*w + 32*ks + 8*j)*LH + 16*nb + n, LH);
  for (int i = tid; i < 2*QC*GR; i += 512) (&dgb[0][0])[i] = 0;
  float dc = 0.f;
  const bool rev = (reverse_mask >> grp) & 1;
  const float* dyrow = dy + (long long)(live ? cl : per_group - 1)*T*dy_ld + grp*dy_goff + u;
  struct Saved { Raw a; float cv, cp, dyv; };
  auto fetch = [&](int t, Saved& v) {
    const int tc = t > 0 ? t : 0;
    const long long st = chain*T + tc;
    const int tt = rev ? T - 1 - tc : tc;
    q4_load(v.a, act, st*4*LH + 4*u);
    q4_load1(v.cv, cs, st*LH + u);
    q4_load1(v.cp, cs, (tc > 0 ? st - 1 : st)*LH + u);
    q4_load1(v.dyv, dyrow, (long long)tt*dy_ld);
  };
  // per step: 4 loads (step t - 3), then one store: 12 younger loads may stay in flight
  auto wait_set = [](Saved& v) {
    asm volatile("s_waitcnt vmcnt(12)" : "+v"(v.a), "+v"(v.cv), "+v"(v.cp), "+v"(v.dyv) :: "memory");
  };
  const int arow = (lane & 3)*GR + 64*w + 8*j;
  auto step = [&](Saved& cur, int t) {
    // this wave's slice of the gate gradients of step t + 1 (zeros for the last step), all four chains
    const uint16_t* gp = dgb[(t + 1) & 1] + arow;
    const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(gp), a1 = *reinterpret_cast<const bf16x8*>(gp + 32);
    f32x4 p[8];
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
      p[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, wf[nb][0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) p[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, wf[nb][1], p[nb], 0, 0, 0);
    // every lane group holds all eight blocks and (registers 0 .. 3) all four chains: group j hands over blocks
    // 2 j, 2 j + 1
    const f32x4 e0 = j == 0 ? p[0] : j == 1 ? p[2] : j == 2 ? p[4] : p[6];
    const f32x4 e1 = j == 0 ? p[1] : j == 1 ? p[3] : j == 2 ? p[5] : p[7];
#pragma unroll
    for (int cc = 0; cc < QC; ++cc) { part[w][cc][32*j + n] = e0[cc]; part[w][cc][32*j + 16 + n] = e1[cc]; }
    q4_barrier();
    wait_set(cur);
    const int tt = rev ? T - 1 - t : t;
    float dht = cur.dyv;
#pragma unroll
    for (int i = 0; i < 8; ++i) dht += part[i][j][u];
    const f32x4 a = q4_value(cur.a);
    const float ig = a[0], fg = a[1], gg = a[2], og = a[3];
    const float cprev = t > 0 ? cur.cp : 0.f;
    const float tc = tanh_fast(cur.cv);
    const float dct = dc + dht*og*(1.f - tc*tc);
    const f32x4 d = {dct*gg*ig*(1.f - ig), dct*cprev*fg*(1.f - fg), dct*ig*(1.f - gg*gg), dht*tc*og*(1.f - og)};
    uint16_t* gw = dgb[t & 1] + j*GR + u;
#pragma unroll
    for (int g = 0; g < 4; ++g) gw[g*LH] = brv::f2bf(d[g]);
    if (live) store4(dgates, (chain*T + tt)*4*LH + 4*u, d, IO16);
    dc = dct*fg;
    q4_barrier();
  };
  Saved s0, s1, s2, s3;
  fetch(T - 1, s0); fetch(T - 2, s1); fetch(T - 3, s2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int t = T - 1;
  while (true) {
    fetch(t - 3, s3); step(s0, t); if (--t < 0) break;
    fetch(t - 3, s0); step(s1, t); if (--t < 0) break;
    fetch(t - 3, s1); step(s2, t); if (--t < 0) break;
    fetch(t - 3, s2); step(s3, t); if (--t < 0) break;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

#ifndef BRV_LSTM_Q4
#define BRV_LSTM_Q4 1          // diagnostic builds: 0 = always the 16-chain kernels, 2 = always four chains per workgroup
#endif
// Four chains per workgroup or sixteen? One workgroup per CU either way (8 waves at 150 - 190 registers); a step
// costs ~0.72 us with four chains and ~2.4 us with sixteen (measured on TF-GridNet's two shapes: 1 056 chains x 126
// steps: 181 against 277 us forward; 4 032 chains x 33 steps: 144 against 103 us), and a launch runs
// ceil(workgroups / CUs) rounds of T steps.
int q4_pays(int64_t B, int64_t groups) {
  if (BRV_LSTM_Q4 != 1) return BRV_LSTM_Q4 == 2;
  static int cus = 0;
  if (!cus) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        n < 1) n = 256;
    cus = n;
  }
  const int64_t per_group = B/groups;
  const int64_t wg4 = groups*((per_group + QC - 1)/QC), wg16 = groups*((per_group + LC - 1)/LC);
  const int64_t r4 = (wg4 + cus - 1)/cus, r16 = (wg16 + cus - 1)/cus;
  return 0.72*(double)r4 < 2.4*(double)r16;
}

}  // namespace

extern "C" {

int brv_lstm_tile_supported(int64_t H) { return H == LH ? 1 : 0; }

int brv_lstm_tile_forward(const float* gates_in, const float* w_hh, const float* bias, float* y,
                          float* act, float* cs, int64_t B, int64_t T, int64_t H, int64_t groups,
                          int64_t reverse_mask, int64_t y_ld, int64_t y_group_offset, int lowp,
                          brv_stream_t stream) {
  if (H != LH || B < 1 || T < 1 || groups < 1 || groups > 30 || B % groups) return -1;
  const int per_group = (int)(B/groups);
  const unsigned grid = (unsigned)(groups*((per_group + LC - 1)/LC));
  if (q4_pays(B, groups) && lowp) {
    const unsigned g4 = (unsigned)(groups*((per_group + QC - 1)/QC));
    if (lowp == 2)
      hipLaunchKernelGGL(lstm_q4_fwd_kernel<true>, dim3(g4), dim3(512), 0, (hipStream_t)stream, gates_in, w_hh, bias,
                         y, act, cs, per_group, (int)T, (int)reverse_mask, (long long)y_ld, (long long)y_group_offset);
    else
      hipLaunchKernelGGL(lstm_q4_fwd_kernel<false>, dim3(g4), dim3(512), 0, (hipStream_t)stream, gates_in, w_hh, bias,
                         y, act, cs, per_group, (int)T, (int)reverse_mask, (long long)y_ld, (long long)y_group_offset);
    return (int)hipGetLastError();
  }
  if (lowp == 2)
    hipLaunchKernelGGL(lstm_tile_fwd_bf16_kernel<true>, dim3(grid), dim3(512), 0, (hipStream_t)stream,
                       gates_in, w_hh, bias, y, act, cs, per_group, (int)T, (int)reverse_mask,
                       (long long)y_ld, (long long)y_group_offset);
  else if (lowp)
    hipLaunchKernelGGL(lstm_tile_fwd_bf16_kernel<false>, dim3(grid), dim3(512), 0, (hipStream_t)stream,
                       gates_in, w_hh, bias, y, act, cs, per_group, (int)T, (int)reverse_mask,
                       (long long)y_ld, (long long)y_group_offset);
  else
  hipLaunchKernelGGL(lstm_tile_fwd_kernel, dim3(grid), dim3(512), 0, (hipStream_t)stream, gates_in,
                     w_hh, bias, y, act, cs, per_group, (int)T, (int)reverse_mask, (long long)y_ld,
                     (long long)y_group_offset);
  return (int)hipGetLastError();
}

int brv_lstm_tile_backward(const float* act, const float* cs, const float* w_hh, const float* dy,
                           float* dgates, int64_t B, int64_t T, int64_t H, int64_t groups,
                           int64_t reverse_mask, int64_t dy_ld, int64_t dy_group_offset, int lowp,
                           brv_stream_t stream) {
  if (H != LH || B < 1 || T < 1 || groups < 1 || groups > 30 || B % groups) return -1;
  const int per_group = (int)(B/groups);
  const unsigned grid = (unsigned)(groups*((per_group + LC - 1)/LC));
  if (q4_pays(B, groups) && lowp) {
    const unsigned g4 = (unsigned)(groups*((per_group + QC - 1)/QC));
    if (lowp == 2)
      hipLaunchKernelGGL(lstm_q4_bwd_kernel<true>, dim3(g4), dim3(512), 0, (hipStream_t)stream, act, cs, w_hh, dy,
                         dgates, per_group, (int)T, (int)reverse_mask, (long long)dy_ld, (long long)dy_group_offset);
    else
      hipLaunchKernelGGL(lstm_q4_bwd_kernel<false>, dim3(g4), dim3(512), 0, (hipStream_t)stream, act, cs, w_hh, dy,
                         dgates, per_group, (int)T, (int)reverse_mask, (long long)dy_ld, (long long)dy_group_offset);
    return (int)hipGetLastError();
  }
  if (lowp == 2)
    hipLaunchKernelGGL(lstm_tile_bwd_bf16_kernel<true>, dim3(grid), dim3(512), 0, (hipStream_t)stream,
                       act, cs, w_hh, dy, dgates, per_group, (int)T, (int)reverse_mask,
                       (long long)dy_ld, (long long)dy_group_offset);
  else if (lowp)
    hipLaunchKernelGGL(lstm_tile_bwd_bf16_kernel<false>, dim3(grid), dim3(512), 0, (hipStream_t)stream,
                       act, cs, w_hh, dy, dgates, per_group, (int)T, (int)reverse_mask,
                       (long long)dy_ld, (long long)dy_group_offset);
  else
  hipLaunchKernelGGL(lstm_tile_bwd_kernel, dim3(grid), dim3(512), 0, (hipStream_t)stream, act, cs,
                     w_hh, dy, dgates, per_group, (int)T, (int)reverse_mask, (long long)dy_ld,
                     (long long)dy_group_offset);
  return (int)hipGetLastError();
}

}  // extern "C"
