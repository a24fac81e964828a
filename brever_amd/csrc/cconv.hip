// The (5, 2) / stride (2, 1) / padding (2, 0) convolutions of DCCRN (ComplexWrapper(nn.Conv2d |
// nn.ConvTranspose2d), reference brever/models/dccrn/dccrn.py:225-235, 238-292) on fp32 (B, C, H, W)
// images, W = frames contiguous, as implicit GEMMs on v_mfma_f32_32x32x16_bf16: no column matrix in
// HBM, no scatter pass.
//
//   strided form  (encoder forward, decoder data gradient):
//       out[m][r][w] = bias[m] + sum_{c,i,j} W[m][c][i][j] in[c][2r - 2 + i][w + j]        Hout = Hin/2, Wout = Win - 1
//   transposed form (decoder forward, encoder data gradient):
//       out[m][r][w] = bias[m] + sum_{c, i = r (mod 2), j} W[m][c][i][j] in[c][(r + 2 - i)/2][w - j]   Hout = 2 Hin, Wout = Win + 1
//
// One workgroup (512 threads, 8 waves) = one output row r of one batch item x NT output frames x MT
// output channels. Per chunk of 8 input channels each contributing input row ("tap" i) is staged ONCE:
// fp32 rows are read along the frames (one wave = 1 KB of one channel row), rounded to bf16 and kept in
// LDS as TWO images, the row itself and the row shifted by one frame (the j = 1 tap) -- one MFMA k step
// (16) = 8 channels x 2 frame taps, so the frame shift never meets the 8-byte alignment rule of the
// transposing LDS reads (cdna_hip_programming.md T10) that turn the frame-contiguous rows into B
// operands. The weights are prepacked into A-operand fragments (cconv_pack_kernel) and go from L2
// straight into registers, two k steps ahead of their use. LDS images are double-buffered: one
// barrier per chunk (5 / 3 / 2 k steps of MF x NF MFMAs per wave). Two loaders (cconv_tile, cconv_tile_lean) and,
// for the transposed form with M <= 128, a pair form (both output rows of a pair from the three input rows they
// share) are selected per launch shape in cconv_launch; workgroups are dealt to the 8 XCDs in contiguous runs.
// Weight gradient: cconv_wgrad_kernel below. Measurements and what was tried: DESIGN.md 5b.
#include <string.h>

#include <type_traits>

#include "common.cuh"
#include "../../include/brever_hip.h"

namespace {
using namespace brv;

constexpr int CC_THREADS = 512;
#ifndef CC_DMA
#define CC_DMA 1       // bf16 images staged by LDS-DMA (0: through registers, cconv_tile_lean -- for A/B runs)
#endif
#ifndef CC_ABL
#define CC_ABL 0       // diagnostic builds (tools/cconv_bench.py): 1 no image loads, 2 no weight loads, 4 no MFMAs, 8 no stores,
                       // 16 image loads from addresses rounded down to 16 bytes, 32 every chunk re-reads the first rows of the item
#endif
#ifndef CC_PAIR
#define CC_PAIR 1      // transposed form with M <= 128: two output rows per workgroup (0: one, for A/B runs)
#endif
constexpr int CC_KH = 5;

struct CConvParams {
  const void* in; const void* in2;   // fp32 or bf16 elements (the kernel's TI)
  const uint4* wp; const float* bias;
  void* out; void* out2;       // fp32, or bf16 with out_bf16 (the block's batch norm reads it: brv_batchnorm2d_*_bf16io)
  int out_bf16;
  int in_seg, out_seg;         // > 0: channels = [t[:seg] | t2[:seg] | t[seg:] | t2[seg:]] of two (B, 2 seg, H, W) tensors
  int B, C, M, Hin, Win, Hout, Wout;
  long long in_bs, out_bs;
  int mode;                    // 0 strided, 1 transposed
  int ncc;                     // chunks of 8 input channels
  int mtiles, ftiles, mfrags;   // mfrags: 32-row groups the packed weights hold
};

__device__ __forceinline__ int cc_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
// byte offset of 16-byte chunk `ch` (0..15) of row `row` of a [16 rows][128 x bf16] image (layout (b) of T10)
__device__ __forceinline__ int cc_off(int row, int ch) { return 256*row + 16*(ch ^ cc_swz(row)); }

// B fragment of v_mfma_f32_32x32x16_bf16 from one image: lane (n = lane & 31, g = lane >> 5) gets
// image[8 g + e][col0 + n], e = 0..7 (col0 a multiple of 32 inside the 128-column image)
__device__ __forceinline__ bf16x8 cc_frag(const unsigned char* img, int col0, int lane) {
  const int g4 = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
  const int chunk = (col0 >> 3) + 2*(g4 & 1) + (pp >> 1);
  const int row = 8*(g4 >> 1) + q;
  typedef __attribute__((address_space(3))) s16x4* lds_p;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + cc_off(row, chunk) + 8*(pp & 1)));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + cc_off(row + 4, chunk) + 8*(pp & 1)));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

// ---- four neighbouring frames of one input row in registers, for both element types the images come in: fp32
// (rounded to bf16 on the way into LDS) or bf16 as the batch-norm passes store them under use_amp
// (brv_batchnorm2d_forward_bf16 / brv_batchnorm2d_backward_bf16: the values the fp32 form rounds to here, at half
// the bytes -- these kernels are bound by what a CU can take in, DESIGN.md 5b "Round 6, DCCRN")
template <typename T> struct CQuad;
template <> struct CQuad<float> {
  typedef float4 type;
  typedef float vec_u __attribute__((ext_vector_type(4), aligned(4)));
  static __device__ __forceinline__ type zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
  static __device__ __forceinline__ type load(const void* p) {
    const vec_u w = *reinterpret_cast<const vec_u*>(p);
    return make_float4(w.x, w.y, w.z, w.w);
  }
  // frames f .. f + 3 of a row of n frames, those outside as zeros
  static __device__ __forceinline__ type gather(const float* src, int f, int n) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (f >= 0 && f < n) v.x = src[f];
    if (f + 1 >= 0 && f + 1 < n) v.y = src[f + 1];
    if (f + 2 >= 0 && f + 2 < n) v.z = src[f + 2];
    if (f + 3 >= 0 && f + 3 < n) v.w = src[f + 3];
    return v;
  }
  // kind 6: up by one frame; 3 / 4 / 5: down by one / two / three (cconv_tile_lean)
  static __device__ __forceinline__ type moved(type v, int k) {
    if (k == 6) return make_float4(0.f, v.x, v.y, v.z);
    if (k == 3) return make_float4(v.y, v.z, v.w, 0.f);
    if (k == 4) return make_float4(v.z, v.w, 0.f, 0.f);
    return make_float4(v.w, 0.f, 0.f, 0.f);
  }
  static __device__ __forceinline__ uint2 packed(type v) { return make_uint2(pack2(v.x, v.y), pack2(v.z, v.w)); }
  // (frame before the quad, x, y, z) as bf16: that frame is the last one of lane - 1's quad (DPP row_shr:1), or
  // `left` where `edge` says the neighbouring lane holds another row
  static __device__ __forceinline__ uint2 shifted(type v, float left, bool edge) {
    float prev = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v.w), 0x111, 0xf, 0xf, false));
    if (edge) prev = left;
    return make_uint2(pack2(prev, v.x), pack2(v.y, v.z));
  }
};
template <> struct CQuad<bf16_t> {
  typedef uint2 type;
  typedef unsigned int vec_u __attribute__((ext_vector_type(2), aligned(2)));
  static __device__ __forceinline__ type zero() { return make_uint2(0u, 0u); }
  static __device__ __forceinline__ type load(const void* p) {
    const vec_u w = *reinterpret_cast<const vec_u*>(p);
    return make_uint2(w.x, w.y);
  }
  static __device__ __forceinline__ type gather(const bf16_t* src, int f, int n) {
    unsigned int e[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) e[k] = (f + k >= 0 && f + k < n) ? (unsigned int)src[f + k] : 0u;
    return make_uint2(e[0] | e[1] << 16, e[2] | e[3] << 16);
  }
  static __device__ __forceinline__ type moved(type v, int k) {
    const unsigned long long w = (unsigned long long)v.x | (unsigned long long)v.y << 32;
    const unsigned long long r = k == 6 ? w << 16 : k == 3 ? w >> 16 : k == 4 ? w >> 32 : w >> 48;
    return make_uint2((unsigned int)r, (unsigned int)(r >> 32));
  }
  static __device__ __forceinline__ uint2 packed(type v) { return v; }
  static __device__ __forceinline__ uint2 shifted(type v, bf16_t left, bool edge) {
    unsigned int prev = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v.y, 0x111, 0xf, 0xf, false) >> 16;
    if (edge) prev = left;
    return make_uint2(prev | v.x << 16, v.x >> 16 | v.y << 16);
  }
};

// wc: fp32 matrix with W[m][c][i][j] = wc[m*sm + c*sk + 2 i + j]; wp[((mf*ncc + cc)*5 + i)*64 + lane] =
// the 8 channels 8 cc .. 8 cc + 7 of output row 32 mf + (lane & 31), frame tap j = lane >> 5
__global__ __launch_bounds__(256) void cconv_pack_kernel(const float* wc, uint4* wp, int M, int C, long long sm,
                                                         long long sk, int ncc, long long total) {
  const long long e = (long long)blockIdx.x*256 + threadIdx.x;
  if (e >= total) return;
  const int lane = (int)(e & 63);
  long long t = e >> 6;
  const int i = (int)(t % CC_KH); t /= CC_KH;
  const int cc = (int)(t % ncc);
  const int mf = (int)(t / ncc);
  const int m = 32*mf + (lane & 31), j = lane >> 5;
  float v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = 8*cc + k;
    v[k] = (m < M && c < C) ? wc[m*sm + c*sk + 2*i + j] : 0.f;
  }
  wp[e] = pack8(v);
}

// One launch for everything a complex layer's forward AND backward need of its weights (three launches in front of
// every forward and one in front of every data gradient before -- 44 one-wave-sized launches per DCCRN step,
// profiles/r05_dccrn_trace.txt): the real matrix wc (2R x 2C) = [[wr, -s wi], [s wi, wr]] of
// brv_complex_weight_pack, the bias [br - bi | br + bi] of brv_complex_bias_pack, and the operand fragments of
// brv_cconv_pack for up to two (M, C, m_stride, c_stride) readings of wc, built from wr / wi directly.
struct CPackJob { uint4* wp; int M, C, ncc; unsigned int sm, sk, total; };
struct CPackAllParams {
  const float* wr; const float* wi; const float* br; const float* bi;
  float* wc; float* bias; int R, C, Cb; float s;
  CPackJob job[2];
};
__device__ __forceinline__ float cpack_wc(const CPackAllParams& p, unsigned int idx) {
  const unsigned int ld = 2u*p.C;
  const unsigned int row = idx / ld, col = idx - row*ld;
  const unsigned int r = row >= (unsigned)p.R ? row - p.R : row, c = col >= (unsigned)p.C ? col - p.C : col;
  const bool lower = row >= (unsigned)p.R, right = col >= (unsigned)p.C;
  const unsigned int src = r*p.C + c;
  return lower == right ? p.wr[src] : (right ? -p.s : p.s)*p.wi[src];
}
__global__ __launch_bounds__(256) void cconv_pack_all_kernel(const CPackAllParams p) {
  const unsigned int e = blockIdx.x*256u + threadIdx.x;
  if (e < 4u*p.R*p.C) p.wc[e] = cpack_wc(p, e);
  if (e < (unsigned)p.Cb) { p.bias[e] = p.br[e] - p.bi[e]; p.bias[p.Cb + e] = p.br[e] + p.bi[e]; }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const CPackJob& j = p.job[q];
    if (!j.wp || e >= j.total) continue;
    const unsigned int lane = e & 63u;
    unsigned int t = e >> 6;
    const unsigned int i = t % CC_KH; t /= CC_KH;
    const unsigned int cc = t % (unsigned)j.ncc, mf = t / (unsigned)j.ncc;
    const unsigned int m = 32u*mf + (lane & 31u), jj = lane >> 5;
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const unsigned int c = 8u*cc + k;
      v[k] = (m < (unsigned)j.M && c < (unsigned)j.C) ? cpack_wc(p, m*j.sm + c*j.sk + 2u*i + jj) : 0.f;
    }
    j.wp[e] = pack8(v);
  }
}

template <typename TI, int MF, int NF, int WM, int WN, int NTAP, bool SEG>
__device__ __forceinline__ void cconv_tile(const CConvParams& p, unsigned char* lds, int b, int r, int ftile,
                                           int mtile, const int (&tap_i)[NTAP], const int (&tap_row)[NTAP],
                                           int shift) {
  constexpr int NT = 32*NF*WN;               // output frames per workgroup
  constexpr int TILES = NT/128;              // 128-column images side by side
  constexpr int IT = NT/128;                 // 16-byte staging items per thread and tap
  constexpr int TAPB = TILES*4096, BUFB = CC_KH*TAPB;
  static_assert(WM*WN == 8 && NT % 128 == 0, "8 waves, whole images");
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int f0 = ftile*NT;
  typedef CQuad<TI> Q;
  const TI* in_b = static_cast<const TI*>(p.in) + (long long)b*p.in_bs;
  const TI* in2_b = SEG && p.in_seg > 0 ? static_cast<const TI*>(p.in2) + (long long)b*p.in_bs : nullptr;

  // ---- staging items: (image, channel, frame quad) -> source frame and LDS byte offset
  int s_c[IT], s_f[IT], s_off[IT];
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int e = tid + CC_THREADS*it;
    const int img = e / (2*NT), rem = e % (2*NT);
    const int c = rem / (NT/4), q = rem % (NT/4);
    const int col = 4*q, cw = col & 127;
    s_c[it] = c;
    s_f[it] = f0 + col + shift*img;
    s_off[it] = (col >> 7)*4096 + cc_off(8*img + c, cw >> 3) + 8*((cw >> 2) & 1);
  }
  typename Q::type st[NTAP][IT];
  auto stage_load = [&](int cc) {
    // source tensor and channel offset of this chunk of 8 (never straddles two segments: seg % 8 == 0)
    const TI* base = in_b;
    int ch0 = 8*cc;
    if (SEG && p.in_seg > 0) {
      const int sg = (ch0 >= p.in_seg) + (ch0 >= 2*p.in_seg) + (ch0 >= 3*p.in_seg);
      ch0 -= ((sg + 1) >> 1)*p.in_seg;
      if (sg & 1) base = in2_b;
    }
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
      const int row = tap_row[t];
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        const int f = s_f[it];
        typename Q::type v = Q::zero();
        if (row >= 0 && row < p.Hin && 8*cc + s_c[it] < p.C) {
          const TI* src = base + ((long long)(ch0 + s_c[it])*p.Hin + row)*p.Win;
          if (f >= 0 && f + 3 < p.Win) v = Q::load(src + f);
          else v = Q::gather(src, f, p.Win);
        }
        st[t][it] = v;
      }
    }
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
      for (int it = 0; it < IT; ++it)
        *reinterpret_cast<uint2*>(lds + buf*BUFB + t*TAPB + s_off[it]) = Q::packed(st[t][it]);
  };

  // ---- weights: fragments of this wave's MF row groups, (chunk, tap) two k steps ahead
  const int mfrag0 = (mtile*WM + wm)*MF;
  const uint4* wq[MF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) {         // row groups past M: any packed group (their outputs are not stored)
    const int fr = mfrag0 + mf < p.mfrags ? mfrag0 + mf : p.mfrags - 1;
    wq[mf] = p.wp + (long long)fr*p.ncc*CC_KH*64 + lane;
  }
  auto a_load = [&](int cc, int t, uint4 (&dst)[MF]) {
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      if (CC_ABL & 2) dst[mf] = make_uint4(cc, t, mf, 0); else
      dst[mf] = wq[mf][(cc*CC_KH + tap_i[t])*64];
    }
  };

  f32x16 acc[MF][NF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mf][nf][i] = 0.f;

  uint4 a0[MF], a1[MF], a2[MF];
  a_load(0, 0, a0);
  if (NTAP > 1) a_load(0, 1 % NTAP, a1); else if (p.ncc > 1) a_load(1, 0, a1);
  stage_load(0);
  stage_store(0);
  __syncthreads();

  for (int cc = 0; cc < p.ncc; ++cc) {
    const unsigned char* cur = lds + (cc & 1)*BUFB;
    const bool more = cc + 1 < p.ncc;
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
      if (t + 2 < NTAP) a_load(cc, (t + 2) % NTAP, a2);
      else if (more) a_load(cc + 1, (t + 2) % NTAP, a2);
      if (t == 0 && more) stage_load(cc + 1);
      bf16x8 bq[NF];
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        const int col0 = 32*(wn*NF + nf);
        bq[nf] = cc_frag(cur + t*TAPB + (col0 >> 7)*4096, col0 & 127, lane);
      }
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        const bf16x8 af = __builtin_bit_cast(bf16x8, a0[mf]);
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
          acc[mf][nf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bq[nf], acc[mf][nf], 0, 0, 0);
      }
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) { a0[mf] = a1[mf]; a1[mf] = a2[mf]; }
    }
    if (more) stage_store((cc + 1) & 1);
    __syncthreads();
  }

  // ---- D[m][frame] -> out[b][m][r][frame] (+ bias)
  const int oes = p.out_bf16 ? 2 : 4;        // bytes per output element
  char* out_b = static_cast<char*>(p.out) + (long long)b*p.out_bs*oes;
  char* out2_b = SEG && p.out_seg > 0 ? static_cast<char*>(p.out2) + (long long)b*p.out_bs*oes : nullptr;
#pragma unroll
  for (int mf = 0; mf < MF; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      const int w = f0 + 32*(wn*NF + nf) + (lane & 31);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int m = 32*(mfrag0 + mf) + (i & 3) + 8*(i >> 2) + 4*(lane >> 5);
        if (m < p.M && w < p.Wout) {
          float v = acc[mf][nf][i];
          if (p.bias) v += p.bias[m];
          char* dst = out_b;
          int mc = m;
          if (SEG && p.out_seg > 0) {
            const int sg = (m >= p.out_seg) + (m >= 2*p.out_seg) + (m >= 3*p.out_seg);
            mc -= ((sg + 1) >> 1)*p.out_seg;
            if (sg & 1) dst = out2_b;
          }
          const long long idx = ((long long)mc*p.Hout + r)*p.Wout + w;
            if (p.out_bf16) *reinterpret_cast<bf16_t*>(dst + idx*2) = f2bf(v);
            else *reinterpret_cast<float*>(dst + idx*4) = v;
        }
      }
    }
}

// ---- the same tile with a leaner loader (LEAN): every staging item is ONE unconditional 16-byte load (quads that
// meet the row end load the last four frames of the row and are moved into place at store time, rows and quads
// outside the image are never stored: their LDS slots keep the zeros of the prologue), so that there is no
// second load path into the same registers and no control flow between loads -- either makes hipcc drain the
// whole load queue (s_waitcnt vmcnt(0)) in front of every load. Measured per launch shape (tools/dccrn_conv_bench.py):
// faster wherever the input comes from two tensors or M <= 64, slower on the 256-row tiles.
// PAIR (transposed form, M <= 128): one workgroup computes BOTH output rows 2r and 2r + 1 from the three input
// rows r + 1, r, r - 1 they share (slots 0..2; products t = 0..4 = weight taps 0..4, tap t reads slot t >> 1 and
// adds to the accumulator set t & 1): 1.5 staged rows per output row instead of 2.5.
template <typename TI, int MF, int NF, int WM, int WN, int NTAP, bool SEG, bool TINY, bool PAIR = false>
__device__ __forceinline__ void cconv_tile_lean(const CConvParams& p, unsigned char* lds, int b, int r, int ftile,
                                           int mtile, const int (&tap_i)[NTAP], const int (&tap_row)[PAIR ? 3 : NTAP],
                                           int shift) {
  constexpr int NSLOT = PAIR ? 3 : NTAP, NSET = PAIR ? 2 : 1;
  static_assert(!PAIR || NTAP == 5, "pair form: five taps");
  constexpr int NT = 32*NF*WN;               // output frames per workgroup
  constexpr int TILES = NT/128;              // 128-column images side by side
  constexpr int IT = NT/128;                 // 16-byte staging items per thread and tap
  constexpr int TAPB = TILES*4096, BUFB = CC_KH*TAPB;
  static_assert(WM*WN == 8 && NT % 128 == 0, "8 waves, whole images");
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int f0 = ftile*NT;
  typedef CQuad<TI> Q;
  constexpr int ES = (int)sizeof(TI);
  const TI* in_b = static_cast<const TI*>(p.in) + (long long)b*p.in_bs;
  const TI* in2_b = SEG && p.in_seg > 0 ? static_cast<const TI*>(p.in2) + (long long)b*p.in_bs : nullptr;

  // ---- staging items: (image, channel, frame quad). Everything that does not depend on the chunk or the tap
  // is computed once, and every item is ONE 16-byte load under its lane mask: a second (element-wise) path
  // into the same registers makes hipcc drain the load queue (s_waitcnt vmcnt(0)) in front of every load.
  // kind 0: no frame of the quad exists (its LDS slot keeps the zeros of the prologue); 1: the whole quad;
  // 2 + d: the row ends inside the quad: the last four frames of the row are loaded and moved down by d at
  // store time; 6: the quad starts at frame -1 (transposed form): frames 0..3 are loaded and moved up by one.
  const int plane = p.Hin*p.Win;
  int s_c[IT], s_off[IT], s_voff[IT], s_kind[IT];
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int e = tid + CC_THREADS*it;
    const int img = e / (2*NT), rem = e % (2*NT);
    const int c = rem / (NT/4), q = rem % (NT/4);
    const int col = 4*q, cw = col & 127;
    const int f = f0 + col + shift*img;
    int kind = 0, fl = 0;
    if (TINY) { kind = 1; fl = f; }
    else if (f >= 0 && f + 3 < p.Win) { kind = 1; fl = f; }
    else if (f >= 0 && f < p.Win) { kind = 2 + f - (p.Win - 4); fl = p.Win - 4; }
    else if (f == -1) { kind = 6; fl = 0; }
    s_c[it] = c;
    s_kind[it] = kind;
    s_voff[it] = kind ? (c*plane + fl)*ES : 0;      // lanes with nothing to load read the row base (ignored)
    s_off[it] = (col >> 7)*4096 + cc_off(8*img + c, cw >> 3) + 8*((cw >> 2) & 1);
  }
  bool tap_ok[NSLOT];
#pragma unroll
  for (int t = 0; t < NSLOT; ++t) tap_ok[t] = tap_row[t] >= 0 && tap_row[t] < p.Hin;
  // rows / quads that are never written hold zeros
  for (int e = tid; e < 2*BUFB/16; e += CC_THREADS) reinterpret_cast<uint4*>(lds)[e] = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();

  // bf16 images on the 256-row tile: the staged quads are half the registers, so TWO chunks are kept in flight there
  // (235 against 271 us on the decoder's 512 -> 256 layers; set = chunk & 1)
  constexpr int DEPTH = sizeof(TI) == 2 && MF*NF >= 8 ? 2 : 1;     // (the smaller tiles lose more to the lower occupancy)
  typename Q::type st[DEPTH][NSLOT][IT];
  auto stage_load = [&](int cc, auto set_tag) {
    constexpr int set = decltype(set_tag)::value;
    // source tensor and channel offset of this chunk of 8 (never straddles two segments: seg % 8 == 0)
    const TI* base = in_b;
    int ch0 = 8*cc;
    if (SEG && p.in_seg > 0) {
      const int sg = (ch0 >= p.in_seg) + (ch0 >= 2*p.in_seg) + (ch0 >= 3*p.in_seg);
      ch0 -= ((sg + 1) >> 1)*p.in_seg;
      if (sg & 1) base = in2_b;
    }
    const int nch = p.C - 8*cc;                     // channels of this chunk that exist (>= 8 but for the last one)
    const char* cbase = reinterpret_cast<const char*>(base + (long long)ch0*plane);
#pragma unroll
    for (int t = 0; t < NSLOT; ++t) {
      // no branch around a load (rows outside the image read row 0 and are not stored): with control flow
      // between loads hipcc falls back to s_waitcnt vmcnt(0) in front of each of them
      const char* rbase = cbase + (long long)(tap_ok[t] ? tap_row[t] : 0)*p.Win*ES;
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        if (TINY) {                                 // images narrower than a quad: element by element
          const TI* src = reinterpret_cast<const TI*>(rbase + (long long)s_c[it]*plane*ES);
          const int f = s_voff[it]/ES - s_c[it]*plane;
          st[set][t][it] = s_c[it] < nch ? Q::gather(src, f, p.Win) : Q::zero();
        } else {
          const int vo = s_c[it] < nch ? s_voff[it] : 0;
          if (CC_ABL & 1) st[set][t][it] = Q::zero(); else
          if (CC_ABL & 32) st[set][t][it] = Q::load(reinterpret_cast<const char*>(in_b) + s_voff[it]); else   // every chunk re-reads chunk 0, row 0
          if (CC_ABL & 16) st[set][t][it] = Q::load(rbase + (vo & ~15)); else                                  // 16-byte aligned addresses
          st[set][t][it] = Q::load(rbase + vo);
        }
      }
    }
  };
  auto stage_store = [&](int buf, int cc, auto set_tag) {
    constexpr int set = decltype(set_tag)::value;
    const int nch = p.C - 8*cc;
#pragma unroll
    for (int t = 0; t < NSLOT; ++t) {
      if (!tap_ok[t]) continue;
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        if (s_kind[it] == 0) continue;
        typename Q::type v = st[set][t][it];
        if (!TINY) {
          if (s_kind[it] > 1) v = Q::moved(v, s_kind[it]);                      // a lane or two per row
          if (nch < 8 && s_c[it] >= nch) v = Q::zero();                         // last chunk of an odd channel count
        }
        *reinterpret_cast<uint2*>(lds + buf*BUFB + t*TAPB + s_off[it]) = Q::packed(v);
      }
    }
  };

  // ---- weights: fragments of this wave's MF row groups, (chunk, tap) two k steps ahead
  const int mfrag0 = (mtile*WM + wm)*MF;
  const uint4* wq[MF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) {         // row groups past M: any packed group (their outputs are not stored)
    const int fr = mfrag0 + mf < p.mfrags ? mfrag0 + mf : p.mfrags - 1;
    wq[mf] = p.wp + (long long)fr*p.ncc*CC_KH*64 + lane;
  }
  auto a_load = [&](int cc, int t, uint4 (&dst)[MF]) {
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      if (CC_ABL & 2) dst[mf] = make_uint4(cc, t, mf, 0); else
      dst[mf] = wq[mf][(cc*CC_KH + tap_i[t])*64];
    }
  };

  f32x16 acc[NSET][MF][NF];
#pragma unroll
  for (int st_ = 0; st_ < NSET; ++st_)
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[st_][mf][nf][i] = 0.f;

  // weight fragments of tap t live in slot t: the slot is refilled with the NEXT chunk's fragment as soon as
  // its MFMAs are issued, i.e. NTAP k steps (one chunk) ahead of its use -- L2 latency, not bandwidth, is what
  // the MFMAs wait for otherwise (and a rotating ring costs 16 v_mov per k step)
  uint4 a0[MF], a1[MF], a2[MF];
  a_load(0, 0, a0);
  if (NTAP > 1) a_load(0, 1 % NTAP, a1); else if (p.ncc > 1) a_load(1, 0, a1);
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, DEPTH - 1>;
  stage_load(0, S0{});
  if (DEPTH == 2) stage_load(p.ncc > 1 ? 1 : 0, S1{});
  stage_store(0, 0, S0{});
  __syncthreads();

  // one chunk: its MFMAs on LDS buffer cc & 1; the loads of chunk cc + DEPTH go out behind the first tap's weights,
  // chunk cc + 1 (register set `nx`) moves into the other buffer at the end
  auto body = [&](int cc, auto ld_tag, auto nx_tag) {
    const unsigned char* cur = lds + (cc & 1)*BUFB;
    const bool more = cc + 1 < p.ncc;
    const int nxt = more ? cc + 1 : cc;            // the last chunk is requested once more instead of branching
    const int far = cc + DEPTH < p.ncc ? cc + DEPTH : cc;
    bf16x8 bq[NF];
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
      if (t + 2 < NTAP) a_load(cc, (t + 2) % NTAP, a2);
      else a_load(nxt, (t + 2) % NTAP, a2);
      if (t == 0) stage_load(far, ld_tag);
      const int slot = PAIR ? (t >> 1) : t, set = PAIR ? (t & 1) : 0;
      if (!PAIR || (t & 1) == 0) {                  // (the odd tap of a pair reads the fragments of the even one)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
          const int col0 = 32*(wn*NF + nf);
          bq[nf] = cc_frag(cur + slot*TAPB + (col0 >> 7)*4096, col0 & 127, lane);
        }
      }
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        const bf16x8 af = __builtin_bit_cast(bf16x8, a0[mf]);
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
          if (CC_ABL & 4) { acc[set][mf][nf][0] += __builtin_bit_cast(float, (int)af[0]) + __builtin_bit_cast(float, (int)bq[nf][0]); continue; }
          acc[set][mf][nf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bq[nf], acc[set][mf][nf], 0, 0, 0);
        }
      }
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) { a0[mf] = a1[mf]; a1[mf] = a2[mf]; }
    }
    if (more) stage_store((cc + 1) & 1, cc + 1, nx_tag);
    __syncthreads();
  };
  if (DEPTH == 2) {
#pragma unroll 1
    for (int cc = 0; cc < p.ncc; cc += 2) {
      body(cc, S0{}, S1{});                        // chunk cc + 2 -> set 0 (chunk cc's, stored already), chunk cc + 1 from set 1
      if (cc + 1 < p.ncc) body(cc + 1, S1{}, S0{});
    }
  } else {
#pragma unroll 1
    for (int cc = 0; cc < p.ncc; ++cc) body(cc, S0{}, S0{});
  }

  // ---- D[m][frame] -> out[b][m][row][frame] (+ bias)
  const int oes = p.out_bf16 ? 2 : 4;        // bytes per output element
  char* out_b = static_cast<char*>(p.out) + (long long)b*p.out_bs*oes;
  char* out2_b = SEG && p.out_seg > 0 ? static_cast<char*>(p.out2) + (long long)b*p.out_bs*oes : nullptr;
#pragma unroll
  for (int set = 0; set < NSET; ++set) {
    const int orow = PAIR ? 2*r + set : r;
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        const int w = f0 + 32*(wn*NF + nf) + (lane & 31);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int m = 32*(mfrag0 + mf) + (i & 3) + 8*(i >> 2) + 4*(lane >> 5);
          if (m < p.M && w < p.Wout && (!(CC_ABL & 8) || acc[set][mf][nf][i] == 1234.5f)) {
            float v = acc[set][mf][nf][i];
            if (p.bias) v += p.bias[m];
            char* dst = out_b;
            int mc = m;
            if (SEG && p.out_seg > 0) {
              const int sg = (m >= p.out_seg) + (m >= 2*p.out_seg) + (m >= 3*p.out_seg);
              mc -= ((sg + 1) >> 1)*p.out_seg;
              if (sg & 1) dst = out2_b;
            }
            const long long idx = ((long long)mc*p.Hout + orow)*p.Wout + w;
            if (p.out_bf16) *reinterpret_cast<bf16_t*>(dst + idx*2) = f2bf(v);
            else *reinterpret_cast<float*>(dst + idx*4) = v;
          }
        }
      }
  }
}

template <int MF, int NF, int WM, int WN, bool SEG, bool LEAN, bool TINY = false, bool PAIR = false, typename TI = float>
__global__ __launch_bounds__(CC_THREADS) void cconv_rows_kernel(const CConvParams p) {
  constexpr int NT = 32*NF*WN;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2*CC_KH*(NT/128)*4096];
  // XCD-aware order: workgroups are dealt to the 8 XCDs (one L2 each) round-robin by linear id; the remap gives
  // every XCD a CONTIGUOUS run of (frame tile, output row) pairs, so that the input rows neighbouring output
  // rows share (5 of them per row in the transposed form) are fetched into one L2 once instead of into several
  // (measured fabric traffic was 2 - 4.5x the algorithmic bytes without it: profiles/r04_rows_dccrn_bf16_pmc_hbm_traffic_before_xcd_order.json)
  int ftile = blockIdx.x, r = blockIdx.y, bz = blockIdx.z;
  {
    const unsigned total = gridDim.x*gridDim.y*gridDim.z;
    if ((total & 7u) == 0) {
      const unsigned L = blockIdx.x + gridDim.x*(blockIdx.y + gridDim.y*blockIdx.z);
      unsigned g = (L & 7u)*(total >> 3) + (L >> 3);
      ftile = (int)(g % gridDim.x); g /= gridDim.x;
      r = (int)(g % gridDim.y); bz = (int)(g / gridDim.y);
    }
  }
  const int b = bz / p.mtiles, mtile = bz % p.mtiles;
  if (p.mode == 0) {
    const int ti[5] = {0, 1, 2, 3, 4};
    const int tr[5] = {2*r - 2, 2*r - 1, 2*r, 2*r + 1, 2*r + 2};
    if (LEAN) cconv_tile_lean<TI, MF, NF, WM, WN, 5, SEG, TINY>(p, lds, b, r, ftile, mtile, ti, tr, 1);
    else cconv_tile<TI, MF, NF, WM, WN, 5, SEG>(p, lds, b, r, ftile, mtile, ti, tr, 1);
  } else if (PAIR) {                 // output rows 2r and 2r + 1 from input rows r + 1, r, r - 1
    const int ti[5] = {0, 1, 2, 3, 4};
    const int tr[3] = {r + 1, r, r - 1};
    cconv_tile_lean<TI, MF, NF, WM, WN, 5, SEG, TINY, true>(p, lds, b, r, ftile, mtile, ti, tr, -1);
  } else if (r & 1) {
    const int ti[2] = {1, 3};
    const int tr[2] = {(r + 1) >> 1, (r - 1) >> 1};
    if (LEAN) cconv_tile_lean<TI, MF, NF, WM, WN, 2, SEG, TINY>(p, lds, b, r, ftile, mtile, ti, tr, -1);
    else cconv_tile<TI, MF, NF, WM, WN, 2, SEG>(p, lds, b, r, ftile, mtile, ti, tr, -1);
  } else {
    const int ti[3] = {0, 2, 4};
    const int tr[3] = {(r >> 1) + 1, r >> 1, (r >> 1) - 1};
    if (LEAN) cconv_tile_lean<TI, MF, NF, WM, WN, 3, SEG, TINY>(p, lds, b, r, ftile, mtile, ti, tr, -1);
    else cconv_tile<TI, MF, NF, WM, WN, 3, SEG>(p, lds, b, r, ftile, mtile, ti, tr, -1);
  }
}

// ---- weight gradient of both forms: dW[a][c][i][j] = sum_{b,h,w} small[b][a][h][w] big[b][c][2h - 2 + i][w + j]
// (Conv2d: small = dy, big = x; ConvTranspose2d: small = x, big = dy; big is (B, C, 2 Hs, Ws + 1)), written as
// the fp32 matrix out[a][10 c + 2 i + j] the complex weight unpacking reads. A GEMM whose k axis is the FRAME
// axis both images are contiguous along: per stage of 64 frames of one (b, h) the 128 rows of `small` (twice:
// as they are and shifted by one frame, which is the j = 1 tap -- every fragment read stays 16-byte aligned)
// and the 5 x 64 rows of `big` are rounded to bf16 into LDS images with 128-byte rows (16-byte chunk ^ (row >> 1) & 7:
// conflict-free ds_read_b128 fragments). Wave (a half, c half, j) keeps 2 x 5 accumulator tiles (its 64 rows of
// `small`, 32 channels of `big`, 5 rows i): 7 fragment reads per 10 MFMAs. A workgroup walks a contiguous
// range of (b, h) pairs and adds its tile to the gradient once at the end.
struct CWgradParams {
  const void* small; const void* small2; const void* big;   // fp32 or bf16 elements (the kernel's T), all three alike
  float* part;
  int B, A, C, Hs, Ws, Hb, Wb;
  int seg;                       // small = [s[:seg] | s2[:seg] | s[seg:] | s2[seg:]] along the channels (0: one source)
  long long small_bs, small2_bs, big_bs;
  int atiles, ctiles, npairs, pairs_per, nstage, ldo;
};

constexpr int WG_A = 128, WG_C = 32, WG_F = 64;
constexpr int WG_THREADS = 512, WG_NS = WG_A*16/WG_THREADS, WG_NB = CC_KH*WG_C*16/WG_THREADS;
constexpr int WG_SMALLB = WG_A*128, WG_BIGB = CC_KH*WG_C*128, WG_BUFB = 2*WG_SMALLB + WG_BIGB;

__device__ __forceinline__ int wg_swz(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ int wg_off(int row, int q) { return 128*row + 16*((q >> 1) ^ wg_swz(row)) + 8*(q & 1); }

template <typename T>
__global__ __launch_bounds__(WG_THREADS) void cconv_wgrad_kernel(const CWgradParams p) {
  typedef CQuad<T> Q;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2*WG_BUFB];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int j = wid & 1, afr = wid >> 1;
  // XCD-aware order (as cconv_rows_kernel): the tiles of one split read the same rows of both images
  unsigned bx = blockIdx.x, by = blockIdx.y;
  {
    const unsigned total = gridDim.x*gridDim.y;
    if ((total & 7u) == 0) {
      const unsigned L = blockIdx.x + gridDim.x*blockIdx.y;
      const unsigned g = (L & 7u)*(total >> 3) + (L >> 3);
      bx = g % gridDim.x; by = g / gridDim.x;
    }
  }
  const int atile = bx % p.atiles, ctile = bx / p.atiles;
  const int pair0 = by*p.pairs_per;
  const int pair1 = pair0 + p.pairs_per < p.npairs ? pair0 + p.pairs_per : p.npairs;
  const int nitems = (pair1 - pair0)*p.nstage;
  const int q = tid & 15;

  // ---- staging rows of this thread: 4 of `small` (row = channel), 10 of `big` (row = (i, channel))
  const T* s_ptr[WG_NS]; long long s_bs[WG_NS];
#pragma unroll
  for (int r = 0; r < WG_NS; ++r) {
    const int a = atile*WG_A + ((tid + WG_THREADS*r) >> 4);
    s_ptr[r] = nullptr; s_bs[r] = 0;
    if (a < p.A) {
      if (p.seg > 0) {
        const int sg = a / p.seg, ch = (sg >> 1)*p.seg + a % p.seg;
        s_ptr[r] = static_cast<const T*>((sg & 1) ? p.small2 : p.small) + (long long)ch*p.Hs*p.Ws;
        s_bs[r] = (sg & 1) ? p.small2_bs : p.small_bs;
      } else { s_ptr[r] = static_cast<const T*>(p.small) + (long long)a*p.Hs*p.Ws; s_bs[r] = p.small_bs; }
    }
  }
  // (bf16 images: two stages in flight in the registers one stage of fp32 quads takes -- as cconv_tile_lean)
  constexpr int DEPTH = sizeof(T) == 2 ? 2 : 1;
  typename Q::type sv[DEPTH][WG_NS]; T sl[DEPTH][WG_NS];    // frames 4q .. 4q+3 of the stage and the frame before the stage
  typename Q::type bv[DEPTH][WG_NB];
  auto load_small = [&](int it, auto set_tag) {
    constexpr int set = decltype(set_tag)::value;
    const int pr = pair0 + it / p.nstage, stg = it % p.nstage;
    const int b = pr / p.Hs, h = pr % p.Hs, f = stg*WG_F + 4*q;
#pragma unroll
    for (int r = 0; r < WG_NS; ++r) {
      typename Q::type v = Q::zero();
      T left = 0;
      if (s_ptr[r]) {
        const T* src = s_ptr[r] + b*s_bs[r] + (long long)h*p.Ws;
        if (f + 3 < p.Ws) v = Q::load(src + f);
        else v = Q::gather(src, f, p.Ws);
        if (q == 0 && f > 0 && f - 1 < p.Ws) left = src[f - 1];
      }
      sv[set][r] = v; sl[set][r] = left;
    }
  };
  auto load_big = [&](int it, auto set_tag) {
    constexpr int set = decltype(set_tag)::value;
    const int pr = pair0 + it / p.nstage, stg = it % p.nstage;
    const int b = pr / p.Hs, h = pr % p.Hs, f = stg*WG_F + 4*q;
#pragma unroll
    for (int r = 0; r < WG_NB; ++r) {
      const int rs = (tid + WG_THREADS*r) >> 4, i = rs >> 5, c = ctile*WG_C + (rs & 31);
      const int row = 2*h - 2 + i;
      typename Q::type v = Q::zero();
      if (c < p.C && row >= 0 && row < p.Hb) {
        const T* src = static_cast<const T*>(p.big) + b*p.big_bs + ((long long)c*p.Hb + row)*p.Wb;
        if (f + 3 < p.Wb) v = Q::load(src + f);
        else v = Q::gather(src, f, p.Wb);
      }
      bv[set][r] = v;
    }
  };
  auto store_small = [&](int buf, auto set_tag) {
    constexpr int set = decltype(set_tag)::value;
    unsigned char* base = lds + buf*WG_BUFB;
#pragma unroll
    for (int r = 0; r < WG_NS; ++r) {
      const int row = (tid + WG_THREADS*r) >> 4;
      // the frame before this quad: lane q - 1 of the same row (16 lanes = one DPP row), or the stage's left neighbour
      *reinterpret_cast<uint2*>(base + wg_off(row, q)) = Q::packed(sv[set][r]);
      *reinterpret_cast<uint2*>(base + WG_SMALLB + wg_off(row, q)) = Q::shifted(sv[set][r], sl[set][r], q == 0);
    }
  };
  auto store_big = [&](int buf, auto set_tag) {
    constexpr int set = decltype(set_tag)::value;
    unsigned char* base = lds + buf*WG_BUFB + 2*WG_SMALLB;
#pragma unroll
    for (int r = 0; r < WG_NB; ++r) {
      const int rs = (tid + WG_THREADS*r) >> 4;
      *reinterpret_cast<uint2*>(base + wg_off(rs, q)) = Q::packed(bv[set][r]);
    }
  };

  f32x16 acc[CC_KH];
#pragma unroll
  for (int i = 0; i < CC_KH; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, DEPTH - 1>;
  if (nitems > 0) {
    load_small(0, S0{}); load_big(0, S0{});
    if (DEPTH == 2 && nitems > 1) { load_big(1, S1{}); load_small(1, S1{}); }
    store_small(0, S0{}); store_big(0, S0{});
  }
  __syncthreads();
  const int m = lane & 31, g = lane >> 5;
  // one stage: its MFMAs on LDS buffer it & 1; stage it + DEPTH is requested first, stage it + 1 (register set `nx`)
  // moves into the other buffer at the end
  auto body = [&](int it, auto ld_tag, auto nx_tag) {
    const unsigned char* cur = lds + (it & 1)*WG_BUFB;
    const bool more = it + 1 < nitems;
    if (it + DEPTH < nitems) { load_big(it + DEPTH, ld_tag); load_small(it + DEPTH, ld_tag); }
#pragma unroll
    for (int ks = 0; ks < WG_F/16; ++ks) {
      const int ch = 2*ks + g;
      bf16x8 bf[CC_KH];
      const int arow = 32*afr + m;
      const bf16x8 af = *reinterpret_cast<const bf16x8*>(cur + j*WG_SMALLB + 128*arow + 16*(ch ^ wg_swz(arow)));
#pragma unroll
      for (int i = 0; i < CC_KH; ++i) {
        const int row = 32*i + m;
        bf[i] = *reinterpret_cast<const bf16x8*>(cur + 2*WG_SMALLB + 128*row + 16*(ch ^ wg_swz(row)));
      }
#pragma unroll
      for (int i = 0; i < CC_KH; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf[i], acc[i], 0, 0, 0);
    }
    if (more) { store_big((it + 1) & 1, nx_tag); store_small((it + 1) & 1, nx_tag); }
    __syncthreads();
  };
  if (DEPTH == 2) {
#pragma unroll 1
    for (int it = 0; it < nitems; it += 2) {
      body(it, S0{}, S1{});
      if (it + 1 < nitems) body(it + 1, S1{}, S0{});
    }
  } else {
#pragma unroll 1
    for (int it = 0; it < nitems; ++it) body(it, S0{}, S0{});
  }

  // ---- D[a][c] of tap (i, j) -> part[split][2 i + j][a][c]: plain stores, 128 bytes per 32 lanes; the splits
  // are summed in order by cconv_wgrad_reduce_kernel (device-scope atomics cost more than the products here)
  const int c = ctile*WG_C + m;
  if (c < p.C) {
    float* part = p.part + (long long)by*10*p.A*p.C;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = atile*WG_A + 32*afr + (e & 3) + 8*(e >> 2) + 4*g;
      if (row < p.A) {
#pragma unroll
        for (int i = 0; i < CC_KH; ++i) part[((long long)(2*i + j)*p.A + row)*p.C + c] = acc[i][e];
      }
    }
  }
}

// out[a][10 c + tap] = sum_split part[split][tap][a][c]: 64 outputs per workgroup, the splits dealt to 4 thread
// groups with 8 loads in flight each, partial sums combined in a fixed order
__global__ __launch_bounds__(256) void cconv_wgrad_reduce_kernel(const float* part, float* out, int A, int C,
                                                                  int nsplit) {
  __shared__ float sh[4][64];
  const long long n = (long long)10*A*C;
  const long long e = (long long)blockIdx.x*64 + (threadIdx.x & 63);
  const int grp = threadIdx.x >> 6;
  float s = 0.f;
  if (e < n) {
    int k = grp;
    for (; k + 28 < nsplit; k += 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(k + 4*u)*n + e];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < nsplit; k += 4) s += part[k*n + e];
  }
  sh[grp][threadIdx.x & 63] = s;
  __syncthreads();
  if (grp == 0 && e < n) {
    s = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
    const int c = (int)(e % C);
    const long long t = e / C;
    const int a = (int)(t % A), tap = (int)(t / A);
    out[(long long)a*10*C + 10*c + tap] = s;
  }
}

#include "cconv_dma.cuh"
#include "cconv_wgrad_dma.cuh"

// which tile function a launch shape runs (measured per layer, tools/dccrn_conv_bench.py): the lean loader for
// two-source inputs / split outputs and for M <= 64; the pair form of the transposed convolution for M <= 128
template <typename TI, int MF, int NF, int WM, int WN>
static void cconv_launch(CConvParams& p, int64_t B, int64_t M, bool transposed, bool seg, hipStream_t st) {
  constexpr int MT = 32*MF*WM, NT = 32*NF*WN;
  p.mtiles = (int)((M + MT - 1)/MT); p.ftiles = (p.Wout + NT - 1)/NT;
  const dim3 grid(p.ftiles, p.Hout, (unsigned)(B*p.mtiles)), block(CC_THREADS);
  if constexpr (sizeof(TI) == 2 && NT == 256) {
    // bf16 images in whole chunks of 8 channels: staged by LDS-DMA (cconv_dma.cuh); pair form as below
    // (measured per layer, tools/cconv_bench.py / profiles/r06_cconv_ablation.txt: every transposed launch and the
    // 128-row strided tile; the 256-row strided tile keeps its weights for five taps twice over and spills, the 64- and
    // 32-row strided tiles are as fast through registers)
    if (CC_DMA && p.C % 8 == 0 && (long long)B*p.in_bs < (1LL << 29) && (transposed || MT == 128)) {
      if (!transposed) {
        if (seg) hipLaunchKernelGGL((cconv_rows_dma_kernel<MF, NF, WM, WN, true, 0>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((cconv_rows_dma_kernel<MF, NF, WM, WN, false, 0>), grid, block, 0, st, p);
        return;
      }
      if constexpr (MT <= 128 && CC_PAIR) {
        if (MT <= 32 || (MT == 64) != seg) {
          const dim3 gp(p.ftiles, p.Hout/2, (unsigned)(B*p.mtiles));
          if (seg) hipLaunchKernelGGL((cconv_rows_dma_kernel<MF, NF, WM, WN, true, 2>), gp, block, 0, st, p);
          else hipLaunchKernelGGL((cconv_rows_dma_kernel<MF, NF, WM, WN, false, 2>), gp, block, 0, st, p);
          return;
        }
      }
      if (seg) hipLaunchKernelGGL((cconv_rows_dma_kernel<MF, NF, WM, WN, true, 1>), grid, block, 0, st, p);
      else hipLaunchKernelGGL((cconv_rows_dma_kernel<MF, NF, WM, WN, false, 1>), grid, block, 0, st, p);
      return;
    }
  }
  if constexpr (MT <= 128 && CC_PAIR) {
    // (measured exceptions: M = 128 from one source and M = 64 from two sources are as fast / faster row by row)
    if (transposed && (MT <= 32 || (MT == 64) != seg)) {
      const dim3 gp(p.ftiles, p.Hout/2, (unsigned)(B*p.mtiles));
      if (seg) hipLaunchKernelGGL((cconv_rows_kernel<MF, NF, WM, WN, true, true, false, true, TI>), gp, block, 0, st, p);
      else hipLaunchKernelGGL((cconv_rows_kernel<MF, NF, WM, WN, false, true, false, true, TI>), gp, block, 0, st, p);
      return;
    }
  }
  if (seg) hipLaunchKernelGGL((cconv_rows_kernel<MF, NF, WM, WN, true, true, false, false, TI>), grid, block, 0, st, p);
  else if (MT <= 64) hipLaunchKernelGGL((cconv_rows_kernel<MF, NF, WM, WN, false, true, false, false, TI>), grid, block, 0, st, p);
  else hipLaunchKernelGGL((cconv_rows_kernel<MF, NF, WM, WN, false, sizeof(TI) == 2 && MF*NF >= 8, false, false, TI>), grid, block, 0, st, p);
}

template <typename TI>
static int cconv_rows_any(const void* in, const void* in2, int64_t in_seg, const void* wp, const float* bias, void* out,
                          void* out2, int64_t out_seg, int64_t B, int64_t C, int64_t M, int64_t Hin, int64_t Win,
                          int32_t transposed, brv_stream_t stream, int out_bf16 = 0) {
  if (!in || !wp || !out || B < 1 || C < 1 || M < 1 || Hin < 1 || Win < 1) return -1;
  if (!transposed && ((Hin & 1) || Win < 2)) return -1;
  if (in_seg < 0 || (in_seg > 0 && (!in2 || C != 4*in_seg || (in_seg & 7)))) return -1;
  if (out_seg < 0 || (out_seg > 0 && (!out2 || M != 4*out_seg))) return -1;
  CConvParams p;
  p.in = in; p.in2 = in2; p.wp = (const uint4*)wp; p.bias = bias; p.out = out; p.out2 = out2; p.out_bf16 = out_bf16;
  p.in_seg = (int)in_seg; p.out_seg = (int)out_seg;
  p.B = (int)B; p.C = (int)C; p.M = (int)M; p.Hin = (int)Hin; p.Win = (int)Win;
  p.Hout = transposed ? (int)(2*Hin) : (int)(Hin/2);
  p.Wout = transposed ? (int)(Win + 1) : (int)(Win - 1);
  p.in_bs = (in_seg > 0 ? 2*in_seg : C)*Hin*Win; p.out_bs = (out_seg > 0 ? 2*out_seg : M)*(long long)p.Hout*p.Wout;
  p.mode = transposed ? 1 : 0;
  p.ncc = (int)((C + 7)/8); p.mfrags = (int)((M + 31)/32);
  hipStream_t st = (hipStream_t)stream;
#define CC_LAUNCH(MF_, NF_, WM_, WN_) cconv_launch<TI, MF_, NF_, WM_, WN_>(p, B, M, transposed != 0, in_seg > 0 || out_seg > 0, st)
  if (Win < 4) {            // narrower than a staging quad: the element-wise loader, one workgroup shape
    constexpr int MT = 64, NT = 256;
    p.mtiles = (int)((M + MT - 1)/MT); p.ftiles = (p.Wout + NT - 1)/NT;
    hipLaunchKernelGGL((cconv_rows_kernel<2, 1, 1, 8, true, true, true, false, TI>), dim3(p.ftiles, p.Hout, (unsigned)(B*p.mtiles)),
                       dim3(CC_THREADS), 0, st, p);
  } else if (M > 128) {
    const long long wgs = (long long)((p.Wout + 255)/256)*p.Hout*B*((M + 255)/256);
    // (strided launches on bf16 images with up to 256 output rows: two passes of the 128-row LDS-DMA tile beat the
    // 256-row register tile -- 150 -> 137, 92 -> 61, 209 -> 198 us; with 512 rows they do not: profiles/r06_cconv_ablation.txt)
    if (CC_DMA && sizeof(TI) == 2 && !transposed && C % 8 == 0 && M <= 256) CC_LAUNCH(2, 2, 2, 4);
    else if (wgs < 256) CC_LAUNCH(2, 2, 4, 2); else CC_LAUNCH(2, 4, 4, 2);
  } else if (M > 64) CC_LAUNCH(2, 2, 2, 4);
  else if (M > 32) CC_LAUNCH(2, 1, 1, 8);
  else CC_LAUNCH(1, 1, 1, 8);
#undef CC_LAUNCH
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace

extern "C" {

int64_t brv_cconv_packed_bytes(int64_t M, int64_t C) {
  return ((M + 31)/32)*((C + 7)/8)*CC_KH*64*16;
}

int brv_cconv_pack(const float* wc, void* wp, int64_t M, int64_t C, int64_t m_stride, int64_t c_stride,
                   brv_stream_t stream) {
  if (!wc || !wp || M < 1 || C < 1) return -1;
  const int ncc = (int)((C + 7)/8);
  const long long total = ((M + 31)/32)*(long long)ncc*CC_KH*64;
  hipLaunchKernelGGL(cconv_pack_kernel, dim3((unsigned)((total + 255)/256)), dim3(256), 0, (hipStream_t)stream,
                     wc, (uint4*)wp, (int)M, (int)C, (long long)m_stride, (long long)c_stride, ncc, total);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

int brv_cconv_pack_complex(const float* wr, const float* wi, const float* br, const float* bi, int64_t R, int64_t C,
                           int64_t Cb, float sign, float* wc, float* bias, void* wp1, int64_t M1, int64_t C1,
                           int64_t m_stride1, int64_t c_stride1, void* wp2, int64_t M2, int64_t C2,
                           int64_t m_stride2, int64_t c_stride2, brv_stream_t stream) {
  if (!wr || !wi || !br || !bi || !wc || !bias || !wp1 || R < 1 || C < 1 || Cb < 1 || M1 < 1 || C1 < 1) return -1;
  if (wp2 && (M2 < 1 || C2 < 1)) return -1;
  if (4*R*C >= (1LL << 31)) return -1;
  CPackAllParams p; memset(&p, 0, sizeof(p));
  p.wr = wr; p.wi = wi; p.br = br; p.bi = bi; p.wc = wc; p.bias = bias;
  p.R = (int)R; p.C = (int)C; p.Cb = (int)Cb; p.s = sign;
  long long most = 4*R*C > Cb ? 4*R*C : Cb;
  const int64_t Ms[2] = {M1, M2}, Cs[2] = {C1, C2}, sm[2] = {m_stride1, m_stride2}, sk[2] = {c_stride1, c_stride2};
  void* wps[2] = {wp1, wp2};
  for (int q = 0; q < 2; ++q) {
    if (!wps[q]) continue;
    // every index the job forms stays inside wc
    if ((Ms[q] - 1)*sm[q] + (Cs[q] - 1)*sk[q] + 2*(CC_KH - 1) + 1 >= 4*R*C) return -1;
    CPackJob& j = p.job[q];
    j.wp = (uint4*)wps[q]; j.M = (int)Ms[q]; j.C = (int)Cs[q]; j.ncc = (int)((Cs[q] + 7)/8);
    j.sm = (unsigned int)sm[q]; j.sk = (unsigned int)sk[q];
    const long long total = ((Ms[q] + 31)/32)*(long long)j.ncc*CC_KH*64;
    if (total >= (1LL << 31)) return -1;
    j.total = (unsigned int)total;
    if (total > most) most = total;
  }
  hipLaunchKernelGGL(cconv_pack_all_kernel, dim3((unsigned)((most + 255)/256)), dim3(256), 0, (hipStream_t)stream, p);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

int brv_cconv_rows(const float* in, const float* in2, int64_t in_seg, const void* wp, const float* bias, float* out,
                   float* out2, int64_t out_seg, int64_t B, int64_t C, int64_t M, int64_t Hin, int64_t Win,
                   int32_t transposed, brv_stream_t stream) {
  return cconv_rows_any<float>(in, in2, in_seg, wp, bias, out, out2, out_seg, B, C, M, Hin, Win, transposed, stream);
}

int brv_cconv_rows_bf16(const void* in, const void* in2, int64_t in_seg, const void* wp, const float* bias, float* out,
                        float* out2, int64_t out_seg, int64_t B, int64_t C, int64_t M, int64_t Hin, int64_t Win,
                        int32_t transposed, brv_stream_t stream) {
  return cconv_rows_any<bf16_t>(in, in2, in_seg, wp, bias, out, out2, out_seg, B, C, M, Hin, Win, transposed, stream);
}

int brv_cconv_rows_ex(const void* in, const void* in2, int64_t in_seg, const void* wp, const float* bias, void* out,
                      void* out2, int64_t out_seg, int64_t B, int64_t C, int64_t M, int64_t Hin, int64_t Win,
                      int32_t transposed, int32_t in_bf16, int32_t out_bf16, brv_stream_t stream) {
  if (in_bf16) return cconv_rows_any<bf16_t>(in, in2, in_seg, wp, bias, out, out2, out_seg, B, C, M, Hin, Win, transposed,
                                             stream, out_bf16 != 0);
  return cconv_rows_any<float>(in, in2, in_seg, wp, bias, out, out2, out_seg, B, C, M, Hin, Win, transposed, stream,
                               out_bf16 != 0);
}

static void cconv_wgrad_plan(int64_t B, int64_t A, int64_t C, int64_t Hs, int& atiles, int& ctiles, int& nsplit,
                             int& pairs_per) {
  atiles = (int)((A + WG_A - 1)/WG_A); ctiles = (int)((C + WG_C - 1)/WG_C);
  const int tiles = atiles*ctiles, npairs = (int)(B*Hs);
  nsplit = (2*256 + tiles - 1)/tiles;
  if (nsplit > npairs) nsplit = npairs;
  pairs_per = (npairs + nsplit - 1)/nsplit;
  nsplit = (npairs + pairs_per - 1)/pairs_per;
}

int64_t brv_cconv_wgrad_workspace_bytes(int64_t B, int64_t A, int64_t C, int64_t Hs) {
  if (B < 1 || A < 1 || C < 1 || Hs < 1) return -1;
  int atiles, ctiles, nsplit, pairs_per;
  cconv_wgrad_plan(B, A, C, Hs, atiles, ctiles, nsplit, pairs_per);
  return (int64_t)nsplit*10*A*C*4;
}

}  // extern "C"

template <typename T>
static int cconv_wgrad_any(const void* small, const void* small2, const void* big, float* out, void* workspace,
                           int64_t B, int64_t A, int64_t C, int64_t Hs, int64_t Ws, int64_t seg, brv_stream_t stream) {
  if (!small || !big || !out || !workspace || B < 1 || A < 1 || C < 1 || Hs < 1 || Ws < 1) return -1;
  if (seg < 0 || (seg > 0 && (!small2 || A != 4*seg))) return -1;
  CWgradParams p;
  p.small = small; p.small2 = small2; p.big = big; p.part = (float*)workspace;
  p.B = (int)B; p.A = (int)A; p.C = (int)C; p.Hs = (int)Hs; p.Ws = (int)Ws; p.Hb = (int)(2*Hs); p.Wb = (int)(Ws + 1);
  p.seg = (int)seg;
  p.small_bs = (seg > 0 ? 2*seg : A)*Hs*Ws; p.small2_bs = p.small_bs; p.big_bs = C*(long long)p.Hb*p.Wb;
  int nsplit;
  cconv_wgrad_plan(B, A, C, Hs, p.atiles, p.ctiles, nsplit, p.pairs_per);
  p.npairs = (int)(B*Hs); p.nstage = (p.Wb + WG_F - 1)/WG_F; p.ldo = (int)(10*C);
  hipStream_t st = (hipStream_t)stream;
  // bf16 images: staged by LDS-DMA (cconv_wgrad_dma.cuh) when a unit of 8 rows never straddles two sources
  if (sizeof(T) == 2 && CC_DMA && (seg == 0 || seg % 8 == 0) && B*p.small_bs < (1LL << 29) && B*p.big_bs < (1LL << 29))
    hipLaunchKernelGGL(cconv_wgrad_dma_kernel, dim3((unsigned)(p.atiles*p.ctiles), (unsigned)nsplit), dim3(WG_THREADS), 0,
                       st, p);
  else
  hipLaunchKernelGGL(cconv_wgrad_kernel<T>, dim3((unsigned)(p.atiles*p.ctiles), (unsigned)nsplit), dim3(WG_THREADS), 0,
                     st, p);
  const long long n = 10*A*C;
  hipLaunchKernelGGL(cconv_wgrad_reduce_kernel, dim3((unsigned)((n + 63)/64)), dim3(256), 0, st,
                     (const float*)workspace, out, (int)A, (int)C, nsplit);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

extern "C" {

int brv_cconv_wgrad(const float* small, const float* small2, const float* big, float* out, void* workspace,
                    int64_t B, int64_t A, int64_t C, int64_t Hs, int64_t Ws, int64_t seg, brv_stream_t stream) {
  return cconv_wgrad_any<float>(small, small2, big, out, workspace, B, A, C, Hs, Ws, seg, stream);
}

int brv_cconv_wgrad_bf16(const void* small, const void* small2, const void* big, float* out, void* workspace,
                         int64_t B, int64_t A, int64_t C, int64_t Hs, int64_t Ws, int64_t seg, brv_stream_t stream) {
  return cconv_wgrad_any<bf16_t>(small, small2, big, out, workspace, B, A, C, Hs, Ws, seg, stream);
}

}  // extern "C"
