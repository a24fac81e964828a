// Weight-gradient GEMM: reduction over frames of two channels-last operands.
//
//   D[n][k] += sum_{b,t} G[b][t][n] * H[b][t][k]       n < Gp, k < Hp
//
// G is the output-side gradient (bf16, optionally two concatenated tensors),
// H the layer input, reconstructed on the fly from what the forward pass saved
// (PReLU and/or global-layer-norm affine applied while staging; or the framed
// waveform). Both operands are contiguous in the channel index, i.e. *strided*
// in the reduction index t, so MFMA fragments are fetched from row-major LDS
// tiles with the transposing read ds_read_b64_tr_b16 (gfx950).
//
// One workgroup = 128 (n) x BH (k) output tile over a contiguous range of
// 64-frame chunks; partial tiles are added to the fp32 gradient with atomics.
// Also produces the bias gradient (column sums of G) when requested.
//
// Replaces (reference): the autograd weight/bias gradients of the nn.Conv1d /
// nn.ConvTranspose1d layers in brever/models/convtasnet/convtasnet.py.
#pragma once
#include "gemm_rows.cuh"

namespace brv {

struct WgradParams {
  ASpec g;                 // always bf16 rows (A_BF16), optional CAT2, no transform
  ASpec h;
  int B, T;
  int Gp, Hp;              // padded channel counts
  float* out0; float* out1; int G0p;   // rows n < G0p -> out0, else out1 (n - G0p)
  int N0, N1;              // true row counts of the two parts
  int Kout, ldo;           // true column count / leading dim of D
  float* gbias0; float* gbias1;        // nullable: column sums of G
  int nsplit;
};

// Grouped launch: blockIdx.z selects one of up to kWgMaxProb problems that share all
// dimensions and differ only in their tensors (the same layer of different TCN blocks).
// Deferring the weight gradients of all blocks to one launch per layer kind makes the
// frame range per workgroup ~20x longer, i.e. ~20x fewer partial-tile atomics.
constexpr int kWgMaxProb = 24;
struct WgradProb {
  const void* g0; const void* g1; const void* h;
  float* out0; float* out1; float* gbias0; float* gbias1;
  const float* slope; const double* stats; const float* gamma; const float* beta;
};
struct WgradGroupParams {
  WgradParams base;
  int nprob;
  WgradProb prob[kWgMaxProb];
};

constexpr int WG_BT = 64;                  // frames per chunk
constexpr int WG_BG = 128;                 // G channels per tile

typedef __attribute__((ext_vector_type(8))) short s16x8;

__device__ __forceinline__ bf16x8 tr_frag(const bf16_t* tile, int ld, int row0,
                                          int col0, int lane) {
  // fragment for v_mfma_f32_32x32x16_bf16: lane (r = lane&31, h = lane>>5) gets
  // tile[row0 + 8h + j][col0 + r], j = 0..7, through two transposing reads of
  // 4 rows x 16 columns each (per 16-lane group).
  const int g4 = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
  const int col = col0 + 16*(g4 & 1) + 4*pp;
  const int row = row0 + 8*(g4 >> 1) + q;
  const bf16_t* p0 = tile + row*ld + col;
#ifndef BRV_WGRAD_SCALAR_FRAG
  typedef __attribute__((address_space(3))) s16x4* lds_p;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p0 + 4*ld));
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
#else
  const int r = lane & 31, h = lane >> 5;
  s16x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (short)tile[(row0 + 8*h + j)*ld + col0 + r];
  return __builtin_bit_cast(bf16x8, v);
#endif
}

template <int BH, int HK>
__global__ __launch_bounds__(256) void gemm_wgrad_kernel(const WgradGroupParams gp) {
  WgradParams p = gp.base;
  // XCD-aware order: the tiles of one (frame split, problem) pair all stream the same H rows
  // (and, for several H tiles, the same G rows). Dispatch ids congruent mod 8 share an XCD and
  // its L2: with the tile index fastest in the plain grid order the 4 G tiles of the first 1x1
  // convolution's weight gradient landed on 4 different XCDs and each fetched H from HBM
  // (PMC: 1.6x the algorithmic bytes). Here the tiles of a pair take consecutive slots of ONE XCD.
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  {
    const int tiles = gridDim.x, pairs = gridDim.y*gridDim.z;
    if ((pairs & 7) == 0) {
      const int L = bx + tiles*(by + (int)gridDim.y*bz);
      const int xcd = L & 7, slot = L >> 3;
      const int pair = (slot / tiles)*8 + xcd;
      bx = slot % tiles; by = pair % (int)gridDim.y; bz = pair / (int)gridDim.y;
    }
  }
  if (gp.nprob > 0) {
    const WgradProb& q = gp.prob[bz];
    p.g.p0 = q.g0; p.g.p1 = q.g1; p.h.p0 = q.h;
    p.out0 = q.out0; p.out1 = q.out1; p.gbias0 = q.gbias0; p.gbias1 = q.gbias1;
    p.h.slope = q.slope; p.h.stats = q.stats; p.h.gamma = q.gamma; p.h.beta = q.beta;
  }
  constexpr int LDG = WG_BG + 32;          // +64 B: conflict-free transposing reads
  constexpr int LDH = BH + 32;
  constexpr int NT = BH/32;
  constexpr int HCH = BH/32;               // H chunks per thread (BH*64/8/256)
  constexpr int HCPR = BH/8;               // H chunks per row
  __shared__ __attribute__((aligned(16))) bf16_t Gs[WG_BT*LDG];
  __shared__ __attribute__((aligned(16))) bf16_t Hs[WG_BT*LDH];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int n_htiles = p.Hp/BH;
  const int gtile = bx / n_htiles, htile = bx % n_htiles;
  const int n0 = gtile*WG_BG, k0 = htile*BH;
  const int T = p.T;
  const int cpi = ceil_div(T, WG_BT);
  const int total = p.B*cpi;
  const int per = ceil_div(total, p.nsplit);
  const int c_begin = by*per;
  const int c_end = min(total, c_begin + per);

  // staging geometry: fixed channel chunk per thread
  const int gcc = tid & 15, grow = tid >> 4;          // rows grow + 16*ci
  const int hcc = tid % HCPR, hrow = tid / HCPR;      // rows hrow + (256/HCPR)*ci
  constexpr int HRS = 256/HCPR;

  const ASpec& G = p.g;
  const ASpec& H = p.h;
  const int gch = n0 + gcc*8;                          // global G channel of the chunk
  const bool g_first = gch < G.K0;
  const bf16_t* gsrc = reinterpret_cast<const bf16_t*>(g_first ? G.p0 : G.p1);
  const int gld = g_first ? G.ld0 : G.ld1;
  const long long gbs = g_first ? G.bs0 : G.bs1;
  const int gk = g_first ? gch : gch - G.K0;
  const bool g_ok = gch < p.Gp;

  const int hch = k0 + hcc*8;
  const bool h_ok = hch < p.Hp;
  const bool h_aff = H.stats != nullptr;
  const bool h_prelu = H.slope != nullptr;
  const float hslope = h_prelu ? *H.slope : 1.f;
  float hg[8], hb[8];
  load8_masked(h_aff ? H.gamma : nullptr, hch, H.C, hg);
  load8_masked(h_aff ? H.beta : nullptr, hch, H.C, hb);

  uint4 graw[4];
  uint4 hraw_b[HCH];
  float hraw_f[(HK == A_BF16) ? 1 : HCH][8];
  float bias_acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bias_acc[j] = 0.f;

  auto load_chunk = [&](int c) {
    const int b = c / cpi, t0 = (c % cpi)*WG_BT;
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
      const int t = t0 + grow + 16*ci;
      graw[ci] = (g_ok && t < T)
          ? *reinterpret_cast<const uint4*>(gsrc + (long long)b*gbs + (long long)t*gld + gk)
          : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int ci = 0; ci < HCH; ++ci) {
      const int t = t0 + hrow + HRS*ci;
      if (HK == A_BF16) {
        hraw_b[ci] = (h_ok && t < T)
            ? *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(H.p0)
                                              + (long long)b*H.bs0 + (long long)t*H.ld0 + hch)
            : make_uint4(0, 0, 0, 0);
      } else if (HK == A_F32) {
        float4 lo = make_float4(0, 0, 0, 0), hi = lo;
        if (h_ok && t < T) {
          const float* s = reinterpret_cast<const float*>(H.p0) + (long long)b*H.bs0 + (long long)t*H.ld0 + hch;
          lo = *reinterpret_cast<const float4*>(s);
          hi = *reinterpret_cast<const float4*>(s + 4);
        }
        float* d = hraw_f[(HK == A_BF16) ? 0 : ci];
        d[0] = lo.x; d[1] = lo.y; d[2] = lo.z; d[3] = lo.w;
        d[4] = hi.x; d[5] = hi.y; d[6] = hi.z; d[7] = hi.w;
      } else {
        const float* wav = reinterpret_cast<const float*>(H.p0) + (long long)b*H.wav_stride;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = hch + j;
          const long long idx = (long long)t*H.hop + k;
          float v = 0.f;
          if (t < T && k < H.Kf && idx < H.wav_len) v = wav[idx];
          hraw_f[(HK == A_BF16) ? 0 : ci][j] = v;
        }
      }
    }
  };

  auto store_chunk = [&](int c) {
    const int b = c / cpi, t0 = (c % cpi)*WG_BT;
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
      *reinterpret_cast<uint4*>(Gs + (grow + 16*ci)*LDG + gcc*8) = graw[ci];
      if (p.gbias0 || p.gbias1) {
        float f[8]; unpack8(graw[ci], f);
#pragma unroll
        for (int j = 0; j < 8; ++j) bias_acc[j] += f[j];
      }
    }
    NormStat ns = {0.f, 1.f};
    if (h_aff) ns = norm_stat(H.stats, b, H.inv_n, H.eps);
#pragma unroll
    for (int ci = 0; ci < HCH; ++ci) {
      const int row = hrow + HRS*ci;
      uint4 q;
      if (HK == A_BF16 && !h_aff && !h_prelu) {
        q = hraw_b[ci];
      } else {
        float f[8];
        if (HK == A_BF16) unpack8(hraw_b[ci], f);
        else {
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = hraw_f[(HK == A_BF16) ? 0 : ci][j];
        }
        if (h_prelu) {
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = prelu(f[j], hslope);
        }
        if (h_aff) {
#pragma unroll
          for (int j = 0; j < 8; ++j)
            f[j] = f[j]*(ns.rstd*hg[j]) + (hb[j] - ns.mean*ns.rstd*hg[j]);
        }
        if (t0 + row >= T) {
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = 0.f;
        }
        q = pack8(f);
      }
      *reinterpret_cast<uint4*>(Hs + row*LDH + hcc*8) = q;
    }
  };

  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;

  if (c_begin < c_end) load_chunk(c_begin);
  for (int c = c_begin; c < c_end; ++c) {
    __syncthreads();
    store_chunk(c);
    __syncthreads();
    if (c + 1 < c_end) load_chunk(c + 1);
#pragma unroll
    for (int s = 0; s < WG_BT/16; ++s) {
      const bf16x8 af = tr_frag(Gs, LDG, 16*s, 32*wid, lane);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const bf16x8 bfr = tr_frag(Hs, LDH, 16*s, 32*j, lane);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[j], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: atomically add the partial tile ---------------------------
  const int fr = lane & 31, fh = lane >> 5;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int k = k0 + 32*j + fr;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int n = n0 + 32*wid + (i & 3) + 8*(i >> 2) + 4*fh;
      if (k < p.Kout) {
        if (n < p.G0p) {
          if (n < p.N0) atomic_add_f32(p.out0 + (long long)n*p.ldo + k, acc[j][i]);
        } else {
          const int n1 = n - p.G0p;
          if (n1 < p.N1) atomic_add_f32(p.out1 + (long long)n1*p.ldo + k, acc[j][i]);
        }
      }
    }
  }
  if ((p.gbias0 || p.gbias1) && htile == 0) {
    __syncthreads();
    float* sc = reinterpret_cast<float*>(Gs);          // [16][128]
#pragma unroll
    for (int j = 0; j < 8; ++j) sc[grow*WG_BG + gcc*8 + j] = bias_acc[j];
    __syncthreads();
    if (tid < WG_BG) {
      float s = 0.f;
      for (int r = 0; r < 16; ++r) s += sc[r*WG_BG + tid];
      const int n = n0 + tid;
      if (n < p.G0p) { if (p.gbias0 && n < p.N0) atomic_add_f32(p.gbias0 + n, s); }
      else if (p.gbias1 && n - p.G0p < p.N1) atomic_add_f32(p.gbias1 + (n - p.G0p), s);
    }
  }
}

}  // namespace brv
