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
// barrier per chunk (5 / 3 / 2 k steps of MF x NF MFMAs per wave).
#include "common.cuh"
#include "../../include/brever_hip.h"

namespace {
using namespace brv;

constexpr int CC_THREADS = 512;
constexpr int CC_KH = 5;

struct CConvParams {
  const float* in; const uint4* wp; const float* bias; float* out;
  int B, C, M, Hin, Win, Hout, Wout;
  long long in_bs, out_bs;
  int mode;                    // 0 strided, 1 transposed
  int ncc;                     // chunks of 8 input channels
  int mtiles, ftiles;
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

template <int MF, int NF, int WM, int WN, int NTAP>
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
  const float* in_b = p.in + (long long)b*p.in_bs;

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
  float4 st[NTAP][IT];
  auto stage_load = [&](int cc) {
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
      const int row = tap_row[t];
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        const int ch = 8*cc + s_c[it], f = s_f[it];
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row >= 0 && row < p.Hin && ch < p.C) {
          const float* src = in_b + ((long long)ch*p.Hin + row)*p.Win;
          if (f >= 0 && f + 3 < p.Win) __builtin_memcpy(&v, src + f, 16);
          else {
            if (f >= 0 && f < p.Win) v.x = src[f];
            if (f + 1 >= 0 && f + 1 < p.Win) v.y = src[f + 1];
            if (f + 2 >= 0 && f + 2 < p.Win) v.z = src[f + 2];
            if (f + 3 >= 0 && f + 3 < p.Win) v.w = src[f + 3];
          }
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
        *reinterpret_cast<uint2*>(lds + buf*BUFB + t*TAPB + s_off[it]) =
            make_uint2(pack2(st[t][it].x, st[t][it].y), pack2(st[t][it].z, st[t][it].w));
  };

  // ---- weights: fragments of this wave's MF row groups, (chunk, tap) two k steps ahead
  const int mfrag0 = (mtile*WM + wm)*MF;
  const uint4* wq[MF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) wq[mf] = p.wp + (long long)(mfrag0 + mf)*p.ncc*CC_KH*64 + lane;
  auto a_load = [&](int cc, int t, uint4 (&dst)[MF]) {
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) dst[mf] = wq[mf][(cc*CC_KH + tap_i[t])*64];
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
  float* out_b = p.out + (long long)b*p.out_bs;
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
          out_b[((long long)m*p.Hout + r)*p.Wout + w] = v;
        }
      }
    }
}

template <int MF, int NF, int WM, int WN>
__global__ __launch_bounds__(CC_THREADS) void cconv_rows_kernel(const CConvParams p) {
  constexpr int NT = 32*NF*WN;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2*CC_KH*(NT/128)*4096];
  const int ftile = blockIdx.x, r = blockIdx.y;
  const int b = blockIdx.z / p.mtiles, mtile = blockIdx.z % p.mtiles;
  if (p.mode == 0) {
    const int ti[5] = {0, 1, 2, 3, 4};
    const int tr[5] = {2*r - 2, 2*r - 1, 2*r, 2*r + 1, 2*r + 2};
    cconv_tile<MF, NF, WM, WN, 5>(p, lds, b, r, ftile, mtile, ti, tr, 1);
  } else if (r & 1) {
    const int ti[2] = {1, 3};
    const int tr[2] = {(r + 1) >> 1, (r - 1) >> 1};
    cconv_tile<MF, NF, WM, WN, 2>(p, lds, b, r, ftile, mtile, ti, tr, -1);
  } else {
    const int ti[3] = {0, 2, 4};
    const int tr[3] = {(r >> 1) + 1, r >> 1, (r >> 1) - 1};
    cconv_tile<MF, NF, WM, WN, 3>(p, lds, b, r, ftile, mtile, ti, tr, -1);
  }
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

int brv_cconv_rows(const float* in, const void* wp, const float* bias, float* out, int64_t B, int64_t C,
                   int64_t M, int64_t Hin, int64_t Win, int32_t transposed, brv_stream_t stream) {
  if (!in || !wp || !out || B < 1 || C < 1 || M < 1 || Hin < 1 || Win < 1) return -1;
  if (!transposed && ((Hin & 1) || Win < 2)) return -1;
  CConvParams p;
  p.in = in; p.wp = (const uint4*)wp; p.bias = bias; p.out = out;
  p.B = (int)B; p.C = (int)C; p.M = (int)M; p.Hin = (int)Hin; p.Win = (int)Win;
  p.Hout = transposed ? (int)(2*Hin) : (int)(Hin/2);
  p.Wout = transposed ? (int)(Win + 1) : (int)(Win - 1);
  p.in_bs = C*Hin*Win; p.out_bs = M*(long long)p.Hout*p.Wout;
  p.mode = transposed ? 1 : 0;
  p.ncc = (int)((C + 7)/8);
  hipStream_t st = (hipStream_t)stream;
#define CC_LAUNCH(MF_, NF_, WM_, WN_) do { \
    constexpr int MT = 32*MF_*WM_, NT = 32*NF_*WN_; \
    p.mtiles = (int)((M + MT - 1)/MT); p.ftiles = (p.Wout + NT - 1)/NT; \
    hipLaunchKernelGGL((cconv_rows_kernel<MF_, NF_, WM_, WN_>), dim3(p.ftiles, p.Hout, (unsigned)(B*p.mtiles)), \
                       dim3(CC_THREADS), 0, st, p); } while (0)
  if (M > 128) {
    const long long wgs = (long long)((p.Wout + 255)/256)*p.Hout*B*((M + 255)/256);
    if (wgs < 256) CC_LAUNCH(2, 2, 4, 2); else CC_LAUNCH(2, 4, 4, 2);
  } else if (M > 64) CC_LAUNCH(2, 2, 2, 4);
  else if (M > 32) CC_LAUNCH(2, 1, 1, 8);
  else CC_LAUNCH(1, 1, 1, 8);
#undef CC_LAUNCH
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // extern "C"
