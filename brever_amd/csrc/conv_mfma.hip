// Stride-1 "same" 2-D convolution (1x1 / 3x3) as an implicit GEMM on the fp16 MFMA, fp32
// accumulation, fp32 NCHW activations in HBM: the hot operator of the SGMSE+ score network
// (UNetBlock.conv_1 / conv_2 / skip_conv, AttentionBlock 1x1 projections; reference
// brever/models/sgmse/net.py:352-422, run by the reference under fp16 autocast,
// sgmse.py:190-193).
//
//   D[co][pixel] = sum_{ci, tap} W[co][ci][tap] * X[ci][pixel + tap offset]
//
// Workgroup = 4 waves, output tile 128 channels x (8 rows x 32 columns); wave = 64 channels x
// (4 rows x 32 columns) = 2 x 4 accumulators of 32x32. Per 32-channel chunk of the reduction
// the (8+2) x (32+2) input patch is loaded once with column-coalesced dword loads (zero
// padding and ragged edges through out-of-range buffer offsets), converted to fp16 and laid
// out in LDS as [row][channel-group of 8][column][8 halves], so that the B fragment of
// v_mfma_f32_32x32x16_f16 for any of the 9 taps is one conflict-free ds_read_b128 at a
// shifted column / row; the weights are pre-packed once per model in A-fragment order
// (brv_conv2d_pack_f16) and stream from L2 as 1-KB coalesced loads. LDS is double
// buffered: one barrier per chunk, the next patch is in flight during the MFMAs.
// Small images (fewer than 128 workgroups) split the reduction over workgroups that add their
// partial sums into a preset output with fp32 atomics.
// Optional fusions on the way in (per-(item, channel) affine [+ SiLU] = a folded GroupNorm)
// and out (bias, residual add, scale).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include "../../include/brever_hip.h"
#include "common.cuh"

using namespace brv;

namespace {

#define CM_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return (int)e_; } while (0)

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef CM_PF
#define CM_PF 4       // 32x32 pixel fragments (= output rows) per wave
#endif
#ifndef CM_HALF
#define CM_HALF 1     // 1: fetch the patch in two halves through one small register set
#endif
constexpr int CM_ROWS = 2*CM_PF; // output rows per workgroup
constexpr int CM_COLS = 32;      // output columns per workgroup (= MFMA N)
constexpr int CM_CK = 32;        // reduction channels per chunk (2 MFMA k-steps)
constexpr int CM_MAX_FOLD = 1024; // padded input channels whose folded norm fits the LDS table
#ifndef CM_ABL
#define CM_ABL 0   // compile-time ablation bits of tools/convbench.hip: 1 no patch loads, 2 no A
                   // reloads, 4 no MFMA, 8 no LDS commit, 16 no stores, 32 no B reads
#endif
#ifndef CM_RING
#define CM_RING 3
#endif

struct ConvMfmaParams {
  const float* x; const h8* wp; const float* bias; const float* res; float* y;
  const float* in_scale; const float* in_shift;      // (B, Cin) each, nullable: silu(a*x + b)
  int B, Cin, Cout, H, W, n_wt, n_tiles, n_chunks;
  int n_split, split_chunks;     // split-K over workgroups (small images): atomics into y
  long long x_bs, y_bs;
  float out_scale;
  int in_silu;
};

__device__ __forceinline__ float silu_f(float v) { return v/(1.f + __expf(-v)); }

#ifndef CM_OCC
#define CM_OCC 2     // resident workgroups per CU the register budget is sized for
#endif
#ifndef CM_BPRE
#define CM_BPRE 1    // read the B fragments one step ahead
#endif
template <int KS>
__global__ __launch_bounds__(256, CM_OCC) void conv_mfma_kernel(ConvMfmaParams p) {
  constexpr int PAD = KS/2, TAPS = KS*KS;
  constexpr int PR = CM_ROWS + KS - 1, PC = CM_COLS + KS - 1;
  constexpr int NITEMS = PR*4*PC;                 // (row, channel group, column) 16-byte slots
  constexpr int ROUNDS = (NITEMS + 255)/256;
  __shared__ h8 patch[2][PR*4*PC];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wco = wave >> 1, wpx = wave & 1;
  // XCD-aware order: workgroups go round-robin over the 8 XCDs (each with its own L2), so
  // XCD k takes the k-th contiguous eighth of the tiles -- neighbours share halo rows and the
  // cache lines that straddle tile edges in ONE L2 instead of fetching them 2-3 times from HBM
  const int per_xcd = gridDim.x >> 3;
  const int tile = (blockIdx.x & 7)*per_xcd + (blockIdx.x >> 3);
  if (tile >= p.n_tiles) return;
  const int wt = tile % p.n_wt, ht = tile / p.n_wt;
  const int w0 = wt*CM_COLS, h0 = ht*CM_ROWS;
  const int b = blockIdx.z / p.n_split, split = blockIdx.z % p.n_split;
  const int chunk_lo = split*p.split_chunks;
  const int chunk_hi = min(p.n_chunks, chunk_lo + p.split_chunks);
  const int co_blk = blockIdx.y*2 + wco;
  const bool co_active = co_blk*64 < p.Cout;
  const long long HW = (long long)p.H*p.W;

  const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x + (long long)b*p.x_bs, (long long)p.Cin*HW*4);

  // folded GroupNorm of the input: the item's (scale, shift) table lives in LDS
  __shared__ float fold_tab[2][CM_MAX_FOLD];
  const bool folded = p.in_scale != nullptr;
  if (folded) {
    for (int c = tid; c < p.n_chunks*CM_CK; c += 256) {
      fold_tab[0][c] = c < p.Cin ? p.in_scale[(long long)b*p.Cin + c] : 0.f;
      fold_tab[1][c] = c < p.Cin ? p.in_shift[(long long)b*p.Cin + c] : 0.f;
    }
  }

  // per-thread patch slots: byte offset of channel 0 of the slot's pixel (clamped to 0 when
  // the pixel is padding: the value is discarded at commit time) and the channel group;
  // channels >= Cin are beyond the descriptor's range and read as zero
  unsigned int pix_off[ROUNDS];
  int kg_of[ROUNDS];
  bool pix_ok[ROUNDS];
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const int item = tid + r*256;
    const int row = item/(4*PC), rem = item % (4*PC);
    const int kg = rem/PC, col = rem % PC;
    const int h = h0 + row - PAD, w = w0 + col - PAD;
    pix_ok[r] = item < NITEMS && h >= 0 && h < p.H && w >= 0 && w < p.W;
    pix_off[r] = pix_ok[r] ? (unsigned int)(((long long)h*p.W + w)*4) : 0u;
    kg_of[r] = kg;
  }
  const unsigned int ch_stride = (unsigned int)(HW*4);

  // the patch is fetched in two halves (rounds [0, RH) and [RH, ROUNDS)) through ONE set of
  // RH x 8 staging registers: the first half is committed to LDS in the middle of the chunk
  constexpr int RH = CM_HALF ? (ROUNDS + 1)/2 : ROUNDS;
  float stage[RH][8];
  auto issue = [&](int chunk, int r0) {
    if (CM_ABL & 1) return;
#pragma unroll
    for (int q = 0; q < RH; ++q) {
      const int r = r0 + q;
      if (r >= ROUNDS) continue;
      const unsigned int base = pix_off[r] + (unsigned int)(chunk*CM_CK + kg_of[r]*8)*ch_stride;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        stage[q][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
            xr, (int)(base + (unsigned int)j*ch_stride), 0, 0));
    }
  };
  auto commit = [&](int chunk, int buf, int r0) {
    if (CM_ABL & 8) return;
#pragma unroll
    for (int q = 0; q < RH; ++q) {
      const int r = r0 + q;
      if (r >= ROUNDS) continue;
      const int item = tid + r*256;
      if (item >= NITEMS) continue;
      // the 8 values are converted as a vector: four v_cvt_pk_f16_f32 (element-wise casts compile
      // to a conversion per element plus the packing ors)
      f32x8 tv;
      if (folded) {
        const int c0 = chunk*CM_CK + kg_of[r]*8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float t = fold_tab[0][c0 + j]*stage[q][j] + fold_tab[1][c0 + j];
          const float ts = silu_f(t);
          t = p.in_silu ? ts : t;
          tv[j] = pix_ok[r] ? t : 0.f;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) tv[j] = pix_ok[r] ? stage[q][j] : 0.f;
      }
      patch[buf][item] = __builtin_convertvector(tv, h8);
    }
  };

  f32x16 acc[2][CM_PF];
#pragma unroll
  for (int cf = 0; cf < 2; ++cf)
#pragma unroll
    for (int pf = 0; pf < CM_PF; ++pf)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[cf][pf][i] = 0.f;

  // A fragments stream from L2 through a register ring RD steps deep (a step = one tap x one
  // k-step = 2 fragments); the packed layout makes the whole reduction one contiguous stream.
  constexpr int STEPS = TAPS*2;
  constexpr int RD = KS == 3 ? CM_RING : 2;
  static_assert(STEPS % RD == 0, "ring depth must divide the steps of a chunk");
  const long long total_steps = (long long)p.n_chunks*STEPS;     // of this 64-channel block
  const h8* wa = p.wp + (long long)(co_active ? co_blk : 0)*total_steps*2*64 + lane;
  const long long step_lo = (long long)chunk_lo*STEPS;
  h8 ring[RD][2];
#pragma unroll
  for (int d = 0; d < RD; ++d) {
    const long long f = step_lo + d < total_steps ? step_lo + d : total_steps - 1;
    ring[d][0] = wa[(f*2 + 0)*64];
    ring[d][1] = wa[(f*2 + 1)*64];
  }
  issue(chunk_lo, 0);
  __syncthreads();                 // fold_tab
  commit(chunk_lo, 0, 0);
  if (CM_HALF) { issue(chunk_lo, RH); commit(chunk_lo, 0, RH); }
  __syncthreads();
  const int n32 = lane & 31, khalf = lane >> 5;
  for (int chunk = chunk_lo; chunk < chunk_hi; ++chunk) {
    const int buf = (chunk - chunk_lo) & 1;
    const bool more = chunk + 1 < chunk_hi;
    // issued even after the last chunk (a discarded reload) so that the wait counts of the
    // ring loads do not depend on a branch
    const int nxt = more ? chunk + 1 : chunk;
    issue(nxt, 0);
    // B fragments are read from LDS one step ahead of the MFMAs that consume them
    h8 bf[2][CM_PF];
    auto load_b = [&](h8 (&dst)[CM_PF], int st) {
      const int tap = st >> 1, ks = st & 1;
      const int kh = tap/KS, kw = tap % KS;
#pragma unroll
      for (int pf = 0; pf < CM_PF; ++pf)
        dst[pf] = patch[(CM_ABL & 32) ? 0 : buf][(CM_ABL & 32) ? lane : ((wpx*CM_PF + pf + kh)*4 + ks*2 + khalf)*PC + n32 + kw];
    };
    load_b(bf[0], 0);
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
      if (CM_BPRE) { if (st + 1 < STEPS) load_b(bf[(st + 1) & 1], st + 1); }
      else if (st > 0) load_b(bf[st & 1], st);
      const h8 a0 = ring[st % RD][0], a1 = ring[st % RD][1];
      if (CM_ABL & 4) {
#pragma unroll
        for (int pf = 0; pf < CM_PF; ++pf) asm volatile("" :: "v"(bf[st & 1][pf]), "v"(a0), "v"(a1));
      } else
#pragma unroll
      for (int pf = 0; pf < CM_PF; ++pf) {
        acc[0][pf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bf[st & 1][pf], acc[0][pf], 0, 0, 0);
        acc[1][pf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, bf[st & 1][pf], acc[1][pf], 0, 0, 0);
      }
      long long f = (long long)chunk*STEPS + st + RD;
      if (f >= total_steps) f = total_steps - 1;
      if (!(CM_ABL & 2)) {
      ring[st % RD][0] = wa[(f*2 + 0)*64];
      ring[st % RD][1] = wa[(f*2 + 1)*64];
      }
      if (CM_HALF && st == STEPS/2 - 1) {    // first half lands in the other buffer, second half goes out
        if (more) commit(nxt, buf ^ 1, 0);
        issue(nxt, RH);
      }
      __builtin_amdgcn_sched_barrier(0);     // keep the ring RD steps deep
    }
    if (more) commit(nxt, buf ^ 1, CM_HALF ? RH : 0);
    __syncthreads();
  }

  if (!co_active) return;
  float* yb = p.y + (long long)b*p.y_bs;
  const float* rb = p.res ? p.res + (long long)b*p.y_bs : nullptr;
  const int w = w0 + n32;
  if (p.n_split > 1) {            // y was preset to out_scale*(bias + res) by conv_init_kernel
#pragma unroll
    for (int cf = 0; cf < 2; ++cf)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int co = co_blk*64 + cf*32 + (i >> 2)*8 + khalf*4 + (i & 3);
#pragma unroll
        for (int pf = 0; pf < CM_PF; ++pf) {
          const int h = h0 + wpx*CM_PF + pf;
          if (co < p.Cout && h < p.H && w < p.W)
            atomicAdd(yb + ((long long)co*p.H + h)*p.W + w, acc[cf][pf][i]*p.out_scale);
        }
      }
    return;
  }
#pragma unroll
  for (int cf = 0; cf < 2; ++cf) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int co = co_blk*64 + cf*32 + (i >> 2)*8 + khalf*4 + (i & 3);
      if (co >= p.Cout) continue;
      const float bias = p.bias ? p.bias[co] : 0.f;
#pragma unroll
      for (int pf = 0; pf < CM_PF; ++pf) {
        const int h = h0 + wpx*CM_PF + pf;
        if (h >= p.H || w >= p.W) continue;
        if ((CM_ABL & 16) && acc[cf][pf][i] != 12345.f) continue;
        const long long o = ((long long)co*p.H + h)*p.W + w;
        float v = acc[cf][pf][i] + bias;
        if (rb) v += rb[o];
        yb[o] = v*p.out_scale;
      }
    }
  }
}

// y = out_scale*(bias[co] + res): the value the split-K workgroups add their partial sums to
__global__ __launch_bounds__(256) void conv_init_kernel(const float* bias, const float* res, float* y,
                                                        int Cout, long long HW, long long y_bs,
                                                        float out_scale) {
  const long long n = (long long)Cout*HW;
  const long long bo = (long long)blockIdx.y*y_bs;
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < n; i += (long long)gridDim.x*256) {
    float v = bias ? bias[i / HW] : 0.f;
    if (res) v += res[bo + i];
    y[bo + i] = v*out_scale;
  }
}

// wp[co_blk][chunk][tap][kstep][cofrag][lane][8] <- w[co][ci][tap] (zero beyond Cout / Cin)
__global__ __launch_bounds__(256) void conv_pack_kernel(const float* w, _Float16* wp, int Cout,
                                                        int Cin, int taps, int n_chunks,
                                                        long long total) {
  for (long long idx = (long long)blockIdx.x*256 + threadIdx.x; idx < total;
       idx += (long long)gridDim.x*256) {
    long long r = idx;
    const int j = (int)(r % 8); r /= 8;
    const int lane = (int)(r % 64); r /= 64;
    const int cf = (int)(r % 2); r /= 2;
    const int ks = (int)(r % 2); r /= 2;
    const int tap = (int)(r % taps); r /= taps;
    const int chunk = (int)(r % n_chunks); r /= n_chunks;
    const int co = (int)r*64 + cf*32 + (lane & 31);
    const int ci = chunk*CM_CK + ks*16 + (lane >> 5)*8 + j;
    float v = 0.f;
    if (co < Cout && ci < Cin) v = w[((long long)co*Cin + ci)*taps + tap];
    wp[idx] = (_Float16)v;
  }
}

}  // namespace

extern "C" {

int64_t brv_conv2d_packed_size(int64_t Cout, int64_t Cin, int64_t ksize) {
  if (Cout < 1 || Cin < 1 || (ksize != 1 && ksize != 3)) return -1;
  return ((Cout + 63)/64)*64*((Cin + CM_CK - 1)/CM_CK)*CM_CK*ksize*ksize;
}

int brv_conv2d_pack_f16(const float* w, void* wp, int64_t Cout, int64_t Cin, int64_t ksize,
                        brv_stream_t stream) {
  const int64_t total = brv_conv2d_packed_size(Cout, Cin, ksize);
  if (total < 0) return -1;
  long long g = (total + 255)/256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(conv_pack_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w,
                     (_Float16*)wp, (int)Cout, (int)Cin, (int)(ksize*ksize),
                     (int)((Cin + CM_CK - 1)/CM_CK), (long long)total);
  CM_OK(hipGetLastError());
  return 0;
}

int brv_conv2d_mfma_forward(const float* x, const void* wp, const float* bias, const float* res,
                            const float* in_scale, const float* in_shift, int in_silu, float* y,
                            int64_t B, int64_t Cin, int64_t H, int64_t W, int64_t Cout,
                            int64_t ksize, int64_t x_batch_stride, int64_t y_batch_stride,
                            float out_scale, brv_stream_t stream) {
  if (B < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1 || (ksize != 1 && ksize != 3)) return -1;
  const int64_t n_chunks = (Cin + CM_CK - 1)/CM_CK;
  if ((n_chunks*CM_CK + 1)*H*W*4 >= (1LL << 32)) return -2;      // 32-bit buffer offsets
  if (in_scale != nullptr && n_chunks*CM_CK > CM_MAX_FOLD) return -3;
  ConvMfmaParams p;
  p.x = x; p.wp = (const h8*)wp; p.bias = bias; p.res = res; p.y = y;
  p.in_scale = in_scale; p.in_shift = in_shift; p.in_silu = in_silu;
  p.B = (int)B; p.Cin = (int)Cin; p.Cout = (int)Cout; p.H = (int)H; p.W = (int)W;
  p.n_wt = (int)((W + CM_COLS - 1)/CM_COLS); p.n_chunks = (int)n_chunks;
  p.n_tiles = (int)(p.n_wt*((H + CM_ROWS - 1)/CM_ROWS));
  p.x_bs = x_batch_stride; p.y_bs = y_batch_stride; p.out_scale = out_scale;

  // small images leave most CUs idle and run the whole reduction as one latency chain: split it
  const int64_t n_wg = p.n_wt*((H + CM_ROWS - 1)/CM_ROWS)*((Cout + 127)/128)*B;
  int64_t n_split = 1;
  if (n_wg < 128 && n_chunks > 1) {
    n_split = 256/n_wg;
    if (n_split > n_chunks) n_split = n_chunks;
  }
  p.split_chunks = (int)((n_chunks + n_split - 1)/n_split);
  p.n_split = (int)((n_chunks + p.split_chunks - 1)/p.split_chunks);
  if (p.n_split > 1) {
    long long g = (Cout*H*W + 255)/256;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(conv_init_kernel, dim3((unsigned)g, (unsigned)B), dim3(256), 0,
                       (hipStream_t)stream, bias, res, y, (int)Cout, (long long)(H*W),
                       (long long)y_batch_stride, out_scale);
  }
  const dim3 grid((unsigned)((p.n_tiles + 7)/8*8), (unsigned)((Cout + 127)/128),
                  (unsigned)(B*p.n_split));
  if (ksize == 3)
    hipLaunchKernelGGL(conv_mfma_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL(conv_mfma_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, p);
  CM_OK(hipGetLastError());
  return 0;
}

}  // extern "C"
