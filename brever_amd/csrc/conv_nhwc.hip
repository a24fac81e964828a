// Stride-1 "same" 3x3 convolution on channels-last fp16 activations, fp16 MFMA with fp32
// accumulation: the hot operator of the SGMSE+ score network under use_amp (UNetBlock.conv_1 /
// conv_2, AuxiliaryUp/Down 3x3; reference brever/models/sgmse/net.py:352-422, run by the
// reference under fp16 autocast, sgmse.py:190-193 -- there, too, the activations between the
// convolutions are fp16).
//
//   Y[b][h][w][co] = out_scale*(bias[co] + res[b][h][w][co]
//                               + sum_{ci,kh,kw} W[co][ci][kh][kw]*act(X)[b][h+kh-1][w+kw-1][ci])
//   act(X) = X, or silu?(in_scale[b][ci]*X + in_shift[b][ci]) -- a GroupNorm folded to an affine
//   map; X may be the channel concatenation of two tensors (the U-Net's skip connections).
//
// Design (one workgroup of 8 waves per CU, persistent over a contiguous range of tiles):
//  * output tile 128 channels x (16 rows x 32 columns); wave (wco, wpx) owns 64 channels x
//    (4 rows x 32 columns) = 2 x 4 accumulators of v_mfma_f32_32x32x16_f16 (D[co][pixel]);
//  * the reduction runs in chunks of 32 input channels x 9 taps. Per chunk the (18 x 34)-pixel
//    input patch (64 B per pixel) goes global -> LDS by LDS-DMA (global_load_lds_dwordx4: no
//    staging registers, no conversion pass); it is laid out pixel-major with the four 16-byte
//    channel octets of a pixel XOR-swizzled by ((pixel >> 2) & 3) -- applied on the SOURCE
//    address, the DMA destination is lane-linear -- so that the B fragment of any tap is one
//    conflict-free ds_read_b128; zero padding = lanes pointed at a block of zeros;
//  * the weights, pre-packed once per model in A-fragment order (brv_conv_nhwc_pack), stream
//    through a 4-slot LDS ring (8 KB per tap) three taps ahead of their use: every wave of the
//    workgroup reads the same fragments from LDS instead of each streaming them from L2;
//  * one raw s_barrier per tap (16 MFMAs per wave); DMAs stay in flight across the barriers
//    behind counted s_waitcnt vmcnt(N); every LDS access of the loop is inline asm with counted
//    lgkmcnt waits (hipcc would drain vmcnt(0) before any ds_read it can see);
//  * a folded GroupNorm (+SiLU) is applied in LDS: each lane rewrites the slots its own DMAs
//    filled (ordered by its own vmcnt), spread over the taps of the previous chunk;
//  * the next tile's first patch and weights are prefetched during the current tile's last
//    chunk; bias + residual + scale on the way out, fp16 channels-last.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <stdlib.h>
#include <string.h>
#include <type_traits>

#include "../../include/brever_hip.h"
#include "common.cuh"

using namespace brv;

namespace {

#define CN_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return (int)e_; } while (0)

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* cn_lds_p;
typedef __attribute__((address_space(1))) const void* cn_glb_p;

constexpr int CN_WAVES = 8, CN_THREADS = 64*CN_WAVES;
constexpr int CN_COLS = 32;                   // output columns of a tile (rows: 4*PF)
constexpr int CN_CK = 32;                     // input channels per chunk (2 k-steps)
#ifndef CN_DA
#define CN_DA 4                               // the weight stream runs this many taps ahead
#endif
#ifndef CN_DA_SMALL
#define CN_DA_SMALL 8                         // ... for the 8- and 4-row tiles: their taps are short, the
#endif                                       // weight stream is bound by what is in flight
constexpr int CN_ASLOT = 8192;                // [k-step 2][co fragment 4][lane 64][8 halves]
constexpr int CN_MAXC = 1024;                 // input channels of a self-folding launch (FOLD == 2)
constexpr int CN_TAB = 1024;                  // per wave and parity: 32 scales + 32 shifts + 128 biases (fp32)
#ifndef CN_VAR
#define CN_VAR 4    // experiment bits: 1 wco-0 waves issue their DMAs between the k-steps, 2 s_setprio around MFMAs
#endif
#ifndef CN_ABL
#define CN_ABL 0    // ablation bits of tools/convbench2.hip: 1 no patch DMA, 2 no weight DMA,
                    // 4 no MFMA, 8 no transform, 16 no stores, 32 no fragment reads
#endif

struct ConvNhwcParams {
  const _Float16* x1; const _Float16* x2;     // (B, H, W, C1s) [, (B, H, W, C2s)]
  const unsigned char* wp;                    // packed weights
  const float* bias; const _Float16* res; _Float16* y;
  const float* in_scale; const float* in_shift;   // (B, Cin) each (FOLD == 1)
  // FOLD == 2: the GroupNorm is folded by the workgroup itself from the per-channel sums
  const double* sums1; const double* sums2;       // (B, C1, 2), (B, C2, 2)
  const float* gn_add; const float* gn_gamma; const float* gn_beta;
  const float* adm_scale; const float* adm_shift; // (B, Cin) nullable
  int C1, C2, groups; float eps;
  const unsigned char* zeros;                 // >= 16 bytes of zeros
  double* stats;                              // nullable (B, Cout, 2): += per-channel (sum, sum of squares) of y
  int B, H, W, C1s, C2s, n_chunks1, n_chunks, Cin, Cout, Cys, Crs;
  int n_wt, n_ht, n_cob, n_tiles;
  float out_scale; int in_silu;
  int stagger;          // start delay per XCD phase, in units of ~0.5 us (s_sleep 16)
#ifdef CN_DIAG
  unsigned long long* dbg;   // [wave 8][tap 9][stamp 4] of workgroup 0, chunk 2; then [2] realtime/cycle span
#endif
};

// v*sigmoid(v): v_exp + v_rcp (a plain '/' expands to the ~11-instruction IEEE division; the
// reciprocal's 1 ulp is far below the fp16 rounding of the result)
__device__ __forceinline__ float cn_silu(float v) {
  return v*__builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f*v));
}

__device__ __forceinline__ void cn_glds16(const void* g, unsigned char* l) {
  __builtin_amdgcn_global_load_lds((cn_glb_p)g, (cn_lds_p)l, 16, 0, 0);
}
__device__ __forceinline__ unsigned int cn_lds_addr(const void* p) {
  return (unsigned int)(unsigned long long)p;
}
template <int OFF>
__device__ __forceinline__ u32x4 cn_read16(unsigned int addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
__device__ __forceinline__ void cn_write16(unsigned int addr, const u32x4& v) {
  asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(v) : "memory");
}
template <int N>
__device__ __forceinline__ void cn_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
// the fragments of one k-step become valid at this wait (whole-vector ties: the compiler must
// not read a component before it)
template <int N>
__device__ __forceinline__ void cn_wait_frags(u32x4 (&a)[2], u32x4 (&b)[4]) {
  asm volatile("s_waitcnt lgkmcnt(%6)"
               : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3])
               : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void cn_wait5(u32x4& a, u32x4& b, u32x4& c, u32x4& d, u32x4& e) {
  asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e) : "n"(N) : "memory");
}

template <int N>
__device__ __forceinline__ void cn_wait_tie2(u32x4& a, u32x4& b) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void cn_wait_tie1(u32x4& a) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N) : "memory");
}

template <int V> using cn_int = std::integral_constant<int, V>;

#ifdef CN_DIAG
__device__ __forceinline__ unsigned long long cn_stamp() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  return t;
}
#define CN_STAMP(i) do { if (diag_on) stamps[TP][i] = cn_stamp(); } while (0)
#else
#define CN_STAMP(i) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------
// DMA issue order of one chunk (KS = 3, ROUNDS = 5 patch pieces P, one weight piece A per tap, one
// table piece T): tap 0: T A P0 | tap 1: A P1 | ... | tap 4: A P4 | taps 5-8: A. The same pattern
// every chunk, so every wait count below is a compile-time function of the tap.
template <int ROUNDS, int DA>
struct CnSched {
  static constexpr int P(int tp) { return tp >= 0 && tp < ROUNDS ? 1 : 0; }
  static constexpr int ops(int tp) { return 1 + P(tp) + (tp == 0 ? 1 : 0); }
  static constexpr int mod9(int tp) { return (tp + 90) % 9; }
  // top of tap tp: the weights of tap t+1 (issued DA-1 taps ago, followed in their tap only
  // by a patch piece) have landed when at most this many younger DMAs are pending
  static constexpr int top(int tp) {
    int n = P(mod9(tp + 1 - DA));
    for (int k = tp + 2 - DA; k <= tp - 1; ++k) n += ops(mod9(k));
    return n;
  }
  // after the issues of tap tp: patch piece r = tp - 3 (last DMA of tap r) has landed
  static constexpr int piece(int tp) { return ops(tp - 2) + ops(tp - 1) + ops(tp); }
};

// PF = accumulator rows per wave: the tile is (4*PF rows) x 32 columns x 128 channels. PF = 4 for
// images with enough tiles to fill the chip; smaller PF gives small images 2-4x the workgroups
// (each still streams all the weights: their taps are then paced by the DMA, not the MFMAs).
// FOLD: 0 plain input; 1 scale / shift (B, Cin) from memory (brv_groupnorm_fold_chan); 2 the fold is
// computed in the prologue from the per-channel sums (every tile of the workgroup belongs to one
// item): no fold launch between the producing convolution and this one.
template <int KS, int FOLD, int PF>
__global__ __launch_bounds__(CN_THREADS) void conv_nhwc_kernel(const ConvNhwcParams p) {
  static_assert(KS == 3, "3x3 only");
  constexpr int CN_ROWS = 4*PF;
  constexpr int TAPS = KS*KS, PADK = KS/2;
  constexpr int PR = CN_ROWS + KS - 1, PC = CN_COLS + KS - 1;
  constexpr int NSLOT = PR*PC*4;
  constexpr int ROUNDS = (NSLOT + CN_THREADS - 1)/CN_THREADS;
  // (a patch buffer doubles as the epilogue's staging area: 128 pixel rows of 272 bytes)
  constexpr int PBYTES = ROUNDS*CN_THREADS*16 > 35*1024 ? ROUNDS*CN_THREADS*16 : 35*1024;
  constexpr int OFF_A = 2*PBYTES;
  // weight prefetch depth, ring slots (one slot less where the self-fold table needs the LDS)
  constexpr int DA = PF == 4 ? CN_DA : (FOLD == 2 ? CN_DA_SMALL - 1 : CN_DA_SMALL), NA = DA + 1;
  constexpr int OFF_TAB = OFF_A + NA*CN_ASLOT;
  constexpr int OFF_FOLD = OFF_TAB + 2*CN_WAVES*CN_TAB;
  constexpr int SMEM = OFF_FOLD + (FOLD == 2 ? 2*CN_MAXC*4 : 0);
  using S = CnSched<ROUNDS, DA>;
  // ONE shared array (a second object makes hipcc drain vmcnt before LDS reads)
  __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform: scalar branches
  const int wco = wave >> 2, wpx = wave & 3;
  const int n32 = lane & 31, khalf = lane >> 5;
  const unsigned int smem_a = cn_lds_addr(smem);

  // ---- this workgroup's tiles: XCD k (own L2) takes the k-th contiguous eighth of the tiles,
  // a workgroup a contiguous run of them: halo rows / columns are shared in one L2
  const int G = gridDim.x;
  const int rank = (G & 7) == 0 ? (blockIdx.x & 7)*(G >> 3) + (blockIdx.x >> 3) : blockIdx.x;
  const int tile_lo = (int)((long long)rank*p.n_tiles/G);
  const int tile_hi = (int)((long long)(rank + 1)*p.n_tiles/G);
  if (tile_lo >= tile_hi) return;
  // Start skew: all tiles take the same time, so without it every workgroup reaches its epilogue at
  // the same moment and the output burst of the whole chip (128 KB per workgroup) queues on the
  // memory system while no matrix pipe runs. The 32 workgroups of an XCD (an XCD's write path is
  // its own) start in 8 groups a little apart instead.
  if (tile_hi - tile_lo > 1)
    for (int i = (int)((blockIdx.x >> 3) & 7)*p.stagger; i > 0; --i) __builtin_amdgcn_s_sleep(16);
  const int n_chunks = p.n_chunks;
  const int n_work = (tile_hi - tile_lo)*n_chunks;
  const int taps_per_tile = n_chunks*TAPS;

  struct Tile { int b, h0, w0, cob; };
  auto decode = [&](int t) {
    Tile q;
    q.cob = t % p.n_cob; t /= p.n_cob;
    q.w0 = (t % p.n_wt)*CN_COLS; t /= p.n_wt;
    q.h0 = (t % p.n_ht)*CN_ROWS; q.b = t / p.n_ht;
    return q;
  };

  // the tile after q in the workgroup's run (divisions only once, in decode)
  auto next_tile = [&](Tile& q) {
    if (++q.cob < p.n_cob) return;
    q.cob = 0; q.w0 += CN_COLS;
    if (q.w0 < p.n_wt*CN_COLS) return;
    q.w0 = 0; q.h0 += CN_ROWS;
    if (q.h0 < p.n_ht*CN_ROWS) return;
    q.h0 = 0; ++q.b;
  };

  // ---- patch pieces: slot = (r*8 + wave)*64 + lane = 4*pixel + q holds channel octet
  // kg = q ^ ((pcol >> 2) & 3) of patch pixel (prow, pcol). Per piece ONE register for the tile
  // whose chunks are being prefetched: (pixel index inside the item) << 2 | kg, negative = padding
  int nx_pk[ROUNDS];
  const _Float16 *nx_x1 = nullptr, *nx_x2 = nullptr;
  const float *nx_sc = nullptr, *nx_sh = nullptr, *nx_bias = nullptr;
  int nx_bias_co = 0;
  auto set_prefetch_tile = [&](const Tile& q) {
    int tv = tid;                      // opaque: the per-round constants below are cheap to
    asm volatile("" : "+v"(tv));       // recompute, hoisted out of the main loop they are spilled
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int slot = r*CN_THREADS + tv;
      const int px = slot >> 2;
      const int kg = (slot & 3) ^ (((px % PC) >> 2) & 3);
      const int h = q.h0 + px / PC - PADK, w = q.w0 + px % PC - PADK;
      const bool ok = slot < NSLOT && h >= 0 && h < p.H && w >= 0 && w < p.W;
      nx_pk[r] = ((ok ? h*p.W + w : -1) << 2) | kg;
    }
    const long long hw = (long long)p.H*p.W;
    nx_x1 = p.x1 + (long long)q.b*hw*p.C1s;
    nx_x2 = p.x2 ? p.x2 + (long long)q.b*hw*p.C2s : nullptr;
    if (FOLD == 1) { nx_sc = p.in_scale + (long long)q.b*p.Cin; nx_sh = p.in_shift + (long long)q.b*p.Cin; }
    nx_bias_co = q.cob*128;
    nx_bias = p.bias + nx_bias_co;
  };
  auto issue_patch = [&](int r, int chunk, int par) {
    if (CN_ABL & 1) return;
    const bool second = chunk >= p.n_chunks1;
    const _Float16* base = second ? nx_x2 : nx_x1;
    const int cs = second ? p.C2s : p.C1s;
    const int c0 = (second ? chunk - p.n_chunks1 : chunk)*CN_CK + (nx_pk[r] & 3)*8;
    const bool ok = nx_pk[r] >= 0 && c0 < cs;
    const void* src = ok ? (const void*)(base + (long long)(nx_pk[r] >> 2)*cs + c0) : (const void*)p.zeros;
    cn_glds16(src, smem + par*PBYTES + (r*CN_WAVES + wave)*1024);
  };
  // the table piece of a chunk, private to the wave (ordered by its own vmcnt): lanes 0-7 the 32
  // scales, 8-15 the 32 shifts of the chunk's channels (FOLD), lanes 16-47 the 128 biases of the
  // chunk's tile (no VGPR-destination loads inside the loop: hipcc would wait vmcnt(0) for them)
  auto issue_table = [&](int chunk, int par) {
    const float* src = (const float*)p.zeros;
    if (FOLD == 1 && lane < 16) src = (lane < 8 ? nx_sc : nx_sh) + chunk*CN_CK + (lane & 7)*4;
    if (lane >= 16 && lane < 48 && p.bias && nx_bias_co + (lane - 16)*4 + 4 <= p.Cout)
      src = nx_bias + (lane - 16)*4;
    cn_glds16(src, smem + OFF_TAB + (par*CN_WAVES + wave)*CN_TAB);
  };
  // rewrite own slot r of the patch with parity par: silu?(scale*x + shift), zero for padding.
  // Split in two so that the arithmetic sits behind the MFMAs of the tap: reads (5 DS
  // operations), then math + write.
  struct XfRaw { u32x4 raw, s0, s1, t0, t1; };
  auto transform_reads = [&](int r, int par, int chunk, XfRaw& v) {
    const unsigned int sa = smem_a + par*PBYTES + (r*CN_WAVES + wave)*1024 + lane*16;
    v.raw = cn_read16<0>(sa);
    if (FOLD == 2) {                               // the workgroup's own table: [scale Cin | shift Cin]
      const unsigned int ta = smem_a + OFF_FOLD + (chunk*CN_CK + (nx_pk[r] & 3)*8)*4;
      v.s0 = cn_read16<0>(ta); v.s1 = cn_read16<16>(ta);
      v.t0 = cn_read16<CN_MAXC*4>(ta); v.t1 = cn_read16<CN_MAXC*4 + 16>(ta);
    } else {                                       // this wave's table piece of the chunk
      const unsigned int ta = smem_a + OFF_TAB + (par*CN_WAVES + wave)*CN_TAB + (nx_pk[r] & 3)*32;
      v.s0 = cn_read16<0>(ta); v.s1 = cn_read16<16>(ta);
      v.t0 = cn_read16<128>(ta); v.t1 = cn_read16<144>(ta);
    }
  };
  auto transform_write = [&](int r, int par, const XfRaw& v) {
    const unsigned int sa = smem_a + par*PBYTES + (r*CN_WAVES + wave)*1024 + lane*16;
    const h8 xv = __builtin_bit_cast(h8, v.raw);
    const f32x8 xf = __builtin_convertvector(xv, f32x8);
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sc[j] = __uint_as_float(v.s0[j]); sc[4 + j] = __uint_as_float(v.s1[j]);
      sh[j] = __uint_as_float(v.t0[j]); sh[4 + j] = __uint_as_float(v.t1[j]);
    }
    const bool ok = nx_pk[r] >= 0;
    f32x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float t = sc[j]*xf[j] + sh[j];
      const float ts = cn_silu(t);
      t = p.in_silu ? ts : t;
      o[j] = ok ? t : 0.f;
    }
    const h8 ov = __builtin_convertvector(o, h8);
    cn_write16(sa, __builtin_bit_cast(u32x4, ov));
  };

  // ---- weight stream (CN_DA taps ahead): piece `wave` of the tap's 8 KB
  int a_tile = tile_lo, a_pos = 0;
  const unsigned int a_lane = wave*1024 + lane*16;
  int a_cob = decode(tile_lo).cob;
  unsigned int a_off = (unsigned int)a_cob*(unsigned int)taps_per_tile*CN_ASLOT;
  auto issue_weights = [&](int slot) {
    if (!(CN_ABL & 2)) cn_glds16(p.wp + a_off + a_lane, smem + OFF_A + slot*CN_ASLOT + wave*1024);
    a_off += CN_ASLOT;
    if (++a_pos == taps_per_tile) {
      a_pos = 0;
      if (a_tile + 1 < tile_hi) { ++a_tile; a_cob = a_cob + 1 < p.n_cob ? a_cob + 1 : 0; }   // past the end: harmless reloads
      a_off = (unsigned int)a_cob*(unsigned int)taps_per_tile*CN_ASLOT;
    }
  };

  // ---- fragment reads
  // B fragment of (pf, tap (kh, kw), k-step): pixel (wpx*PF + pf + kh, n32 + kw), octet
  // 2*kstep + khalf at slot octet ^ ((pcol >> 2) & 3): the swizzle depends on the column only, so
  // the address is one of three per-lane column terms (kw) + an immediate row offset, and the
  // second k-step is the first XOR 32 bytes
  unsigned int b_col[KS];
#pragma unroll
  for (int kw = 0; kw < KS; ++kw) {
    const int pcol = n32 + kw;
    b_col[kw] = smem_a + ((wpx*PF)*PC + pcol)*64 + ((khalf ^ ((pcol >> 2) & 3)) << 4);
  }
  const unsigned int a_rd = smem_a + OFF_A + wco*2048 + lane*16;
  auto read_frags = [&](auto tapc, auto ksc, int par, int slot, u32x4 (&a)[2], u32x4 (&b)[4]) {
    constexpr int TAP = decltype(tapc)::value, KSTEP = decltype(ksc)::value;
    constexpr int kh = TAP / KS, kw = TAP % KS;
    if (CN_ABL & 32) return;
    const unsigned int ab = a_rd + slot*CN_ASLOT;
    a[0] = cn_read16<KSTEP*4096>(ab);
    a[1] = cn_read16<KSTEP*4096 + 1024>(ab);
    const unsigned int bb = (b_col[kw] ^ (KSTEP*32)) + par*PBYTES;
    b[0] = cn_read16<(0 + kh)*PC*64>(bb);
    if constexpr (PF > 1) b[1] = cn_read16<(1 + kh)*PC*64>(bb);
    if constexpr (PF > 2) {
      b[2] = cn_read16<(2 + kh)*PC*64>(bb);
      b[3] = cn_read16<(3 + kh)*PC*64>(bb);
    }
  };

  f32x16 acc[2][PF];
  auto zero_acc = [&]() {
#pragma unroll
    for (int cf = 0; cf < 2; ++cf)
#pragma unroll
      for (int pf = 0; pf < PF; ++pf)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[cf][pf][i] = 0.f;
  };
  auto mfma_step = [&](u32x4 (&a)[2], u32x4 (&b)[4], int pf_lo, int pf_hi) {
    const h8 a0 = __builtin_bit_cast(h8, a[0]), a1 = __builtin_bit_cast(h8, a[1]);
#pragma unroll
    for (int pf = 0; pf < PF; ++pf) {
      if (pf < pf_lo || pf >= pf_hi) continue;
      const h8 bv = __builtin_bit_cast(h8, b[pf]);
      if (CN_ABL & 4) { asm volatile("" :: "v"(a0), "v"(a1), "v"(bv)); continue; }
      acc[0][pf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bv, acc[0][pf], 0, 0, 0);
      acc[1][pf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, bv, acc[1][pf], 0, 0, 0);
    }
  };

  // Epilogue through LDS (the patch buffer of the chunk just finished): four passes, one per
  // accumulator row group pf. The waves write fp16 [pixel][128 channels] rows (272-byte stride:
  // 8-byte writes of 32 lanes fall on distinct bank pairs), then every thread moves 16-byte
  // pieces: + bias + residual, * scale, fully coalesced 256-byte pixel rows on the way out.
  constexpr int ESTRIDE = 272;
  static_assert(128*ESTRIDE <= PBYTES, "epilogue staging fits a patch buffer");
  auto store_tile = [&](const Tile& q, int par) {
    if (CN_ABL & 16) {
      float s = 0.f;
#pragma unroll
      for (int cf = 0; cf < 2; ++cf)
#pragma unroll
        for (int pf = 0; pf < PF; ++pf)
#pragma unroll
          for (int i = 0; i < 16; ++i) s += acc[cf][pf][i];
      if (s == 12345.678f) p.y[0] = (_Float16)s;
      return;
    }
#ifdef CN_DIAG
    unsigned long long es[13]; int esn = 0;
    es[esn++] = cn_stamp();
#endif
    const unsigned int stg = smem_a + par*PBYTES;
    int tv = tid;                      // opaque: keeps the epilogue's address arithmetic out of the
    asm volatile("" : "+v"(tv));       // main loop's register budget
    const long long hw = (long long)p.H*p.W;
    _Float16* yb = p.y + (long long)q.b*hw*p.Cys;
    const _Float16* rb = p.res ? p.res + (long long)q.b*hw*p.Crs : nullptr;
    const int c8 = tv & 15, co = q.cob*128 + c8*8;
    // biases of the tile: this wave's table piece of the chunk just finished
    const unsigned int ba = smem_a + OFF_TAB + (par*CN_WAVES + wave)*CN_TAB + 256 + c8*32;
    u32x4 b0 = cn_read16<0>(ba), b1 = cn_read16<16>(ba);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b0), "+v"(b1) :: "memory");
    float bv[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { bv[j] = __uint_as_float(b0[j]); bv[4 + j] = __uint_as_float(b1[j]); }
    // per-channel statistics of the stored values (the next GroupNorm's): 16 pixels per thread
    float st_s[8], st_q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { st_s[j] = 0.f; st_q[j] = 0.f; }
#pragma unroll
    for (int pf = 0; pf < PF; ++pf) {
      // the residual pieces of this pass first: they travel while the accumulators are staged
      u32x4 raw[4], rr[4];
      bool ok[4]; long long pix[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int px = (it*CN_THREADS + tv) >> 4;             // 0..127: row group px >> 5
        const int h = q.h0 + (px >> 5)*PF + pf, w = q.w0 + (px & 31);
        ok[it] = h < p.H && w < p.W && co < p.Cout;
        pix[it] = (long long)h*p.W + w;
        rr[it] = u32x4{0u, 0u, 0u, 0u};
      }
      if (rb) {
#pragma unroll
        for (int it = 0; it < 4; ++it)
          if (ok[it]) rr[it] = *reinterpret_cast<const u32x4*>(rb + pix[it]*p.Crs + co);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();              // last readers of the buffer / of the previous pass
#pragma unroll
      for (int cf = 0; cf < 2; ++cf)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          h4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = (_Float16)acc[cf][pf][g*4 + j];
          const unsigned int a = stg + (wpx*32 + n32)*ESTRIDE + (wco*64 + cf*32 + g*8 + khalf*4)*2;
          asm volatile("ds_write_b64 %0, %1" :: "v"(a), "v"(o) : "memory");
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#ifdef CN_DIAG
      es[esn++] = cn_stamp();
#endif
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int px = (it*CN_THREADS + tv) >> 4;
        raw[it] = cn_read16<0>(stg + px*ESTRIDE + c8*16);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]) :: "memory");
#ifdef CN_DIAG
      es[esn++] = cn_stamp();
#endif
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const f32x8 af = __builtin_convertvector(__builtin_bit_cast(h8, raw[it]), f32x8);
        const f32x8 rf = __builtin_convertvector(__builtin_bit_cast(h8, rr[it]), f32x8);
        f32x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (af[j] + bv[j] + rf[j])*p.out_scale;
        const h8 o = __builtin_convertvector(v, h8);
        if (p.stats && ok[it]) {
          const f32x8 w = __builtin_convertvector(o, f32x8);
#pragma unroll
          for (int j = 0; j < 8; ++j) { st_s[j] += w[j]; st_q[j] = fmaf(w[j], w[j], st_q[j]); }
        }
        if (ok[it]) {
          if (co + 8 <= p.Cout) *reinterpret_cast<h8*>(yb + pix[it]*p.Cys + co) = o;
          else {
#pragma unroll
            for (int j = 0; j < 8; ++j) if (co + j < p.Cout) yb[pix[it]*p.Cys + co + j] = o[j];
          }
        }
      }
#ifdef CN_DIAG
      es[esn++] = cn_stamp();
#endif
    }
#ifdef CN_DIAG
    if (blockIdx.x == 0 && lane == 0)
      for (int a = 0; a < 13; ++a) p.dbg[300 + wave*13 + a] = es[a];
#endif
    if (p.stats) {
      // threads tv, tv + 16, ... hold the same 8 channels: [32 threads][16 octets][16 values] in
      // the staging buffer, then one fp64 atomic per (channel, moment) and tile
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const unsigned int sa = stg + ((tv >> 4)*16 + c8)*64;
      u32x4 w0, w1, w2, w3;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        w0[j] = __float_as_uint(st_s[j]); w1[j] = __float_as_uint(st_s[4 + j]);
        w2[j] = __float_as_uint(st_q[j]); w3[j] = __float_as_uint(st_q[4 + j]);
      }
      cn_write16(sa, w0); cn_write16(sa + 16, w1); cn_write16(sa + 32, w2); cn_write16(sa + 48, w3);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (tv < 256) {
        const int oc = tv >> 4, k = tv & 15;            // octet, value (0-7 sums, 8-15 squares)
        double a = 0.0;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          unsigned int u[16];
#pragma unroll
          for (int l = 0; l < 16; ++l)
            asm volatile("ds_read_b32 %0, %1" : "=v"(u[l]) : "v"(stg + (((half*16 + l)*16 + oc)*16 + k)*4) : "memory");
          asm volatile("s_waitcnt lgkmcnt(0)"
                       : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]),
                         "+v"(u[8]), "+v"(u[9]), "+v"(u[10]), "+v"(u[11]), "+v"(u[12]), "+v"(u[13]), "+v"(u[14]), "+v"(u[15])
                       :: "memory");
#pragma unroll
          for (int l = 0; l < 16; ++l) a += (double)__uint_as_float(u[l]);
        }
        const int ch = q.cob*128 + oc*8 + (k & 7);
        if (ch < p.Cout) atomicAdd(&p.stats[(((long long)q.b*p.Cout + ch) << 1) + (k >> 3)], a);
      }
    }
  };

  // ---- FOLD == 2: GroupNorm(x + add) [ADM-modulated] as scale / shift per channel, the arithmetic of
  // chan_fold_kernel (nhwc.hip), for the ONE item this workgroup's tiles belong to
  if (FOLD == 2) {
    float* ftab = reinterpret_cast<float*>(smem + OFF_FOLD);
    const int b0 = decode(tile_lo).b;
    const int cpg = p.Cin/p.groups;
    const double hw = (double)p.H*(double)p.W;
    for (int c = tid; c < p.Cin; c += CN_THREADS) {
      const int g0 = (c/cpg)*cpg;
      double s1 = 0.0, s2 = 0.0;
      for (int k = 0; k < cpg; ++k) {
        const int ch = g0 + k;
        const double* sp = ch < p.C1 ? p.sums1 + (((long long)b0*p.C1 + ch) << 1)
                                     : p.sums2 + (((long long)b0*p.C2 + ch - p.C1) << 1);
        const double e = p.gn_add ? (double)p.gn_add[(long long)b0*p.Cin + ch] : 0.0;
        const double cs = sp[0], cq = sp[1];
        s1 += cs + hw*e;
        s2 += cq + 2.0*e*cs + hw*e*e;
      }
      const double n = (double)cpg*hw, mean = s1/n;
      double var = s2/n - mean*mean;
      if (var < 0) var = 0;
      const float rstd = (float)(1.0/sqrt(var + (double)p.eps));
      const long long idx = (long long)b0*p.Cin + c;
      float sc = rstd*p.gn_gamma[c];
      float sh = p.gn_beta[c] + ((p.gn_add ? p.gn_add[idx] : 0.f) - (float)mean)*sc;
      if (p.adm_scale) { const float m = 1.f + p.adm_scale[idx]; sc *= m; sh = sh*m + p.adm_shift[idx]; }
      ftab[c] = sc; ftab[CN_MAXC + c] = sh;
    }
    __syncthreads();
  }

  // ---- prologue: chunk 0 of the first tile + the first CN_DA taps of weights
  Tile cur = decode(tile_lo);
  Tile nxt = cur;
  set_prefetch_tile(cur);
  issue_table(0, 0);
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) issue_patch(r, 0, 0);
#pragma unroll
  for (int d = 0; d < DA; ++d) issue_weights(d);
  cn_wait_vm<0>();
  if (FOLD && !(CN_ABL & 8)) {
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      XfRaw v;
      transform_reads(r, 0, 0, v);
      cn_wait5<0>(v.raw, v.s0, v.s1, v.t0, v.t1);
      transform_write(r, 0, v);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // prefetch position: the chunk after the one being computed
  int nx_tile = tile_lo, nx_chunk = 0;
  auto advance_prefetch = [&]() {
    if (++nx_chunk == n_chunks) {
      nx_chunk = 0;
      if (nx_tile + 1 < tile_hi) { ++nx_tile; next_tile(nxt); }    // past the end: harmless reloads
      set_prefetch_tile(nxt);
    }
  };

  u32x4 fa[2][2], fb[2][4];                     // fragments of the k-step in flight / in use
  read_frags(cn_int<0>{}, cn_int<0>{}, 0, 0, fa[0], fb[0]);
  cn_wait_frags<0>(fa[0], fb[0]);
  zero_acc();
  int chunk = 0, ring = 0;                      // ring = slot of the tap being computed

#ifdef CN_DIAG
  unsigned long long stamps[9][4];
  bool diag_on = false;
  const unsigned long long k_t0 = cn_stamp();
  const unsigned long long k_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  auto tap_body = [&](auto tpc, int par) {
    constexpr int TP = decltype(tpc)::value;
    CN_STAMP(0);
    // (1) own DMAs of the next tap's weights (and, at tap 8, of the whole next patch) landed;
    //     own fragment reads returned; then everybody's
    cn_wait_vm<S::top(TP)>();
    CN_STAMP(1);
    cn_wait_frags<0>(fa[0], fb[0]);
    __builtin_amdgcn_s_barrier();
    CN_STAMP(2);
    // (2) The two waves of a SIMD (wave w and w + 4: wco 0 / 1) run the tap in opposite orders, so
    //     that one feeds the matrix pipe while the other issues its DMAs / rewrites its patch piece:
    //       wco 0:  reads(k-step 1)  16 MFMAs (reads of the next tap between the k-steps)  DMAs + transform
    //       wco 1:  DMAs + transform  reads(k-step 1)  16 MFMAs
    //     Scheduling barriers pin the MFMAs where they are written.
    constexpr bool XF = FOLD && !(CN_ABL & 8) && TP >= 3 && TP - 3 < ROUNDS;
    auto dma_block = [&]() {
      if (TP == 0) { advance_prefetch(); issue_table(nx_chunk, par ^ 1); }
      issue_weights(ring + DA >= NA ? ring + DA - NA : ring + DA);
      if (TP < ROUNDS) issue_patch(TP, nx_chunk, par ^ 1);
      if constexpr (XF) {                      // piece TP-3 of the next patch: own DMA landed
        XfRaw xv;
        cn_wait_vm<S::piece(TP)>();
        transform_reads(TP - 3, par ^ 1, nx_chunk, xv);
        cn_wait5<0>(xv.raw, xv.s0, xv.s1, xv.t0, xv.t1);
        transform_write(TP - 3, par ^ 1, xv);
      }
    };
    const int nring = ring + 1 == NA ? 0 : ring + 1;
    // CN_VAR & 4: the 12 fragment reads of the tap go out ONE per MFMA gap instead of in two bursts
    // of six (a burst stalls the issuing wave on the LDS queue while its MFMA slots pass)
    auto one_mfma = [&](int cf, int pf, u32x4& a, u32x4& b) {
      const h8 av = __builtin_bit_cast(h8, a), bv = __builtin_bit_cast(h8, b);
      if (CN_ABL & 4) { asm volatile("" :: "v"(av), "v"(bv)); return; }
      acc[cf][pf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc[cf][pf], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    };
    auto mfma_block_fine = [&]() {
      constexpr int kh = TP / KS, kw = TP % KS;
      constexpr int NT = (TP + 1) % TAPS, nkh = NT / KS, nkw = NT % KS;
      const int npar = TP + 1 < TAPS ? par : par ^ 1;
      const unsigned int ab1 = a_rd + ring*CN_ASLOT, bb1 = (b_col[kw] ^ 32) + par*PBYTES;
      const unsigned int ab0 = a_rd + nring*CN_ASLOT, bb0 = b_col[nkw] + npar*PBYTES;
      // k-step 0 on fa[0] / fb[0]; k-step 1's fragments requested in the order their MFMAs need them
      one_mfma(0, 0, fa[0][0], fb[0][0]); fa[1][0] = cn_read16<4096>(ab1);
      one_mfma(1, 0, fa[0][1], fb[0][0]); fb[1][0] = cn_read16<(0 + kh)*PC*64>(bb1);
      one_mfma(0, 1, fa[0][0], fb[0][1]); fa[1][1] = cn_read16<4096 + 1024>(ab1);
      one_mfma(1, 1, fa[0][1], fb[0][1]); fb[1][1] = cn_read16<(1 + kh)*PC*64>(bb1);
      one_mfma(0, 2, fa[0][0], fb[0][2]); fb[1][2] = cn_read16<(2 + kh)*PC*64>(bb1);
      one_mfma(1, 2, fa[0][1], fb[0][2]); fb[1][3] = cn_read16<(3 + kh)*PC*64>(bb1);
      one_mfma(0, 3, fa[0][0], fb[0][3]);
      one_mfma(1, 3, fa[0][1], fb[0][3]);
      // k-step 1; the next tap's k-step 0 fragments (tap 8: first tap of the next chunk)
      cn_wait_tie2<4>(fa[1][0], fb[1][0]);
      one_mfma(0, 0, fa[1][0], fb[1][0]); fa[0][0] = cn_read16<0>(ab0);
      cn_wait_tie1<4>(fa[1][1]);
      one_mfma(1, 0, fa[1][1], fb[1][0]); fb[0][0] = cn_read16<(0 + nkh)*PC*64>(bb0);
      cn_wait_tie1<4>(fb[1][1]);
      one_mfma(0, 1, fa[1][0], fb[1][1]); fa[0][1] = cn_read16<1024>(ab0);
      one_mfma(1, 1, fa[1][1], fb[1][1]); fb[0][1] = cn_read16<(1 + nkh)*PC*64>(bb0);
      cn_wait_tie1<5>(fb[1][2]);
      one_mfma(0, 2, fa[1][0], fb[1][2]); fb[0][2] = cn_read16<(2 + nkh)*PC*64>(bb0);
      one_mfma(1, 2, fa[1][1], fb[1][2]); fb[0][3] = cn_read16<(3 + nkh)*PC*64>(bb0);
      cn_wait_tie1<6>(fb[1][3]);
      one_mfma(0, 3, fa[1][0], fb[1][3]);
      one_mfma(1, 3, fa[1][1], fb[1][3]);
    };
    auto mfma_block = [&](bool dma_mid) {
      read_frags(tpc, cn_int<1>{}, par, ring, fa[1], fb[1]);
      __builtin_amdgcn_sched_barrier(0);
      if (CN_VAR & 2) __builtin_amdgcn_s_setprio(1);
      mfma_step(fa[0], fb[0], 0, 4);
      if (CN_VAR & 2) __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      if (dma_mid) { dma_block(); __builtin_amdgcn_sched_barrier(0); }
      // the next tap's k-step 0 reads (tap 8: first tap of the next chunk)
      if constexpr (TP + 1 < TAPS) read_frags(cn_int<(TP + 1) % TAPS>{}, cn_int<0>{}, par, nring, fa[0], fb[0]);
      else read_frags(cn_int<0>{}, cn_int<0>{}, par ^ 1, nring, fa[0], fb[0]);
      cn_wait_frags<2 + PF>(fa[1], fb[1]);
      if (CN_VAR & 2) __builtin_amdgcn_s_setprio(1);
      mfma_step(fa[1], fb[1], 0, 4);
      if (CN_VAR & 2) __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
    };
    // (an opaque scalar copy: a loop-invariant condition would make hipcc clone the whole loop)
    int wsel;
    asm volatile("s_mov_b32 %0, %1" : "=s"(wsel) : "s"(wco));
    if (wsel != 0) dma_block();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr ((CN_VAR & 4) != 0 && PF == 4) {
      mfma_block_fine();
      CN_STAMP(3);
      // (fragments in flight are made real before the long DMA / transform block: a register copy
      // or spill hipcc might place there would otherwise copy bits that have not arrived)
      if (wsel == 0) { cn_wait_frags<0>(fa[0], fb[0]); dma_block(); }
    } else if constexpr ((CN_VAR & 1) != 0) {
      mfma_block(wsel == 0);
      CN_STAMP(3);
    } else {
      mfma_block(false);
      CN_STAMP(3);
      if (wsel == 0) { cn_wait_frags<0>(fa[0], fb[0]); dma_block(); }
    }
    // the fragments read for the next chunk are live across the loop back edge (and the epilogue):
    // make them real first -- a register copy hipcc inserts before the wait would copy stale bits
    if constexpr (TP + 1 == TAPS) cn_wait_frags<0>(fa[0], fb[0]);
    ring = nring;
  };

#pragma unroll 1
  for (int uc = 0; uc < n_work; ++uc) {
    const int par = uc & 1;
#ifdef CN_DIAG
    diag_on = blockIdx.x == 0 && uc == 6;
#endif
    tap_body(cn_int<0>{}, par); tap_body(cn_int<1>{}, par); tap_body(cn_int<2>{}, par);
    tap_body(cn_int<3>{}, par); tap_body(cn_int<4>{}, par); tap_body(cn_int<5>{}, par);
    tap_body(cn_int<6>{}, par); tap_body(cn_int<7>{}, par); tap_body(cn_int<8>{}, par);
#ifdef CN_DIAG
    if (diag_on && lane == 0)
      for (int a = 0; a < 9; ++a) for (int c = 0; c < 4; ++c) p.dbg[(wave*9 + a)*4 + c] = stamps[a][c];
#endif
    if (++chunk == n_chunks) {
      chunk = 0;
      store_tile(cur, par);
      zero_acc();
      next_tile(cur);
    }
  }
  cn_wait_vm<0>();
#ifdef CN_DIAG
  if (blockIdx.x == 0 && tid == 0) {
    p.dbg[8*9*4] = cn_stamp() - k_t0;
    p.dbg[8*9*4 + 1] = __builtin_amdgcn_s_memrealtime() - k_r0;
  }
#endif
}

#include "conv_nhwc_splitk.cuh"

// wp[co block of 128][chunk][tap][k-step 2][co fragment 4][lane 64][8] <- w[co][ci][tap]
__global__ __launch_bounds__(256) void conv_nhwc_pack_kernel(const float* w, _Float16* wp, int Cout,
                                                             int Cin, int taps, int n_chunks,
                                                             long long total) {
  for (long long idx = (long long)blockIdx.x*256 + threadIdx.x; idx < total;
       idx += (long long)gridDim.x*256) {
    long long r = idx;
    const int j = (int)(r % 8); r /= 8;
    const int lane = (int)(r % 64); r /= 64;
    const int cf = (int)(r % 4); r /= 4;
    const int ks = (int)(r % 2); r /= 2;
    const int tap = (int)(r % taps); r /= taps;
    const int chunk = (int)(r % n_chunks); r /= n_chunks;
    const int co = (int)r*128 + cf*32 + (lane & 31);
    const int ci = chunk*CN_CK + ks*16 + (lane >> 5)*8 + j;
    float v = 0.f;
    if (co < Cout && ci < Cin) v = w[((long long)co*Cin + ci)*taps + tap];
    wp[idx] = (_Float16)v;
  }
}

__device__ __attribute__((aligned(256))) unsigned char cn_zero_block[256];

}  // namespace

#ifdef CN_DIAG
unsigned long long* brv_conv_nhwc_dbg = nullptr;
#endif

extern "C" {

int64_t brv_conv_nhwc_packed_size(int64_t Cout, int64_t Cin, int64_t ksize) {
  if (Cout < 1 || Cin < 1 || ksize != 3) return -1;
  return ((Cout + 127)/128)*128*((Cin + CN_CK - 1)/CN_CK)*CN_CK*ksize*ksize;
}

int brv_conv_nhwc_pack(const float* w, void* wp, int64_t Cout, int64_t Cin, int64_t ksize,
                       brv_stream_t stream) {
  const int64_t total = brv_conv_nhwc_packed_size(Cout, Cin, ksize);
  if (total < 0) return -1;
  long long g = (total + 255)/256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(conv_nhwc_pack_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w,
                     (_Float16*)wp, (int)Cout, (int)Cin, (int)(ksize*ksize),
                     (int)((Cin + CN_CK - 1)/CN_CK), (long long)total);
  CN_OK(hipGetLastError());
  return 0;
}

// shared launcher: `gn` selects the self-folding kernel when every workgroup's tiles lie in one item
struct ConvNhwcNorm {
  const double* sums1; const double* sums2; const float* add; const float* gamma; const float* beta;
  const float* adm_scale; const float* adm_shift; int groups; float eps; float* fold_ws;
};
}  // extern "C"
extern "C" int brv_groupnorm_fold_chan2(const double* sums1, int64_t C1, const double* sums2, int64_t C2,
                                        const float* add_bc, const float* gamma, const float* beta,
                                        const float* adm_scale, const float* adm_shift, float* scale_bc,
                                        float* shift_bc, int64_t B, int64_t HW, int64_t groups,
                                        float eps, brv_stream_t stream);
namespace {
int conv_nhwc_launch(const void* x1, int64_t C1, int64_t C1s, const void* x2, int64_t C2, int64_t C2s,
                     const void* wp, const float* bias, const void* res, int64_t Crs,
                     const float* in_scale, const float* in_shift, const ConvNhwcNorm* gn, int in_silu,
                     void* y, int64_t Cys, int64_t B, int64_t H, int64_t W, int64_t Cout, int64_t ksize,
                     float out_scale, double* stats, brv_stream_t stream,
                     float* split_ws = nullptr, long long split_floats = 0) {
  if (B < 1 || H < 1 || W < 1 || C1 < 1 || Cout < 1 || ksize != 3) return -1;
  if ((C1s & 7) || C1 > C1s || (Cys & 7) || (res && (Crs & 7)) || (Cout & 3)) return -2;
  if (x2 && ((C2s & 7) || C2 > C2s || C2 < 1 || (C1 % CN_CK) != 0)) return -2;
  const int64_t Cin = C1 + (x2 ? C2 : 0);
  if ((in_scale || gn) && (Cin % CN_CK) != 0) return -3;
  if (gn && (gn->groups < 1 || Cin % gn->groups || !gn->fold_ws)) return -3;
  static const unsigned char* zeros = nullptr;
  if (!zeros) {
    void* z = nullptr;
    CN_OK(hipGetSymbolAddress(&z, HIP_SYMBOL(cn_zero_block)));
    zeros = (const unsigned char*)z;
  }
  ConvNhwcParams p;
  memset(&p, 0, sizeof(p));
  p.x1 = (const _Float16*)x1; p.x2 = (const _Float16*)x2; p.wp = (const unsigned char*)wp;
  p.bias = bias; p.res = (const _Float16*)res; p.y = (_Float16*)y;
  p.in_scale = in_scale; p.in_shift = in_shift; p.zeros = zeros; p.stats = stats;
  p.B = (int)B; p.H = (int)H; p.W = (int)W; p.C1s = (int)C1s; p.C2s = (int)C2s;
  p.n_chunks1 = (int)((C1 + CN_CK - 1)/CN_CK);
  p.n_chunks = p.n_chunks1 + (x2 ? (int)((C2 + CN_CK - 1)/CN_CK) : 0);
  p.Cin = (int)Cin; p.Cout = (int)Cout; p.Cys = (int)Cys; p.Crs = (int)Crs;
  p.C1 = (int)C1; p.C2 = x2 ? (int)C2 : 0;
  p.n_wt = (int)((W + CN_COLS - 1)/CN_COLS);
  p.n_cob = (int)((Cout + 127)/128);
  // Low-resolution launches (few tiles): the reduction split over workgroups, partial sums through the caller's
  // scratch (conv_nhwc_splitk.cuh). Without scratch -- the entry points of rounds 2 - 5 -- the pixel-parallel kernel.
  if (split_ws) {
    int cpw = 0, n_split = 1;
    const long long need = conv_split_plan(B, H, W, p.n_chunks, p.n_cob, cpw, n_split);
    if (need > 0 && need <= split_floats && (!gn || (Cin % gn->groups) == 0)) {
      ConvSplitParams sp;
      memset(&sp, 0, sizeof(sp));
      sp.x1 = p.x1; sp.x2 = p.x2; sp.wp = p.wp; sp.in_scale = in_scale; sp.in_shift = in_shift;
      if (gn) {
        sp.sums1 = gn->sums1; sp.sums2 = gn->sums2; sp.gn_add = gn->add; sp.gn_gamma = gn->gamma; sp.gn_beta = gn->beta;
        sp.adm_scale = gn->adm_scale; sp.adm_shift = gn->adm_shift; sp.groups = gn->groups; sp.eps = gn->eps;
      }
      sp.C1 = p.C1; sp.C2 = p.C2; sp.part = split_ws;
      sp.B = p.B; sp.H = p.H; sp.W = p.W; sp.C1s = p.C1s; sp.C2s = p.C2s; sp.n_chunks1 = p.n_chunks1;
      sp.n_chunks = p.n_chunks; sp.Cin = p.Cin; sp.n_wt = p.n_wt; sp.n_ht = (int)((H + CS_ROWS - 1)/CS_ROWS);
      sp.n_cob = p.n_cob; sp.cpw = cpw; sp.in_silu = in_silu;
      const dim3 sgrid((unsigned)(B*sp.n_ht*sp.n_wt), (unsigned)p.n_cob, (unsigned)n_split);
      const hipStream_t sst = (hipStream_t)stream;
      if (gn) hipLaunchKernelGGL((conv_nhwc_splitk_kernel<2>), sgrid, dim3(CN_THREADS), 0, sst, sp);
      else if (in_scale) hipLaunchKernelGGL((conv_nhwc_splitk_kernel<1>), sgrid, dim3(CN_THREADS), 0, sst, sp);
      else hipLaunchKernelGGL((conv_nhwc_splitk_kernel<0>), sgrid, dim3(CN_THREADS), 0, sst, sp);
      CN_OK(hipGetLastError());
      ConvCombineParams cp;
      memset(&cp, 0, sizeof(cp));
      cp.part = split_ws; cp.n_split = n_split; cp.cq_n = p.n_cob*32; cp.npix = (long long)B*H*W; cp.hw = (long long)H*W;
      cp.bias = bias; cp.res = (const _Float16*)res; cp.y = (_Float16*)y; cp.stats = stats;
      cp.Cout = (int)Cout; cp.Crs = (int)Crs; cp.Cys = (int)Cys; cp.out_scale = out_scale;
      const dim3 cgrid((unsigned)((H*W + 255)/256), (unsigned)(B*cp.cq_n));
      hipLaunchKernelGGL(conv_nhwc_combine_kernel, cgrid, dim3(256), 0, sst, cp);
      CN_OK(hipGetLastError());
      return 0;
    }
  }
  // rows per tile: 16 when that fills the chip, else 8 or 4 (2-4x the workgroups)
#ifndef BRV_CONV_PF
#define BRV_CONV_PF 0          // diagnostic builds: force the rows per tile (1, 2 or 4 x 4 rows)
#endif
  constexpr int force_pf = BRV_CONV_PF;
  // (BRV_CONV_MIN_TILES, diagnostic builds: the tile count below which the rows per tile are halved -- every workgroup
  // streams the whole weight set of its channel block from L2 per TILE, so more, smaller tiles buy occupancy with
  // weight traffic; swept 32 .. 1 024 at batch 1 and 8: 192 is the optimum, profiles/r05_sgmse_tiles.txt)
#ifndef BRV_CONV_MIN_TILES
#define BRV_CONV_MIN_TILES 192
#endif
  constexpr long long min_tiles = BRV_CONV_MIN_TILES;
  int pf = 4;
  while (pf > 1 && (long long)B*((H + 4*pf - 1)/(4*pf))*p.n_wt*p.n_cob < min_tiles) pf >>= 1;
  if (force_pf == 1 || force_pf == 2 || force_pf == 4) pf = force_pf;
  p.n_ht = (int)((H + 4*pf - 1)/(4*pf));
  const long long n_tiles = (long long)B*p.n_ht*p.n_wt*p.n_cob;
  if (n_tiles > 0x7fffffffLL) return -2;
  p.n_tiles = (int)n_tiles;
  p.out_scale = out_scale; p.in_silu = in_silu;
#ifndef BRV_CONV_STAGGER
#define BRV_CONV_STAGGER 0     // diagnostic builds: start skew of the workgroups (measured: no effect)
#endif
  p.stagger = BRV_CONV_STAGGER;
#ifdef CN_DIAG
  static unsigned long long* dbg = nullptr;
  if (!dbg) CN_OK(hipMalloc(&dbg, 4096));
  p.dbg = dbg;
  brv_conv_nhwc_dbg = dbg;
#endif
  const long long G = n_tiles < 256 ? n_tiles : 256;
  int fold = in_scale ? 1 : 0;
  if (gn) {
    // self-fold: every workgroup's run of tiles inside one item (the launch's own rank -> range map)
#ifndef BRV_CONV_NO_SELF_FOLD
#define BRV_CONV_NO_SELF_FOLD 0   // diagnostic builds: GroupNorm fold always as a launch of its own
#endif
    constexpr int no_self = BRV_CONV_NO_SELF_FOLD;
    const long long tpi = (long long)p.n_ht*p.n_wt*p.n_cob;
    bool single = !no_self && Cin <= CN_MAXC;
    for (long long r = 0; single && r < G; ++r) {
      const long long lo = r*n_tiles/G, hi = (r + 1)*n_tiles/G;
      if (hi > lo && lo/tpi != (hi - 1)/tpi) single = false;
    }
    if (single) {
      fold = 2;
      p.sums1 = gn->sums1; p.sums2 = gn->sums2; p.gn_add = gn->add; p.gn_gamma = gn->gamma;
      p.gn_beta = gn->beta; p.adm_scale = gn->adm_scale; p.adm_shift = gn->adm_shift;
      p.groups = gn->groups; p.eps = gn->eps;
    } else {
      const int rc = brv_groupnorm_fold_chan2(gn->sums1, C1, gn->sums2, x2 ? C2 : 0, gn->add, gn->gamma,
                                              gn->beta, gn->adm_scale, gn->adm_shift, gn->fold_ws,
                                              gn->fold_ws + B*Cin, B, H*W, gn->groups, gn->eps, stream);
      if (rc) return rc;
      fold = 1;
      p.in_scale = gn->fold_ws; p.in_shift = gn->fold_ws + B*Cin;
    }
  }
  const dim3 grid((unsigned)G), block(CN_THREADS);
  const hipStream_t st = (hipStream_t)stream;
#define CN_LAUNCH(F, P) hipLaunchKernelGGL((conv_nhwc_kernel<3, F, P>), grid, block, 0, st, p)
#define CN_LAUNCH_PF(F) do { if (pf == 4) CN_LAUNCH(F, 4); else if (pf == 2) CN_LAUNCH(F, 2); else CN_LAUNCH(F, 1); } while (0)
  if (fold == 2) CN_LAUNCH_PF(2); else if (fold == 1) CN_LAUNCH_PF(1); else CN_LAUNCH_PF(0);
#undef CN_LAUNCH_PF
#undef CN_LAUNCH
  CN_OK(hipGetLastError());
  return 0;
}
}  // namespace
extern "C" {

int brv_conv_nhwc_forward(const void* x1, int64_t C1, int64_t C1s, const void* x2, int64_t C2,
                          int64_t C2s, const void* wp, const float* bias, const void* res,
                          int64_t Crs, const float* in_scale, const float* in_shift, int in_silu,
                          void* y, int64_t Cys, int64_t B, int64_t H, int64_t W, int64_t Cout,
                          int64_t ksize, float out_scale, double* stats, brv_stream_t stream) {
  return conv_nhwc_launch(x1, C1, C1s, x2, C2, C2s, wp, bias, res, Crs, in_scale, in_shift, nullptr,
                          in_silu, y, Cys, B, H, W, Cout, ksize, out_scale, stats, stream);
}

// The same two entry points with a caller-provided scratch for the launches whose reduction is split over workgroups
// (conv_nhwc_splitk.cuh): brv_conv_nhwc_split_ws_bytes says how much a launch of these sizes would use (0: none).
int64_t brv_conv_nhwc_split_ws_bytes(int64_t B, int64_t H, int64_t W, int64_t C1, int64_t C2, int64_t Cout) {
  if (B < 1 || H < 1 || W < 1 || C1 < 1 || Cout < 1) return 0;
  int cpw = 0, n_split = 1;
  const long long n_chunks = (C1 + CN_CK - 1)/CN_CK + (C2 > 0 ? (C2 + CN_CK - 1)/CN_CK : 0);
  return 4*conv_split_plan(B, H, W, n_chunks, (Cout + 127)/128, cpw, n_split);
}

int brv_conv_nhwc_forward_ws(const void* x1, int64_t C1, int64_t C1s, const void* x2, int64_t C2,
                             int64_t C2s, const void* wp, const float* bias, const void* res,
                             int64_t Crs, const float* in_scale, const float* in_shift, int in_silu,
                             void* y, int64_t Cys, int64_t B, int64_t H, int64_t W, int64_t Cout,
                             int64_t ksize, float out_scale, double* stats, void* split_ws,
                             int64_t split_ws_bytes, brv_stream_t stream) {
  return conv_nhwc_launch(x1, C1, C1s, x2, C2, C2s, wp, bias, res, Crs, in_scale, in_shift, nullptr,
                          in_silu, y, Cys, B, H, W, Cout, ksize, out_scale, stats, stream,
                          (float*)split_ws, split_ws_bytes/4);
}

int brv_conv_nhwc_forward_gn_ws(const void* x1, int64_t C1, int64_t C1s, const void* x2, int64_t C2,
                                int64_t C2s, const void* wp, const float* bias, const void* res,
                                int64_t Crs, const double* sums1, const double* sums2,
                                const float* add_bc, const float* gamma, const float* beta,
                                const float* adm_scale, const float* adm_shift, int64_t groups, float eps,
                                float* fold_ws, int in_silu, void* y, int64_t Cys, int64_t B, int64_t H,
                                int64_t W, int64_t Cout, int64_t ksize, float out_scale, double* stats,
                                void* split_ws, int64_t split_ws_bytes, brv_stream_t stream) {
  if (!sums1 || !gamma || !beta || (x2 && !sums2)) return -1;
  ConvNhwcNorm gn = {sums1, sums2, add_bc, gamma, beta, adm_scale, adm_shift, (int)groups, eps, fold_ws};
  return conv_nhwc_launch(x1, C1, C1s, x2, C2, C2s, wp, bias, res, Crs, nullptr, nullptr, &gn, in_silu,
                          y, Cys, B, H, W, Cout, ksize, out_scale, stats, stream,
                          (float*)split_ws, split_ws_bytes/4);
}

int brv_conv_nhwc_forward_gn(const void* x1, int64_t C1, int64_t C1s, const void* x2, int64_t C2,
                             int64_t C2s, const void* wp, const float* bias, const void* res,
                             int64_t Crs, const double* sums1, const double* sums2,
                             const float* add_bc, const float* gamma, const float* beta,
                             const float* adm_scale, const float* adm_shift, int64_t groups, float eps,
                             float* fold_ws, int in_silu, void* y, int64_t Cys, int64_t B, int64_t H,
                             int64_t W, int64_t Cout, int64_t ksize, float out_scale, double* stats,
                             brv_stream_t stream) {
  if (!sums1 || !gamma || !beta || (x2 && !sums2)) return -1;
  ConvNhwcNorm gn = {sums1, sums2, add_bc, gamma, beta, adm_scale, adm_shift, (int)groups, eps, fold_ws};
  return conv_nhwc_launch(x1, C1, C1s, x2, C2, C2s, wp, bias, res, Crs, nullptr, nullptr, &gn, in_silu,
                          y, Cys, B, H, W, Cout, ksize, out_scale, stats, stream);
}

}  // extern "C"
