// Large products of fp32 matrices at fp32 accuracy: the 1x1 convolutions, data gradients and weight
// gradients of the fp32 Conv-TasNet path (reference: brever/models/convtasnet/convtasnet.py:225-262
// run without autocast) and the brv_gemm_f32 calls with 16-byte aligned operands that fill the chip
// without a reduction split. One kernel template, two arithmetic forms:
//
// fp32 MFMA (v_mfma_f32_32x32x2_f32, 256 FLOP / clk / CU). One workgroup = 8 wavefronts = a 256 x 128
// (or 128 x 256) tile of D, 64 x 64 per wavefront (2 x 2 MFMA accumulators); the reduction runs in
// tiles of 32 through two LDS stages: 16-byte global loads into registers while the MFMAs of the
// previous tile run, one barrier per tile. An operand whose storage is contiguous in k sits in LDS as
// [row][32 + 4] and is read as 16-byte fragments (a lane takes 4 consecutive k: MFMA step i of a group
// of 8 multiplies k = 8j + 4h + i of both operands, h = lane / 32 -- any pairing of k works as long
// as both operands use the same one); an operand contiguous along its rows sits as [k][rows + 8] and
// is read by dword. Both layouts are bank-conflict free for these reads.
//
// Split bf16 (X3; v_mfma_f32_32x32x16_bf16, 4096 FLOP / clk / CU): every fp32 operand is split on its way
// into LDS into three bf16 pieces and a k-step is six MFMAs -- fp32 accuracy at 2.7x the matrix rate.
// 128 x 128 tile per 4-wavefront workgroup, two workgroups per CU, one LDS stage of six bf16 planes:
// [row][32 + 8] for operands contiguous in k (forward, data gradients), [k][rows + 8] read through
// ds_read_b64_tr_b16 for operands contiguous along their rows (weight gradients).
//
// Common: workgroups are persistent and walk a flat (tile, k-tile) sequence, the first loads of the
// next tile are issued before the epilogue of the present one; tiles are dealt XCD-aware; an operand
// can be transformed while it is staged (norm + PReLU of a layer whose normalised tensor is never
// stored), a second B can serve the columns past n_split, the rows past m_split can go elsewhere;
// the epilogue addresses through buffer descriptors (one add per element). Long reductions over few
// output tiles (weight gradients) are split over workgroups: partial tiles go to a scratch buffer
// and are added in split order by a second kernel (no atomics: the result does not depend on the
// order of arrival). Measurements and what was tried: DESIGN.md section 5a.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.cuh"
#include "gemm_f32_big.h"

// Compile-time ablations of tools/gemm_f32_bench.py (bit mask): 1 no MFMA, 2 no global loads,
// 4 no epilogue stores, 8 no LDS operand reads, 16 (split-bf16 kernel) no operand split
#ifndef BRV_BIG_ABL
#define BRV_BIG_ABL 0
#endif

namespace brv {
namespace {

struct BigDev {
  BigGemm g;
  int m_tiles, n_tiles, mn_padded, xcd_perm, ksplit, ktiles, n_work, b_scalar, fast_epi, split_xcd;
  long long total_t, per_t;
};

constexpr int kBK = 32;          // reduction tile of the fp32-MFMA form
#ifndef BRV_X3_BK
// Reduction tile of the split-bf16 form: 32 = two workgroups per CU (61 KB of LDS, ~240 registers each).
// 16 (three workgroups per CU at <= 168 registers) was measured: the 128 -> 512 product 84 -> 75 us, the
// 512 -> 128 one 78 -> 85, the weight-gradient form 83 -> 115 (its transposing reads spill at 168
// registers), and the whole fp32 step 16x slower because the spilling instantiations need scratch.
#define BRV_X3_BK 32
#endif
constexpr int kLDK = kBK + 4;    // floats per LDS row of a k-contiguous operand

__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// y = (prelu(z) - mean) rstd gain + bias on 4 consecutive channels
__device__ __forceinline__ float4 norm_pro(float4 v, float mean, float rstd, float a, bool act,
                                           const float4& g, const float4& b) {
  float4 o;
  o.x = (((v.x > 0.f || !act) ? v.x : a*v.x) - mean)*rstd*g.x + b.x;
  o.y = (((v.y > 0.f || !act) ? v.y : a*v.y) - mean)*rstd*g.y + b.y;
  o.z = (((v.z > 0.f || !act) ? v.z : a*v.z) - mean)*rstd*g.z + b.z;
  o.w = (((v.w > 0.f || !act) ? v.w : a*v.w) - mean)*rstd*g.w + b.w;
  return o;
}
__device__ __forceinline__ float4 ld4s(const float* p) { return make_float4(p[0], p[1], p[2], p[3]); }

// Barrier of the product kernels: LDS traffic of this wavefront complete, then s_barrier. The barriers
// here only order LDS stages; __syncthreads() also waits for every global access in flight (vmcnt(0)
// counts loads AND stores on gfx9), i.e. for the epilogue's stores at every tile end.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// MFMA fragment from a k-major LDS plane [k][rows] (bf16, row stride ld): lane (r = lane & 31, h = lane >> 5)
// gets plane[k0 + 8h + j][c0 + r], j = 0 .. 7, through two transposing reads (ds_read_b64_tr_b16: 4 rows x
// 16 columns per 16-lane group) -- the operands of a weight gradient are contiguous along their rows, the
// reduction index is the strided one
typedef __attribute__((ext_vector_type(8))) short big_s16x8;
__device__ __forceinline__ bf16x8 tr_frag16(const bf16_t* plane, int ld, int k0, int c0, int lane) {
  const int g4 = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
  const bf16_t* p0 = plane + (k0 + 8*(g4 >> 1) + q)*ld + c0 + 16*(g4 & 1) + 4*pp;
  typedef __attribute__((address_space(3))) s16x4* lds_p;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p0 + 4*ld));
  const big_s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

struct Work { int valid, z, split, m0, n0, kb, k0, left, fresh; int q; };

// X3: the product runs on the bf16 matrix pipe with every fp32 operand split into three bf16 pieces
// (x = hi + mid + lo exactly: 3 x 8 significand bits) and six MFMAs per k-step (hi hi, hi mid,
// mid hi, hi lo, lo hi, mid mid: every bf16 x bf16 product is exact in the fp32 accumulator, the
// dropped terms are below 2^-24 of |x||y|) -- fp32 accuracy at 2.7x the rate of the fp32 MFMA.
// Both operands contiguous in k only (!TA, TB); 128 x 128 tile per 4-wavefront workgroup, two
// workgroups per compute unit (one splits / stages while the other multiplies), one LDS stage of
// six bf16 planes [row][32 + 8].
template <int WM, int WN, bool TA, bool TB, int PRO, bool X3 = false>
__global__ __launch_bounds__(64*WM*WN, (X3 && BRV_X3_BK == 16) ? 3 : 1) void gemm_f32_big_kernel(const BigDev p) {
  constexpr int TM = 64*WM, TN = 64*WN, NT = 64*WM*WN;
  constexpr int BK = X3 ? BRV_X3_BK : kBK;    // reduction tile
  constexpr int KQ = BK/4;                    // 16-byte pieces per row of a k-contiguous operand
  constexpr int LDA = TA ? TM + 8 : kLDK, LDB = TB ? kLDK : TN + 8;
  constexpr int A_FLOATS = TA ? BK*LDA : TM*LDA;
  constexpr int B_FLOATS = TB ? TN*LDB : BK*LDB;
  constexpr int STAGE = A_FLOATS + B_FLOATS;
  constexpr int NA = TM*KQ/NT, NB = TN*KQ/NT;   // 16-byte loads per thread and k-tile
  constexpr int LDH = BK + 8;                 // X3: bf16 elements per LDS row (80 or 48 bytes: conflict-free 16-byte reads)
  constexpr int LDT = TM + 8;                 // X3 weight-gradient form: bf16 elements per LDS row of a [k][rows] plane
  // X3 planes: an operand contiguous in k is staged row-major ([rows][LDH]), one contiguous along its rows k-major
  // ([BK][rows + 8], fragments through the transposing LDS read) -- each operand by its own storage order
  constexpr int X3_PA = TA ? BK*LDT : TM*LDH, X3_PB = TB ? TN*LDH : BK*(TN + 8);
  constexpr int LDS_FLOATS = X3 ? 3*(X3_PA + X3_PB)/2 : 2*STAGE;
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
  const BigGemm& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int r32 = lane & 31, kh = lane >> 5;

  // work item q -> (batch item, reduction split, tile); all integers are 32-bit (checked on the host)
  auto decode = [&](int q) {
    Work w; w.valid = 0; w.q = q; w.z = 0; w.split = 0; w.m0 = 0; w.n0 = 0; w.kb = 0; w.k0 = 0; w.left = 0; w.fresh = 1;
    for (; q < p.n_work; q += gridDim.x) {
      int zs = q / p.mn_padded, r = q % p.mn_padded;
      if (p.split_xcd) {
        // split reductions: the tiles of ONE split read the same rows of both operands, so they take
        // consecutive slots of one XCD (work items congruent mod 8 run on one XCD and share its L2; with
        // the tiles of a split dealt round-robin every XCD fetched every operand row: 541 MB instead of
        // 196 MB per [res | skip] weight gradient, PMC)
        const int xcd = q & 7, slot = q >> 3;
        zs = (slot / p.mn_padded)*8 + xcd;
        r = slot % p.mn_padded;
        if (zs >= p.ksplit) continue;
      }
      int mt, nt;
      if (p.xcd_perm) { const int xcd = r & 7, slot = r >> 3; mt = (slot / p.n_tiles)*8 + xcd; nt = slot % p.n_tiles; }
      else { mt = r / p.n_tiles; nt = r % p.n_tiles; }
      if (mt >= p.m_tiles) continue;
      w.split = zs % p.ksplit; w.z = zs / p.ksplit;
      const int t_lo = w.split*(int)p.per_t;
      const int t_hi = t_lo + (int)p.per_t < (int)p.total_t ? t_lo + (int)p.per_t : (int)p.total_t;
      if (t_lo >= t_hi) continue;
      w.left = t_hi - t_lo;
      w.kb = t_lo / p.ktiles; w.k0 = (t_lo % p.ktiles)*BK;
      w.m0 = mt*TM; w.n0 = nt*TN; w.q = q; w.valid = 1;
      return w;
    }
    w.q = q;
    return w;
  };
  // the k-tile after w (same tile), or the first one of this workgroup's next work item
  auto advance = [&](const Work& w) {
    if (w.left > 1) {
      Work n = w; n.left = w.left - 1; n.k0 = w.k0 + BK; n.fresh = 0;
      if (n.k0 >= g.K) { n.k0 = 0; n.kb = w.kb + 1; n.fresh = 1; }
      return n;
    }
    return decode(w.q + (int)gridDim.x);
  };

  // per-thread constants of the staging maps: element (row-ish, k) of load i inside the tile
  int a_r[NA], a_k[NA], b_r[NB], b_k[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int f = tid + NT*i;
    if (TA) { a_k[i] = f/(TM/4); a_r[i] = 4*(f % (TM/4)); } else { a_r[i] = f / KQ; a_k[i] = 4*(f % KQ); }
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int f = tid + NT*i;
    if (TB) { b_r[i] = f / KQ; b_k[i] = 4*(f % KQ); } else { b_k[i] = f/(TN/4); b_r[i] = 4*(f % (TN/4)); }
  }
  const float* pa[NA]; const float* pb[NB];
  bool oka[NA], okb[NB];
  float4 ra[NA], rb[NB];
  // operand transform: what does not change over the k-tiles of a tile is loaded once per tile
  float pm[PRO == 1 ? NA : 1], pr[PRO == 1 ? NA : 1];      // PRO 1: (mean, rstd) of the thread's rows
  float4 pg = zero4(), pbias = zero4();                    // PRO 2: gain / bias of the thread's columns
  const NormPro& np = PRO == 2 ? g.pb : g.pa;
  const bool pact = PRO != 0 && np.slope != nullptr;
  const float pslope = pact ? *np.slope : 1.f;
  auto fetch = [&](const Work& w) {
    if (BRV_BIG_ABL & 2) {
#pragma unroll
      for (int i = 0; i < NA; ++i) ra[i] = make_float4((float)tid, 1.f, 2.f, (float)w.k0);
#pragma unroll
      for (int i = 0; i < NB; ++i) rb[i] = make_float4((float)tid, 1.f, 2.f, (float)w.k0);
      return;
    }
    const bool csec = g.n_split > 0 && w.n0 >= g.n_split;       // tile of the second B (uniform)
    const int cshift = csec ? g.n_split : 0;
    const float* A = g.A + (long long)w.z*g.a_bs + (long long)w.kb*g.a_kbs;
    const float* B = (csec ? g.B2 : g.B) + (long long)w.z*g.b_bs + (long long)w.kb*g.b_kbs;
    if (w.fresh) {
      // first k-tile of a tile (or of the next operand pair): pointers from scratch; afterwards
      // they advance by one k-tile per call
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int m = w.m0 + a_r[i];
        oka[i] = m < g.M;
        pa[i] = TA ? A + (long long)(w.k0 + a_k[i])*g.lda + m : A + (long long)m*g.lda + (w.k0 + a_k[i]);
        if (PRO == 1) { const int mc = oka[i] ? m : 0; pm[i] = np.table[2*mc]; pr[i] = np.table[2*mc + 1]; }
      }
      if (PRO == 2) {
        const int n = w.n0 + b_r[0];            // the same columns for all of the thread's B loads
        const int nc = n < g.N ? n : 0;
        pg = ld4s(np.gain + nc); pbias = ld4s(np.bias + nc);
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int n = w.n0 + b_r[i] - cshift;
        okb[i] = n + cshift < g.N;
        pb[i] = TB ? B + (long long)n*g.ldb + (w.k0 + b_k[i]) : B + (long long)(w.k0 + b_k[i])*g.ldb + n;
      }
    }
    if (PRO == 1) {
      const int kc = w.k0 + a_k[0];             // the same 4 channels for all of the thread's A loads
      const int kcc = kc < g.K ? kc : 0;
      pg = ld4s(np.gain + kcc); pbias = ld4s(np.bias + kcc);
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const bool ok = oka[i] && w.k0 + a_k[i] < g.K;
      float4 v = *reinterpret_cast<const float4*>(ok ? pa[i] : A);
      if (PRO == 1) v = norm_pro(v, pm[i], pr[i], pslope, pact, pg, pbias);
      ra[i] = ok ? v : zero4();
      pa[i] += TA ? (long long)BK*g.lda : BK;
    }
    if (p.b_scalar) {
      // weights inside a flat parameter buffer: any alignment, any extent; 4 dwords per 16-byte piece with
      // element-wise bounds (small, L2-resident operand); the 4 NB loads are issued together
      float e[NB][4]; bool okc[NB][4];
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int lim = TB ? g.K - (w.k0 + b_k[i]) : g.N - (w.n0 + b_r[i]);     // elements left along the contiguous axis
        const bool rowok = TB ? okb[i] : w.k0 + b_k[i] < g.K;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          okc[i][c] = rowok && c < lim;
          e[i][c] = (okc[i][c] ? pb[i] : B)[okc[i][c] ? c : 0];
        }
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        rb[i] = make_float4(okc[i][0] ? e[i][0] : 0.f, okc[i][1] ? e[i][1] : 0.f, okc[i][2] ? e[i][2] : 0.f,
                            okc[i][3] ? e[i][3] : 0.f);
        pb[i] += TB ? BK : (long long)BK*g.ldb;
      }
    } else {
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const bool ok = okb[i] && w.k0 + b_k[i] < g.K;
        float4 v = *reinterpret_cast<const float4*>(ok ? pb[i] : B);
        if (PRO == 2) {
          const int kc = w.k0 + b_k[i] < g.K ? w.k0 + b_k[i] : 0;
          v = norm_pro(v, np.table[2*kc], np.table[2*kc + 1], pslope, pact, pg, pbias);
        }
        rb[i] = ok ? v : zero4();
        pb[i] += TB ? BK : (long long)BK*g.ldb;
      }
    }
  };
  // LDS staging addresses are per-thread constants
  int sa[NA], sb[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) sa[i] = TA ? a_k[i]*LDA + a_r[i] : a_r[i]*LDA + a_k[i];
#pragma unroll
  for (int i = 0; i < NB; ++i) sb[i] = A_FLOATS + (TB ? b_r[i]*LDB + b_k[i] : b_k[i]*LDB + b_r[i]);
  auto stash = [&](int buf) {
    float* S = lds + buf*STAGE;
#pragma unroll
    for (int i = 0; i < NA; ++i) *reinterpret_cast<float4*>(S + sa[i]) = ra[i];
#pragma unroll
    for (int i = 0; i < NB; ++i) *reinterpret_cast<float4*>(S + sb[i]) = rb[i];
  };

  f32x16 acc[2][2];
  auto clear = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };
  // operand fragment addresses inside a stage (per-thread constants)
  const int fa = TA ? 4*kh*LDA + 64*wm + r32 : (64*wm + r32)*LDA + 4*kh;
  const int fb = A_FLOATS + (TB ? (64*wn + r32)*LDB + 4*kh : 4*kh*LDB + 64*wn + r32);
  auto compute = [&](int buf) {
    const float* S = lds + buf*STAGE;
#pragma unroll
    for (int j = 0; j < BK/8; ++j) {
      float a[2][4], b[2][4];
      if (BRV_BIG_ABL & 8) {
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int i = 0; i < 4; ++i) { a[f][i] = (float)(lane + i + j); b[f][i] = (float)(lane - i + f); }
      } else {
#pragma unroll
        for (int fi = 0; fi < 2; ++fi) {
          if (TA) {
#pragma unroll
            for (int i = 0; i < 4; ++i) a[fi][i] = S[fa + (8*j + i)*LDA + 32*fi];
          } else {
            const float4 v = *reinterpret_cast<const float4*>(S + fa + 32*fi*LDA + 8*j);
            a[fi][0] = v.x; a[fi][1] = v.y; a[fi][2] = v.z; a[fi][3] = v.w;
          }
        }
#pragma unroll
        for (int fj = 0; fj < 2; ++fj) {
          if (TB) {
            const float4 v = *reinterpret_cast<const float4*>(S + fb + 32*fj*LDB + 8*j);
            b[fj][0] = v.x; b[fj][1] = v.y; b[fj][2] = v.z; b[fj][3] = v.w;
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) b[fj][i] = S[fb + (8*j + i)*LDB + 32*fj];
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int fi = 0; fi < 2; ++fi)
#pragma unroll
          for (int fj = 0; fj < 2; ++fj)
            if (BRV_BIG_ABL & 1) acc[fi][fj][i] += a[fi][i]*b[fj][i];
            else acc[fi][fj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[fi][i], b[fj][i], acc[fi][fj], 0, 0, 0);
    }
  };
  auto epilogue = [&](const Work& w) {
    // D element (row, col) of a 32 x 32 block: col = lane & 31, row = (i & 3) + 8 (i >> 2) + 4 (lane >> 5).
    // Fast path (every block that does not straddle m_split, results below 2 GB): one buffer descriptor
    // per side, the lane's byte offset computed once per block, ONE add per element -- rows past the end
    // fall outside the descriptor's range and are dropped / read as zero by the hardware. The
    // first version spent ~20 instructions per element on 64-bit address arithmetic: 4 us per tile,
    // more than the MFMAs of a K = 128 tile (without global memory: 58 -> 47 us per 8.4 GFLOP product).
    // Column split: the whole tile belongs to one side (columns local to the side).
    const bool csec = g.n_split > 0 && w.n0 >= g.n_split;
    const int cshift = csec ? g.n_split : 0;
    float* Da = (csec ? g.D2 : g.D) + (long long)w.z*g.d_bs;                 // rows below m_split
    float* Db = g.n_split > 0 ? Da : g.D2 + (long long)w.z*g.d_bs;           // rows from m_split on
    const float* addc = csec ? g.add2 : g.add;
    const float* Aa = addc ? addc + (long long)w.z*g.add_bs : nullptr;
    const float* Ab = g.n_split > 0 ? Aa : (g.add ? g.add2 + (long long)w.z*g.add_bs : nullptr);
    const float* biasc = csec ? g.bias2 : g.bias;
#pragma unroll
    for (int fi = 0; fi < 2; ++fi)
#pragma unroll
      for (int fj = 0; fj < 2; ++fj) {
        const int col = w.n0 + 64*wn + 32*fj + r32;
        const bool colok = col < g.N;
        const int cl = (colok ? col : g.N - 1) - cshift;            // column inside the side
        const int rb0 = w.m0 + 64*wm + 32*fi;
        const int rbase = rb0 + 4*kh;
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = acc[fi][fj][i];
        if (p.ksplit > 1) {
          float* sc = g.scratch + ((long long)w.split*g.batch + w.z)*g.M*g.N + (colok ? col : g.N - 1);
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int row = rbase + (i & 3) + 8*(i >> 2);
            if (colok && row < g.M) sc[row*g.N] = v[i];
          }
          continue;
        }
        if (biasc) {
          if (g.col_bias) {
            const float bc = biasc[cl];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] += bc;
          } else {
            float br[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const int row = rbase + (i & 3) + 8*(i >> 2);
              br[i] = biasc[row < g.M ? row : g.M - 1];
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] += br[i];
          }
        }
        const bool cross = g.m_split < g.M && rb0 < g.m_split && rb0 + 32 > g.m_split;
        if (p.fast_epi && !cross) {
          const bool sec = rb0 >= g.m_split;                           // uniform over the block
          const int rl0 = rbase - (sec ? g.m_split : 0);
          const int m_side = sec ? g.M - g.m_split : (g.m_split < g.M ? g.m_split : g.M);
          const __amdgpu_buffer_rsrc_t rd = make_rsrc(sec ? Db : Da, (long long)m_side*g.ldd*4);
          // a lane outside the columns gets an offset past every range (and cannot wrap: ranges < 2 GB)
          const unsigned int vo = colok ? (unsigned int)(rl0*g.ldd + cl)*4u : 0x80000000u;
          if (Aa) {
            const __amdgpu_buffer_rsrc_t rad = make_rsrc(sec ? Ab : Aa, (long long)m_side*g.ldadd*4);
            const unsigned int va = colok ? (unsigned int)(rl0*g.ldadd + cl)*4u : 0x80000000u;
            float ad[16];
#pragma unroll
            for (int i = 0; i < 16; ++i)
              ad[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                          rad, (int)(va + (unsigned int)(((i & 3) + 8*(i >> 2))*g.ldadd)*4u), 0, 0));
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] += ad[i];
          }
          if (!(BRV_BIG_ABL & 4)) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v[i]), rd,
                                                    (int)(vo + (unsigned int)(((i & 3) + 8*(i >> 2))*g.ldd)*4u), 0, 0);
          }
          continue;
        }
        // general path: a block that straddles m_split (or a result too large for 32-bit byte offsets)
        if (Aa) {
          float ad[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            int row = rbase + (i & 3) + 8*(i >> 2);
            if (row >= g.M) row = g.M - 1;
            const bool second = row >= g.m_split;
            ad[i] = (second ? Ab : Aa)[(long long)(second ? row - g.m_split : row)*g.ldadd + cl];
          }
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] += ad[i];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = rbase + (i & 3) + 8*(i >> 2);
          const bool second = row >= g.m_split;
          if (colok && row < g.M && !(BRV_BIG_ABL & 4))
            (second ? Db : Da)[(long long)(second ? row - g.m_split : row)*g.ldd + cl] = v[i];
        }
      }
  };

  if constexpr (X3) {
    bf16_t* H = reinterpret_cast<bf16_t*>(lds);
    // planes: A hi, mid, lo, then B hi, mid, lo; row-major [rows][LDH] for an operand contiguous in k (!TA / TB),
    // k-major [BK][rows + 8] for one contiguous along its rows (TA / !TB)
    constexpr int LDTB = TN + 8;
    constexpr int PA = X3_PA, PB = X3_PB;
    auto split_store = [&](bf16_t* plane0, int plane_stride, int off, const float4& v) {
      // three bf16 pieces of 4 values -> one 8-byte store per plane
      const uint32_t h01 = pack2(v.x, v.y), h23 = pack2(v.z, v.w);
      const float r0 = v.x - __uint_as_float(h01 << 16), r1 = v.y - __uint_as_float(h01 & 0xffff0000u);
      const float r2 = v.z - __uint_as_float(h23 << 16), r3 = v.w - __uint_as_float(h23 & 0xffff0000u);
      const uint32_t m01 = pack2(r0, r1), m23 = pack2(r2, r3);
      const float s0 = r0 - __uint_as_float(m01 << 16), s1 = r1 - __uint_as_float(m01 & 0xffff0000u);
      const float s2 = r2 - __uint_as_float(m23 << 16), s3 = r3 - __uint_as_float(m23 & 0xffff0000u);
      const uint32_t l01 = pack2(s0, s1), l23 = pack2(s2, s3);
      *reinterpret_cast<uint2*>(plane0 + off) = make_uint2(h01, h23);
      *reinterpret_cast<uint2*>(plane0 + plane_stride + off) = make_uint2(m01, m23);
      *reinterpret_cast<uint2*>(plane0 + 2*plane_stride + off) = make_uint2(l01, l23);
    };
    auto stash3 = [&]() {
      if (BRV_BIG_ABL & 16) {          // no split: one plane written, the others left as they are
#pragma unroll
        for (int i = 0; i < NA; ++i) *reinterpret_cast<uint2*>(H + a_r[i]*LDH + a_k[i]) = make_uint2(__float_as_uint(ra[i].x), __float_as_uint(ra[i].y));
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<uint2*>(H + 3*PA + b_r[i]*LDH + b_k[i]) = make_uint2(__float_as_uint(rb[i].x), __float_as_uint(rb[i].y));
        return;
      }
#pragma unroll
      for (int i = 0; i < NA; ++i) split_store(H, PA, TA ? a_k[i]*LDT + a_r[i] : a_r[i]*LDH + a_k[i], ra[i]);
#pragma unroll
      for (int i = 0; i < NB; ++i) split_store(H + 3*PA, PB, TB ? b_r[i]*LDH + b_k[i] : b_k[i]*LDTB + b_r[i], rb[i]);
    };
    const int ha = (64*wm + r32)*LDH + 8*kh, hb = 3*PA + (64*wn + r32)*LDH + 8*kh;
    auto compute3 = [&]() {
#pragma unroll
      for (int ks = 0; ks < BK/16; ++ks) {
        bf16x8 a[2][3], b[2][3];
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) {
            if (BRV_BIG_ABL & 8) {
              const uint4 q = make_uint4(lane + pl, f, ks, 1);
              a[f][pl] = __builtin_bit_cast(bf16x8, q); b[f][pl] = __builtin_bit_cast(bf16x8, q);
              continue;
            }
            if (TA) a[f][pl] = tr_frag16(H + pl*PA, LDT, 16*ks, 64*wm + 32*f, lane);
            else a[f][pl] = *reinterpret_cast<const bf16x8*>(H + ha + pl*PA + 32*f*LDH + 16*ks);
            if (!TB) b[f][pl] = tr_frag16(H + 3*PA + pl*PB, LDTB, 16*ks, 64*wn + 32*f, lane);
            else b[f][pl] = *reinterpret_cast<const bf16x8*>(H + hb + pl*PB + 32*f*LDH + 16*ks);
          }
        if (BRV_BIG_ABL & 1) {
#pragma unroll
          for (int fi = 0; fi < 2; ++fi)
#pragma unroll
            for (int fj = 0; fj < 2; ++fj)
#pragma unroll
              for (int pl = 0; pl < 3; ++pl) acc[fi][fj][pl] += (float)a[fi][pl][0]*(float)b[fj][pl][1];
          continue;
        }
        // small terms first; consecutive MFMAs go to different accumulators (a dependent MFMA waits for
        // the whole pass count of its predecessor)
        constexpr int PA_[6] = {2, 0, 1, 1, 0, 0}, PB_[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
          for (int fi = 0; fi < 2; ++fi)
#pragma unroll
            for (int fj = 0; fj < 2; ++fj)
              acc[fi][fj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[fi][PA_[t]], b[fj][PB_[t]], acc[fi][fj], 0, 0, 0);
      }
    };
    Work cur = decode(blockIdx.x);
    if (!cur.valid) return;
    clear();
    fetch(cur);
    while (true) {
      stash3();
      lds_barrier();
      const Work nxt = advance(cur);
      if (nxt.valid) fetch(nxt);
      compute3();
      if (cur.left == 1) { epilogue(cur); clear(); }
      if (!nxt.valid) break;
      lds_barrier();
      cur = nxt;
    }
    return;
  }
  Work cur = decode(blockIdx.x);
  if (!cur.valid) return;
  clear();
  fetch(cur);
  stash(0);
  lds_barrier();
  int buf = 0;
  while (true) {
    const Work nxt = advance(cur);
    if (nxt.valid) fetch(nxt);
    compute(buf);
    if (cur.left == 1) { epilogue(cur); clear(); }
    if (!nxt.valid) break;
    stash(buf ^ 1);
    lds_barrier();
    buf ^= 1;
    cur = nxt;
  }
}

// D = bias + add + partial[0] + partial[1] + ... : a workgroup owns 64 elements, its 4 wavefronts
// take the splits s = w, w + 4, ... (4 loads in flight each), the 4 sums are added in wavefront
// order -- the association is fixed by the shape, not by the order of arrival
__global__ __launch_bounds__(256) void gemm_f32_big_reduce_kernel(const BigDev p) {
  __shared__ float part[4][64];
  const BigGemm& g = p.g;
  const long long per = (long long)g.M*g.N;
  const long long n = (long long)g.batch*per;
  const int c = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long long i = (long long)blockIdx.x*64 + c;
  const long long ic = i < n ? i : n - 1;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int s = w;
  for (; s + 12 < p.ksplit; s += 16) {
    const float v0 = g.scratch[(long long)s*n + ic], v1 = g.scratch[(long long)(s + 4)*n + ic];
    const float v2 = g.scratch[(long long)(s + 8)*n + ic], v3 = g.scratch[(long long)(s + 12)*n + ic];
    s0 += v0; s1 += v1; s2 += v2; s3 += v3;
  }
  for (; s < p.ksplit; s += 4) s0 += g.scratch[(long long)s*n + ic];
  part[w][c] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (w != 0 || i >= n) return;
  float v = ((part[0][c] + part[1][c]) + part[2][c]) + part[3][c];
  const int z = (int)(i / per); const long long r = i % per;
  const int row = (int)(r / g.N), col = (int)(r % g.N);
  if (g.bias) v += g.bias[g.col_bias ? col : row];
  const bool second = row >= g.m_split;
  const int rr = second ? row - g.m_split : row;
  const float* ad = second ? g.add2 : g.add;
  if (ad) v += ad[(long long)z*g.add_bs + (long long)rr*g.ldadd + col];
  (second ? g.D2 : g.D)[(long long)z*g.d_bs + (long long)rr*g.ldd + col] = v;
}

inline bool q4(long long v) { return (v & 3) == 0; }
inline bool a16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

inline bool b_vector(const BigGemm& g) {
  return a16(g.B) && (!g.B2 || a16(g.B2)) && q4(g.ldb) && q4(g.b_bs) && q4(g.b_kbs) &&
         q4(g.tb ? g.K : g.N) && (g.tb || q4(g.n_split));
}

// the split-bf16 kernel takes the products with both operands contiguous in k and a long M
inline bool use_x3(const BigGemm& g) {
#ifdef BRV_GEMM_NO_X3
  return false;
#else
  // enough 128 x 128 tiles to fill the chip without splitting the reduction
  const long long tiles = (long long)((g.M + 127)/128)*((g.N + 127)/128)*(g.batch > 0 ? g.batch : 1);
  if (!g.x3) return false;
  // (x3 & 2: the caller brings scratch for an ordered reduction split -- long reductions over few tiles, the
  // weight gradients of convolutions whose column matrix and output gradient are both contiguous in k)
  if (!g.ta && g.tb)
    return !g.pb.table && (tiles >= 192 || ((g.x3 & 2) && (long long)g.K*(g.kbatch > 1 ? g.kbatch : 1) >= 8192));
  // weight-gradient form (both operands contiguous along their rows): a long reduction split over the chip
  if (g.ta && !g.tb)
    return !g.pa.table && b_vector(g) && ((long long)g.K*(g.kbatch > 1 ? g.kbatch : 1) >= 8192 ||
                                          ((g.x3 & 4) && tiles >= 192 && !g.pb.table));
  // a contiguous in k, b along its rows (a convolution as weights x column matrix): no split
  if (!g.ta && !g.tb) return (g.x3 & 4) && !g.pa.table && !g.pb.table && b_vector(g) && tiles >= 192;
  return false;
#endif
}

struct Plan { int wm, wn, m_tiles, n_tiles, mn_padded, xcd_perm, ksplit, ktiles; long long total_t, per_t; };

int device_cus() {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) return 256;
  return cus;
}

Plan make_plan(const BigGemm& g) {
  Plan pl;
  auto waste = [&](long long tm, long long tn) {
    return ((g.M + tm - 1)/tm*tm)*((g.N + tn - 1)/tn*tn);
  };
  bool tall = waste(256, 128) <= waste(128, 256);
  if (g.n_split > 0 && g.n_split % 256 != 0) tall = true;      // whole 128-wide tiles on either side of the split
  pl.wm = tall ? 4 : 2; pl.wn = tall ? 2 : 4;
  if (use_x3(g)) { pl.wm = 2; pl.wn = 2; }
  const int tm = 64*pl.wm, tn = 64*pl.wn;
  pl.m_tiles = (int)((g.M + tm - 1)/tm); pl.n_tiles = (int)((g.N + tn - 1)/tn);
  pl.xcd_perm = pl.m_tiles >= 16 && pl.n_tiles > 1;
  pl.mn_padded = pl.xcd_perm ? (pl.m_tiles + 7)/8*8*pl.n_tiles : pl.m_tiles*pl.n_tiles;
  const int bk = use_x3(g) ? BRV_X3_BK : kBK;
  pl.ktiles = (g.K + bk - 1)/bk;
  pl.total_t = (long long)(g.kbatch > 1 ? g.kbatch : 1)*pl.ktiles;
  const long long tiles = (long long)g.batch*pl.m_tiles*pl.n_tiles;
  long long ks = 1;
  const int cus = device_cus();
  const bool x3w = use_x3(g) && (g.ta || (g.x3 & 2));          // several 4-wavefront workgroups per CU
  constexpr int kX3Wgs = BRV_X3_BK == 16 ? 3 : 2;
  if (tiles*2 <= cus*(x3w ? kX3Wgs : 1) && pl.total_t >= 32 && (!use_x3(g) || x3w)) {
    ks = cus*(x3w ? kX3Wgs : 1)/tiles;
    if (ks > pl.total_t/8) ks = pl.total_t/8;
    if (ks < 1) ks = 1;
  } else if ((g.x3 & 8) && tiles <= 32 && pl.total_t >= 8 && (!use_x3(g) || x3w)) {
    // (x3 bit 8: set by the entry points that bring scratch for the partial tiles -- brv_gemm_f32_ws)
    // a handful of tiles and a SHORT reduction (round 6: the attention products of the SGMSE+ U-Net at batch 1 --
    // 512 x 512 x 256 and 256 x 512 x 512: 4 - 16 tiles, 8 - 16 k-tiles -- ran 47 / 78 us on 4 - 16 workgroups): split
    // down to four k-tiles per workgroup
    ks = pl.total_t/4;
    if (ks*tiles > 4LL*cus) ks = 4LL*cus/tiles;
    if (ks < 1) ks = 1;
  }
  pl.per_t = (pl.total_t + ks - 1)/ks;
  pl.ksplit = (int)((pl.total_t + pl.per_t - 1)/pl.per_t);
  return pl;
}

}  // namespace

bool gemm_f32_big_ok(const BigGemm& g) {
  if (g.M < 1 || g.N < 1 || g.K < 1 || g.batch < 1) return false;
  // A: 16-byte loads (aligned base and strides, the extent along its contiguous axis a whole number
  // of 16-byte pieces); B takes the dword loader when it does not qualify (b_vector)
  if (!a16(g.A) || !q4(g.lda) || !q4(g.a_bs) || !q4(g.a_kbs) || !q4(g.ta ? g.M : g.K)) return false;
  if (g.pb.table && !b_vector(g)) return false;
  if ((g.pa.table && g.ta) || (g.pb.table && (g.tb || g.kbatch > 1)) || (g.pa.table && g.pb.table)) return false;
  if ((g.pa.table && !g.tb) || (g.pb.table && !g.ta)) return false;       // instantiated pairs only
  if ((g.pa.table || g.pb.table) && g.batch != 1) return false;            // frame index = row / k index
  // 32-bit offsets inside one batch item of the result / addend, 32-bit work-item counts
  if ((long long)g.M*g.ldd >= (1LL << 31) || (g.add && (long long)g.M*g.ldadd >= (1LL << 31)) ||
      (long long)g.M*g.N >= (1LL << 31)) return false;
  if (g.n_split > 0) {
    // second B: whole tiles on either side, no row split, no reduction split (long-M shapes only)
    if (!g.B2 || !g.D2 || g.m_split > 0 || g.n_split >= g.N || g.pb.table) return false;
    const Plan pl = make_plan(g);
    if (g.n_split % (64*pl.wn) != 0 || pl.ksplit > 1) return false;
    return true;
  }
  if (g.D2 && (g.m_split < 1 || g.m_split >= g.M)) return false;
  return true;
}

long long gemm_f32_big_scratch(const BigGemm& g) {
  const Plan pl = make_plan(g);
  return pl.ksplit > 1 ? (long long)pl.ksplit*g.batch*g.M*g.N : 0;
}

int gemm_f32_big(const BigGemm& g_in, hipStream_t st) {
  if (!gemm_f32_big_ok(g_in)) return -1;
  BigDev p;
  p.g = g_in;
  if (!p.g.D2 || p.g.n_split > 0) { p.g.m_split = p.g.M; if (!p.g.D2) { p.g.D2 = p.g.D; p.g.add2 = p.g.add; } }
  if (p.g.kbatch < 1) p.g.kbatch = 1;
  const Plan pl = make_plan(p.g);
  p.m_tiles = pl.m_tiles; p.n_tiles = pl.n_tiles; p.mn_padded = pl.mn_padded; p.xcd_perm = pl.xcd_perm;
  p.ksplit = pl.ksplit; p.ktiles = pl.ktiles; p.total_t = pl.total_t; p.per_t = pl.per_t;
  if (p.ksplit > 1 && (!p.g.scratch || p.g.scratch_floats < (long long)p.ksplit*p.g.batch*p.g.M*p.g.N)) {
    // no scratch from the caller: one workgroup per tile walks the whole reduction
    p.ksplit = 1; p.per_t = p.total_t;
  }
  p.split_xcd = p.ksplit >= 8 && p.g.batch == 1 && !p.xcd_perm;
  p.n_work = p.split_xcd ? (p.ksplit + 7)/8*8*p.mn_padded : p.g.batch*p.ksplit*p.mn_padded;
  p.b_scalar = b_vector(p.g) ? 0 : 1;
  p.fast_epi = (long long)p.g.M*p.g.ldd*4 < (1LL << 31) && (!p.g.add || (long long)p.g.M*p.g.ldadd*4 < (1LL << 31));
  const bool x3 = use_x3(p.g) && (p.ksplit == 1 || p.g.ta || (p.g.x3 & 2));
  const int wgs = device_cus()*(x3 ? (BRV_X3_BK == 16 ? 3 : 2) : 1);
  const int grid = p.n_work < wgs ? p.n_work : wgs;
  const int pro = p.g.pa.table ? 1 : (p.g.pb.table ? 2 : 0);
#define BRV_BIG(WM_, WN_, TA_, TB_, PRO_) \
  hipLaunchKernelGGL((gemm_f32_big_kernel<WM_, WN_, TA_, TB_, PRO_>), dim3(grid), dim3(64*WM_*WN_), 0, st, p)
#define BRV_BIG_SHAPE(WM_, WN_)                                       \
  do {                                                                \
    if (pro == 1) BRV_BIG(WM_, WN_, false, true, 1);                  \
    else if (pro == 2) BRV_BIG(WM_, WN_, true, false, 2);             \
    else if (p.g.ta && p.g.tb) BRV_BIG(WM_, WN_, true, true, 0);      \
    else if (p.g.ta) BRV_BIG(WM_, WN_, true, false, 0);               \
    else if (p.g.tb) BRV_BIG(WM_, WN_, false, true, 0);               \
    else BRV_BIG(WM_, WN_, false, false, 0);                          \
  } while (0)
  if (x3 && p.g.ta) {
    if (pro == 2) hipLaunchKernelGGL((gemm_f32_big_kernel<2, 2, true, false, 2, true>), dim3(grid), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((gemm_f32_big_kernel<2, 2, true, false, 0, true>), dim3(grid), dim3(256), 0, st, p);
  } else if (x3 && !p.g.tb) {
    hipLaunchKernelGGL((gemm_f32_big_kernel<2, 2, false, false, 0, true>), dim3(grid), dim3(256), 0, st, p);
  } else if (x3) {
    if (pro == 1) hipLaunchKernelGGL((gemm_f32_big_kernel<2, 2, false, true, 1, true>), dim3(grid), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((gemm_f32_big_kernel<2, 2, false, true, 0, true>), dim3(grid), dim3(256), 0, st, p);
  } else if (pl.wm == 4) BRV_BIG_SHAPE(4, 2); else BRV_BIG_SHAPE(2, 4);
#undef BRV_BIG_SHAPE
#undef BRV_BIG
  if (p.ksplit > 1) {
    const long long n = (long long)p.g.batch*p.g.M*p.g.N;
    hipLaunchKernelGGL(gemm_f32_big_reduce_kernel, dim3((unsigned)((n + 63)/64)), dim3(256), 0, st, p);
  }
  return (int)hipGetLastError();
}

}  // namespace brv
