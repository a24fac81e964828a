// Fused forward of the second half of a TCN block, second organisation (round 5). Same contract as
// dwpw2_fused_kernel (dwpw2_fused.cuh): z2 = dconv(gLN_1(PReLU_1(z1))) + bias stored, statistics of
// p = PReLU_2(z2), u = (W gamma_2) p stored as bf16; reference brever/models/convtasnet/convtasnet.py:240-260.
//
// Why a second form. The ablations of the first one (profiles/r05_dwpw2_ablation.txt) show its data-movement
// skeleton alone -- tap loads, LDS slabs, barriers, no arithmetic, no stores -- takes 26 us for 65.5 MB of z1
// (2.5 TB/s): a thread owns one channel octet of a 64-channel slab and two frames, its six 16-byte tap loads go
// out ONE slab (48 KB per CU) ahead and there are eight barriers per 128 frames; the phases (loads, stencil,
// MFMA, stores) then add up to 55 us where the bytes alone need 30. Here
//   * a WAVE owns whole frames: its 64 lanes are the 64 channel octets of a 1 KB row, so every load / store
//     instruction moves one full row, the per-channel coefficients of a lane never change and live in
//     REGISTERS (no table reads: 65 KB of LDS traffic per slab in the first form), and the taps of a frame
//     are requested FOUR frames (12 KB per wave, 96 KB per CU) before they are used;
//   * p of a whole 64-frame tile (all 512 channels) sits in LDS, double buffered: the [res | skip] product of
//     tile n (every wave: 32 outputs x 64 frames x K = 512, weights from L2 one k-group ahead) and the stencil
//     of tile n + 1 are one instruction stream, with ONE barrier per tile instead of eight per 128 frames;
//   * u leaves straight from the accumulators (8 bytes per lane and store: four consecutive outputs of a frame).
#pragma once
#include "dwpw2_fused.cuh"

namespace brv {

constexpr int D2_TT = 64;                          // frames per tile
constexpr int D2_LDP = DP_H + 8;                   // halves per p row: 1040 B = 4 banks mod 64, conflict-free b128 reads
constexpr int D2_PBYTES = D2_TT*D2_LDP*2;          // 66 560
constexpr int D2_OFF_TAB = 2*D2_PBYTES;
constexpr int D2_SMEM = D2_OFF_TAB + 4*DP_H*4 + 8*32*(64 + 16);   // + [wc0 | wc1 | wc2 | bias] of the item being staged + the waves' u patches (8 x 2560 or 4 x 4608 B): 161 792 B
#ifndef D2_SCHED
#define D2_SCHED 1
#endif
#ifndef D2_ABL
#define D2_ABL 0     // ablation bits (diagnostic builds): 1 no MFMA, 2 no z2 store, 4 no u store, 8 no stage arithmetic
#endif

__device__ __forceinline__ void buf_store8(__amdgpu_buffer_rsrc_t r, unsigned int off, uint2 q) {
  typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
  u32x2 v; v.x = q.x; v.y = q.y;
  __builtin_amdgcn_raw_buffer_store_b64(v, r, (int)off, 0, 0);
}

// NW = 8: two waves per SIMD (256 registers each); NW = 4: one wave per SIMD with the whole register file (no
// spills with eight frames of taps in flight), every wave two 32-output slices.
//
// SEQ (diagnostic, NW = 4): ONE p buffer (the u patches reuse it), the stencil and the product of a tile one after the
// other, 75 KB of LDS and 256 registers, so that TWO workgroups share a CU and one's product runs beside the other's
// stencil -- the phases overlap between workgroups instead of inside one instruction stream.
template <int NW, int AHEAD, bool SEQ = false>
__global__ __launch_bounds__(64*NW) __attribute__((amdgpu_waves_per_eu(SEQ ? 2 : 1, SEQ ? 2 : NW/4)))
void dwpw2_v2_kernel(const DwPw2Params p) {
  constexpr int D2_FPW = D2_TT/NW;                 // frames per wave and tile
  constexpr int NSL = 8/NW;                        // 32-output slices per wave
  constexpr int KPJ = 32/D2_FPW;                   // k-steps beside one frame of the stencil
  constexpr int D2_AHEAD = AHEAD;
  static_assert(!SEQ || NW == 4, "the sequential form is written for four waves");
  constexpr int OFF_TAB = SEQ ? D2_PBYTES : D2_OFF_TAB;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SEQ ? D2_PBYTES + 4*DP_H*4 : D2_SMEM];
  float* tabs = reinterpret_cast<float*>(smem + OFF_TAB);         // [4][512]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n32 = lane & 31, khalf = lane >> 5;
  const int T = p.T, dil = p.dil, left = p.left;
  const int tpi = (T + D2_TT - 1)/D2_TT;
  const int n_tiles = tpi*p.B;
  const float a1 = *p.slope1, a2 = *p.slope2;
  const float c1r = 0.5f*(1.f + a1), c2 = 0.5f*(1.f - a1);
  const float c1 = __builtin_fabsf(c1r) < 0x1p-40f ? 0x1p-40f : c1r;     // (as the first form)
  const float rho = c2/c1;
  const float d1 = 0.5f*(1.f + a2), d2 = 0.5f*(1.f - a2);
  const int cb = lane*8;                                          // this lane's channel octet, every frame

  // tiles of this workgroup: XCD k takes the k-th contiguous eighth (dilated taps of neighbouring tiles from one L2) ...
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = (n_tiles + 7) >> 3;
  const int wg_per_xcd = gridDim.x >> 3;
  // ... and a workgroup a contiguous run of that eighth: consecutive frames of one item (the coefficients change with
  // the item: rarely; the halo rows of a tile are the neighbour's centre rows)
  const int run = (per_xcd + wg_per_xcd - 1)/wg_per_xcd;
  const int first = xcd*per_xcd + slot*run;
  int n_my = min(run, min(per_xcd - slot*run, n_tiles - first));
  if (n_my <= 0) return;
  auto tile_of = [&](int i) { return first + i; };

  // ---- per-lane coefficients of the item being staged ------------------------------------------------
  float wa[3][8], bs7[8];
  int coef_item = -1;
  auto load_coefs = [&](int b) {
    const NormStat ns = norm_stat(p.stats1, b, p.inv_n, p.eps);
    float wc[3][8], bia[8], gam[8], bet[8], tp[24];
    // unconditional (clamped) loads + selects: guarded per-element loads compile to 48 load / branch / wait
    // sequences in a row (2 us per call); whole octets inside the tensor take the vector loads
    if (cb + 8 <= p.C) {
      const float4* g4 = reinterpret_cast<const float4*>(p.gamma1 + cb);
      const float4* b4 = reinterpret_cast<const float4*>(p.beta1 + cb);
      const float4* d4 = reinterpret_cast<const float4*>(p.dbias + cb);
      const float4* t4 = reinterpret_cast<const float4*>(p.taps + cb*3);
      const float4 g0 = g4[0], g1 = g4[1], b0 = b4[0], b1 = b4[1], d0 = d4[0], d1_ = d4[1];
      const float4 t0 = t4[0], t1 = t4[1], t2 = t4[2], t3 = t4[3], t4_ = t4[4], t5 = t4[5];
      const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      const float dd[8] = {d0.x, d0.y, d0.z, d0.w, d1_.x, d1_.y, d1_.z, d1_.w};
      const float tt[24] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w, t2.x, t2.y, t2.z, t2.w,
                            t3.x, t3.y, t3.z, t3.w, t4_.x, t4_.y, t4_.z, t4_.w, t5.x, t5.y, t5.z, t5.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) { gam[j] = gg[j]; bet[j] = bb[j]; bia[j] = dd[j]; }
#pragma unroll
      for (int j = 0; j < 24; ++j) tp[j] = tt[j];
    } else {
      load8_masked(p.gamma1, cb, p.C, gam); load8_masked(p.beta1, cb, p.C, bet); load8_masked(p.dbias, cb, p.C, bia);
#pragma unroll
      for (int j = 0; j < 24; ++j) {
        const int idx = cb*3 + j;
        const float x = p.taps[idx < 3*p.C ? idx : 0];
        tp[j] = idx < 3*p.C ? x : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float scv = ns.rstd*gam[j], shv = bet[j] - ns.mean*ns.rstd*gam[j];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float w = tp[3*j + k];
        wa[k][j] = w*c1*scv; wc[k][j] = w*shv;
      }
      bs7[j] = bia[j] + wc[0][j] + wc[1][j] + wc[2][j];
    }
    // edge frames read these back; every wave writes the same values into the slots its own lanes read,
    // so no barrier is needed (a wave's LDS operations complete in order)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      *reinterpret_cast<float4*>(tabs + k*DP_H + cb) = make_float4(wc[k][0], wc[k][1], wc[k][2], wc[k][3]);
      *reinterpret_cast<float4*>(tabs + k*DP_H + cb + 4) = make_float4(wc[k][4], wc[k][5], wc[k][6], wc[k][7]);
    }
    *reinterpret_cast<float4*>(tabs + 3*DP_H + cb) = make_float4(bia[0], bia[1], bia[2], bia[3]);
    *reinterpret_cast<float4*>(tabs + 3*DP_H + cb + 4) = make_float4(bia[4], bia[5], bia[6], bia[7]);
    coef_item = b;
  };

  // statistics of p, per item
  int stat_item = -1;
  float ts = 0.f, tq = 0.f;
  auto flush_stats = [&]() {
    if (stat_item < 0) return;
    const double r0 = wave_sum((double)ts), r1 = wave_sum((double)tq);
    if (lane == 0) {
      atomic_add_f64(p.stats2 + stat_sum(stat_item), r0);
      atomic_add_f64(p.stats2 + stat_sq(stat_item), r1);
    }
    ts = 0.f; tq = 0.f;
  };

  // ---- tile bookkeeping: (item, first frame) of the tiles this workgroup stages next, advanced without divisions
  struct TilePos { int b, t0, live; };
  const int t_items = tpi*D2_TT, t_step = D2_TT;
  auto pos_of = [&](int i) {                      // (one division: start-up only)
    TilePos q; q.live = i < n_my;
    const int tile = q.live ? tile_of(i) : 0;
    q.b = tile / tpi; q.t0 = (tile % tpi)*D2_TT;
    return q;
  };
  int n_seen = 0;
  auto advance = [&](TilePos& q, int i) {         // q = tile index i - 1 -> tile index i
    q.live = i < n_my;
    q.t0 += t_step;
    while (q.t0 >= t_items) { q.t0 -= t_items; ++q.b; }
  };

  // ---- tap requests ----------------------------------------------------------------------------------
  // frame j of this wave in the tile at `q`: t = t0 + wid + NW j. A tile past the end requests nothing (descriptor
  // of zero records: the loads return zeros at once and keep the in-flight count uniform).
  const unsigned int lane_off = (unsigned int)(lane*16);
  auto request = [&](const TilePos& q, int j, uint4 (&raw)[3]) {
    const int t = q.t0 + wid + NW*j;
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(p.z1 + (long long)q.b*T*DP_H, q.live ? (long long)T*DP_H*2 : 0);
#pragma unroll
    for (int k = 0; k < 3; ++k)      // taps outside [0, T) wrap beyond the descriptor: zeros
      raw[k] = buf_load16(rin, (unsigned int)(t + k*dil - left)*(unsigned int)(DP_H*2) + lane_off);
  };

  // ---- stencil of one frame: z2 out, p into the LDS tile ----------------------------------------------
  auto stage = [&](const TilePos& q, int j, const uint4 (&raw)[3], bf16_t* pb) {
    const int fr = wid + NW*j, t = q.t0 + fr;
    const __amdgpu_buffer_rsrc_t rz2 = make_rsrc(p.z2 + (long long)q.b*T*DP_H, (long long)T*DP_H*2);
    const bool in0 = t - left >= 0 && t - left < T;
    const bool in1 = t + dil - left >= 0 && t + dil - left < T;
    const bool in2 = t + 2*dil - left >= 0 && t + 2*dil - left < T;
    float acc[8];
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) acc[jj] = bs7[jj];
    if (!(in0 && in1 && in2)) {                    // (wave-uniform, rare: frames within a dilation of the item's ends)
      // the constant terms of the taps that fall outside leave again
      const float4 b0 = *reinterpret_cast<const float4*>(tabs + 3*DP_H + cb);
      const float4 b1 = *reinterpret_cast<const float4*>(tabs + 3*DP_H + cb + 4);
      acc[0] = b0.x; acc[1] = b0.y; acc[2] = b0.z; acc[3] = b0.w;
      acc[4] = b1.x; acc[5] = b1.y; acc[6] = b1.z; acc[7] = b1.w;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float in = (k == 0 ? in0 : k == 1 ? in1 : in2) ? 1.f : 0.f;
        const float4 w0 = *reinterpret_cast<const float4*>(tabs + k*DP_H + cb);
        const float4 w1 = *reinterpret_cast<const float4*>(tabs + k*DP_H + cb + 4);
        const float wc[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) acc[jj] = __builtin_fmaf(in, wc[jj], acc[jj]);
      }
    }
#pragma unroll
    for (int k = 0; k < ((D2_ABL & 8) ? 1 : 3); ++k) {
      float f[8];
      unpack8(raw[k], f);
#pragma unroll
      for (int jj = 0; jj < 8; ++jj)
        acc[jj] = __builtin_fmaf(wa[k][jj], __builtin_fmaf(rho, __builtin_fabsf(f[jj]), f[jj]), acc[jj]);
    }
    const uint4 qz = pack8(acc);
    if (!(D2_ABL & 2)) buf_store16(rz2, (unsigned int)t*(unsigned int)(DP_H*2) + lane_off, qz);   // t >= T: dropped
    float rr[8], pv[8];
    unpack8(qz, rr);
    const float live = t < T ? 1.f : 0.f;
    float fs = 0.f, fq = 0.f;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      pv[jj] = live*__builtin_fmaf(d2, __builtin_fabsf(rr[jj]), d1*rr[jj]);
      fs += pv[jj]; fq = __builtin_fmaf(pv[jj], pv[jj], fq);
    }
    ts += fs; tq += fq;
    if (!(D2_ABL & 16)) *reinterpret_cast<uint4*>(pb + fr*D2_LDP + cb) = pack8(pv);
    else if (pv[0] == 123.456f) tabs[lane] = pv[1];
  };

  // ---- the product: wave wid = outputs 32 NSL wid .. x 64 frames ---------------------------------------
  f32x16 acc[NSL][2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[sl][f][i] = 0.f;
  };
  if (!SEQ) zero_acc();
  // weight fragments by buffer loads: ONE offset register (lane * 16) for all of them, the fragment's place as the
  // scalar offset (as global loads every fragment 4 KB apart needed an address pair of its own: 60 registers)
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.Wp + (long long)wid*NSL*32*DP_H, (long long)NSL*32*DP_H*2);
  bf16x8 wfr[2][NSL*KPJ];
  auto load_w = [&](int g, bf16x8 (&w)[NSL*KPJ]) {                      // k-steps KPJ g .. KPJ g + KPJ - 1
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
      for (int ks = 0; ks < KPJ; ++ks) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rw, (int)lane_off, (sl*32*DP_H + (KPJ*g + ks)*512)*2, 0);
        w[sl*KPJ + ks] = __builtin_bit_cast(bf16x8, v);
      }
  };

  uint4 ring[D2_AHEAD][3];
  static_assert((D2_AHEAD == 4 || D2_AHEAD == 2 || D2_AHEAD == 8) && D2_AHEAD <= D2_FPW, "ring slots are indexed with j & (D2_AHEAD - 1)");
  TilePos qs = pos_of(0);                          // tile being staged
  TilePos qn = pos_of(1);                          // the one after it (its first frames are requested meanwhile)
  TilePos qm = qs;                                 // tile being multiplied
#pragma unroll
  for (int j = 0; j < D2_AHEAD; ++j) request(qs, j, ring[j]);
  if (!SEQ) load_w(0, wfr[0]);

  // one pass over the wave's frames: the product of tile `qm` (MF) beside the stencil of tile `qs` (ST)
  auto pass = [&](auto mf_tag, auto st_tag, int par) {
    constexpr bool MF = decltype(mf_tag)::value, ST = decltype(st_tag)::value;
    if (ST && qs.b != coef_item && !((D2_ABL & 64) && coef_item >= 0)) load_coefs(qs.b);
    if (ST && qs.b != stat_item) { flush_stats(); stat_item = qs.b; }
    bf16_t* pst = reinterpret_cast<bf16_t*>(smem + (SEQ ? 0 : (par ^ 1)*D2_PBYTES));
    const bf16_t* pmm = reinterpret_cast<const bf16_t*>(smem + (SEQ ? 0 : par*D2_PBYTES)) + n32*D2_LDP + khalf*8;
    if (SEQ && MF) { zero_acc(); load_w(0, wfr[0]); }      // (neither lives through the stencil pass)
#pragma unroll
    for (int j = 0; j < D2_FPW; ++j) {
      if (MF) {
        if (!SEQ || j + 1 < D2_FPW) load_w((j + 1) % D2_FPW, wfr[(j + 1) & 1]);   // (last j: k-group 0 again, for the next tile)
        if (!(D2_ABL & 1)) {
#pragma unroll
          for (int ks = 0; ks < KPJ; ++ks) {
            const int kk = KPJ*j + ks;
#pragma unroll
            for (int f = 0; f < 2; ++f) {
              const bf16x8 bv = *reinterpret_cast<const bf16x8*>(pmm + 32*f*D2_LDP + kk*16);
#pragma unroll
              for (int sl = 0; sl < NSL; ++sl)
                acc[sl][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfr[j & 1][sl*KPJ + ks], bv, acc[sl][f], 0, 0, 0);
            }
          }
        }
      }
      if (ST) {
        stage(qs, j, ring[j & (D2_AHEAD - 1)], pst);
        const int jn = j + D2_AHEAD;                        // next request of this ring slot
        if (jn < D2_FPW) request(qs, jn, ring[j & (D2_AHEAD - 1)]);
        else request(qn, jn - D2_FPW, ring[j & (D2_AHEAD - 1)]);
      }
#if D2_SCHED
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
    if (MF) {
      // u: accumulators (lane = frame n32 of half f, registers 4 g .. 4 g + 3 = outputs 8 g + 4 khalf .. + 3 of a slice)
      // -> this wave's private LDS patch [32 frames][32 NSL outputs] -> 16 bytes per lane, 64 NSL contiguous bytes
      // per frame. (Straight from the accumulators -- 8-byte pieces, 32 rows per instruction -- the stores took
      // 26 us of a 74 us launch: 4 M partial-line requests, the L2s take ~16 per clock and XCD.)
      const __amdgpu_buffer_rsrc_t ru = make_rsrc(p.u + (long long)qm.b*T*DP_N, (long long)T*DP_N*2);
      constexpr int UROW = 64*NSL + 16;            // bytes per staged frame (padded)
      constexpr int PPR = 4*NSL;                   // 16-byte pieces per frame
      constexpr int FPI = 64/PPR;                  // frames per store instruction
      unsigned char* ust = SEQ ? smem + wid*(32*UROW) : smem + D2_OFF_TAB + 4*DP_H*4 + wid*(32*UROW);
      if (SEQ) __syncthreads();                    // (every wave is done reading p before the patches overwrite it)
      if (!(D2_ABL & 4)) {
#pragma unroll
        for (int f = 0; f < 2; ++f) {
#pragma unroll
          for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              uint2 q;
              q.x = pack2(acc[sl][f][4*g], acc[sl][f][4*g + 1]); q.y = pack2(acc[sl][f][4*g + 2], acc[sl][f][4*g + 3]);
              *reinterpret_cast<uint2*>(ust + n32*UROW + (32*sl + 8*g + 4*khalf)*2) = q;
            }
#pragma unroll
          for (int r = 0; r < 32/FPI; ++r) {
            const int fr = lane/PPR + FPI*r, pc = lane % PPR;
            const uint4 q = *reinterpret_cast<const uint4*>(ust + fr*UROW + pc*16);
            buf_store16(ru, (unsigned int)(qm.t0 + 32*f + fr)*(unsigned int)(DP_N*2)
                            + (unsigned int)(64*NSL*wid + 16*pc), q);
          }
        }
      }
      if (!SEQ) zero_acc();
    }
    if (!(D2_ABL & 32)) __syncthreads();
  };
  using T_ = std::true_type; using F_ = std::false_type;
  if (SEQ) {
#pragma unroll 1
    for (int n = 0; n < n_my; ++n) {
      pass(F_{}, T_{}, 0);
      qm = qs;
      pass(T_{}, F_{}, 0);
      qs = qn; advance(qn, n + 2);
    }
    flush_stats();
    return;
  }
  // tile 0 is staged alone, the last tile multiplied alone; in between both streams run side by side
  pass(F_{}, T_{}, 1);                             // stages into buffer 0
  int par = 0;
#pragma unroll 1
  for (int n = 0; n + 1 < n_my; ++n, par ^= 1) {
    qm = qs; qs = qn; advance(qn, n + 2);
    pass(T_{}, T_{}, par);
  }
  qm = qs;
  pass(T_{}, F_{}, par);
  flush_stats();
}

}  // namespace brv
