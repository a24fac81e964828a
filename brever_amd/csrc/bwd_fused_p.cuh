// Persistent form of dwconv_bwd_fused_kernel (bwd_fused.cuh; same phases, same arithmetic per element): a workgroup walks
// NT tiles of its channel group instead of one. Why (round 5): with one tile per workgroup (~2 000 workgroups per launch)
// every tile paid the per-channel parameter loads, two barrier pairs around the cross-wave reductions of its eight
// per-channel gradient vectors and 512 + 8 global atomics -- 4 us of a 75 us launch for the atomics alone, 6.5 us for the
// folds (profiles/r03_fused_bwd_ablation.txt, BF_ABL = 32 / 64). Here the per-wave partial sums of the eight vectors stay
// in the 8 KB reduction scratch for the whole tile range (a lane adds into its own slots: no barrier), the scalar sums in
// registers, and the cross-wave sums and atomics run ONCE per workgroup. The accumulators of a tile still leave their
// registers after each phase (the register budget for three workgroups per CU is unchanged).
#pragma once
#include "bwd_fused.cuh"

namespace brv {

template <int P, int KG>
__global__ __launch_bounds__(256, BF_WPS) void dwconv_bwd_fused_p_kernel(const BwdFusedParams fp, const int NT) {
  const DwParams& p = fp.d;
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
  bf16_t* win = reinterpret_cast<bf16_t*>(dyn_lds);
  float* red = reinterpret_cast<float*>(dyn_lds + BF_ROWS*BF_LDW*2);   // 32*HL_CG floats
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int T = p.T, d = p.dil;
  const int R = fp.R, K = fp.K;
  // r / R and r % R for r < 512: shifts when R is a power of two (dilations <= 32 at the BASELINE length),
  // else a float multiplication ((r + 0.5) / R is never within 1e-3 of an integer: exact)
  const float invR = 1.f/(float)R;
  const bool r_pow2 = (R & (R - 1)) == 0;
  const int lgR = 31 - __builtin_clz(R);
  auto divR = [&](int r, int& rem) {
    if (r_pow2) { rem = r & (R - 1); return r >> lgR; }
    const int q = (int)(((float)r + 0.5f)*invR);
    rem = r - q*R;
    return q;
  };
  const int n_rt = ceil_div(d, R);                       // residue groups
  const int n_teeth = (T - 1)/d + 1;
  const int n_qt = ceil_div(n_teeth, K);                 // tooth groups
  const int n_tt = n_rt*n_qt, n_cg = p.Cp/HL_CG;
  // Workgroup -> (tile, channel group): the channel groups of one tile all multiply the SAME g rows,
  // so they must share an L2: ids congruent mod 8 run on one XCD, hence XCD x takes tile 8 j + x of
  // every run of 8 tiles and the tile's channel groups sit in consecutive slots of that XCD. (With the
  // channel group fastest in the id every XCD fetched every g row: 8 x 32.8 MB per launch, the kernel
  // ran at 108 us whatever phase 0 looked like.) The grid is padded to whole runs of 8 tiles.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int cg = slot % n_cg;
  // PERSISTENT form: this workgroup walks the NT tiles id = ((slot / n_cg) NT + k) 8 + xcd, k < NT, of its channel
  // group (the same XCD and its L2 as before; consecutive k are 8 tiles apart, nearly always inside one item)
  const int id0 = (slot / n_cg)*NT*8 + xcd;
  if (id0 >= n_tt*p.B) return;                         // (whole workgroup)
  __shared__ float ptab[(3 + P)*HL_CG];
  {
    // per-channel parameters of the group: once per workgroup (gamma_2, gamma_1, beta_1, the P taps)
    for (int idx = tid; idx < (3 + P)*HL_CG; idx += 256) {
      const int which = idx >> 6, c = cg*HL_CG + (idx & 63);
      ptab[idx] = which == 0 ? fp.gamma2[c] : which == 1 ? p.gamma1[c] : which == 2 ? p.beta1[c]
                             : p.taps[(long long)c*P + (which - 3)];
    }
    // the per-wave partial sums of the eight per-channel vectors live in `red` for the whole tile range
    for (int idx = tid; idx < 32*HL_CG; idx += 256) red[idx] = 0.f;
  }
  const int rep_off = id0 % kReplicas;
  float da2_run = 0.f, l1_run = 0.f, l2_run = 0.f;     // per lane, folded once (da2) / per item (l1, l2)
  int sum_item = -1;
  auto flush_item = [&]() {
    if (sum_item < 0) return;
    const float s1 = wave_sum(l1_run), s2 = wave_sum(l2_run);
    if (lane == 0 && !(BF_ABL & 32)) {
      atomic_add_f64(p.sums1 + stat_sum(sum_item), (double)s1);
      atomic_add_f64(p.sums1 + stat_sq(sum_item), (double)s2);
    }
    l1_run = 0.f; l2_run = 0.f;
  };
#pragma unroll 1
  for (int kt = 0; kt < NT; ++kt) {
  const int id = id0 + kt*8;
  if (id >= n_tt*p.B) break;
  // (opaque thread index per tile: hoisted out of the tile loop, the lane-dependent address arithmetic of the three
  // phases stayed live across it and spilled 109 registers at the three-workgroups-per-CU budget)
  int tv_ = threadIdx.x;
  asm volatile("" : "+v"(tv_));
  const int tid = tv_, lane = tid & 63, wid = tid >> 6;
  const int cl = (tid & 7)*8;                          // channel offset inside the group
  const int c0 = cg*HL_CG + cl;
  const int rslot = tid >> 3;                          // 32 row slots per pass
  const int b = id / n_tt;
  const int tile = id % n_tt;
  const int r0 = (tile % n_rt)*R, q0 = (tile / n_rt)*K;
  if (b != sum_item) { flush_item(); sum_item = b; }
  if (kt > 0) __syncthreads();                           // the previous tile's phase 2 has read the window
  BF_MARK(0);
  const int W = (K + P - 1)*R;                         // rows of the window (<= BF_ROWS)
  const int KR = K*R;                                  // output rows of the tile
  const int qbase = q0 - (P - 1) + p.left/d;           // tooth of window row 0
  // window row r -> frame: tooth qbase + r / R, residue r0 + r % R
  auto frame_of = [&](int r, bool& ok) {
    int ri;
    const int qi = divR(r, ri);
    const int q = qbase + qi;
    ok = r < W && q >= 0 && r0 + ri < d;
    return q*d + r0 + ri;
  };

  // z2 rows of phase 1 (BF_PIPE: requested across the phase boundaries, two batches of four rows per thread;
  // the window has at most 256 rows = 8 per thread)
  const __amdgpu_buffer_rsrc_t rz2 = make_rsrc(p.z2in + (long long)b*T*p.Cp, (long long)T*p.Cp*2);
  const unsigned int row = (unsigned int)(p.Cp*2), coff = (unsigned int)(c0*2);
  uint4 qzA[4];
  // z1 rows of phase 2 (centre rows of the tile), first batch requested inside phase 1
  const __amdgpu_buffer_rsrc_t rz1 = make_rsrc(p.z1 + (long long)b*T*p.Cp, (long long)T*p.Cp*2);
  uint4 qz1A[4];
  // output row i -> frame: tooth q0 + i / R, residue r0 + i % R
  auto out_frame = [&](int i, bool& ok) {
    int ri;
    const int qi = divR(i, ri);
    const int ro = r0 + ri;
    ok = i < KR && ro < d;
    return (q0 + qi)*d + ro;
  };
  // ---- phase 0: e = W^T g of the window -> LDS ---------------------------------------------------
  // Every wave works alone on its own 64 window rows (no workgroup barrier before the end of the
  // phase): the reduction runs in chunks of 64 g columns; a chunk of the wave's rows is fetched by
  // whole-line loads (8 rows x 128 B per instruction, the next chunk's while this one multiplies),
  // parked in the wave's part of the (not yet used) window buffer and read back as B fragments; the
  // A fragments (W^T in fragment order: 1 KB per wave-load, L1 / L2 hits -- every workgroup of the
  // channel group reads the same 32 KB) go straight to registers, refilled one chunk ahead.
  // Measured steps (us per launch, BASELINE size): B fragments straight from global memory (32 rows
  // x 32 B per instruction, every line four times through the L1) 119; g and W chunks through LDS
  // with two workgroup barriers per chunk 108.
  {
    const int n32 = lane & 31, h = lane >> 5;
    const int oct = lane & 7, rsub = lane >> 3;
    const int widu = __builtin_amdgcn_readfirstlane(wid);          // scalar: plain branches, not exec masks
    const int ngrp = (W + 31) >> 5;
    const bool act0 = 2*widu < ngrp, act1 = 2*widu + 1 < ngrp;
    const __amdgpu_buffer_rsrc_t rg =
        make_rsrc(fp.g + (long long)b*T*fp.ldg, ((long long)(T - 1)*fp.ldg + KG)*2);
    unsigned int offG[8];                                  // rows 64 wid + 8 i + rsub of the window (kOob: none)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      bool ok;
      const int tf = frame_of(64*widu + 8*i + rsub, ok);
      offG[i] = (ok && tf < T) ? (unsigned int)tf*(unsigned int)(fp.ldg*2) + (unsigned int)(oct*16) : kOob;
    }
    constexpr int nkc = KG >> 6;                           // chunks of 64 k
    const bf16_t* wsrc0 = fp.Wp + (long long)(cg*2)*32*KG + lane*8;
    const bf16_t* wsrc1 = wsrc0 + (long long)32*KG;
    const uint4 z4 = make_uint4(0, 0, 0, 0);
    uint4 gq0 = z4, gq1 = z4, gq2 = z4, gq3 = z4, gq4 = z4, gq5 = z4, gq6 = z4, gq7 = z4;
    auto gl = [&](int i, int kc) {
      return buf_load16(rg, offG[i] == kOob ? kOob : offG[i] + (unsigned int)(kc*128));
    };
    auto gload = [&](int kc) {
      gq0 = gl(0, kc); gq1 = gl(1, kc); gq2 = gl(2, kc); gq3 = gl(3, kc);
      gq4 = gl(4, kc); gq5 = gl(5, kc); gq6 = gl(6, kc); gq7 = gl(7, kc);
    };
    bf16_t* gp = win + (64*widu + rsub)*BF_LDW + oct*8;
    auto gstore = [&]() {
      *reinterpret_cast<uint4*>(gp) = gq0;
      *reinterpret_cast<uint4*>(gp + 8*BF_LDW) = gq1;
      *reinterpret_cast<uint4*>(gp + 16*BF_LDW) = gq2;
      *reinterpret_cast<uint4*>(gp + 24*BF_LDW) = gq3;
      *reinterpret_cast<uint4*>(gp + 32*BF_LDW) = gq4;
      *reinterpret_cast<uint4*>(gp + 40*BF_LDW) = gq5;
      *reinterpret_cast<uint4*>(gp + 48*BF_LDW) = gq6;
      *reinterpret_cast<uint4*>(gp + 56*BF_LDW) = gq7;
    };
    // A fragments of k-step s of chunk kc (two 32-channel slices)
    bf16x8 a0_0, a0_1, a0_2, a0_3, a1_0, a1_1, a1_2, a1_3;
    auto al = [&](const bf16_t* w, int kc, int s) {
      return *reinterpret_cast<const bf16x8*>(w + kc*2048 + s*512);
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int ms = 0; ms < 2; ++ms)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[q][ms][i] = 0.f;
    const bf16_t* brow = win + (64*widu + n32)*BF_LDW + h*8;
    auto step = [&](int s, const bf16x8& a0, const bf16x8& a1) {
      if (BF_ABL & 1) return;
      if (act0) {
        const bf16x8 bv = *reinterpret_cast<const bf16x8*>(brow + s*16);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bv, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bv, acc[0][1], 0, 0, 0);
      }
      if (act1) {
        const bf16x8 bv = *reinterpret_cast<const bf16x8*>(brow + 32*BF_LDW + s*16);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bv, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bv, acc[1][1], 0, 0, 0);
      }
    };
    if (act0 && !(BF_ABL & 4)) {
      if (!(BF_ABL & 2)) {
        gload(0);
        a0_0 = al(wsrc0, 0, 0); a1_0 = al(wsrc1, 0, 0); a0_1 = al(wsrc0, 0, 1); a1_1 = al(wsrc1, 0, 1);
        a0_2 = al(wsrc0, 0, 2); a1_2 = al(wsrc1, 0, 2); a0_3 = al(wsrc0, 0, 3); a1_3 = al(wsrc1, 0, 3);
      }
#pragma unroll
      for (int kc = 0; kc < nkc; ++kc) {
        gstore();                          // own rows: ordered behind this wave's reads of the last chunk
        const bool more = kc + 1 < nkc && !(BF_ABL & 2);
        if (more) gload(kc + 1);
        step(0, a0_0, a1_0);
        if (more) { a0_0 = al(wsrc0, kc + 1, 0); a1_0 = al(wsrc1, kc + 1, 0); }
        step(1, a0_1, a1_1);
        if (more) { a0_1 = al(wsrc0, kc + 1, 1); a1_1 = al(wsrc1, kc + 1, 1); }
        step(2, a0_2, a1_2);
        if (more) { a0_2 = al(wsrc0, kc + 1, 2); a1_2 = al(wsrc1, kc + 1, 2); }
        step(3, a0_3, a1_3);
        if (more) { a0_3 = al(wsrc0, kc + 1, 3); a1_3 = al(wsrc1, kc + 1, 3); }
      }
    }
    // the first four z2 rows of this thread (phase 1) are requested now: they arrive behind the barrier
    if (BF_PIPE) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        bool ok;
        const int tf = frame_of(rslot + 32*u, ok);
        qzA[u] = buf_load16(rz2, (ok && tf < T) ? (unsigned int)tf*row + coff : kOob);
      }
    }
    // D[channel][frame]: lane = frame n32, registers = channels 8 (i >> 2) + 4 h + (i & 3)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (!(q == 0 ? act0 : act1)) continue;
      const int wr = 64*widu + 32*q + n32;
#pragma unroll
      for (int ms = 0; ms < 2; ++ms)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          uint2 v;
          v.x = pack2(acc[q][ms][4*gq], acc[q][ms][4*gq + 1]);
          v.y = pack2(acc[q][ms][4*gq + 2], acc[q][ms][4*gq + 3]);
          *reinterpret_cast<uint2*>(win + wr*BF_LDW + ms*32 + 8*gq + 4*h) = v;
        }
    }
  }
  BF_MARK(1);
  __syncthreads();                                       // (window rows of phase 0 complete; first tile: ptab / red too)
  BF_MARK(2);

  // ---- phase 1: dz2 of the window, in place in LDS --------------------------------------------------
  const double mean2 = p.stats2[stat_sum(b)]*p.inv_n;
  double var2 = p.stats2[stat_sq(b)]*p.inv_n - mean2*mean2;
  if (var2 < 0.0) var2 = 0.0;
  const double rstd2d = 1.0/sqrt(var2 + (double)p.eps);
  const float mu2 = (float)mean2, rs2 = (float)rstd2d;
  const double S1 = p.sums2[stat_sum(b)], S2 = p.sums2[stat_sq(b)];
  const float m1 = (float)(S1*p.inv_n);
  const float m2 = (float)(rstd2d*(S2 - mean2*S1)*p.inv_n);
  const float a2 = *p.slope2;
  const float ya = 0.5f*(1.f + a2)*rs2, yb = 0.5f*(1.f - a2)*rs2, yc = -mu2*rs2;
  const float R2 = rs2, K0 = -m1*rs2, M2R = -m2*rs2;
  float da2 = 0.f;
  f32x2 dbia[4], dgam2[4], dbet2[4], g2[4];
  {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      g2[j] = f32x2{ptab[cl + 2*j], ptab[cl + 2*j + 1]};
      dbia[j] = f32x2{0.f, 0.f}; dgam2[j] = f32x2{0.f, 0.f}; dbet2[j] = f32x2{0.f, 0.f};
    }
  }
  // one window row of this thread (8 channels): dz2 in place + the per-channel sums
  auto phase1_row = [&](int r, const uint4& qzv) {
      bool ok;
      const int tf = frame_of(r, ok);
      const bool in = ok && tf < T;
      float e[8], z[8], g[8];
      unpack8(*reinterpret_cast<const uint4*>(win + r*BF_LDW + cl), e);
      unpack8(qzv, z);
      const float on = in ? 1.f : 0.f;
      const float rr = on*R2, k0 = on*K0, mm = on*M2R;
      // each element is "owned" by the tile whose teeth [q0, q0 + K) contain it
      int rem_;
      const int q = qbase + divR(r, rem_);
      const bool centre = in && q >= q0 && q < q0 + K;
      float uz = 0.f;                                    // sum over the row's 8 channels of uu min(z, 0)
      f32x2 xg[4], ee[4];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xh2 = __builtin_fmaf(yb, __builtin_fabsf(z[j]), __builtin_fmaf(ya, z[j], yc));
        const float gj = (j & 1) ? g2[j >> 1].y : g2[j >> 1].x;
        const float uu = __builtin_fmaf(mm, xh2, __builtin_fmaf(e[j]*gj, rr, k0));
        g[j] = uu*(z[j] > 0.f ? 1.f : a2);               // PReLU_2'
        uz = __builtin_fmaf(uu, z[j] - __builtin_fabsf(z[j]), uz);    // 2 min(z, 0) (fminf canonicalises first)
        if (j & 1) { xg[j >> 1].y = xh2; ee[j >> 1].y = e[j]; } else { xg[j >> 1].x = xh2; ee[j >> 1].x = e[j]; }
      }
      const uint4 q4 = pack8(g);
      *reinterpret_cast<uint4*>(win + r*BF_LDW + cl) = q4;
      if (centre) {                                      // (row-uniform per thread: one branch, packed math)
        da2 = __builtin_fmaf(0.5f, uz, da2);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          dbia[j] += f32x2{g[2*j], g[2*j + 1]};          // bias gradient = sum of dz2 (before its bf16 rounding)
          dgam2[j] += ee[j]*xg[j]; dbet2[j] += ee[j];
        }
      }
  };
  if (BF_PIPE) {
    // rows rslot + 32 u: the first four were requested before the barrier, the other four go out now and
    // arrive while the first four are worked on
    uint4 qzB[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      bool ok;
      const int tf = frame_of(rslot + 128 + 32*u, ok);
      qzB[u] = buf_load16(rz2, (ok && tf < T) ? (unsigned int)tf*row + coff : kOob);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int r = rslot + 32*u; if (r < W) phase1_row(r, qzA[u]); }
    // the first z1 rows of phase 2 are requested before the second half and the folds
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      bool ok;
      const int t = out_frame(rslot + 32*u, ok);
      qz1A[u] = buf_load16(rz1, (ok && t < T) ? (unsigned int)t*row + coff : kOob);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int r = rslot + 128 + 32*u; if (r < W) phase1_row(r, qzB[u]); }
  } else {
  // BF_AHEAD1 rows of a thread requested before the first is consumed
  constexpr int NU = BF_AHEAD1;
  for (int rw = rslot; rw < W; rw += 32*NU) {
    uint4 qz[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      bool ok;
      const int tf = frame_of(rw + 32*u, ok);
      qz[u] = buf_load16(rz2, (ok && tf < T) ? (unsigned int)tf*row + coff : kOob);
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int r = rw + 32*u;
      if (r >= W) break;
      phase1_row(r, qz[u]);
    }
  }
  }
  // ---- per-channel reductions. A thread holds 8 channels of its row slot; the 8 row slots of a wave are
  // folded with lane shuffles (lanes 8 apart share a channel octet), so LDS only carries ONE row per wave
  // and vector: all vectors of a phase go through it together behind a single barrier pair. (One vector
  // at a time through a [32 slots][64] image, summed by 64 threads: 9 x 2 barriers and 32 dependent LDS
  // reads each -- 20 us of the 100 us launch.) The phase-1 vectors are folded right away so that their
  // registers are free during phase 2.
  // replica of the per-channel gradient block: by TILE, not by workgroup id -- the id also encodes the
  // channel group, so `blockIdx % 64` sent all adds of a channel to 8 of the 64 replicas: 1024 - 2048
  // same-line atomics (~12 ns each, serialised) per line and launch = 24 us (dilation 1) to 44 us (128)
  // fold8: the wave's sum over its 8 row slots of the 8 channels a lane holds, on the VALU alone. A swap
  // of lane halves (rows) between TWO registers followed by one add reduces both at once, each result
  // living in one half (row pair): 8 values -> 4 (v_permlane32_swap) -> 2 (v_permlane16_swap) -> the
  // DPP rotate by 8 inside a 16-lane row. Afterwards lane L (row r = L >> 4, octet L & 7) holds in `xa`
  // channel kA[r] of its octet and in `xb` channel 4 + kA[r], kA = {0, 2, 1, 3}. (As 3 ds_bpermute per
  // value -- 210 LDS-pipe instructions per wave -- the folds cost 9 to 19 us of the launch.)
  auto swap32 = [](float& a, float& b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
  };
  auto swap16 = [](float& a, float& b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
  };
  auto ror8_add = [](float x) {
    return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x128, 0xf, 0xf, false));
  };
  auto fold8 = [&](const f32x2 (&v)[4], float& xa, float& xb) {
    float p[4], q[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { p[j] = v[j].x; q[j] = v[j].y; swap32(p[j], q[j]); p[j] += q[j]; }
    // p[j]: lanes 0-31 channel 2j, lanes 32-63 channel 2j + 1 (summed over lane bit 5)
    swap16(p[0], p[1]); xa = p[0] + p[1];          // rows: channels 0, 2, 1, 3
    swap16(p[2], p[3]); xb = p[2] + p[3];          // rows: channels 4, 6, 5, 7
    xa = ror8_add(xa); xb = ror8_add(xb);
  };
  const int fold_ch = (lane & 7)*8 + (((lane >> 4) & 1)*2 + (lane >> 5));   // octet * 8 + kA[row]
  auto put8 = [&](int vec, const f32x2 (&v)[4]) {
    float xa, xb;
    fold8(v, xa, xb);
    if (!(lane & 8)) {
      float* dst = red + (vec*4 + wid)*HL_CG + fold_ch;        // (this lane's own slots, every tile: a plain add)
      dst[0] += xa; dst[4] += xb;
    }
  };
  if (BF_ABL & 64) {
    float keep = da2;
#pragma unroll
    for (int j = 0; j < 4; ++j) keep += dbia[j].x*dbia[j].y + dgam2[j].x*dgam2[j].y + dbet2[j].x*dbet2[j].y;
    if (keep == 123.456f) red[tid] = keep;
  }
  if (!(BF_ABL & (16 | 64))) {
    da2_run += da2;
    put8(0, dbia); put8(1, dgam2); put8(2, dbet2);
  }
  __syncthreads();                                       // dz2 window complete
  BF_MARK(4);

  // ---- phase 2: transposed stencil out of LDS -------------------------------------------------
  const NormStat ns = norm_stat(p.stats1, b, p.inv_n, p.eps);
  const float a1 = *p.slope1;
  const float xa = 0.5f*(1.f + a1)*ns.rstd, xb = 0.5f*(1.f - a1)*ns.rstd, xc = -ns.mean*ns.rstd;
  f32x2 gm[4], be[4], w[P][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    gm[j] = f32x2{ptab[HL_CG + cl + 2*j], ptab[HL_CG + cl + 2*j + 1]};
    be[j] = f32x2{ptab[2*HL_CG + cl + 2*j], ptab[2*HL_CG + cl + 2*j + 1]};
#pragma unroll
    for (int k = 0; k < P; ++k)
      w[k][j] = f32x2{ptab[(3 + k)*HL_CG + cl + 2*j], ptab[(3 + k)*HL_CG + cl + 2*j + 1]};
  }
  f32x2 dgam[4], dbet[4], dtap[P][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    dgam[j] = f32x2{0.f, 0.f}; dbet[j] = f32x2{0.f, 0.f};
#pragma unroll
    for (int k = 0; k < P; ++k) dtap[k][j] = f32x2{0.f, 0.f};
  }
  const __amdgpu_buffer_rsrc_t re1 = make_rsrc(p.e1 + (long long)b*T*p.Cp, (long long)T*p.Cp*2);
  float l1 = 0.f, l2 = 0.f;
  // one centre row of this thread (8 channels): transposed stencil out of the window + the per-channel sums
  auto phase2_row = [&](int i, const uint4& qz) {
    bool ok;
    const int t = out_frame(i, ok);
    if (!(ok && t < T)) return;                          // frames past the end: nothing to store or add
    float zc[8];
    unpack8(qz, zc);
    f32x2 xh[4], hn[4], dh[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      xh[j].x = __builtin_fmaf(xb, __builtin_fabsf(zc[2*j]), __builtin_fmaf(xa, zc[2*j], xc));
      xh[j].y = __builtin_fmaf(xb, __builtin_fabsf(zc[2*j + 1]), __builtin_fmaf(xa, zc[2*j + 1], xc));
      hn[j] = gm[j]*xh[j] + be[j];                       // gLN_1 output at frame t
      dh[j] = f32x2{0.f, 0.f};
    }
#pragma unroll
    for (int k = 0; k < P; ++k) {
      // output frame that reads frame t through tap k: window tooth (i / R) + P - 1 - k, same residue
      const int r = i + (P - 1 - k)*R;                 // same residue, tooth + P - 1 - k
      float g[8];
      unpack8(*reinterpret_cast<const uint4*>(win + r*BF_LDW + cl), g);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x2 gk = {g[2*j], g[2*j + 1]};
        dh[j] += w[k][j]*gk;
        dtap[k][j] += gk*hn[j];
      }
    }
    f32x2 o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x2 dl = dh[j];
      o[j] = gm[j]*dl;
      dgam[j] += dl*xh[j]; dbet[j] += dl;
    }
    buf_store16(re1, (unsigned int)t*row + coff, pack8v(o));
  };
  if (BF_PIPE) {
    uint4 qz1B[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      bool ok;
      const int t = out_frame(rslot + 128 + 32*u, ok);
      qz1B[u] = buf_load16(rz1, (ok && t < T) ? (unsigned int)t*row + coff : kOob);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int i = rslot + 32*u; if (i < KR) phase2_row(i, qz1A[u]); }
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int i = rslot + 128 + 32*u; if (i < KR) phase2_row(i, qz1B[u]); }
  } else {
  constexpr int NU2 = BF_AHEAD2;
  for (int i0 = rslot; i0 < KR; i0 += 32*NU2) {
   uint4 qz4[NU2];
#pragma unroll
   for (int u = 0; u < NU2; ++u) {
     bool ok;
     const int t = out_frame(i0 + 32*u, ok);
     qz4[u] = buf_load16(rz1, (ok && t < T) ? (unsigned int)t*row + coff : kOob);   // outside: zeros
   }
#pragma unroll
   for (int u = 0; u < NU2; ++u) {
    const int i = i0 + 32*u;
    if (i >= KR) break;
    phase2_row(i, qz4[u]);
   }
  }
  }
  BF_MARK(5);
  // the tile's sums of e1 and e1 xh_1 (layer-norm backward means of the first norm) from the per-channel
  // partials: sum_t gamma_1 dl = gamma_1 sum_t dl -- a tile lies inside one item, so no per-element adds
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    l1 += gm[j].x*dbet[j].x + gm[j].y*dbet[j].y;
    l2 += gm[j].x*dgam[j].x + gm[j].y*dgam[j].y;
  }

  if (BF_ABL & 64) {
    float keep = l1 + l2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      keep += dgam[j].x*dgam[j].y + dbet[j].x*dbet[j].y;
#pragma unroll
      for (int k = 0; k < P; ++k) keep += dtap[k][j].x*dtap[k][j].y;
    }
    if (keep == 123.456f) red[tid] = keep;
  }
  if (!(BF_ABL & (16 | 64))) {
    l1_run += l1; l2_run += l2;
    put8(3, dgam); put8(4, dbet);
#pragma unroll
    for (int k = 0; k < P; ++k) put8(5 + k, dtap[k]);
  }
  BF_MARK(6);
  }   // tiles of this workgroup
  flush_item();
  // ---- once per workgroup: the four waves' partial sums -> the replicated per-channel gradients ------------------
  {
    const float s = wave_sum(da2_run);
    if (lane == 0) ptab[wid] = s;                        // (ptab is no longer read: every wave is past its last phase 2)
  }
  __syncthreads();
  if (!(BF_ABL & 32)) {
    auto vec_sum = [&](int vec, int ch) {
      const float* src = red + vec*4*HL_CG + ch;
      return (src[0] + src[HL_CG]) + (src[2*HL_CG] + src[3*HL_CG]);
    };
    const long long ro = (long long)rep_off*p.rep_stride;
    for (int idx = tid; idx < (5 + P)*HL_CG; idx += 256) {
      const int vec = idx >> 6, ch = idx & 63, c = cg*HL_CG + ch;
      if (c >= p.C) continue;
      const float sum = vec_sum(vec, ch);
      float* dst = vec == 0 ? p.dbias + ro + c : vec == 1 ? fp.dgamma2 + ro + c : vec == 2 ? fp.dbeta2 + ro + c
                 : vec == 3 ? p.dgamma1 + ro + c : vec == 4 ? p.dbeta1 + ro + c
                 : p.dtaps + ro + (long long)c*P + (vec - 5);
      atomic_add_f32(dst, sum);
    }
    if (tid == 255) atomic_add_f32(p.dslope2 + ro, (ptab[0] + ptab[1]) + (ptab[2] + ptab[3]));
  }
}


}  // namespace brv
