// Backward mirror of the fused forward stage (dwpw2_fused.cuh), default widths:
//
//   e   = W_rs^T [g_out | g_skip]                     [res | skip] data gradient (MFMA, K = 256)
//   dz2 = PReLU_2'(z2) rstd_2 (gamma_2 e - m1 - xh_2 m2)   gLN_2 / PReLU_2 backward
//   e1  = gamma_1 dconv^T(dz2)                        transposed dilated depthwise stencil
//
// in ONE kernel: the 512-wide e2 = gamma_2 e tensor that used to cross a launch boundary (written
// by pw2_dgrad, read by the depthwise backward: 3 passes of 65.5 MB per block at the BASELINE size)
// never exists. What made the boundary necessary are the item-wide means m1 = mean(e2),
// m2 = mean(e2 xh_2) of the layer-norm backward; with the fused forward's stored u = (W gamma_2) p
// they are functions of the 256-wide g and u alone,
//   sum e2       = sum_t <g_t, v1>,           v1 = (W gamma_2) 1   (lazy-norm table of the forward)
//   sum e2 xh_2  = rstd_2 (sum_t <g_t, u_t> - mean_2 sum_t <g_t, v1>),
// so the kernel that PRODUCES g accumulates the two dot products (gemm_rows E_ADD epilogue of the
// next block's first-conv data gradient; gu_dots_kernel for the last block) and they are known
// before this kernel starts.
//
// Structure = dwconv_bwd_halo_kernel (comb tiles, tcn_kernels.cuh) with a phase 0 in front:
//   phase 0: e of the window (tile + halo teeth, <= 256 rows x 64 channels) on the matrix pipe:
//            every wave multiplies its own 64 rows with no workgroup barrier: A = W^T fragments
//            (fragment order, straight from L1 / L2 to registers), B = the rows' g, chunk by chunk
//            through the wave's part of the window buffer (whole-line loads), result -> the same
//            LDS window (bf16, 144-byte rows);
//   phase 1: dz2 of the window in place in LDS (+ d gamma_2, d beta_2, PReLU_2 slope, bias grads);
//   phase 2: transposed stencil out of LDS, gLN_1 partial sums, tap / gamma_1 / beta_1 gradients.
// Reference: autograd of brever/models/convtasnet/convtasnet.py:240-260 (Conv1DBlock.forward).
#pragma once
#include "tcn_kernels.cuh"

namespace brv {

#ifndef BF_ABL
#define BF_ABL 0     // ablation bits (diagnostic builds, results wrong): 1 no MFMAs, 2 no g / W loads, 4 no phase 0,
#endif               // 16 no per-channel reductions (accumulators die too), 32 no atomics, 64 no folds but live accumulators

#ifndef BF_AHEAD1
#define BF_AHEAD1 4        // rows of a thread whose z2 loads are in flight together in phase 1 (4 or 8)
#endif
#ifndef BF_AHEAD2
#define BF_AHEAD2 4        // the same for the z1 loads of phase 2
#endif
#ifndef BF_PIPE
#define BF_PIPE 1          // z2 / z1 rows requested across the phase boundaries in two batches of four (0: per phase)
#endif

struct BwdFusedParams {
  DwParams d;              // as dwconv_bwd_halo_kernel; d.dz2 unused, d.sums2 = {sum <g, v1>, sum <g, u>}
  const bf16_t* g;         // [B][T][ldg]: first used column of [g_out | g_skip]
  int ldg, Kg;             // row stride (elements); reduction length (256, or 128 without residual conv)
  const bf16_t* Wp;        // W^T [Hp][Kg] in fragment order, slices of 32 channels (p_rs_bp)
  const float* gamma2;
  float* dgamma2; float* dbeta2;    // replicated like the other per-channel gradients
  int K, R;                // centre teeth per tile, rows per tooth (host: bf_tile_shape)
#ifdef BF_STAMP
  long long* dbg;          // diagnostic build: 8 cycle stamps per workgroup (tools/stamp_bwd.py)
#endif
};

#ifndef BF_ROWS_N
#define BF_ROWS_N 256
#endif
#ifndef BF_WPS
#define BF_WPS 3     // resident workgroups per CU the register budget is set for (diagnostic builds: 4)
#endif
constexpr int BF_ROWS = BF_ROWS_N;         // window rows (teeth x rows per tooth), 8 MFMA groups of 32
constexpr int BF_LDW = HL_CG + 8;          // halves per window row: 144 B
constexpr int BF_LDS = BF_ROWS*BF_LDW*2 + 32*HL_CG*4;   // window + the reduction scratch

// Tile shape (K centre teeth x R rows per tooth, window (K + P - 1) R <= BF_ROWS rows): the pair with the
// fewest tiles per item, teeth balanced over the tooth groups. (Round 6: MORE, smaller tiles -- so that the launch's
// 2 048 workgroups become a whole number of rounds of the 768 resident ones -- measured slower: 224 / 200 / 168 output
// rows per tile 70.9 / 71.3 / 75.5 us against 68.8 at 250.) R need not be a power of two: at dilation
// 128 and T = 3999 (32 teeth) 7 rows x 32 teeth give 19 tiles where 8 x 16 gave 32 -- the per-workgroup
// costs (parameter table, folds, atomics) made that dilation 94 us against 73 us at dilation 1.
inline void bf_tile_shape(int T, int dil, int P, int& K, int& R) {
  const int n_teeth = (T - 1)/dil + 1;
  long long best = -1;
  K = 1; R = 1;
  for (int r = 1; r <= HL_RMAX*2 && r <= dil; ++r) {
    const int kmax = BF_ROWS/r - (P - 1);
    if (kmax < 1) break;
    const int n_qt = ceil_div(n_teeth, kmax);
    const int k = ceil_div(n_teeth, n_qt);
    const long long tiles = (long long)n_qt*ceil_div(dil, r);
    // fewest tiles; among equals the fewest window rows in total (halo), then the longer runs of frames
    const long long cost = tiles*100000 + tiles*(k + P - 1)*r;
    if (best < 0 || cost <= best) { best = cost; K = k; R = r; }
  }
}

// KG: reduction length of phase 0 (compile time: the chunk loop is straight-line code, so every wait of
// its load pipeline is a counted vmcnt -- as a run-time loop hipcc drained ALL loads, the just-issued
// prefetch included, in front of the first MFMA of every chunk)
// (The channel count is a multiple of 64 on this path -- hidden_channels = 512 -- so no channel masks.)
#ifdef BF_STAMP
#define BF_MARK(i) do { long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                        if (threadIdx.x == 0 && fp.dbg) fp.dbg[(long long)blockIdx.x*8 + (i)] = t_; } while (0)
#else
#define BF_MARK(i) do { } while (0)
#endif
// Row -> frame tables (round 6): see the kernel. A row outside the item / the window maps to kRowBad, a frame
// index whose byte offset (x 512 B for g, x 1 KB for z2 / z1 / e1) lies beyond every descriptor of this kernel.
constexpr int kRowBad = 1 << 21;
constexpr int kRowCentre = 1 << 30;
constexpr int kRowFrame = kRowCentre - 1;
inline bool bf_frames_ok(long long T) { return T < kRowBad; }      // host: longer items take the three-launch path

template <int P, int KG>
__global__ __launch_bounds__(256, BF_WPS) void dwconv_bwd_fused_kernel(const BwdFusedParams fp) {
  const DwParams& p = fp.d;
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
  bf16_t* win = reinterpret_cast<bf16_t*>(dyn_lds);
  float* red = reinterpret_cast<float*>(dyn_lds + BF_ROWS*BF_LDW*2);   // 32*HL_CG floats
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int T = p.T, d = p.dil;
  const int R = fp.R, K = fp.K;
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
  const int id = (slot / n_cg)*8 + xcd;
  if (id >= n_tt*p.B) return;                          // (whole workgroup: no barrier is skipped by part of it)
  const int b = id / n_tt;
  const int tile = id % n_tt;
  const int r0 = (tile % n_rt)*R, q0 = (tile / n_rt)*K;
  BF_MARK(0);
  const int cl = (tid & 7)*8;                          // channel offset inside the group
  const int c0 = cg*HL_CG + cl;
  const int rslot = tid >> 3;                          // 32 row slots per pass
  const int W = (K + P - 1)*R;                         // rows of the window (<= BF_ROWS)
  const int KR = K*R;                                  // output rows of the tile
  const int qbase = q0 - (P - 1) + p.left/d;           // tooth of window row 0
  // ---- row -> frame tables. A thread visits 8 window rows and 8 output rows, once to load and once to
  // compute; until round 6 every visit mapped its row to a frame again (a division by R behind a run-time
  // branch on R being a power of two, two range tests, a 64-bit multiply-add for the offset): ~550 of the
  // kernel's 3 165 vector instructions per thread and ~60 branches that cut the stream into small blocks.
  // Thread r maps window row r and output row r ONCE; every later visit is one LDS word (8 lanes of a row
  // slot read the same word: no conflicts):
  //   wtab[r] = frame of window row r: tooth qbase + r / R, residue r0 + r % R (| kRowCentre when the tooth
  //             is one of this tile's own K), or kRowBad for rows outside the item / the window: loads of
  //             such rows are out of range of their descriptors and return zeros by themselves;
  //   otab[i] = frame of output row i: tooth q0 + i / R, same residue; or kRowBad.
  __shared__ int wtab[BF_ROWS];
  __shared__ int otab[BF_ROWS];
  {
    int qi, ri;
    if ((R & (R - 1)) == 0) { ri = tid & (R - 1); qi = tid >> (31 - __builtin_clz(R)); }
    else {     // ((r + 0.5) / R is never within 1e-3 of an integer for r < 512: exact)
      qi = (int)(((float)tid + 0.5f)*(1.f/(float)R)); ri = tid - qi*R;
    }
    const int q = qbase + qi;
    const int tf = q*d + r0 + ri;
    const bool in = tid < W && q >= 0 && r0 + ri < d && tf < T;
    wtab[tid] = in ? (tf | ((q >= q0 && q < q0 + K) ? kRowCentre : 0)) : kRowBad;
    const int to = (q0 + qi)*d + r0 + ri;
    otab[tid] = (tid < KR && r0 + ri < d && to < T) ? to : kRowBad;
  }
  // Per-channel parameters of the group (gamma_2, gamma_1, beta_1, the P taps): requested now, parked
  // in LDS behind the barrier that ends phase 0 and read from there by phases 1 and 2. (Loaded by every
  // thread where they are used -- 8 + 40 dependent scalar loads -- they put two L2 round trips into
  // every workgroup's critical path.)
  __shared__ float ptab[(3 + P)*HL_CG];
  float pv0 = 0.f, pv1 = 0.f;
  auto ptab_src = [&](int idx) -> const float* {
    const int which = idx >> 6, c = cg*HL_CG + (idx & 63);
    return which == 0 ? fp.gamma2 + c : which == 1 ? p.gamma1 + c : which == 2 ? p.beta1 + c
                      : p.taps + (long long)c*P + (which - 3);
  };
  {
    const float* s0 = ptab_src(tid);
    if (s0) pv0 = *s0;
    if (tid + 256 < (3 + P)*HL_CG) { const float* s1 = ptab_src(tid + 256); if (s1) pv1 = *s1; }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // tables visible; LDS only: pv0 / pv1 stay in flight

  // z2 rows of phase 1 and z1 rows of phase 2: requested across the phase boundaries in two batches of four
  // rows per thread (the window has at most 256 rows = 8 per thread)
  const __amdgpu_buffer_rsrc_t rz2 = make_rsrc(p.z2in + (long long)b*T*p.Cp, (long long)T*p.Cp*2);
  const __amdgpu_buffer_rsrc_t rz1 = make_rsrc(p.z1 + (long long)b*T*p.Cp, (long long)T*p.Cp*2);
  const unsigned int row = (unsigned int)(p.Cp*2), coff = (unsigned int)(c0*2);
  // (24-bit multiply: one full-rate v_mad_u32_u24 -- frames < 2^22, row = 1 KB; kRowBad lands out of range)
  auto row_off = [&](int v) { return __umul24((unsigned int)(v & kRowFrame), row) + coff; };
  uint4 qzA[4], qz1A[4];
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
    unsigned int offG[8];                                  // rows 64 wid + 8 i + rsub of the window
#pragma unroll
    for (int i = 0; i < 8; ++i)
      offG[i] = __umul24((unsigned int)(wtab[64*widu + 8*i + rsub] & kRowFrame), (unsigned int)(fp.ldg*2)) + (unsigned int)(oct*16);
    constexpr int nkc = KG >> 6;                           // chunks of 64 k
    const bf16_t* wsrc0 = fp.Wp + (long long)(cg*2)*32*KG + lane*8;
    const bf16_t* wsrc1 = wsrc0 + (long long)32*KG;
    const uint4 z4 = make_uint4(0, 0, 0, 0);
    uint4 gq0 = z4, gq1 = z4, gq2 = z4, gq3 = z4, gq4 = z4, gq5 = z4, gq6 = z4, gq7 = z4;
    auto gl = [&](int i, int kc) { return buf_load16(rg, offG[i] + (unsigned int)(kc*128)); };
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
#pragma unroll
    for (int u = 0; u < 4; ++u) qzA[u] = buf_load16(rz2, row_off(wtab[rslot + 32*u]));
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
  ptab[tid] = pv0;
  if (tid + 256 < (3 + P)*HL_CG) ptab[tid + 256] = pv1;
  __syncthreads();
  BF_MARK(2);

  // ---- phase 1: dz2 of the window, in place in LDS --------------------------------------------------
  // Per-item constants: the variance (a difference of two nearly equal sums) in fp64, its inverse square root
  // in fp32 (rstd_f32: the fp64 square root + division were ~45 half-rate instructions in every thread).
  const double mean2 = p.stats2[stat_sum(b)]*p.inv_n;
  double var2 = p.stats2[stat_sq(b)]*p.inv_n - mean2*mean2;
  if (var2 < 0.0) var2 = 0.0;
  const float rs2 = rstd_f32(var2 + (double)p.eps);
  const float mu2 = (float)mean2;
  const double S1 = p.sums2[stat_sum(b)], S2 = p.sums2[stat_sq(b)];
  const float m1 = (float)(S1*p.inv_n);
  const float m2 = rs2*(float)((S2 - mean2*S1)*p.inv_n);
  const float a2 = *p.slope2;
  // PReLU_2 through the sign s = +-1 of z (one v_and_or per element): |z| = s z, PReLU' = c0 + c1 s, so
  //   xh_2 = rstd (PReLU(z) - mean) = (ya + yb s) z + yc,   dz2 = uu (c0 + c1 s),
  //   slope gradient = sum uu min(z, 0) = 0.5 sum (uu - uu s) z     -- all packed two-channel operations
  // (PReLU'(+0) is 1 here where torch's is the slope: a set of measure zero that carries no gradient mass).
  const float c0s = 0.5f*(1.f + a2), c1s = 0.5f*(1.f - a2);
  const float ya = c0s*rs2, yb = c1s*rs2, yc = -mu2*rs2;
  const float R2 = rs2, K0 = -m1*rs2, M2R = -m2*rs2;
  f32x2 daz = {0.f, 0.f};
  f32x2 dbia[4], dgam2[4], dbet2[4], g2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    g2[j] = f32x2{ptab[cl + 2*j], ptab[cl + 2*j + 1]};
    dbia[j] = f32x2{0.f, 0.f}; dgam2[j] = f32x2{0.f, 0.f}; dbet2[j] = f32x2{0.f, 0.f};
  }
  // one window row of this thread (8 channels): dz2 in place + the per-channel sums
  auto phase1_row = [&](int r, const uint4& qzv) {
      const int v = wtab[r];
      const float on = (v & kRowBad) ? 0.f : 1.f;        // rows outside the item: dz2 = 0
      const bool centre = v >= kRowCentre;                // owned by this tile's teeth [q0, q0 + K)
      const f32x2 rr = {on*R2, on*R2}, k0 = {on*K0, on*K0}, mm = {on*M2R, on*M2R};
      const f32x2 yav = {ya, ya}, ybv = {yb, yb}, ycv = {yc, yc}, c0v = {c0s, c0s}, c1v = {c1s, c1s};
      float e[8], z[8];
      unpack8(*reinterpret_cast<const uint4*>(win + r*BF_LDW + cl), e);
      unpack8(qzv, z);
      f32x2 gg[4], xg[4], ee[4], dd[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x2 zz = {z[2*j], z[2*j + 1]};
        ee[j] = f32x2{e[2*j], e[2*j + 1]};
        const f32x2 sg = {__uint_as_float((__float_as_uint(zz.x) & 0x80000000u) | 0x3f800000u),
                          __uint_as_float((__float_as_uint(zz.y) & 0x80000000u) | 0x3f800000u)};
        xg[j] = (yav + ybv*sg)*zz + ycv;                 // xh_2
        const f32x2 uu = mm*xg[j] + ((ee[j]*g2[j])*rr + k0);
        const f32x2 us = uu*sg;
        gg[j] = c0v*uu + c1v*us;                         // dz2 = PReLU_2' uu
        dd[j] = (uu - us)*zz;                            // 2 uu min(z, 0)
      }
      uint4 q4;
      q4.x = pack2(gg[0].x, gg[0].y); q4.y = pack2(gg[1].x, gg[1].y);
      q4.z = pack2(gg[2].x, gg[2].y); q4.w = pack2(gg[3].x, gg[3].y);
      *reinterpret_cast<uint4*>(win + r*BF_LDW + cl) = q4;
      if (centre) {                                      // (row-uniform per thread: one branch, packed math)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          daz += dd[j];
          dbia[j] += gg[j];                              // bias gradient = sum of dz2 (before its bf16 rounding)
          dgam2[j] += ee[j]*xg[j]; dbet2[j] += ee[j];
        }
      }
  };
  {
    // rows rslot + 32 u: the first four were requested before the barrier, the other four go out now and
    // arrive while the first four are worked on
    uint4 qzB[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) qzB[u] = buf_load16(rz2, row_off(wtab[rslot + 128 + 32*u]));
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int r = rslot + 32*u; if (r < W) phase1_row(r, qzA[u]); }
    // the first z1 rows of phase 2 are requested before the second half and the folds
#pragma unroll
    for (int u = 0; u < 4; ++u) qz1A[u] = buf_load16(rz1, row_off(otab[rslot + 32*u]));
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int r = rslot + 128 + 32*u; if (r < W) phase1_row(r, qzB[u]); }
  }
  float da2 = 0.5f*(daz.x + daz.y);
  // ---- per-channel reductions. A thread holds 8 channels of its row slot; the 8 row slots of a wave are
  // folded with lane shuffles (lanes 8 apart share a channel octet), so LDS only carries ONE row per wave
  // and vector: all vectors of a phase go through it together behind a single barrier pair. (One vector
  // at a time through a [32 slots][64] image, summed by 64 threads: 9 x 2 barriers and 32 dependent LDS
  // reads each -- 20 us of the 100 us launch.) The phase-1 vectors are folded right away so that their
  // registers are free during phase 2.
  // replica of the per-channel gradient block: by TILE, not by workgroup id -- the id also encodes the
  // channel group, so `blockIdx % 64` sent all adds of a channel to 8 of the 64 replicas: 1024 - 2048
  // same-line atomics (~12 ns each, serialised) per line and launch = 24 us (dilation 1) to 44 us (128)
  const int rep_off = id % kReplicas;
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
      float* dst = red + (vec*4 + wid)*HL_CG + fold_ch;
      dst[0] = xa; dst[4] = xb;
    }
  };
  auto vec_sum = [&](int vec, int ch) {
    const float* src = red + vec*4*HL_CG + ch;
    return (src[0] + src[HL_CG]) + (src[2*HL_CG] + src[3*HL_CG]);
  };
  if (BF_ABL & 64) {
    float keep = da2;
#pragma unroll
    for (int j = 0; j < 4; ++j) keep += dbia[j].x*dbia[j].y + dgam2[j].x*dgam2[j].y + dbet2[j].x*dbet2[j].y;
    if (keep == 123.456f) red[tid] = keep;
  }
  if (!(BF_ABL & (16 | 64))) {
    da2 = wave_sum(da2);
    put8(0, dbia); put8(1, dgam2); put8(2, dbet2);
    if (lane == 0) red[3*4*HL_CG + wid] = da2;
    __syncthreads();
    if (tid < 3*HL_CG) {
      const int vec = tid >> 6, ch = tid & 63, c = cg*HL_CG + ch;
      float* dst = (vec == 0 ? p.dbias : vec == 1 ? fp.dgamma2 : fp.dbeta2) + (long long)rep_off*p.rep_stride;
      if (c < p.C && !(BF_ABL & 32)) atomic_add_f32(dst + c, vec_sum(vec, ch));
    } else if (tid == 3*HL_CG && !(BF_ABL & 32)) {
      const float* sc = red + 3*4*HL_CG;
      atomic_add_f32(p.dslope2 + (long long)rep_off*p.rep_stride, (sc[0] + sc[1]) + (sc[2] + sc[3]));
    }
  }
  __syncthreads();                                       // dz2 window complete; `red` free again
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
    const int v = otab[i];
    if (v & kRowBad) return;                             // frames past the end: nothing to store or add
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
      const int r = i + (P - 1 - k)*R;
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
    buf_store16(re1, __umul24((unsigned int)v, row) + coff, pack8v(o));
  };
  {
    uint4 qz1B[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) qz1B[u] = buf_load16(rz1, row_off(otab[rslot + 128 + 32*u]));
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int i = rslot + 32*u; if (i < KR) phase2_row(i, qz1A[u]); }
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int i = rslot + 128 + 32*u; if (i < KR) phase2_row(i, qz1B[u]); }
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
    l1 = wave_sum(l1); l2 = wave_sum(l2);
    put8(0, dgam); put8(1, dbet);
#pragma unroll
    for (int k = 0; k < P; ++k) put8(2 + k, dtap[k]);
    if (lane == 0) { red[(2 + P)*4*HL_CG + wid] = l1; red[(2 + P)*4*HL_CG + 4 + wid] = l2; }
    __syncthreads();
    for (int idx = tid; idx < (2 + P)*HL_CG; idx += 256) {
      const int vec = idx >> 6, ch = idx & 63, c = cg*HL_CG + ch;
      if (c >= p.C || (BF_ABL & 32)) continue;
      const float sum = vec_sum(vec, ch);
      if (vec == 0) atomic_add_f32(p.dgamma1 + (long long)rep_off*p.rep_stride + c, sum);
      else if (vec == 1) atomic_add_f32(p.dbeta1 + (long long)rep_off*p.rep_stride + c, sum);
      else atomic_add_f32(p.dtaps + (long long)rep_off*p.rep_stride + (long long)c*P + (vec - 2), sum);
    }
    if (tid == 255 && !(BF_ABL & 32)) {
      const float* sc = red + (2 + P)*4*HL_CG;
      atomic_add_f64(p.sums1 + stat_sum(b), ((double)sc[0] + (double)sc[1]) + ((double)sc[2] + (double)sc[3]));
      atomic_add_f64(p.sums1 + stat_sq(b), ((double)sc[4] + (double)sc[5]) + ((double)sc[6] + (double)sc[7]));
    }
  }
  BF_MARK(6);
}

// sum_t <g_t, v1> and sum_t <g_t, u_t> over `ncols` columns of one item (the block without a producer
// kernel for its g: the last one, whose g is g_skip alone). out = the item's {stat_sum, stat_sq} slots.
struct GuDotsParams {
  const bf16_t* g; int ldg; const bf16_t* u; int ldu; const float* v1; int ncols; int B, T;
  double* out;
};
__global__ __launch_bounds__(256) void gu_dots_kernel(const GuDotsParams p) {
  __shared__ double dscr[16];
  const int b = blockIdx.y, tid = threadIdx.x;
  const int cpr = p.ncols/8;
  const long long per_item = (long long)p.T*cpr;
  double s1 = 0.0, s2 = 0.0;
  for (long long i = (long long)blockIdx.x*256 + tid; i < per_item; i += (long long)gridDim.x*256) {
    const int c = (int)(i % cpr)*8; const long long t = i / cpr;
    float gv[8], uv[8];
    unpack8(*reinterpret_cast<const uint4*>(p.g + ((long long)b*p.T + t)*p.ldg + c), gv);
    unpack8(*reinterpret_cast<const uint4*>(p.u + ((long long)b*p.T + t)*p.ldu + c), uv);
    float a = 0.f, q = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a = __builtin_fmaf(gv[j], p.v1[c + j], a); q = __builtin_fmaf(gv[j], uv[j], q); }
    s1 += a; s2 += q;
  }
  const double r1 = block_sum(s1, dscr);
  const double r2 = block_sum(s2, dscr + 8);
  if (tid == 0) {
    atomic_add_f64(p.out + stat_sum(b), r1);
    atomic_add_f64(p.out + stat_sq(b), r2);
  }
}

}  // namespace brv
