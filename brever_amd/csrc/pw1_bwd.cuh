// Backward of the first 1x1 convolution of a TCN block WITHOUT the stored pre-activation:
//
//   z1  = W1 x + b1                 recomputed on the matrix pipe from the 128-wide block input x
//   dz1 = PReLU_1'(z1) rstd_1 (e1 - m1 - xh_1 m2)        gLN_1 / PReLU_1 backward (e1 from dwpw2_bwd)
//   g_x = W1^T dz1 (+ residual path)                      pw1_dgrad_rc_kernel
//   dW1 = dz1^T x, db1 = sum_t dz1                        pw1_wgrad_rc_kernel
//
// Round 3 read z1 (65.5 MB per block at the BASELINE size) in the data-gradient kernel, wrote dz1 back
// (65.5 MB) and read it again in the weight-gradient kernel. Here dz1 never exists in HBM: both kernels
// rebuild z1 = W1 x + b1 with the SAME instruction sequence as pw1_fwd (gemm_ws.cuh: accumulator
// initialised with the bias, k-steps in order, W as the A operand), so the recomputed bf16 values equal
// the stored ones bit for bit, and rebuild dz1 from e1: e1 is read twice instead of
// {z1 read, dz1 written, dz1 read}: 278 -> 164 MB per block for the pair.
// Reference: autograd of brever/models/convtasnet/convtasnet.py:241-243 (conv -> PReLU -> norm).
#pragma once
#include "gemm_rows.cuh"
#include "gemm_wgrad.cuh"

namespace brv {

// ------------------------------------------------------------------------------------------------
// Data gradient. One workgroup (4 waves) = 128 frames of one item x all 512 hidden channels in 8
// chunks of 64; wave w owns frames [32 w, 32 w + 32) for BOTH products, so the z1 / dz1 chunk lives in
// wave-private LDS rows (no workgroup barrier between the recompute, the norm backward and the data-
// gradient MFMAs of a chunk; the two barriers per chunk guard the shared W1^T chunk only).
struct Pw1DgradRcParams {
  const bf16_t* e1; int lde; long long bse;      // (B, T, 512): gamma_1 dconv^T(dz2), from dwpw2_bwd
  const bf16_t* x; int ldx; long long bsx;       // (B, T, 128): block input
  const bf16_t* Wfp;                             // W1 [512][128] in fragment order, slices of 64 rows (p_c1_fp)
  const float* bias;                             // b1 [C]
  const bf16_t* Wb;                              // W1^T [128][512] plain (p_c1_b)
  const double* stats; const double* sums; const float* slope; double inv_n; float eps; int C;
  int T, n_ttiles, batch;
  EpiSpec e;                                     // E_ADD fields (out, add_in, out2, gu_*, dslope, replicas)
};

constexpr int RC_BM = 128, RC_BK = 64, RC_LDK = RC_BK + 8, RC_N = 128, RC_H = 512, RC_KX = 128;
constexpr int RC_LDC = RC_N + 4;
constexpr int RC_SMEM_MAIN = 2*RC_BM*RC_LDK*2;
constexpr int RC_SMEM_EPI = RC_BM*RC_LDC*4;
constexpr int RC_SMEM = RC_SMEM_MAIN > RC_SMEM_EPI ? RC_SMEM_MAIN : RC_SMEM_EPI;

__global__ __launch_bounds__(256, 2) void pw1_dgrad_rc_kernel(const Pw1DgradRcParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[RC_SMEM];
  __shared__ __attribute__((aligned(16))) float bias_s[RC_H];
  bf16_t* As = reinterpret_cast<bf16_t*>(smem);
  bf16_t* Ws = As + RC_BM*RC_LDK;
  float* Cs = reinterpret_cast<float*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int vid = xcd_remap(blockIdx.x, gridDim.x);
  const int b = vid / p.n_ttiles;
  const int t0 = (vid % p.n_ttiles)*RC_BM;
  const int T = p.T;

  for (int c = tid; c < RC_H; c += 256) bias_s[c] = c < p.C ? p.bias[c] : 0.f;

  const NormStat ns = norm_stat(p.stats, b, p.inv_n, p.eps);
  const float slope = *p.slope;
  const float dz_m1 = (float)(p.sums[stat_sum(b)]*p.inv_n);
  const float dz_m2 = (float)(p.sums[stat_sq(b)]*p.inv_n);
  float dz_da = 0.f;

  // x fragments of the wave's 32 frames: the B operand of the recompute for all 8 chunks (frames past
  // the end of the item read zeros)
  bf16x8 xf[RC_KX/16];
  {
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x + (long long)b*p.bsx, (long long)T*p.ldx*2);
    const unsigned int off = (unsigned int)(t0 + 32*wid + fr)*(unsigned int)(p.ldx*2) + (unsigned int)(fh*16);
#pragma unroll
    for (int s = 0; s < RC_KX/16; ++s) {
      const uint4 q = buf_load16(rx, off + (unsigned int)(s*32));
      xf[s] = __builtin_bit_cast(bf16x8, q);
    }
  }

  // staging geometry: a wave stages ITS OWN 32 rows (8 rows x 128 B per load instruction)
  const int sub = lane >> 3, kc = lane & 7;
  const __amdgpu_buffer_rsrc_t re = make_rsrc(p.e1 + (long long)b*p.bse, (long long)T*p.lde*2);
  uint4 eraw[4], wraw[4];
  auto load_chunk = [&](int kt) {
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
      const unsigned int t = (unsigned int)(t0 + 32*wid + sub + 8*ci);
      eraw[ci] = buf_load16(re, t*(unsigned int)(p.lde*2) + (unsigned int)((kt*RC_BK + kc*8)*2));
    }
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
      const int n = (tid + 256*ci) >> 3;
      wraw[ci] = *reinterpret_cast<const uint4*>(p.Wb + (long long)n*RC_H + kt*RC_BK + kc*8);
    }
  };
  bf16x8 wfa[8], wfb[8];
  auto wf_load = [&](int kt, int f, bf16x8 (&w)[8]) {
    const bf16_t* src = p.Wfp + (long long)kt*RC_BK*RC_KX + ((long long)(f*8)*64 + lane)*8;
#pragma unroll
    for (int s = 0; s < 8; ++s) w[s] = *reinterpret_cast<const bf16x8*>(src + s*512);
  };
  // z1 tile f of chunk kt for the wave's frames -> As (bf16): D[channel][frame], lane = frame fr,
  // registers = channels 8 (i >> 2) + 4 fh + (i & 3); accumulator initialised with the bias (as pw1_fwd)
  auto recompute = [&](int kt, int f, const bf16x8 (&w)[8]) {
    f32x16 az;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 bv = *reinterpret_cast<const float4*>(bias_s + kt*RC_BK + 32*f + 8*g + 4*fh);
      az[4*g] = bv.x; az[4*g + 1] = bv.y; az[4*g + 2] = bv.z; az[4*g + 3] = bv.w;
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) az = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[s], xf[s], az, 0, 0, 0);
    bf16_t* dst = As + (32*wid + fr)*RC_LDK + 32*f + 4*fh;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      uint2 v;
      v.x = pack2(az[4*g], az[4*g + 1]); v.y = pack2(az[4*g + 2], az[4*g + 3]);
      *reinterpret_cast<uint2*>(dst + 8*g) = v;
    }
  };

  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;

  load_chunk(0);
  wf_load(0, 0, wfa);
  constexpr int nk = RC_H/RC_BK;
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();                     // the previous chunk's MFMAs have read Ws (and bias_s is complete)
#pragma unroll
    for (int ci = 0; ci < 4; ++ci)
      *reinterpret_cast<uint4*>(Ws + ((tid + 256*ci) >> 3)*RC_LDK + kc*8) = wraw[ci];
    wf_load(kt, 1, wfb);
    recompute(kt, 0, wfa);
    recompute(kt, 1, wfb);
    // the wave's own rows: LDS operations of one wave complete in order, the fence keeps the compiler
    // from moving the reads below above the writes of other lanes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
      const int row = 32*wid + sub + 8*ci;
      bf16_t* zp = As + row*RC_LDK + kc*8;
      float ev[8], zv[8], o[8];
      unpack8(eraw[ci], ev);
      unpack8(*reinterpret_cast<const uint4*>(zp), zv);
      const int kbase = kt*RC_BK + kc*8;
      const bool live = t0 + row < T;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool pos = zv[j] > 0.f;
        const float pv = pos ? zv[j] : slope*zv[j];
        const float xh = (pv - ns.mean)*ns.rstd;
        const float dh = ns.rstd*(ev[j] - dz_m1 - xh*dz_m2);
        const bool ok = live && kbase + j < p.C;
        o[j] = ok ? (pos ? dh : slope*dh) : 0.f;
        if (ok && !pos) dz_da += dh*zv[j];
      }
      *reinterpret_cast<uint4*>(zp) = pack8(o);
    }
    __syncthreads();                     // Ws complete (As rows are wave-private)
    if (kt + 1 < nk) { load_chunk(kt + 1); wf_load(kt + 1, 0, wfa); }
#pragma unroll
    for (int s = 0; s < RC_BK/16; ++s) {
      const bf16x8 af = *reinterpret_cast<const bf16x8*>(As + (32*wid + fr)*RC_LDK + 16*s + 8*fh);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(Ws + (32*j + fr)*RC_LDK + 16*s + 8*fh);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[j], 0, 0, 0);
      }
    }
  }

  // ---- accumulators -> LDS (fp32, [frame][channel]), row-wise E_ADD epilogue (as gemm_rows.cuh) ------
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = 32*wid + (i & 3) + 8*(i >> 2) + 4*fh;
      Cs[row*RC_LDC + 32*j + fr] = acc[j][i];
    }
  __syncthreads();
  const EpiSpec& e = p.e;
  constexpr int NCH = RC_N/8, RG = 256/NCH;
  const int ch = tid % NCH, rg = tid / NCH;
  const int ncol = ch*8;
  double st_sum = 0.0, st_sq = 0.0;
  float guv[16];
  if (e.gu_out) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { guv[j] = e.gu_v1[ncol + j]; guv[8 + j] = e.gu_v1[RC_N + ncol + j]; }
  }
  for (int row = rg; row < RC_BM; row += RG) {
    const int t = t0 + row;
    if (t >= T) continue;
    float v[8];
    {
      const float4 lo = *reinterpret_cast<const float4*>(Cs + row*RC_LDC + ch*8);
      const float4 hi = *reinterpret_cast<const float4*>(Cs + row*RC_LDC + ch*8 + 4);
      v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w;
      v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
    }
    const long long rowi = (long long)b*T + t;
    if (e.add_in) {
      float r[8];
      unpack8(*reinterpret_cast<const uint4*>(e.add_in + rowi*e.ld_add + ncol), r);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += r[j];
    }
    const uint4 qa = pack8(v);
    *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(e.out) + rowi*e.ldo + ncol) = qa;
    if (e.out2) *reinterpret_cast<uint4*>(e.out2 + rowi*e.ld_srcf + ncol) = qa;
    if (e.gu_out) {
      float gr[8], gs[8], ur[8], us[8];
      unpack8(qa, gr);
      unpack8(*reinterpret_cast<const uint4*>(e.gu_gskip + rowi*e.ld_gs + ncol), gs);
      unpack8(*reinterpret_cast<const uint4*>(e.gu_u + rowi*e.ld_gu + ncol), ur);
      unpack8(*reinterpret_cast<const uint4*>(e.gu_u + rowi*e.ld_gu + RC_N + ncol), us);
      float d1 = 0.f, d2 = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        d1 = __builtin_fmaf(gr[j], guv[j], d1);
        d1 = __builtin_fmaf(gs[j], guv[8 + j], d1);
        d2 = __builtin_fmaf(gr[j], ur[j], d2);
        d2 = __builtin_fmaf(gs[j], us[j], d2);
      }
      st_sum += d1; st_sq += d2;
    }
  }
  if (e.gu_out) {
    __syncthreads();
    double* dscr = reinterpret_cast<double*>(smem);
    const double s0 = block_sum(st_sum, dscr);
    const double s1 = block_sum(st_sq, dscr + 8);
    if (tid == 0) { atomic_add_f64(e.gu_out + stat_sum(b), s0); atomic_add_f64(e.gu_out + stat_sq(b), s1); }
  }
  {
    __syncthreads();
    float* fscr = reinterpret_cast<float*>(smem);
    const float s0 = block_sum(dz_da, fscr);
    if (tid == 0 && e.dslope) {
      const long long ro = e.n_rep > 1 ? (long long)(blockIdx.x % e.n_rep)*e.rep_stride : 0;
      atomic_add_f32(e.dslope + ro, s0);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Data gradient, weight-stationary form (the default): ONE persistent workgroup of 8 waves per CU keeps
// BOTH weight operands in registers for its whole tile range, 128 VGPRs per wave, with the waves
// SPECIALISED so that the two operands never sit in one wave's registers:
//   waves 0-3 (producers): wave w holds the W1 fragments of hidden channels [128 w, 128 w + 128) and
//     rebuilds z1^T of a 32-frame tile (lane = frame, registers = channels), reads e1 in that same
//     layout from the staged image and writes dz1 (bf16) into a double-buffered image Ds;
//   waves 4-7 (consumers): wave w holds the W1^T fragments of outputs [32 w, 32 w + 32) over all 512
//     hidden channels, multiplies the dz1 image of the PREVIOUS tile and runs the row-wise epilogue
//     (residual path, layer-norm means of the previous block) on its own 32 outputs.
// A producer and a consumer share every SIMD: the norm-backward VALU work of one runs beside the MFMAs
// of the other. e1 (32 KB) and x (8 KB) of the next tile are staged through registers by all 8 waves.
// ONE workgroup barrier per tile; no weight traffic per tile (the tile kernel above re-reads 256 KB of
// weights per 128 frames from L2). Optionally dz1 is also stored (dz_out: the stored-dz1 weight-gradient
// kernel then runs unchanged).
#ifndef WSD_ABL
#define WSD_ABL 0     // ablation bits (diagnostic builds, results wrong): 1 no norm-backward arithmetic, 2 no epilogue
#endif                // companions / math, 4 no e1 loads, 8 no producer MFMAs, 16 no consumer MFMAs, 32 no output stores
struct Pw1DgradWsParams {
  Pw1DgradRcParams r;            // tensors as for the tile kernel (n_ttiles unused)
  bf16_t* dz_out;                // nullable: dz1 (B, T, 512), row stride r.lde
  long long* dbg;                // -DWSD_STAMP: s_memtime stamps of wave 0 (producer) / 4 (consumer), 64 per workgroup
};
#ifdef WSD_STAMP
#define WSD_MARK(role, idx) do { long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    if (lane == 0 && pp.dbg && (idx) < 32) pp.dbg[(long long)blockIdx.x*64 + (role)*32 + (idx)] = t_; } while (0)
#else
#define WSD_MARK(role, idx) do { } while (0)
#endif

constexpr int WSD_TF = 32;                         // frames per tile
constexpr int WSD_LDX = RC_KX + 8;                 // halves per x row (272 B)
constexpr int WSD_LDE = 128 + 8;                   // halves per row of a producer's private e1 block (272 B)
constexpr int WSD_LDD = RC_H + 8;                  // halves per dz1 row (1040 B: conflict-free b128 B-operand reads)
constexpr int WSD_LDC = 32 + 4;                    // floats per staged output row of one consumer wave
constexpr int WSD_XB = WSD_TF*WSD_LDX*2, WSD_EB = WSD_TF*WSD_LDE*2, WSD_DB = WSD_TF*WSD_LDD*2;
constexpr int WSD_CB = WSD_TF*WSD_LDC*4;           // per consumer wave
constexpr int WSD_SMEM = 2*WSD_XB + 4*WSD_EB + 2*WSD_DB + 4*WSD_CB + RC_H*4 + 2*RC_N*4;
static_assert(WSD_SMEM <= 160*1024, "LDS budget");

template <bool FULLC>
__global__ __launch_bounds__(512) void pw1_dgrad_ws_kernel(const Pw1DgradWsParams pp) {
  const Pw1DgradRcParams& p = pp.r;
  __shared__ __attribute__((aligned(16))) unsigned char smem[WSD_SMEM];
  bf16_t* Xs = reinterpret_cast<bf16_t*>(smem);                        // [2][32][LDX]
  bf16_t* Es = reinterpret_cast<bf16_t*>(smem + 2*WSD_XB);             // [4 producers][32][LDE], wave-private
  bf16_t* Ds = reinterpret_cast<bf16_t*>(smem + 2*WSD_XB + 4*WSD_EB);  // [2][32][LDD]
  float* Cs_all = reinterpret_cast<float*>(smem + 2*WSD_XB + 4*WSD_EB + 2*WSD_DB);   // [4][32][LDC]
  float* bias_s = reinterpret_cast<float*>(smem + 2*WSD_XB + 4*WSD_EB + 2*WSD_DB + 4*WSD_CB);
  float* guv_s = bias_s + RC_H;                      // v1 of the previous block's lazy norm, both halves
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, fh = lane >> 5;
  const int T = p.T;
  const EpiSpec& e = p.e;

  for (int c = tid; c < RC_H; c += 512) bias_s[c] = c < p.C ? p.bias[c] : 0.f;
  if (tid < 2*RC_N) guv_s[tid] = e.gu_out ? e.gu_v1[tid] : 0.f;

  // ---- tile schedule: contiguous range per workgroup ----------------------------------------------
  const int tpi = ceil_div(T, WSD_TF);
  const int total = tpi*p.batch;
  const int per = ceil_div(total, (int)gridDim.x);
  const int t_begin = blockIdx.x*per;
  const int t_end = min(total, t_begin + per);
  const int t_items = tpi*WSD_TF;
  if (t_begin >= t_end) return;

  // staging: a producer stages the e1 block of ITS 128 channels (wave-private image, 256-byte row segments);
  // the consumers stage x (shared, double-buffered, handed over by the barrier)
  auto advance = [&](int& b, int& t) { t += WSD_TF; if (t >= t_items) { t = 0; ++b; } };
  const int n_tiles = t_end - t_begin;
  const int b0 = t_begin / tpi, tf0 = (t_begin % tpi)*WSD_TF;

  // Iteration i (0 .. n_tiles): [all] the staged operands of tile i -> LDS images (i & 1); barrier;
  // [all] request tile i + 1; [producers] dz1 of tile i -> Ds[i & 1]; [consumers] data gradient + epilogue
  // of tile i - 1 from Ds[(i - 1) & 1]. Ds[i & 1] is rewritten in iteration i + 2, behind the barrier of
  // iteration i + 1, which every consumer reaches after its reads; the staged images likewise.
  if (wid < 4) {
    // ======================= producers: z1^T, norm backward -> dz1 =======================
    if (wid == 0) WSD_MARK(0, 0);
    bf16x8 wf[4][8];
    {
      const bf16_t* src = p.Wfp + (long long)wid*128*RC_KX + (long long)lane*8;
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int s = 0; s < 8; ++s) wf[f][s] = *reinterpret_cast<const bf16x8*>(src + (f*8 + s)*512);
    }
    const float slope = *p.slope;
    const float pc1 = 0.5f*(1.f + slope), pc2 = 0.5f*(1.f - slope);
    float da2 = 0.f;
    int cur_b = -1;
    float rstd = 1.f, dz_m1 = 0.f, dz_m2 = 0.f, XA = 0.f, XB = 0.f, XC = 0.f;
    const int ec = lane & 15, er = lane >> 4;      // 16-byte column / row (+ 4 ci) of the wave's e1 block
    bf16_t* es = Es + wid*WSD_TF*WSD_LDE;
    uint4 eraw[8];
    auto load_e = [&](int b, int t0, bool valid) {
      const __amdgpu_buffer_rsrc_t re = make_rsrc(p.e1 + (long long)b*p.bse, (valid && !(WSD_ABL & 4)) ? (long long)T*p.lde*2 : 0);
#pragma unroll
      for (int ci = 0; ci < 8; ++ci)
        eraw[ci] = buf_load16(re, (unsigned int)(t0 + er + 4*ci)*(unsigned int)(p.lde*2)
                                  + (unsigned int)((128*wid + ec*8)*2));
    };
    int b_ld = b0, t_ld = tf0;                     // tile whose operands are in the staging registers
    load_e(b_ld, t_ld, true);
    if (wid == 0) WSD_MARK(0, 1);
#pragma unroll 1
    for (int i = 0; i <= n_tiles; ++i) {
      const int buf = i & 1;
      const int b = b_ld, t0 = t_ld;               // tile i (meaningless for i == n_tiles)
      if (wid == 0) WSD_MARK(0, 2 + 3*i);
#pragma unroll
      for (int ci = 0; ci < 8; ++ci)               // own image: ordered behind this wave's reads of tile i - 1
        *reinterpret_cast<uint4*>(es + (er + 4*ci)*WSD_LDE + ec*8) = eraw[ci];
      if (wid == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); WSD_MARK(0, 3 + 3*i); }
      __syncthreads();                             // x image of tile i complete, Ds[buf] free
      if (wid == 0) WSD_MARK(0, 4 + 3*i);
      if (i == n_tiles) break;
      advance(b_ld, t_ld);
      load_e(b_ld, t_ld, i + 1 < n_tiles);
      if (b != cur_b) {
        const NormStat ns = norm_stat(p.stats, b, p.inv_n, p.eps);
        rstd = ns.rstd;
        XA = pc1*ns.rstd; XB = pc2*ns.rstd; XC = -ns.mean*ns.rstd;
        dz_m1 = (float)(p.sums[stat_sum(b)]*p.inv_n);
        dz_m2 = (float)(p.sums[stat_sq(b)]*p.inv_n);
        cur_b = b;
      }
      const bf16_t* xs = Xs + buf*WSD_TF*WSD_LDX;
      bf16_t* ds = Ds + buf*WSD_TF*WSD_LDD;
      // dz1 = PReLU'(z) R (e - m1 - xh m2), xh = (PReLU(z) - mean) rstd = XA z + XB |z| + XC; the frame's
      // validity (lane = frame) is folded into R: rows past the end of the item give exact zeros
      const float R = (t0 + fr < T) ? rstd : 0.f;
      const float K0 = -dz_m1*R, M2 = -dz_m2*R;
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        f32x16 az;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 bv = *reinterpret_cast<const float4*>(bias_s + 128*wid + 32*f + 8*g + 4*fh);
          az[4*g] = bv.x; az[4*g + 1] = bv.y; az[4*g + 2] = bv.z; az[4*g + 3] = bv.w;
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const bf16x8 xb = *reinterpret_cast<const bf16x8*>(xs + fr*WSD_LDX + 16*s + 8*fh);
          if (!(WSD_ABL & 8)) az = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[f][s], xb, az, 0, 0, 0);
          else az[s] += __builtin_bit_cast(f32x4, xb)[0] + __builtin_bit_cast(f32x4, wf[f][s])[0];
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int kch = 128*wid + 32*f + 8*g + 4*fh;
          const uint2 eq = *reinterpret_cast<const uint2*>(es + fr*WSD_LDE + 32*f + 8*g + 4*fh);
          const float ev[4] = {__uint_as_float(eq.x << 16), __uint_as_float(eq.x & 0xffff0000u),
                               __uint_as_float(eq.y << 16), __uint_as_float(eq.y & 0xffff0000u)};
          // z1 as pw1_fwd stored it (bf16)
          const uint32_t zq0 = pack2(az[4*g], az[4*g + 1]), zq1 = pack2(az[4*g + 2], az[4*g + 3]);
          const float zv[4] = {__uint_as_float(zq0 << 16), __uint_as_float(zq0 & 0xffff0000u),
                               __uint_as_float(zq1 << 16), __uint_as_float(zq1 & 0xffff0000u)};
          float o[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float z = zv[j];
            if (WSD_ABL & 1) { o[j] = z + ev[j]; continue; }
            const float xh = __builtin_fmaf(XB, __builtin_fabsf(z), __builtin_fmaf(XA, z, XC));
            float dh = __builtin_fmaf(xh, M2, __builtin_fmaf(ev[j], R, K0));
            if (!FULLC) dh = kch + j < p.C ? dh : 0.f;
            o[j] = z > 0.f ? dh : slope*dh;
            da2 = __builtin_fmaf(dh, z - __builtin_fabsf(z), da2);     // 2 dh min(z, 0)
          }
          uint2 v;
          v.x = pack2(o[0], o[1]); v.y = pack2(o[2], o[3]);
          *reinterpret_cast<uint2*>(ds + fr*WSD_LDD + kch) = v;
        }
      }
    }
    const float dz_da = 0.5f*da2;
    const float s0 = wave_sum(dz_da);
    if (lane == 0 && e.dslope) {
      const long long ro = e.n_rep > 1 ? (long long)((blockIdx.x*4 + wid) % e.n_rep)*e.rep_stride : 0;
      atomic_add_f32(e.dslope + ro, s0);
    }
  } else {
    // ======================= consumers: W1^T dz1, residual path, <g, u> dots =======================
    const int cw = wid - 4;                          // outputs [32 cw, 32 cw + 32)
    bf16x8 wb[32];
    {
      const bf16_t* srb = p.Wb + (long long)(32*cw + fr)*RC_H + 8*fh;
#pragma unroll
      for (int s = 0; s < 32; ++s) wb[s] = *reinterpret_cast<const bf16x8*>(srb + 16*s);
    }
    float* Cw = Cs_all + cw*WSD_TF*WSD_LDC;
    // epilogue geometry of the wave: rows orow + 16 k (k = 0, 1), 8 outputs from column ocol
    const int orow = lane >> 2, ocol = 32*cw + 8*(lane & 3);
    double gu_s = 0.0, gu_q = 0.0;
    int cur_b = -1;
    auto flush_gu = [&]() {
      if (cur_b < 0 || !e.gu_out) return;
      const double s0 = wave_sum(gu_s), s1 = wave_sum(gu_q);
      if (lane == 0) { atomic_add_f64(e.gu_out + stat_sum(cur_b), s0); atomic_add_f64(e.gu_out + stat_sq(cur_b), s1); }
      gu_s = 0.0; gu_q = 0.0;
    };
    const int ctid = tid - 256;                    // x chunks ctid, ctid + 256: row id >> 4, 16-byte column id & 15
    uint4 xraw[2];
    auto load_x = [&](int b, int t0, bool valid) {
      const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x + (long long)b*p.bsx, valid ? (long long)T*p.ldx*2 : 0);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int id = ctid + 256*k;
        xraw[k] = buf_load16(rx, (unsigned int)(t0 + (id >> 4))*(unsigned int)(p.ldx*2) + (unsigned int)((id & 15)*16));
      }
    };
    int b_ld = b0, t_ld = tf0;
    int b_pv = b0, t_pv = tf0;                     // tile i - 1
    load_x(b_ld, t_ld, true);
    if (wid == 4) { WSD_MARK(1, 0); WSD_MARK(1, 1); }
#pragma unroll 1
    for (int i = 0; i <= n_tiles; ++i) {
      const int buf = i & 1;
      if (wid == 4) WSD_MARK(1, 2 + 3*i);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int id = ctid + 256*k;
        *reinterpret_cast<uint4*>(Xs + buf*WSD_TF*WSD_LDX + (id >> 4)*WSD_LDX + (id & 15)*8) = xraw[k];
      }
      // companion rows of tile i - 1's epilogue: requested BEFORE the barrier, they arrive while this
      // wave waits for the producers
      const int b = b_pv, t0 = t_pv;
      uint4 c_add[2], c_gs[2], c_ur[2], c_us[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int t = t0 + orow + 16*k;
        const long long rowi = (long long)b*T + t;
        c_add[k] = c_gs[k] = c_ur[k] = c_us[k] = make_uint4(0, 0, 0, 0);
        if (i > 0 && t < T && !(WSD_ABL & 2)) {
          if (e.add_in) c_add[k] = *reinterpret_cast<const uint4*>(e.add_in + rowi*e.ld_add + ocol);
          if (e.gu_out) {
            c_gs[k] = *reinterpret_cast<const uint4*>(e.gu_gskip + rowi*e.ld_gs + ocol);
            c_ur[k] = *reinterpret_cast<const uint4*>(e.gu_u + rowi*e.ld_gu + ocol);
            c_us[k] = *reinterpret_cast<const uint4*>(e.gu_u + rowi*e.ld_gu + RC_N + ocol);
          }
        }
      }
      if (wid == 4) WSD_MARK(1, 3 + 3*i);
      __syncthreads();
      if (wid == 4) WSD_MARK(1, 4 + 3*i);
      if (i < n_tiles) {
        advance(b_ld, t_ld);
        load_x(b_ld, t_ld, i + 1 < n_tiles);
      }
      if (i == 0) continue;
      advance(b_pv, t_pv);
      if (b != cur_b) { flush_gu(); cur_b = b; }
      const bf16_t* ds = Ds + (buf ^ 1)*WSD_TF*WSD_LDD;
      if (pp.dz_out != nullptr && !(WSD_ABL & 32)) {        // dz1 for the stored-dz1 weight gradient: whole 1 KiB rows
        const __amdgpu_buffer_rsrc_t rd = make_rsrc(pp.dz_out + (long long)b*p.bse, (long long)T*p.lde*2);
#pragma unroll
        for (int ci = 0; ci < 8; ++ci) {
          const int row = 4*ci + cw;
          buf_store16(rd, (unsigned int)(t0 + row)*(unsigned int)(p.lde*2) + (unsigned int)(lane*16),
                      *reinterpret_cast<const uint4*>(ds + row*WSD_LDD + lane*8));
        }
      }
      f32x16 ag;
#pragma unroll
      for (int q = 0; q < 16; ++q) ag[q] = 0.f;
#pragma unroll
      for (int s = 0; s < 32; ++s) {
        const bf16x8 db = *reinterpret_cast<const bf16x8*>(ds + fr*WSD_LDD + 16*s + 8*fh);
        if (!(WSD_ABL & 16)) ag = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[s], db, ag, 0, 0, 0);
        else ag[s & 15] += __builtin_bit_cast(f32x4, db)[0] + __builtin_bit_cast(f32x4, wb[s])[0];
      }
      // D[output][frame] -> the wave's staging tile [frame][32 outputs] (fp32)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(Cw + fr*WSD_LDC + 8*g + 4*fh) =
            make_float4(ag[4*g], ag[4*g + 1], ag[4*g + 2], ag[4*g + 3]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int row = orow + 16*k, t = t0 + row;
        if (t >= T) continue;
        const long long rowi = (long long)b*T + t;
        float v[8];
        {
          const float4 lo = *reinterpret_cast<const float4*>(Cw + row*WSD_LDC + 8*(lane & 3));
          const float4 hi = *reinterpret_cast<const float4*>(Cw + row*WSD_LDC + 8*(lane & 3) + 4);
          v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w;
          v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
        }
        if (e.add_in) {
          float r[8];
          unpack8(c_add[k], r);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += r[j];
        }
        const uint4 qa = pack8(v);
        if (!(WSD_ABL & 32)) {
          *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(e.out) + rowi*e.ldo + ocol) = qa;
          if (e.out2) *reinterpret_cast<uint4*>(e.out2 + rowi*e.ld_srcf + ocol) = qa;
        } else if (qa.x == 0x12345678u) Cw[0] = 1.f;
        if (e.gu_out && !(WSD_ABL & 2)) {
          float gr[8], gs[8], ur[8], us[8];
          unpack8(qa, gr); unpack8(c_gs[k], gs); unpack8(c_ur[k], ur); unpack8(c_us[k], us);
          float d1 = 0.f, d2 = 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            d1 = __builtin_fmaf(gr[j], guv_s[ocol + j], d1);
            d1 = __builtin_fmaf(gs[j], guv_s[RC_N + ocol + j], d1);
            d2 = __builtin_fmaf(gr[j], ur[j], d2);
            d2 = __builtin_fmaf(gs[j], us[j], d2);
          }
          gu_s += d1; gu_q += d2;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();               // (the staging tile is rewritten by the next tile)
    }
    flush_gu();
  }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient dW1[n][k] = sum_{b,t} dz1[b][t][n] x[b][t][k] (+ db1 = column sums of dz1) with dz1
// rebuilt from e1 and x per 64-frame chunk. One workgroup = 128 hidden channels x all 128 inputs over
// a contiguous range of chunks; wave w keeps the W1 fragments and the bias of ITS 32 channels in
// registers for the whole range. Both LDS tiles are [64 frames][128 x bf16] images in the swizzled
// 256-byte-row layout that serves row reads (B operand of the recompute, 16-byte staging accesses) and
// transposing reads (both operands of the weight-gradient product) without bank conflicts
// (cdna_hip_programming.md T10, layout (b)).
struct Pw1WgradRcProb {
  const bf16_t* e1; const bf16_t* x; const bf16_t* Wfp; const float* bias;
  float* out; float* gbias;
  const double* stats; const double* sums; const float* slope;
};
struct Pw1WgradRcParams {
  int B, T, nsplit, nprob, C, Kout, ldo;         // C: true hidden channels, Kout: true input channels
  int lde, ldx; long long bse, bsx;
  double inv_n; float eps;
  Pw1WgradRcProb prob[kWgMaxProb];
};

__device__ __forceinline__ int rc_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
// byte offset of 16-byte chunk `ch` (0..15) of row `row` of a [rows][128 x bf16] image
__device__ __forceinline__ int rc_off(int row, int ch) { return 256*row + 16*(ch ^ rc_swz(row)); }

// fragment for v_mfma_f32_32x32x16_bf16 from a swizzled image: lane (r = lane & 31, h = lane >> 5) gets
// tile[row0 + 8 h + j][col0 + r], j = 0..7 (col0 a multiple of 32, row0 of 16)
__device__ __forceinline__ bf16x8 rc_tr_frag(const unsigned char* img, int row0, int col0, int lane) {
  const int g4 = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
  const int chunk = (col0 >> 3) + 2*(g4 & 1) + (pp >> 1);
  const int row = row0 + 8*(g4 >> 1) + q;
  typedef __attribute__((address_space(3))) s16x4* lds_p;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + rc_off(row, chunk) + 8*(pp & 1)));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + rc_off(row + 4, chunk) + 8*(pp & 1)));
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

// 4 rows x 16 columns (16-bit) per 16-lane group, delivered column-major (T10): `addr` = byte address of
// this lane's row / 4-column piece
__device__ __forceinline__ s16x4 rc_tr4(const unsigned char* addr) {
  typedef __attribute__((address_space(3))) s16x4* lds_p;
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)addr);
}

// Orientation: z1 is rebuilt as z1[frame][channel] = x W1^T with the frames in the accumulator REGISTERS
// and the channel on the LANE, so that (a) every per-channel quantity (bias, bias gradient) is per lane,
// (b) e1 arrives in the same layout through transposing reads of its row-major image, and (c) dz1, once
// rounded to bf16, IS the A operand of the weight-gradient MFMA (which sums over the frames = the rows of
// the first product's result: no lane movement, no LDS round trip; cdna_hip_programming.md section 3,
// "An accumulator tile as the next MFMA's operand") -- the x fragments of that product are fetched with
// the matching frame permutation. One barrier per 64-frame chunk; both images are double-buffered.
__global__ __launch_bounds__(256, 2) void pw1_wgrad_rc_kernel(const Pw1WgradRcParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char Es[2][WG_BT*256];
  __shared__ __attribute__((aligned(16))) unsigned char Xs[2][WG_BT*256];
  // XCD-aware order (as gemm_wgrad_kernel): the 4 channel tiles of a (split, problem) pair stream the
  // same x rows and take consecutive slots of ONE XCD
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
  const Pw1WgradRcProb& q = p.prob[bz];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int n0 = bx*WG_BG;
  const int T = p.T;
  const int cpi = ceil_div(T, WG_BT);
  const int total = p.B*cpi;
  const int per = ceil_div(total, p.nsplit);
  const int c_begin = by*per;
  const int c_end = min(total, c_begin + per);

  // stationary: W1 fragments of the wave's 32 channels (B operand of the recompute), the lane's bias
  const int nch = n0 + 32*wid + fr;                   // this lane's hidden channel
  bf16x8 wf[8];
  {
    const int nw = n0 + 32*wid;
    const bf16_t* src = q.Wfp + (long long)(nw >> 6)*64*RC_KX + ((long long)(((nw >> 5) & 1)*8)*64 + lane)*8;
#pragma unroll
    for (int s = 0; s < 8; ++s) wf[s] = *reinterpret_cast<const bf16x8*>(src + s*512);
  }
  const bool ch_ok = nch < p.C;
  const float bias = ch_ok ? q.bias[ch_ok ? nch : 0] : 0.f;
  const float slope = *q.slope;
  const float pc1 = 0.5f*(1.f + slope), pc2 = 0.5f*(1.f - slope);

  // staging: thread (row r16 + 16 ci, 16-byte chunk c16) of both [64][128] images
  const int c16 = tid & 15, r16 = tid >> 4;
  uint4 xraw[4], eraw[4];
  auto load_chunk = [&](int c) {
    const int b = c / cpi, t0 = (c % cpi)*WG_BT;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(q.x + (long long)b*p.bsx, (long long)T*p.ldx*2);
    const __amdgpu_buffer_rsrc_t re = make_rsrc(q.e1 + (long long)b*p.bse, (long long)T*p.lde*2);
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
      xraw[ci] = buf_load16(rx, (unsigned int)(t0 + r16 + 16*ci)*(unsigned int)(p.ldx*2) + (unsigned int)(c16*16));
      eraw[ci] = buf_load16(re, (unsigned int)(t0 + r16 + 16*ci)*(unsigned int)(p.lde*2) + (unsigned int)((n0 + c16*8)*2));
    }
  };

  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  float bias_acc = 0.f;

  // transposing-read geometry of this lane (16-lane group g4 = lane >> 4: column block 16 (g4 & 1), row half fh)
  const int i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3, g4 = lane >> 4;
  const int e_chunk = 4*wid + 2*(g4 & 1) + (tp >> 1);          // e1 image: columns 32 wid + 16 (g4 & 1) + 4 tp
  const int tr_byte = 8*(tp & 1);

  int cur_b = -1;
  float R = 0.f, K0 = 0.f, M2 = 0.f, XA = 0.f, XB = 0.f, XC = 0.f;
  if (c_begin < c_end) load_chunk(c_begin);
  for (int c = c_begin; c < c_end; ++c) {
    const int buf = (c - c_begin) & 1;
    const int b = c / cpi, t0 = (c % cpi)*WG_BT;
    if (b != cur_b) {
      const NormStat ns = norm_stat(q.stats, b, p.inv_n, p.eps);
      const float m1 = (float)(q.sums[stat_sum(b)]*p.inv_n), m2 = (float)(q.sums[stat_sq(b)]*p.inv_n);
      R = ch_ok ? ns.rstd : 0.f; K0 = -m1*R; M2 = -m2*R;
      XA = pc1*ns.rstd; XB = pc2*ns.rstd; XC = -ns.mean*ns.rstd;
      cur_b = b;
    }
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
      *reinterpret_cast<uint4*>(Xs[buf] + rc_off(r16 + 16*ci, c16)) = xraw[ci];
      *reinterpret_cast<uint4*>(Es[buf] + rc_off(r16 + 16*ci, c16)) = eraw[ci];
    }
    __syncthreads();      // images of chunk c complete; those of chunk c - 1 (other buffer) are free again
    if (c + 1 < c_end) load_chunk(c + 1);
    const unsigned char* xs = Xs[buf];
    const unsigned char* es = Es[buf];
    const bool tail = t0 + WG_BT > T;                 // frames past the end of the item: only the bias sum minds
#pragma unroll
    for (int ft = 0; ft < 2; ++ft) {
      f32x16 az;
#pragma unroll
      for (int i = 0; i < 16; ++i) az[i] = bias;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const bf16x8 xa = *reinterpret_cast<const bf16x8*>(xs + rc_off(32*ft + fr, 2*s + fh));
        az = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, wf[s], az, 0, 0, 0);
      }
      // register i = frame 32 ft + (i & 3) + 8 (i >> 2) + 4 fh of channel `nch`; e1 likewise: one
      // transposing read per group of 4 frames
      uint32_t dq[8];
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int row = 32*ft + 8*gq + 4*fh + tq;
        const s16x4 e4 = rc_tr4(es + rc_off(row, e_chunk) + tr_byte);
        const uint32_t zq0 = pack2(az[4*gq], az[4*gq + 1]), zq1 = pack2(az[4*gq + 2], az[4*gq + 3]);
        const float zv[4] = {__uint_as_float(zq0 << 16), __uint_as_float(zq0 & 0xffff0000u),
                             __uint_as_float(zq1 << 16), __uint_as_float(zq1 & 0xffff0000u)};
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float z = zv[j];
          const float ev = __uint_as_float(((uint32_t)(uint16_t)e4[j]) << 16);
          const float xh = __builtin_fmaf(XB, __builtin_fabsf(z), __builtin_fmaf(XA, z, XC));
          const float dh = __builtin_fmaf(xh, M2, __builtin_fmaf(ev, R, K0));
          o[j] = z > 0.f ? dh : slope*dh;
        }
        dq[2*gq] = pack2(o[0], o[1]); dq[2*gq + 1] = pack2(o[2], o[3]);
        if (q.gbias) {
          // (the rounded values, as the product sees them)
          float s4 = (__uint_as_float(dq[2*gq] << 16) + __uint_as_float(dq[2*gq] & 0xffff0000u)) +
                     (__uint_as_float(dq[2*gq + 1] << 16) + __uint_as_float(dq[2*gq + 1] & 0xffff0000u));
          if (tail) {
            s4 = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const uint32_t w = dq[2*gq + (j >> 1)];
              const float v = (j & 1) ? __uint_as_float(w & 0xffff0000u) : __uint_as_float(w << 16);
              s4 += (t0 + 32*ft + 8*gq + 4*fh + j < T) ? v : 0.f;
            }
          }
          bias_acc += s4;
        }
      }
      // dW1^T tile: A = dz1 (this lane's channel x 16 frames per k-step, straight from the registers),
      // B = x with the same frame order: element j of lane half fh = frame 16 s + 8 (j >> 2) + 4 fh + (j & 3)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const s16x8 av = {(short)(dq[4*s2] & 0xffff), (short)(dq[4*s2] >> 16), (short)(dq[4*s2 + 1] & 0xffff), (short)(dq[4*s2 + 1] >> 16),
                          (short)(dq[4*s2 + 2] & 0xffff), (short)(dq[4*s2 + 2] >> 16), (short)(dq[4*s2 + 3] & 0xffff), (short)(dq[4*s2 + 3] >> 16)};
        const bf16x8 a2 = __builtin_bit_cast(bf16x8, av);
        const int row = 32*ft + 16*s2 + 4*fh + tq;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int chunk = 4*j + 2*(g4 & 1) + (tp >> 1);
          const s16x4 lo = rc_tr4(xs + rc_off(row, chunk) + tr_byte);
          const s16x4 hi = rc_tr4(xs + rc_off(row + 8, chunk) + tr_byte);
          const s16x8 bv = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, __builtin_bit_cast(bf16x8, bv), acc[j], 0, 0, 0);
        }
      }
    }
  }

  // ---- partial tile -> gradient (atomics, as gemm_wgrad_kernel): D[channel][input] -------------------
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = 32*j + fr;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int n = n0 + 32*wid + (i & 3) + 8*(i >> 2) + 4*fh;
      if (k < p.Kout && n < p.C) atomic_add_f32(q.out + (long long)n*p.ldo + k, acc[j][i]);
    }
  }
  if (q.gbias) {
    bias_acc += __shfl_xor(bias_acc, 32, 64);       // the two frame halves of a channel
    if (lane < 32 && ch_ok) atomic_add_f32(q.gbias + nch, bias_acc);
  }
}

}  // namespace brv
