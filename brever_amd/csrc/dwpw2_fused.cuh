// Fused forward of the second half of a Conv-TasNet TCN block (default widths: 512 hidden, 128 +
// 128 residual / skip channels, kernel 3), reference brever/models/convtasnet/convtasnet.py:
//   z2 = dconv(gLN_1(PReLU_1(z1))) + bias          (stored: the backward pass reads it)
//   p  = PReLU_2(z2)                               (statistics of p accumulated: gLN_2)
//   u  = (W gamma_2) p                             (the [res | skip] product on the UN-normalised p;
//                                                   rstd_2 and the offset are applied by the consumers:
//                                                   gemm_ws AT == 2, skip_combine_kernel)
// in ONE kernel: z2 never comes back from HBM in the forward pass and the fp32 skip accumulation
// is gone (164 MB per launch instead of 131 + 164 MB for dwconv_fwd + pw2_fwd). It supersedes the
// gemm_ws AT == 3 variant (BRV_DWPW2_WS=1), whose weight-stationary register tile (128 VGPRs of
// weights) left the depthwise stage spilling. Measurements and rejected variants: DESIGN.md 5f.
//
// One workgroup of 8 waves per CU, 128 frames x all 512 channels per tile, channel slabs of 64:
//   stage(s):  every thread owns one channel octet of the slab and two frames: three dilated taps
//              of z1 (16-byte loads, one slab ahead), the folded PReLU_1 / gLN_1 / tap arithmetic
//              of dwconv_fwd_kernel, z2 out, p (bf16) into an LDS slab [frame][64];
//   mfma(s):   wave wn = 32 outputs x 128 frames: A = ONE weight fragment per k-step straight from
//              L2 (packed in fragment order), B = the LDS slab, 16 MFMAs per slab;
// stage(s + 1) and mfma(s) are independent instruction streams of one basic block (VALU beside the
// matrix pipe); one barrier per slab. u leaves through LDS as whole 512-byte rows.
#pragma once
#include "common.cuh"

namespace brv {

#ifndef DP_ABL
#define DP_ABL 0     // ablation bits (diagnostic builds): 1 no MFMA slab, 2 no z2 store, 4 no u epilogue, 8 no stage arithmetic
#endif

struct DwPw2Params {
  const bf16_t* z1; bf16_t* z2; bf16_t* u;         // (B, T, 512), (B, T, 512), (B, T, 256)
  const bf16_t* Wp;                                // [256][512] gamma-folded, fragment order (slices of 32)
  const float* slope1; const float* slope2;
  const double* stats1; double* stats2;
  const float* gamma1; const float* beta1; const float* taps; const float* dbias;
  int B, T, dil, left, C;                          // C: true hidden channels (<= 512)
  double inv_n; float eps;
};

constexpr int DP_TT = 128, DP_H = 512, DP_N = 256, DP_SLAB = 64, DP_NSLAB = DP_H/DP_SLAB;
constexpr int DP_LDP = DP_SLAB + 8;                // halves per LDS row (144 B: conflict-free b128 reads)
constexpr int DP_LDU = DP_N + 8;                   // halves per staged u row (528 B)
constexpr int DP_OFF_P = 8*DP_H*4;                // after the per-item tables
constexpr int DP_PBYTES = DP_TT*DP_LDP*2;          // one slab buffer
constexpr int DP_SMEM = DP_OFF_P + 2*DP_PBYTES;

__global__ __launch_bounds__(512) void dwpw2_fused_kernel(const DwPw2Params p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[DP_SMEM];
  static_assert(2*DP_PBYTES >= 64*DP_LDU*2, "u staging fits the slab buffers");
  float* tabs = reinterpret_cast<float*>(smem);                 // [8][512]: wa0-2, wc0-2, bias, bias + wc0-2
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wn = wid;                                           // 32 outputs x all 128 frames per wave
  const int n32 = lane & 31, khalf = lane >> 5;
  const int o = tid & 7, rl = tid >> 3;                         // stage: channel octet, frame lane
  const int T = p.T;
  const int tpi = (T + DP_TT - 1)/DP_TT;
  const int n_tiles = tpi*p.B;
  const float a1 = *p.slope1, a2 = *p.slope2;
  // PReLU_1 -> gLN_1 -> tap k of a channel: w_k (scale (c1 z + c2 |z|) + shift) with ONE slope for all
  // channels, so the tables hold w_k scale c1 and the element is z + rho |z|, rho = c2 / c1 (half the
  // table reads of separate z and |z| coefficients; c1 == 0, a slope of exactly -1, is nudged to
  // 2^-40: relative error 1e-12)
  const float c1r = 0.5f*(1.f + a1), c2 = 0.5f*(1.f - a1);
  const float c1 = __builtin_fabsf(c1r) < 0x1p-40f ? 0x1p-40f : c1r;
  const float rho = c2/c1;
  const float d1 = 0.5f*(1.f + a2), d2 = 0.5f*(1.f - a2);
  const unsigned int rowb = DP_H*2;

  int cur_item = -1;
  double s_sum = 0.0, s_sq = 0.0;
  auto flush_stats = [&]() {
    if (cur_item < 0) return;
    const double r0 = wave_sum(s_sum), r1 = wave_sum(s_sq);
    if (lane == 0) {
      atomic_add_f64(p.stats2 + stat_sum(cur_item), r0);
      atomic_add_f64(p.stats2 + stat_sq(cur_item), r1);
    }
    s_sum = 0.0; s_sq = 0.0;
  };

  // XCD k (workgroup ids congruent k mod 8 share an L2) takes the k-th contiguous eighth of the tiles:
  // the dilated taps of neighbouring tiles are then fetched once per L2, not once per XCD
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = (n_tiles + 7) >> 3;
  const int wg_per_xcd = gridDim.x >> 3;                        // gridDim.x is a multiple of 8 (launch site)
#ifndef DP_CONTIG
#define DP_CONTIG 1
#endif
  // ... and (round 5) a workgroup takes a contiguous RUN of its XCD's tiles: its second tile is then nearly always
  // in the same item, so the per-item tables below are built once per workgroup instead of once per tile (with the
  // strided deal -- tiles 32 apart = the same frames of the NEXT item at the BASELINE size -- every tile paid the
  // dependent global loads and two barriers of the table build: ~2 us of a 55 us launch)
  // (runs as even as the counts allow, the LONGER ones on the lowest slots: those workgroups are dispatched first.
  // With the extra tiles spread over the slots -- so that the last-dispatched workgroup had two -- the two-chain step,
  // where a launch waits for CUs the other chain still holds, lost 0.19 ms.)
  const int run_lo = per_xcd/wg_per_xcd, run_rem = per_xcd % wg_per_xcd;
  const int ti_lo = DP_CONTIG ? slot*run_lo + min(slot, run_rem) : slot;
  const int ti_hi = DP_CONTIG ? ti_lo + run_lo + (slot < run_rem ? 1 : 0) : per_xcd;
  // Round 6 (94 MB read per launch against 70 MB in round 4): with a run of ADJACENT tiles per workgroup
  // all workgroups of an XCD start on every second tile and come back for the tiles in between ~25 us later, when
  // the dilated taps those share with their neighbours (up to 2 x 128 rows per 128-row tile) have left the 4 MB L2
  // again: the halo rows were fetched from HBM twice. When the counts divide evenly (the whole-batch launch:
  // 64 tiles = 2 items per XCD on 32 workgroups) a workgroup's tiles now lie `gap` = tiles-per-item / run apart
  // INSIDE one item -- the workgroups of an item walk a contiguous range of its tiles together in every round
  // (halo rows shared in time, as under the strided deal) and still build the item's tables once.
  const bool even_deal = DP_CONTIG && run_rem == 0 && run_lo > 1 && (n_tiles & 7) == 0 && per_xcd % tpi == 0 &&
                         tpi % run_lo == 0;
  const int gap = even_deal ? tpi/run_lo : 1;
  const int tile0 = xcd*per_xcd + (even_deal ? (slot/gap)*tpi + slot % gap : ti_lo);
  for (int ti_ = ti_lo; ti_ < ti_hi; ti_ += DP_CONTIG ? 1 : wg_per_xcd) {
    const int tile = DP_CONTIG ? tile0 + (ti_ - ti_lo)*gap : xcd*per_xcd + ti_;
    if (tile >= n_tiles) break;
    const int b = tile / tpi, t0 = (tile % tpi)*DP_TT;
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(p.z1 + (long long)b*T*DP_H, (long long)T*DP_H*2);
    const __amdgpu_buffer_rsrc_t rz2 = make_rsrc(p.z2 + (long long)b*T*DP_H, (long long)T*DP_H*2);

    // three taps of the thread's two frames, channel octet o of slab s (taps outside [0, T) wrap to
    // offsets beyond the descriptor: zeros)
    auto load_raw = [&](int s, uint4 (&raw)[2][3]) {
      const unsigned int coff = (unsigned int)((s*DP_SLAB + o*8)*2);
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const unsigned int ti = (unsigned int)(t0 + rl + 64*r + k*p.dil - p.left);
          raw[r][k] = buf_load16(rin, ti*rowb + coff);
        }
    };
    // the first slab's taps go out BEFORE the table build: its loads, barriers and the HBM latency of the taps overlap
    uint4 raw[2][3];
    load_raw(0, raw);
    if (b != cur_item) {
      flush_stats();
      __syncthreads();
      const NormStat ns = norm_stat(p.stats1, b, p.inv_n, p.eps);
      for (int c = tid; c < DP_H; c += 512) {
        const bool ok = c < p.C;
        const float g = ok ? p.gamma1[c] : 0.f, be = ok ? p.beta1[c] : 0.f;
        const float scv = ns.rstd*g, shv = be - ns.mean*ns.rstd*g;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float w = ok ? p.taps[c*3 + k] : 0.f;
          tabs[k*DP_H + c] = w*c1*scv; tabs[(3 + k)*DP_H + c] = w*shv;
        }
        tabs[6*DP_H + c] = ok ? p.dbias[c] : 0.f;
        tabs[7*DP_H + c] = tabs[6*DP_H + c] + tabs[3*DP_H + c] + tabs[4*DP_H + c] + tabs[5*DP_H + c];
      }
      cur_item = b;
      __syncthreads();
    }
    float ts = 0.f, tq = 0.f;
    const bool interior = t0 - p.left >= 0 && t0 + DP_TT - 1 + 2*p.dil - p.left < T;
    auto stage = [&](int s, const uint4 (&raw)[2][3], int buf) {
      const int cb = s*DP_SLAB + o*8;
      auto ld8 = [&](int which, float (&v)[8]) {
        const float4 x0 = *reinterpret_cast<const float4*>(tabs + which*DP_H + cb);
        const float4 x1 = *reinterpret_cast<const float4*>(tabs + which*DP_H + cb + 4);
        v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w;
        v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
      };
      float acc[2][8];
      {
        // constant terms: bias + the taps inside the item -- all three for every frame of an
        // interior tile (workgroup-uniform), else per tap and frame
        float bs[8];
        ld8(interior ? 7 : 6, bs);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[r][j] = bs[j];
      }
#pragma unroll
      for (int k = 0; k < ((DP_ABL & 8) ? 1 : 3); ++k) {
        float wa[8];
        ld8(k, wa);
        if (!interior) {
          float wc[8];
          ld8(3 + k, wc);
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const int ti = t0 + rl + 64*r + k*p.dil - p.left;
            const float in = (ti >= 0 && ti < T) ? 1.f : 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[r][j] = __builtin_fmaf(in, wc[j], acc[r][j]);
          }
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          float f[8];
          unpack8(raw[r][k], f);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            acc[r][j] = __builtin_fmaf(wa[j], __builtin_fmaf(rho, __builtin_fabsf(f[j]), f[j]), acc[r][j]);
          }
        }
      }
      bf16_t* pb = reinterpret_cast<bf16_t*>(smem + DP_OFF_P + buf*DP_PBYTES);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int row = rl + 64*r;
        const uint4 q = pack8(acc[r]);
        if (!(DP_ABL & 2)) buf_store16(rz2, (unsigned int)(t0 + row)*rowb + (unsigned int)(cb*2), q);   // t >= T: dropped
        float rr[8], pv[8];
        unpack8(q, rr);
        const float live = (t0 + row < T) ? 1.f : 0.f;
        float fs = 0.f, fq = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          pv[j] = live*__builtin_fmaf(d2, __builtin_fabsf(rr[j]), d1*rr[j]);
          fs += pv[j]; fq = __builtin_fmaf(pv[j], pv[j], fq);
        }
        ts += fs; tq += fq;
        *reinterpret_cast<uint4*>(pb + row*DP_LDP + o*8) = pack8(pv);
      }
    };

    // wave wn: outputs 32 wn .. + 31 (ONE weight fragment per k-step, fetched by this wave only: the
    // weights are re-read from L2 for every tile, 256 KB per 128 frames -- with two waves per slice
    // that stream, not the matrix pipe, set the pace: 29 us of the launch) x 4 frame fragments
    f32x16 acc[4];
#pragma unroll
    for (int fr = 0; fr < 4; ++fr)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[fr][i] = 0.f;
    // the slab's four weight fragments are requested one slab ahead (an L2 round trip otherwise
    // opens every slab)
    const bf16_t* wa_ = p.Wp + ((long long)wn*32*DP_H) + lane*8;
    bf16x8 wcur[4], wnxt[4];
    auto load_w = [&](int s, bf16x8 (&w)[4]) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        w[ks] = *reinterpret_cast<const bf16x8*>(wa_ + (long long)(s*4 + ks)*64*8);
    };
    auto mfma_slab = [&](const bf16x8 (&w)[4], int buf) {
      if (DP_ABL & 1) return;
      const bf16_t* pb = reinterpret_cast<const bf16_t*>(smem + DP_OFF_P + buf*DP_PBYTES);
#pragma unroll
      for (int ks = 0; ks < DP_SLAB/16; ++ks) {
#pragma unroll
        for (int fr = 0; fr < 4; ++fr) {
          const bf16x8 bv = *reinterpret_cast<const bf16x8*>(pb + (32*fr + n32)*DP_LDP + ks*16 + khalf*8);
          acc[fr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[ks], bv, acc[fr], 0, 0, 0);
        }
      }
    };

    // ---- slab pipeline: z1 taps, weight fragments and the stage one slab ahead of the MFMAs.
    // Measured alternatives: z1 taps two slabs ahead (13 spills, +8 %), one frame per
    // thread with the taps four slabs ahead (+12 %), 64-frame tiles with two workgroups per CU (equal).
    load_w(0, wcur);
    stage(0, raw, 0);
    load_raw(1, raw);
    __syncthreads();
#pragma unroll 1
    for (int s = 0; s < DP_NSLAB; s += 2) {          // two slabs per trip: the buffer parities are static
      load_w(s + 1, wnxt);
      mfma_slab(wcur, 0);
      stage(s + 1, raw, 1);
      if (s + 2 < DP_NSLAB) load_raw(s + 2, raw);
      __syncthreads();
      if (s + 2 < DP_NSLAB) load_w(s + 2, wcur);
      mfma_slab(wnxt, 1);
      if (s + 2 < DP_NSLAB) {
        stage(s + 2, raw, 0);
        load_raw(s + 3, raw);
      }
      __syncthreads();
    }
    s_sum += (double)ts; s_sq += (double)tq;

    // ---- u: D[out][frame] -> LDS [frame][256 outs] (two halves of 64 frames) -> 512-byte rows
    bf16_t* stg = reinterpret_cast<bf16_t*>(smem + DP_OFF_P);
    bf16_t* ub = p.u + (long long)b*T*DP_N;
#pragma unroll
    for (int h = 0; h < ((DP_ABL & 4) ? 0 : 2); ++h) {       // frames 64 h .. 64 h + 63 = fragments 2h, 2h + 1
#pragma unroll
      for (int f2 = 0; f2 < 2; ++f2)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = acc[2*h + f2][g*4 + j];
          uint2 q;
          q.x = pack2(v[0], v[1]); q.y = pack2(v[2], v[3]);
          *reinterpret_cast<uint2*>(stg + (32*f2 + n32)*DP_LDU + 32*wn + g*8 + khalf*4) = q;
        }
      __syncthreads();
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int e = it*512 + tid, fr = e >> 5, c8 = e & 31;
        const int t = t0 + 64*h + fr;
        if (t < T)
          *reinterpret_cast<uint4*>(ub + (long long)t*DP_N + c8*8) =
              *reinterpret_cast<const uint4*>(stg + fr*DP_LDU + c8*8);
      }
      __syncthreads();
    }
  }
  flush_stats();
}

}  // namespace brv
