// Pointwise ("1x1 conv" / filterbank) GEMM over frames, bf16 MFMA, fp32 accumulate.
//
//   D[b][t][n] = sum_k A[b][t][k] * W[n][k]        t < T, n < Np, k < Kp
//
// A is a channels-last activation (or the framed waveform), W a prepared bf16
// weight [Np][Kp] (row n contiguous in k), so both MFMA operands are k-contiguous
// and no transpose is needed. One workgroup (4 waves) owns a 128-frame x BN-channel
// tile of ONE batch item; wave w owns frames [32w, 32w+32) of the tile.
//
// Prologue (A operand, applied while staging global -> registers -> LDS):
//   optional PReLU, optional global-layer-norm affine (per item scale/shift).
// Epilogue (EMODE): the accumulator tile is dumped to LDS as fp32 and re-read
// row-wise so companion tensors are accessed with 16-byte vectors.
//
// Replaces (reference): nn.Conv1d(k=1) in brever/models/convtasnet/convtasnet.py
// :160-164,182-186,209-213,225-234, Encoder conv :107-126, Decoder :136-150, and
// the autograd data-gradients of the same ops.
#pragma once
#include "common.cuh"

namespace brv {

// A_DZ: A = gLN/PReLU backward of (e = p0, z = p1), computed while staging and written back
// over e (the separate gln_prelu_bwd pass of the first 1x1 conv's data gradient, fused)
enum AKind { A_BF16 = 0, A_F32 = 1, A_FRAMES = 2, A_DZ = 3 };
enum EMode {
  E_STORE = 0,     // + bias, store bf16, optional gLN statistics of [prelu](out)
  E_RES_SKIP = 1,  // res: out = res_in + v ; skip: skip (+)= v
  E_MASK = 2,      // m = sigmoid(v + bias); y = m * w
  E_OLA = 3,       // overlap-add of synthesis frames into the waveform (atomics)
  E_GLN_BWD = 4,   // e = gamma*dy, partial sums for the gLN backward, dgamma/dbeta
  E_ADD = 5,       // out = add_in + v
  E_MASK_BWD = 6,  // d_pre = dy*w*m*(1-m); dw1 = dy*m
  E_PRELU_BWD = 7  // g = dy * prelu'(src); dslope
};

struct ASpec {
  const void* p0; const void* p1;
  int ld0, ld1;            // row strides in elements
  long long bs0, bs1;      // batch strides in elements
  int K0;                  // padded channels served by p0 (multiple of 64); rest from p1
  int nsrc;                // > 1: k spans nsrc tensors of ld0 channels laid out as
                           //      batch rows (b*nsrc + s) of p0 (K0/p1 ignored)
  // A_FRAMES: A[t][k] = wav[b*wav_stride + t*hop + k], k < Kf, index < wav_len
  int hop, Kf; long long wav_stride; int wav_len;
  // transform
  const float* slope;      // PReLU slope (scalar) or null
  const double* stats;     // [B][2] or null (no affine)
  const float* gamma; const float* beta; int C;   // true channel count
  double inv_n; float eps;
  const double* sums;      // A_DZ: per-item {sum e, sum e*xh} of the gLN backward
  // gemm_ws AT == 2 (lazy residual): A = p0 + rstd_b*p1 + (lazy_v0 - mean_b*rstd_b*lazy_v1), the
  // finished rows are written to xout (row stride ld0); `stats` are those of the producing
  // block's second norm
  void* xout; const float* lazy_v0; const float* lazy_v1;
  // gemm_ws AT == 3 (fused depthwise stage): A = PReLU_2(z2), z2 = dconv(gLN_1(PReLU_1(p0))) + bias
  // built while staging; z2 goes to z2out (row stride ld0), the statistics of PReLU_2(z2) to stats2_out
  void* z2out; const float* taps; const float* dbias; const float* slope2; double* stats2_out;
  int dil, left;
};

struct EpiSpec {
  void* out; int ldo;
  const float* bias; const float* bias2; int N, Nsplit, N2;
  double* stats_out; const float* stats_slope;
  const bf16_t* res_in; int ld_res; float* skip; int ld_skip; int skip_init;
  const bf16_t* w_in; int ld_w; bf16_t* m_out; int S; int Np_src;
  float* wave_out; int hop, Kf; long long wave_stride; int wave_len;
  const bf16_t* src; int ld_src; const float* src_slope; const double* src_stats;
  double inv_n; float eps;
  const float* gamma; float* dgamma; float* dbeta; double* sums_out;
  long long rep_stride; int n_rep;      // dgamma/dbeta replicas (see tcn_kernels.cuh)
  const bf16_t* add_in; int ld_add;
  const bf16_t* m_in; bf16_t* out2;
  const float* src_f32; int ld_srcf; float* dslope;
  // E_ADD, fused backward (bwd_fused.cuh): `out` is the g_out of the PREVIOUS block, whose layer-norm
  // backward means are functions of g = [g_out | g_skip] and that block's stored u = (W gamma_2) p:
  // gu_out[item] += {sum <g, v1>, sum <g, u>} over both halves (columns ncol and Np + ncol)
  const bf16_t* gu_u; int ld_gu; const bf16_t* gu_gskip; int ld_gs; const float* gu_v1; double* gu_out;
};

struct GemmRowsParams {
  ASpec a;
  const bf16_t* W;         // [Np][Kp]
  const bf16_t* Wp;        // optional: W in MFMA-fragment order for slices of wp_nsl rows
  int wp_nsl;              // (persistent kernels, gemm_ws.cuh)
  int T, Np, Kp;
  int n_ttiles, n_ntiles, batch;   // filled by the launcher (1-D XCD-aware grid)
#ifdef BRV_DIAG                    // make DIAG=1: ablation flags + cycle stamps (tools/ablate.py)
  int dbg;
  long long* dbg_out;
#endif
  EpiSpec e;
};

// 1-D grid -> (frame tile, channel tile, item). Workgroups are dealt round-robin over
// the 8 XCDs (id % 8 share an XCD and its private L2), so the work list is cut into 8
// contiguous chunks, one per XCD, with the channel tile fastest: the channel tiles
// that re-read one A tile then run back to back on the same L2. Speed only -- any
// placement computes the same result.
__device__ __forceinline__ int xcd_remap(int id, int nwg) {
  const int xcd = id & 7, slot = id >> 3;
  const int q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd*(q + 1) : r*(q + 1) + (xcd - r)*q) + slot;
}

constexpr int GR_BM = 128;
constexpr int GR_BK = 64;
constexpr int GR_LDS_K = GR_BK + 8;        // 144-byte rows: conflict-free ds_read_b128

template <int BN>
struct GemmRowsSmem {
  static constexpr int kMain = (GR_BM + BN)*GR_LDS_K*2;
  static constexpr int kLdc = BN + 4;
  static constexpr int kEpi = GR_BM*kLdc*4;
  static constexpr int kBytes = kMain > kEpi ? kMain : kEpi;
};

template <int BN, int AK, int EM>
__global__ __launch_bounds__(256) void gemm_rows_kernel(const GemmRowsParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[GemmRowsSmem<BN>::kBytes];
  bf16_t* As = reinterpret_cast<bf16_t*>(smem);
  bf16_t* Ws = As + GR_BM*GR_LDS_K;
  float* Cs = reinterpret_cast<float*>(smem);
  constexpr int LDC = GemmRowsSmem<BN>::kLdc;
  constexpr int NT = BN/32;                 // 32x32 MFMA tiles per wave
  constexpr int WCH = BN/32;                // W chunks per thread

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int vid = xcd_remap(blockIdx.x, gridDim.x);
  const int n_tile = vid % p.n_ntiles;
  const int rest = vid / p.n_ntiles;
  const int b = rest / p.n_ttiles;
  const int t0 = (rest % p.n_ttiles)*GR_BM;
  const int n0 = n_tile*BN;
  const int T = p.T;

  // ---- per-thread staging geometry --------------------------------------
  const int kc = tid & 7;                   // 8-element k chunk within the k tile
  const int arow = tid >> 3;                // + 32*ci
  const ASpec& a = p.a;

  NormStat ns = {0.f, 1.f};
  const bool affine = AK != A_DZ && a.stats != nullptr;
  if (a.stats != nullptr) ns = norm_stat(a.stats, b, a.inv_n, a.eps);
  const float slope = a.slope ? *a.slope : 1.f;
  const bool has_prelu = AK != A_DZ && a.slope != nullptr;
  // A_DZ: dz = prelu'(z)*rstd*(e - m1 - xh*m2), xh = (prelu(z) - mean)*rstd
  float dz_m1 = 0.f, dz_m2 = 0.f, dz_da = 0.f;
  if (AK == A_DZ) {
    dz_m1 = (float)(a.sums[stat_sum(b)]*a.inv_n);
    dz_m2 = (float)(a.sums[stat_sq(b)]*a.inv_n);
  }
  uint4 araw_z[(AK == A_DZ) ? 4 : 1];

  uint4 araw_b[4];
  float araw_f[(AK == A_BF16) ? 1 : 4][8];
  float asc[8], ash[8];
  uint4 wraw[WCH];

  auto load_tile = [&](int kt) {
    const int kbase = kt*GR_BK + kc*8;
    if (AK == A_FRAMES) {
      const float* wav = reinterpret_cast<const float*>(a.p0) + (long long)b*a.wav_stride;
#pragma unroll
      for (int ci = 0; ci < 4; ++ci) {
        const int t = t0 + arow + 32*ci;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = kbase + j;
          const long long idx = (long long)t*a.hop + k;
          float v = 0.f;
          if (t < T && k < a.Kf && idx < a.wav_len) v = wav[idx];
          araw_f[(AK == A_BF16) ? 0 : ci][j] = v;
        }
      }
    } else {
      bool first = kbase < a.K0;
      int ld = first ? a.ld0 : a.ld1;
      int kk = first ? kbase : kbase - a.K0;
      long long boff = (long long)b*(first ? a.bs0 : a.bs1);
      if (a.nsrc > 1) {
        first = true; ld = a.ld0;
        const int s = kbase / a.ld0;
        kk = kbase - s*a.ld0;
        boff = ((long long)b*a.nsrc + s)*a.bs0;
      }
#pragma unroll
      for (int ci = 0; ci < 4; ++ci) {
        const int t = t0 + arow + 32*ci;
        const long long off = boff + (long long)t*ld + kk;
        if (AK == A_DZ) {
          const long long o0 = (long long)b*a.bs0 + (long long)t*a.ld0 + kbase;
          const long long o1 = (long long)b*a.bs1 + (long long)t*a.ld1 + kbase;
          araw_b[ci] = (t < T) ? *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(a.p0) + o0)
                               : make_uint4(0, 0, 0, 0);
          araw_z[(AK == A_DZ) ? ci : 0] =
              (t < T) ? *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(a.p1) + o1)
                      : make_uint4(0, 0, 0, 0);
        } else if (AK == A_BF16) {
          const bf16_t* src = reinterpret_cast<const bf16_t*>(first ? a.p0 : a.p1);
          araw_b[ci] = (t < T) ? *reinterpret_cast<const uint4*>(src + off)
                               : make_uint4(0, 0, 0, 0);
        } else {
          const float* src = reinterpret_cast<const float*>(first ? a.p0 : a.p1);
          float4 lo = make_float4(0, 0, 0, 0), hi = lo;
          if (t < T) {
            lo = *reinterpret_cast<const float4*>(src + off);
            hi = *reinterpret_cast<const float4*>(src + off + 4);
          }
          float* d = araw_f[(AK == A_BF16) ? 0 : ci];
          d[0] = lo.x; d[1] = lo.y; d[2] = lo.z; d[3] = lo.w;
          d[4] = hi.x; d[5] = hi.y; d[6] = hi.z; d[7] = hi.w;
        }
      }
    }
    if (affine) {
      float g8[8], b8[8];
      load8_masked(a.gamma, kbase, a.C, g8);
      load8_masked(a.beta, kbase, a.C, b8);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        asc[j] = ns.rstd*g8[j];
        ash[j] = b8[j] - ns.mean*ns.rstd*g8[j];
      }
    }
#pragma unroll
    for (int ci = 0; ci < WCH; ++ci) {
      const int c = tid + 256*ci;
      const int wr = c >> 3;
      const int n = n0 + wr;
      wraw[ci] = (n < p.Np)
          ? *reinterpret_cast<const uint4*>(p.W + (long long)n*p.Kp + kt*GR_BK + kc*8)
          : make_uint4(0, 0, 0, 0);
    }
  };

  auto store_tile = [&](int kt) {
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
      const int row = arow + 32*ci;
      uint4 q;
      if (AK == A_DZ) {
        float ev[8], zv[8], o[8];
        unpack8(araw_b[ci], ev);
        unpack8(araw_z[(AK == A_DZ) ? ci : 0], zv);
        const int kbase = kt*GR_BK + kc*8;
        const bool live = t0 + row < T;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const bool pos = zv[j] > 0.f;
          const float pv = pos ? zv[j] : slope*zv[j];
          const float xh = (pv - ns.mean)*ns.rstd;
          const float dh = ns.rstd*(ev[j] - dz_m1 - xh*dz_m2);
          const bool ok = live && kbase + j < a.C;
          o[j] = ok ? (pos ? dh : slope*dh) : 0.f;
          if (ok && !pos) dz_da += dh*zv[j];
        }
        q = pack8(o);
        if (live)       // in place: this workgroup is the only reader of these elements
          *reinterpret_cast<uint4*>(const_cast<bf16_t*>(reinterpret_cast<const bf16_t*>(a.p0))
              + (long long)b*a.bs0 + (long long)(t0 + row)*a.ld0 + kbase) = q;
      } else if (AK == A_BF16 && !affine && !has_prelu) {
        q = araw_b[ci];
      } else {
        float f[8];
        if (AK == A_BF16) unpack8(araw_b[ci], f);
        else {
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = araw_f[(AK == A_BF16) ? 0 : ci][j];
        }
        if (has_prelu) {
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = prelu(f[j], slope);
        }
        if (affine) {
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = f[j]*asc[j] + ash[j];
        }
        if (t0 + row >= T) {
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = 0.f;
        }
        q = pack8(f);
      }
      *reinterpret_cast<uint4*>(As + row*GR_LDS_K + kc*8) = q;
    }
#pragma unroll
    for (int ci = 0; ci < WCH; ++ci) {
      const int c = tid + 256*ci;
      *reinterpret_cast<uint4*>(Ws + (c >> 3)*GR_LDS_K + kc*8) = wraw[ci];
    }
  };

  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;

  const int nk = p.Kp/GR_BK;
  const int fr = lane & 31, fh = lane >> 5;
  load_tile(0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
    store_tile(kt);
    __syncthreads();
    if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
    for (int s = 0; s < GR_BK/16; ++s) {
      const bf16x8 af = *reinterpret_cast<const bf16x8*>(
          As + (32*wid + fr)*GR_LDS_K + 16*s + 8*fh);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(
            Ws + (32*j + fr)*GR_LDS_K + 16*s + 8*fh);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[j], 0, 0, 0);
      }
    }
  }

  // ---- phase 1: accumulators -> LDS (fp32, [frame][channel]) -------------
  __syncthreads();
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = 32*wid + (i & 3) + 8*(i >> 2) + 4*fh;
      Cs[row*LDC + 32*j + fr] = acc[j][i];
    }
  __syncthreads();

  // ---- phase 2: row-wise epilogue ----------------------------------------
  constexpr int NCH = BN/8;                 // 8-column chunks per row
  constexpr int RG = 256/NCH;               // row groups
  const int ch = tid % NCH, rg = tid / NCH;
  const int ncol = n0 + ch*8;               // first global column of the chunk
  const EpiSpec& e = p.e;
  const bool col_ok = ncol < p.Np;

  if (EM == E_OLA && p.n_ntiles == 1 && e.hop > 0) {
    // Overlap-add inside the tile: output sample s (relative to the tile's first frame) is the sum
    // of the frames that cover it, read straight from the staged accumulators. Samples whose frames
    // all lie in this tile are stored (the waveform is zeroed before the launch); the Kf - hop
    // samples shared with a neighbouring tile are added atomically. (One atomic per product element
    // made this launch 80 us: ~64 adds serialised on every 128-byte line of the waveform.)
    const int rows = T - t0 < GR_BM ? T - t0 : GR_BM;
    const int hop = e.hop, Kf = e.Kf;
    const int nsamp = (rows - 1)*hop + Kf;
    float* wo = e.wave_out + (long long)b*e.wave_stride;
    const bool first = t0 == 0, last = t0 + rows >= T;
    for (int s = tid; s < nsamp; s += 256) {
      int r_hi = s / hop; if (r_hi > rows - 1) r_hi = rows - 1;
      int r_lo = s - Kf + 1 <= 0 ? 0 : (s - Kf + hop)/hop;
      float acc_s = 0.f;
      for (int r = r_lo; r <= r_hi; ++r) acc_s += Cs[r*LDC + (s - r*hop)];
      const long long idx = (long long)t0*hop + s;
      if (idx >= e.wave_len) continue;
      const bool owned = (first || s >= Kf - hop) && (last || s < rows*hop);
      if (owned) wo[idx] = acc_s; else atomic_add_f32(wo + idx, acc_s);
    }
    return;
  }

  double st_sum = 0.0, st_sq = 0.0;         // E_STORE stats / E_GLN_BWD sums
  float colA[8], colB[8];                   // per-column partials (E_GLN_BWD)
#pragma unroll
  for (int j = 0; j < 8; ++j) { colA[j] = 0.f; colB[j] = 0.f; }
  float red_f = 0.f;                        // E_PRELU_BWD slope gradient

  float biasv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) biasv[j] = 0.f;
  if (EM == E_STORE || EM == E_RES_SKIP) {
    // Nsplit is a multiple of 64, so an 8-column chunk never straddles it
    if (EM == E_RES_SKIP && ncol >= e.Nsplit) load8_masked(e.bias2, ncol - e.Nsplit, e.N2, biasv);
    else load8_masked(e.bias, ncol, e.N, biasv);
  }
  int msrc = 0, mf = 0;                     // E_MASK: source index / filter index
  if (EM == E_MASK) {
    msrc = ncol / e.Np_src; mf = ncol % e.Np_src;
    load8_masked(e.bias ? e.bias + msrc*e.N : nullptr, mf, e.N, biasv);
  }
  float guv[(EM == E_ADD) ? 16 : 1];        // E_ADD + gu_out: v1 of this chunk's columns, both halves
  if (EM == E_ADD && e.gu_out) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      guv[(EM == E_ADD) ? j : 0] = e.gu_v1[ncol + j];
      guv[(EM == E_ADD) ? 8 + j : 0] = e.gu_v1[p.Np + ncol + j];
    }
  }
  NormStat es = {0.f, 1.f};
  float gam[8];
  if (EM == E_GLN_BWD) {
    es = norm_stat(e.src_stats, b, e.inv_n, e.eps);
    load8_masked(e.gamma, ncol, e.N, gam);
  }
  const float eslope = (EM == E_GLN_BWD && e.src_slope) ? *e.src_slope
                     : (EM == E_PRELU_BWD ? *e.src_slope
                     : (EM == E_STORE && e.stats_slope ? *e.stats_slope : 1.f));

  for (int row = rg; row < GR_BM; row += RG) {
    const int t = t0 + row;
    if (t >= T || !col_ok) continue;
    float v[8];
    {
      const float4 lo = *reinterpret_cast<const float4*>(Cs + row*LDC + ch*8);
      const float4 hi = *reinterpret_cast<const float4*>(Cs + row*LDC + ch*8 + 4);
      v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w;
      v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
    }
    const long long rowi = (long long)b*T + t;
    if (EM == E_STORE) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += biasv[j];
      const uint4 q = pack8(v);
      *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(e.out) + rowi*e.ldo + ncol) = q;
      if (e.stats_out) {
        float r[8]; unpack8(q, r);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (ncol + j < e.N) {
            const float pv = e.stats_slope ? prelu(r[j], eslope) : r[j];
            st_sum += pv; st_sq += (double)pv*pv;
          }
        }
      }
    } else if (EM == E_RES_SKIP) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += biasv[j];
      if (ncol < e.Nsplit) {
        float r[8];
        unpack8(*reinterpret_cast<const uint4*>(e.res_in + rowi*e.ld_res + ncol), r);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += r[j];
        *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(e.out) + rowi*e.ldo + ncol) = pack8(v);
      } else {
        float* sp = e.skip + rowi*e.ld_skip + (ncol - e.Nsplit);
        if (!e.skip_init) {
          const float4 lo = *reinterpret_cast<const float4*>(sp);
          const float4 hi = *reinterpret_cast<const float4*>(sp + 4);
          v[0] += lo.x; v[1] += lo.y; v[2] += lo.z; v[3] += lo.w;
          v[4] += hi.x; v[5] += hi.y; v[6] += hi.z; v[7] += hi.w;
        }
        *reinterpret_cast<float4*>(sp) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(sp + 4) = make_float4(v[4], v[5], v[6], v[7]);
      }
    } else if (EM == E_MASK) {
      float w[8], y[8];
      unpack8(*reinterpret_cast<const uint4*>(e.w_in + rowi*e.ld_w + mf), w);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float m = 1.f/(1.f + __expf(-(v[j] + biasv[j])));
        v[j] = m; y[j] = m*w[j];
      }
      const long long orow = ((long long)b*e.S + msrc)*T + t;
      *reinterpret_cast<uint4*>(e.m_out + orow*e.ldo + mf) = pack8(v);
      *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(e.out) + orow*e.ldo + mf) = pack8(y);
    } else if (EM == E_OLA) {
      float* wo = e.wave_out + (long long)b*e.wave_stride;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int n = ncol + j;
        const long long idx = (long long)t*e.hop + n;
        if (n < e.Kf && idx < e.wave_len) atomic_add_f32(wo + idx, v[j]);
      }
    } else if (EM == E_GLN_BWD) {
      float s[8], o[8];
      unpack8(*reinterpret_cast<const uint4*>(e.src + rowi*e.ld_src + ncol), s);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float pv = e.src_slope ? prelu(s[j], eslope) : s[j];
        const float xh = (pv - es.mean)*es.rstd;
        const float ev = gam[j]*v[j];
        o[j] = ev;
        st_sum += ev; st_sq += (double)ev*xh;
        colA[j] += v[j]*xh; colB[j] += v[j];
      }
      *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(e.out) + rowi*e.ldo + ncol) = pack8(o);
    } else if (EM == E_ADD) {
      if (e.add_in) {
        float r[8];
        unpack8(*reinterpret_cast<const uint4*>(e.add_in + rowi*e.ld_add + ncol), r);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += r[j];
      }
      const uint4 qa = pack8(v);
      *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(e.out) + rowi*e.ldo + ncol) = qa;
      if (e.out2) *reinterpret_cast<uint4*>(e.out2 + rowi*e.ld_srcf + ncol) = qa;   // copy
      if (e.gu_out) {
        float gr[8], gs[8], ur[8], us[8];
        unpack8(qa, gr);                        // the rounded values the consumer will read
        unpack8(*reinterpret_cast<const uint4*>(e.gu_gskip + rowi*e.ld_gs + ncol), gs);
        unpack8(*reinterpret_cast<const uint4*>(e.gu_u + rowi*e.ld_gu + ncol), ur);
        unpack8(*reinterpret_cast<const uint4*>(e.gu_u + rowi*e.ld_gu + p.Np + ncol), us);
        float d1 = 0.f, d2 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          d1 = __builtin_fmaf(gr[j], guv[(EM == E_ADD) ? j : 0], d1);
          d1 = __builtin_fmaf(gs[j], guv[(EM == E_ADD) ? 8 + j : 0], d1);
          d2 = __builtin_fmaf(gr[j], ur[j], d2);
          d2 = __builtin_fmaf(gs[j], us[j], d2);
        }
        st_sum += d1; st_sq += d2;
      }
    } else if (EM == E_MASK_BWD) {
      // batch index of this GEMM is bs = b_item*S + s
      const long long wrow = (long long)(b / e.S)*T + t;
      float w[8], m[8], dp[8], dw[8];
      unpack8(*reinterpret_cast<const uint4*>(e.w_in + wrow*e.ld_w + ncol), w);
      unpack8(*reinterpret_cast<const uint4*>(e.m_in + rowi*e.ldo + ncol), m);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        dp[j] = v[j]*w[j]*m[j]*(1.f - m[j]);
        dw[j] = v[j]*m[j];
      }
      *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(e.out) + rowi*e.ldo + ncol) = pack8(dp);
      *reinterpret_cast<uint4*>(e.out2 + rowi*e.ldo + ncol) = pack8(dw);
    } else if (EM == E_PRELU_BWD) {
      const float* sp = e.src_f32 + rowi*e.ld_srcf + ncol;
      const float4 lo = *reinterpret_cast<const float4*>(sp);
      const float4 hi = *reinterpret_cast<const float4*>(sp + 4);
      const float s[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool pos = s[j] > 0.f;
        o[j] = pos ? v[j] : eslope*v[j];
        if (!pos) red_f += v[j]*s[j];
      }
      *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(e.out) + rowi*e.ldo + ncol) = pack8(o);
    }
  }

  // ---- block-level reductions ----------------------------------------------
  if ((EM == E_STORE && e.stats_out) || EM == E_GLN_BWD || (EM == E_ADD && e.gu_out)) {
    __syncthreads();
    double* dscr = reinterpret_cast<double*>(smem);
    const double s0 = block_sum(st_sum, dscr);
    const double s1 = block_sum(st_sq, dscr + 8);
    double* dst = (EM == E_STORE) ? e.stats_out : (EM == E_ADD ? e.gu_out : e.sums_out);
    if (tid == 0) { atomic_add_f64(dst + stat_sum(b), s0); atomic_add_f64(dst + stat_sq(b), s1); }
  }
  if (AK == A_DZ) {                          // slope gradient of the fused gLN/PReLU backward
    __syncthreads();
    float* fscr = reinterpret_cast<float*>(smem);
    const float s0 = block_sum(dz_da, fscr);
    if (tid == 0 && e.dslope) {
      const long long ro = e.n_rep > 1 ? (long long)(blockIdx.x % e.n_rep)*e.rep_stride : 0;
      atomic_add_f32(e.dslope + ro, s0);
    }
  }
  if (EM == E_PRELU_BWD) {
    __syncthreads();
    float* fscr = reinterpret_cast<float*>(smem);
    const float s0 = block_sum(red_f, fscr);
    if (tid == 0) {
      const long long ro = e.n_rep > 1 ? (long long)(blockIdx.x % e.n_rep)*e.rep_stride : 0;
      atomic_add_f32(e.dslope + ro, s0);
    }
  }
  if (EM == E_GLN_BWD) {
    __syncthreads();
    float* sc = reinterpret_cast<float*>(smem) + 64;   // [2][RG][BN]
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sc[rg*BN + ch*8 + j] = colA[j];
      sc[RG*BN + rg*BN + ch*8 + j] = colB[j];
    }
    __syncthreads();
    if (tid < BN) {
      float a0 = 0.f, b0 = 0.f;
      for (int r = 0; r < RG; ++r) { a0 += sc[r*BN + tid]; b0 += sc[RG*BN + r*BN + tid]; }
      const int n = n0 + tid;
      const long long ro = e.n_rep > 1 ? (long long)(blockIdx.x % e.n_rep)*e.rep_stride : 0;
      if (n < e.N) {
        atomic_add_f32(e.dgamma + ro + n, a0);
        atomic_add_f32(e.dbeta + ro + n, b0);
      }
    }
  }
}

}  // namespace brv
