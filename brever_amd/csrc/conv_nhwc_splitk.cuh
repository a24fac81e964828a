// Low-resolution launches of the channels-last 3x3 convolution: the REDUCTION split over workgroups (round 6).
//
// conv_nhwc_kernel gets its parallelism from pixels: a workgroup owns 128 output channels x (4 .. 16 rows x 32
// columns) and walks ALL input channels, streaming the whole weight set of its channel block (72 KB per 32 input
// channels) once per tile. The inner levels of the SGMSE+ U-Net at batch 1 are 64 x 126 ... 4 x 8 pixels with 256
// channels: 128, 32, 8, 4, 2 tiles = workgroups, each bound by its serial weight stream (0.6 .. 1.2 MB at the
// ~25 GB/s one CU takes in): 37.7 us per launch whatever the level, 888 launches = 42 % of the kernel time of a
// batch-1 `enhance` (profiles/r05_rows_sgmse_b1_kernel_stats.csv; MFMA busy 7 %, 0.55 waves per SIMD).
//
// Here a workgroup takes (tile of 4 rows x 32 columns, block of 128 output channels, `cpw` chunks of 32 input
// channels): a few hundred workgroups each stream 72 .. 288 KB of weights instead of 2 .. 128 streaming 0.6 .. 1.2
// MB: the chip reads the weight set in parallel. Partial sums go to a caller-provided fp32 scratch ([split][channel
// quad][pixel][4]: 512-byte runs per store instruction); conv_nhwc_combine_kernel adds the splits IN ORDER (no
// atomics on the data: bitwise repeatable), then bias + residual, * scale, fp16 channels-last, and the per-channel
// statistics of the next GroupNorm.
//
// Same operand conventions as conv_nhwc_kernel (the packed weights of brv_conv_nhwc_pack are read as they are):
// A = weight fragments [k-step 2][co fragment 4][lane 64][8] straight from L2 to registers, a whole chunk (9 taps) in
// flight (each wave fetches its own 2 x 2 fragments of a tap: 4 KB per wave and tap; the four waves that share a
// channel half hit the same lines in L1); B = the (6 x 34)-pixel patch of the chunk in LDS, pixel-major, the four 16-byte channel
// octets of a pixel XOR-swizzled by ((column >> 2) & 3). The patch goes through REGISTERS here (two 16-byte pieces
// per thread, the next chunk's requested before this chunk's products): the folded GroupNorm (+ SiLU) and the
// zero padding are applied on the way, one pass, no LDS rewrite; two patch buffers, one barrier per chunk.
// Reference: brever/models/sgmse/net.py:352-422 (UNetBlock.conv_1 / conv_2 under fp16 autocast).
#pragma once

struct ConvSplitParams {
  const _Float16* x1; const _Float16* x2;
  const unsigned char* wp;
  const float* in_scale; const float* in_shift;          // FOLD == 1: (B, Cin) each
  const double* sums1; const double* sums2;              // FOLD == 2: per-channel (sum, sum of squares)
  const float* gn_add; const float* gn_gamma; const float* gn_beta;
  const float* adm_scale; const float* adm_shift;
  int C1, C2, groups; float eps;
  float* part;                                           // [n_split][n_cob*32][B*H*W][4]
  int B, H, W, C1s, C2s, n_chunks1, n_chunks, Cin, n_wt, n_ht, n_cob, cpw, in_silu;
};

constexpr int CS_ROWS = 4, CS_PR = CS_ROWS + 2, CS_PC = CN_COLS + 2, CS_NPX = CS_PR*CS_PC, CS_NSLOT = CS_NPX*4;
constexpr int CS_MAXCPW = 8;                              // chunks per workgroup (size of the fold table)
#ifndef BRV_CONV_SPLIT_MAX_BASE
#define BRV_CONV_SPLIT_MAX_BASE 64      // (tiles x channel blocks) of the launch up to which the reduction is split (128 takes
                                        // in the 64 x 126 level at batch 1: 25 MB of partial sums, 56 us against 47: 5.18 against 4.96 ms per evaluation)
#endif
#ifndef BRV_CONV_SPLIT_TARGET
#define BRV_CONV_SPLIT_TARGET 256       // workgroups a split launch aims for (384 / 768: no faster)
#endif
#ifndef BRV_CONV_SPLIT_MAX
#define BRV_CONV_SPLIT_MAX 8            // splits at most (scratch traffic: 1 KB per pixel, split and 256 channels)
#endif

template <int FOLD>
__global__ __launch_bounds__(CN_THREADS) void conv_nhwc_splitk_kernel(const ConvSplitParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char patch[2][CS_NSLOT*16];
  __shared__ __attribute__((aligned(16))) float ftab[2][CS_MAXCPW*CN_CK];      // scale | shift of this workgroup's channels
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wco = wave >> 2, wpx = wave & 3;
  const int n32 = lane & 31, khalf = lane >> 5;
  int t = blockIdx.x;
  const int w0 = (t % p.n_wt)*CN_COLS; t /= p.n_wt;
  const int h0 = (t % p.n_ht)*CS_ROWS;
  const int b = t / p.n_ht;
  const int cob = blockIdx.y, split = blockIdx.z;
  const int c_lo = split*p.cpw, c_hi = c_lo + p.cpw < p.n_chunks ? c_lo + p.cpw : p.n_chunks;
  const int n_my = c_hi - c_lo;
  const long long hw = (long long)p.H*p.W;

  // ---- patch pieces of this thread: slot = r*512 + tid = 4*pixel + position; position q holds channel octet
  // q ^ ((pcol >> 2) & 3) of patch pixel (prow, pcol)
  int pk[2], koct[2];                      // (pixel index inside the item) << 2 | octet; -1 = zero padding, -2 = no slot
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int slot = r*CN_THREADS + tid;
    const int px = slot >> 2;
    const int pcol = px % CS_PC, prow = px / CS_PC;
    const int kg = (slot & 3) ^ ((pcol >> 2) & 3);
    const int h = h0 + prow - 1, w = w0 + pcol - 1;
    const bool ok = slot < CS_NSLOT && h >= 0 && h < p.H && w >= 0 && w < p.W;
    pk[r] = ok ? (((h*p.W + w) << 2) | kg) : (slot < CS_NSLOT ? -1 : -2);
    koct[r] = kg;
  }
  const _Float16* xb1 = p.x1 + (long long)b*hw*p.C1s;
  const _Float16* xb2 = p.x2 ? p.x2 + (long long)b*hw*p.C2s : nullptr;
  auto load_piece = [&](int r, int chunk) {
    u32x4 v = u32x4{0u, 0u, 0u, 0u};
    if (pk[r] >= 0) {
      const bool second = chunk >= p.n_chunks1;
      const _Float16* base = second ? xb2 : xb1;
      const int cs = second ? p.C2s : p.C1s;
      const int c0 = (second ? chunk - p.n_chunks1 : chunk)*CN_CK + (pk[r] & 3)*8;
      if (c0 < cs) v = *reinterpret_cast<const u32x4*>(base + (long long)(pk[r] >> 2)*cs + c0);
    }
    return v;
  };
  auto store_piece = [&](int r, int buf, int ci, const u32x4& raw) {       // ci: chunk index inside the workgroup
    if (pk[r] == -2) return;
    u32x4 out = raw;
    if (FOLD != 0) {
      const f32x8 xf = __builtin_convertvector(__builtin_bit_cast(h8, raw), f32x8);
      const float* sc = ftab[0] + ci*CN_CK + koct[r]*8;
      const float* sh = ftab[1] + ci*CN_CK + koct[r]*8;
      f32x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v = sc[j]*xf[j] + sh[j];
        const float vs = cn_silu(v);
        v = p.in_silu ? vs : v;
        o[j] = pk[r] >= 0 ? v : 0.f;                      // zero padding pads the ACTIVATED input
      }
      out = __builtin_bit_cast(u32x4, __builtin_convertvector(o, h8));
    }
    *reinterpret_cast<u32x4*>(patch[buf] + (r*CN_THREADS + tid)*16) = out;
  };

  // ---- fragments. B of (tap (kh, kw), k-step ks): pixel (wpx + kh, n32 + kw), octet 2 ks + khalf at position
  // octet ^ ((pcol >> 2) & 3); A of (tap, ks, cf): 16 bytes per lane of the packed tap
  unsigned int b_off[3];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    const int pcol = n32 + kw;
    b_off[kw] = (unsigned int)((wpx*CS_PC + pcol)*64 + ((khalf ^ ((pcol >> 2) & 3)) << 4));
  }
  const unsigned char* wlane = p.wp + (long long)cob*p.n_chunks*9*CN_ASLOT + (wco*2)*1024 + lane*16;
  auto load_a = [&](int chunk, int tap, u32x4 (&a)[2][2]) {
    const unsigned char* w = wlane + ((long long)chunk*9 + tap)*CN_ASLOT;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int cf = 0; cf < 2; ++cf) a[ks][cf] = *reinterpret_cast<const u32x4*>(w + ks*4096 + cf*1024);
  };
  f32x16 acc[2];
#pragma unroll
  for (int cf = 0; cf < 2; ++cf)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[cf][i] = 0.f;

  // All 36 weight fragments of a chunk are requested together (144 registers): with one tap in flight ahead of the
  // products a chunk was nine dependent L2 / Infinity-Cache round trips -- 10 us for the 16 workgroups of a 4 x 8 image.
  u32x4 raw[2], a[9][2][2];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) load_a(c_lo, tap, a[tap]);
#pragma unroll
  for (int r = 0; r < 2; ++r) raw[r] = load_piece(r, c_lo);
  // (the weight fragments and the first patch are in flight while the table below is built)
  // ---- the folded GroupNorm of this workgroup's n_my x 32 input channels: scale | shift
  if (FOLD != 0) {
    const int cpg = FOLD == 2 ? p.Cin/p.groups : 1;
    // the group sums: every thread brings the two moments of ITS channel (one memory round trip for the whole table),
    // the channels of a group meet in LDS. (Each thread walking its group's channels itself was a chain of up to 16
    // dependent round trips in front of everything else: 5 us of the 10 us a 4 x 8 image took.) Groups that do not
    // tile a 32-channel chunk fall back to that walk.
    __shared__ double gmom[2][CS_MAXCPW*CN_CK];
    const bool fast = FOLD == 2 && (CN_CK % cpg) == 0;
    const int c = c_lo*CN_CK + tid;
    const bool mine = tid < n_my*CN_CK && c < p.Cin;
    const long long idx = (long long)b*p.Cin + (mine ? c : 0);
    const double n_px = (double)p.H*(double)p.W;
    float e_c = 0.f;
    if (FOLD == 2 && mine) {
      const double* sp = c < p.C1 ? p.sums1 + (((long long)b*p.C1 + c) << 1)
                                  : p.sums2 + (((long long)b*p.C2 + c - p.C1) << 1);
      e_c = p.gn_add ? p.gn_add[idx] : 0.f;
      const double e = (double)e_c, cs = sp[0], cq = sp[1];
      if (fast) { gmom[0][tid] = cs + n_px*e; gmom[1][tid] = cq + 2.0*e*cs + n_px*e*e; }
    }
    if (fast) __syncthreads();
    if (tid < n_my*CN_CK) {
      float sc = 0.f, sh = 0.f;
      if (mine) {
        if (FOLD == 1) { sc = p.in_scale[idx]; sh = p.in_shift[idx]; }
        else {                                            // the arithmetic of chan_fold_kernel (nhwc.hip) / conv_nhwc_kernel
          const int g0 = (c/cpg)*cpg;
          double s1 = 0.0, s2 = 0.0;
          if (fast) {
            const int l0 = g0 - c_lo*CN_CK;
            for (int k = 0; k < cpg; ++k) { s1 += gmom[0][l0 + k]; s2 += gmom[1][l0 + k]; }
          } else {
            for (int k = 0; k < cpg; ++k) {
              const int ch = g0 + k;
              const double* sp = ch < p.C1 ? p.sums1 + (((long long)b*p.C1 + ch) << 1)
                                           : p.sums2 + (((long long)b*p.C2 + ch - p.C1) << 1);
              const double e = p.gn_add ? (double)p.gn_add[(long long)b*p.Cin + ch] : 0.0;
              const double cs = sp[0], cq = sp[1];
              s1 += cs + n_px*e;
              s2 += cq + 2.0*e*cs + n_px*e*e;
            }
          }
          const double n = (double)cpg*n_px, mean = s1/n;
          double var = s2/n - mean*mean;
          if (var < 0) var = 0;
          const float rstd = (float)(1.0/sqrt(var + (double)p.eps));
          sc = rstd*p.gn_gamma[c];
          sh = p.gn_beta[c] + (e_c - (float)mean)*sc;
          if (p.adm_scale) { const float m = 1.f + p.adm_scale[idx]; sc *= m; sh = sh*m + p.adm_shift[idx]; }
        }
      }
      ftab[0][tid] = sc; ftab[1][tid] = sh;
    }
    __syncthreads();
  }

#pragma unroll
  for (int r = 0; r < 2; ++r) store_piece(r, 0, 0, raw[r]);
  __syncthreads();
  for (int ci = 0; ci < n_my; ++ci) {
    const int buf = ci & 1;
    const bool more = ci + 1 < n_my;
    if (more) {
#pragma unroll
      for (int r = 0; r < 2; ++r) raw[r] = load_piece(r, c_lo + ci + 1);
    }
    const unsigned char* pb = patch[buf];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int kh = tap/3, kw = tap % 3;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const h8 bv = *reinterpret_cast<const h8*>(pb + ((b_off[kw] ^ (unsigned int)(ks*32)) + (unsigned int)(kh*CS_PC*64)));
#pragma unroll
        for (int cf = 0; cf < 2; ++cf)
          acc[cf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a[tap][ks][cf]), bv, acc[cf], 0, 0, 0);
      }
      // the next chunk's fragments of this tap go out as soon as this tap's registers are free
      if (more) load_a(c_lo + ci + 1, tap, a[tap]);
    }
    if (more) {
#pragma unroll
      for (int r = 0; r < 2; ++r) store_piece(r, buf ^ 1, ci + 1, raw[r]);
    }
    __syncthreads();
  }

  // ---- partial tile -> scratch: D[co][pixel], lane = column n32, register i = channel (i >> 2)*8 + khalf*4 + (i & 3)
  const int h = h0 + wpx, w = w0 + n32;
  if (h < p.H && w < p.W) {
    const long long npix = (long long)p.B*hw;
    const long long pix = (long long)b*hw + (long long)h*p.W + w;
    const long long cq_n = (long long)p.n_cob*32;
#pragma unroll
    for (int cf = 0; cf < 2; ++cf)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cq = cob*32 + wco*16 + cf*8 + g*2 + khalf;
        float* dst = p.part + (((long long)split*cq_n + cq)*npix + pix)*4;
        *reinterpret_cast<f32x4*>(dst) = f32x4{acc[cf][4*g], acc[cf][4*g + 1], acc[cf][4*g + 2], acc[cf][4*g + 3]};
      }
  }
}

// y = out_scale*(sum over the splits + bias + res), fp16 channels-last; stats (B, Cout, 2) += per-channel sums and
// sums of squares of the ROUNDED outputs (what the next GroupNorm folds, as conv_nhwc_kernel's epilogue takes them).
// One thread = one pixel x one channel quad; a workgroup = 256 consecutive pixels of one (item, quad).
struct ConvCombineParams {
  const float* part; int n_split, cq_n; long long npix, hw;
  const float* bias; const _Float16* res; _Float16* y; double* stats;
  int Cout, Crs, Cys; float out_scale;
};
__global__ __launch_bounds__(256) void conv_nhwc_combine_kernel(const ConvCombineParams p) {
  __shared__ float red[4][8];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int b = blockIdx.y / p.cq_n, cq = blockIdx.y % p.cq_n;
  const int co = cq*4;
  if (co >= p.Cout) return;                                  // (whole workgroup)
  const long long i = (long long)blockIdx.x*256 + tid;
  float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
  if (i < p.hw) {
    const long long pix = (long long)b*p.hw + i;
    // all splits requested together (a run-time loop waited for each load in turn: 8 L2 round trips), added in order
    static_assert(BRV_CONV_SPLIT_MAX <= 8, "combine reads at most 8 splits");
    f32x4 u[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int kk = k < p.n_split ? k : 0;
      u[k] = *reinterpret_cast<const f32x4*>(p.part + (((long long)kk*p.cq_n + cq)*p.npix + pix)*4);
    }
    f32x4 v = u[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) {
      const float on = k < p.n_split ? 1.f : 0.f;
      v += u[k]*on;
    }
    float r[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.res) {
      const h4 rv = *reinterpret_cast<const h4*>(p.res + pix*p.Crs + co);
#pragma unroll
      for (int j = 0; j < 4; ++j) r[j] = (float)rv[j];
    }
    h4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o[j] = (_Float16)((v[j] + (p.bias ? p.bias[co + j] : 0.f) + r[j])*p.out_scale);
      const float wq = (float)o[j];
      s[j] = wq; q[j] = wq*wq;
    }
    *reinterpret_cast<h4*>(p.y + pix*p.Cys + co) = o;
  }
  if (!p.stats) return;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s[j] += __shfl_xor(s[j], o, 64); q[j] += __shfl_xor(q[j], o, 64); }
  }
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[wv][j] = s[j]; red[wv][4 + j] = q[j]; }
  }
  __syncthreads();
  if (tid < 8) {
    const double a = ((double)red[0][tid] + (double)red[1][tid]) + ((double)red[2][tid] + (double)red[3][tid]);
    atomicAdd(&p.stats[(((long long)b*p.Cout + co + (tid & 3)) << 1) + (tid >> 2)], a);
  }
}

// Plan of a launch: the split path pays when the pixel-parallel kernel would leave most of the chip idle.
// Returns the scratch floats needed (0: use conv_nhwc_kernel).
inline long long conv_split_plan(long long B, long long H, long long W, long long n_chunks, long long n_cob,
                                 int& cpw, int& n_split) {
  cpw = (int)n_chunks; n_split = 1;
  const long long base = B*((H + CS_ROWS - 1)/CS_ROWS)*((W + CN_COLS - 1)/CN_COLS)*n_cob;
  if (base > BRV_CONV_SPLIT_MAX_BASE || n_chunks < 2) return 0;
  long long want = (BRV_CONV_SPLIT_TARGET + base - 1)/base;
  if (want > BRV_CONV_SPLIT_MAX) want = BRV_CONV_SPLIT_MAX;
  if (want > n_chunks) want = n_chunks;
  long long c = (n_chunks + want - 1)/want;
  if (c > CS_MAXCPW) c = CS_MAXCPW;
  const long long ns = (n_chunks + c - 1)/c;
  if (ns < 2) return 0;
  cpw = (int)c; n_split = (int)ns;
  return ns*n_cob*32*B*H*W*4;
}
