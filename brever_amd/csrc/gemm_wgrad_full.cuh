// Weight gradient of the residual/skip 1x1 convolutions of ALL TCN blocks in one launch,
// without split-K atomics.
//
//   D_i[n][k] += sum_{b,t} G_i[b][t][n] * H_i[b][t][k]     n < 256 (res | skip), k < Hp
//
// One workgroup owns the FULL G width (256 channels: [g_res_i | g_skip]) and a 64-channel
// slice of H = gLN_2(PReLU_2(z2_i)) over every frame of every item, so
//   * each output element has exactly one owner: plain read-add-store, no atomics
//     (the 128x128 split-K version spent 0.45 ms/step in 12.6 M fp32 atomics);
//   * H is transformed ONCE (it was rebuilt by both 128-wide G tiles), with
//     PReLU + affine folded into  a*z + b*|z| + c  (two FMAs per element);
//   * the 8 workgroups of one block that share G sit on ONE XCD (ids congruent mod 8), so
//     G comes from HBM once and from that XCD's L2 seven times.
// Frames past the end of an item need no masking on H: G is read through a buffer
// descriptor, so its rows there are zero and the products vanish.
//
// Replaces (reference): autograd of res_conv / skip_conv in
// brever/models/convtasnet/convtasnet.py (TCN conv block), weight and bias gradients.
#pragma once
#include <type_traits>
#include "gemm_wgrad.cuh"
#include "gemm_ws.cuh"

namespace brv {

constexpr int W2_BT = 64;                  // frames per chunk
constexpr int W2_G = 256;                  // G channels (two 128-wide sources)
constexpr int W2_BH = 64;                  // H channels per workgroup
constexpr int W2_LDH = W2_BH + 32;         // transformed H rows: +64 B, conflict-free tr reads
constexpr int W2_STAGES = 3;
constexpr int W2_GBYTES = W2_BT*W2_G*2;    // 32 KiB: two half images [64 frames][16 x 16 B]
constexpr int W2_HBYTES = W2_BT*W2_BH*2;   // 8 KiB raw H, lane-linear
constexpr int W2_HT_BYTES = W2_BT*W2_LDH*2;
constexpr int W2_OFF_H = W2_STAGES*W2_GBYTES;
constexpr int W2_OFF_HT = W2_OFF_H + W2_STAGES*W2_HBYTES;
constexpr int W2_SMEM = W2_OFF_HT + 2*W2_HT_BYTES;     // 144 KiB
constexpr int W2_NW = 8;                   // waves per workgroup: two per SIMD, so that one wave's
                                           // DMA issue / LDS latency hides behind the other's MFMAs
constexpr int W2_DMA = 5;                  // LDS-DMA instructions per wave and chunk (4 G + 1 H)

struct WgradFullParams {
  int B, T, nprob, n_htiles;
  int ldg0, ldg1, ldh;           // row strides (elements)
  long long bsg0, bsg1, bsh;     // item strides (elements)
  int N0, N1;                    // true rows of the two G parts (padded to 128 each)
  int Kout, ldo;                 // true H channels / leading dimension of D
  double inv_n; float eps;
  WgradProb prob[kWgMaxProb];    // g0 may be null (block without residual conv)
  // the items are divided over n_split workgroups per (block, H slice) so that the launch
  // fills whole rounds of the 256 CUs (24 x 8 = 192 owners alone leave a quarter of the chip
  // idle); each writes its tile of partial sums to `part` [split][block][256][ldo], which
  // wgrad_full_reduce_kernel adds into the gradients in a fixed order (no atomics)
  int n_split; float* part;
#ifdef BRV_DIAG
  int dbg;                       // 1: no DMA, 2: no fragment reads / MFMA, 4: no transform
#endif
};

typedef __attribute__((address_space(3))) void* lds_void_p;

// 16 bytes per lane, global -> LDS without a VGPR round trip: LDS destination is
// `dst` (wave-uniform) + lane*16; rows outside the descriptor arrive as zeros.
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, unsigned char* dst, unsigned int voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_p)dst, 16, (int)voff, 0, 0, 0);
}

// Inside the pipelined loop EVERY LDS access is inline asm: hipcc orders each ds_read /
// ds_write it can see behind all pending LDS-DMA with s_waitcnt vmcnt(0) (it cannot tell
// the stage being filled from the stage being read), which would drain the two chunks in
// flight once per chunk. The waits are therefore placed by hand: vmcnt(W2_DMA) retires the
// older chunk, lgkmcnt(n) the fragment reads (DS operations return in order).
__device__ __forceinline__ unsigned int lds_addr(const void* p) {
  return (unsigned int)(unsigned long long)p;            // low 32 bits of a flat LDS address
}
__device__ __forceinline__ s16x4 lds_read_tr(unsigned int addr) {
  s16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
// the result is valid only after lds_wait16 (whole-vector operands: a per-component tie
// lets the compiler copy components out BEFORE the wait, i.e. before the data arrived)
__device__ __forceinline__ u32x4 lds_read16(unsigned int addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ void lds_wait16(u32x4& a, u32x4& b) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b) :: "memory");
}
__device__ __forceinline__ uint4 as_uint4(const u32x4& v) { return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void lds_write16(unsigned int addr, const uint4& q) {
  u32x4 v; v.x = q.x; v.y = q.y; v.z = q.z; v.w = q.w;
  asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(v) : "memory");
}
struct TrAddr { unsigned int a0, a1; };                  // the two 4-row reads of one fragment
// fragment (frames row0..row0+15, 32 channels from col0) of the swizzled G image: half image
// = col >> 7, row stride 256 B, 16-byte slot (c ^ ((row & 7) << 1)) -- the 4 rows x 32 B a
// 16-lane group touches fall in distinct bank groups.
__device__ __forceinline__ TrAddr tr_addr_g(unsigned int img, int row0, int col0, int lane) {
  const int g4 = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
  const int col = col0 + 16*(g4 & 1) + 4*pp;
  const int row = row0 + 8*(g4 >> 1) + q;
  const int half = col >> 7, cc = (col & 127) >> 3, in = (col & 7)*2;
  const unsigned int base = img + half*(W2_GBYTES/2) + in;
  TrAddr t;
  t.a0 = base + row*256 + ((cc ^ ((row & 7) << 1)) << 4);
  t.a1 = base + (row + 4)*256 + ((cc ^ (((row + 4) & 7) << 1)) << 4);
  return t;
}
// same fragment shape out of the padded transformed-H image (row stride W2_LDH elements)
__device__ __forceinline__ TrAddr tr_addr_h(unsigned int img, int row0, int col0, int lane) {
  const int g4 = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
  const int col = col0 + 16*(g4 & 1) + 4*pp;
  const int row = row0 + 8*(g4 >> 1) + q;
  TrAddr t;
  t.a0 = img + (row*W2_LDH + col)*2;
  t.a1 = t.a0 + 4*W2_LDH*2;
  return t;
}
struct Frag { s16x4 lo, hi; };
template <int N>
__device__ __forceinline__ void lds_wait16_n(u32x4& a, u32x4& b) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}
__device__ __forceinline__ Frag frag_issue(const TrAddr& t) {
  Frag f; f.lo = lds_read_tr(t.a0); f.hi = lds_read_tr(t.a1); return f;
}
__device__ __forceinline__ bf16x8 frag_value(const Frag& f) {
  const s16x8 v = __builtin_shufflevector(f.lo, f.hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}
// s_waitcnt lgkmcnt(N) that the four fragments of one k-step depend on
template <int N>
__device__ __forceinline__ void frag_wait3(Frag& a, Frag& b, Frag& c) {
  asm volatile("s_waitcnt lgkmcnt(%6)"
               : "+v"(a.lo), "+v"(a.hi), "+v"(b.lo), "+v"(b.hi), "+v"(c.lo), "+v"(c.hi)
               : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void frag_wait(Frag& a, Frag& b, Frag& c, Frag& d) {
  asm volatile("s_waitcnt lgkmcnt(%8)"
               : "+v"(a.lo), "+v"(a.hi), "+v"(b.lo), "+v"(b.hi), "+v"(c.lo), "+v"(c.hi),
                 "+v"(d.lo), "+v"(d.hi)
               : "n"(N) : "memory");
}

__global__ __launch_bounds__(64*W2_NW) void wgrad_full_kernel(const WgradFullParams p) {
  // ONE shared array (a second object makes hipcc drain vmcnt before every ds_read):
  // [3 x G image | 3 x raw H | 2 x transformed H]
  __shared__ __attribute__((aligned(1024))) unsigned char smem[W2_SMEM];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int per_z = p.n_htiles*p.n_split;
  const int z = (slot / per_z)*8 + xcd, htile = (slot % per_z) / p.n_split;
  const int split = slot % p.n_split;
  if (z >= p.nprob) return;
  const WgradProb& q = p.prob[z];
  const int T = p.T;
  const int cpi = ceil_div(T, W2_BT);
  const int b_lo = split*p.B/p.n_split, b_hi = (split + 1)*p.B/p.n_split;   // this split's items
  const int total = (b_hi - b_lo)*cpi;
  const int k0 = htile*W2_BH;

  // LDS starts out as zeros: rows never written (no residual part, frames past the end) must
  // hold finite values
  for (int o = tid*16; o < W2_SMEM; o += 64*W2_NW*16)
    *reinterpret_cast<uint4*>(smem + o) = make_uint4(0, 0, 0, 0);
  __syncthreads();

  // ---- LDS-DMA geometry -----------------------------------------------------------------
  // G: wave w issues the 1 KiB units u = w + 8j (j < 4): unit u = half image u >> 4, rows
  // 4(u & 15) .. +3, lane -> (row = 4(u & 15) + (lane >> 4), slot = lane & 15), and the
  // slot holds channel chunk slot ^ ((row & 7) << 1) (swizzle applied on the SOURCE side).
  const int grow = lane >> 4;                               // + 4*((wid + 8j) & 15)
  const int gsw = ((4*(wid & 1) + grow) & 7) << 1;          // (row & 7) << 1, same for all j
  const int gch = (lane & 15) ^ gsw;                        // channel chunk inside the half
  const bf16_t* g0 = reinterpret_cast<const bf16_t*>(q.g0);
  const bf16_t* g1 = reinterpret_cast<const bf16_t*>(q.g1);
  unsigned int gvoff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int u = wid + 8*j, row = 4*(u & 15) + grow;
    const int ld = j < 2 ? p.ldg0 : p.ldg1;
    gvoff[j] = (unsigned int)(row*ld*2 + gch*16);
  }
  // H: unit u = w: 8 rows x 8 slots, lane -> (row = 8w + (lane >> 3), slot = lane & 7)
  const int hcc = lane & 7, hrow = 8*wid + (lane >> 3);
  const int hch = k0 + hcc*8;
  const bf16_t* hsrc = reinterpret_cast<const bf16_t*>(q.h);

  float hg[8], hb[8];
  load8_masked(q.gamma, hch, p.Kout, hg);
  load8_masked(q.beta, hch, p.Kout, hb);
  const float slope = q.slope ? *q.slope : 1.f;
  const float c1 = 0.5f*(1.f + slope), c2 = 0.5f*(1.f - slope);
  f32x2 ca[4], cb[4], cc[4];                            // per item: a*z + b*|z| + c
  auto item_coefs = [&](int b) {
    NormStat ns = {0.f, 1.f};
    if (q.stats) ns = norm_stat(q.stats, b, p.inv_n, p.eps);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float s0 = q.stats ? ns.rstd*hg[2*j] : 1.f, s1 = q.stats ? ns.rstd*hg[2*j + 1] : 1.f;
      ca[j] = f32x2{c1*s0, c1*s1};
      cb[j] = f32x2{c2*s0, c2*s1};
      cc[j] = q.stats ? f32x2{hb[2*j] - ns.mean*s0, hb[2*j + 1] - ns.mean*s1} : f32x2{0.f, 0.f};
    }
  };

  // issue the 5 DMAs of chunk (b, t0) into stage st; `live` false -> zeros (keeps the
  // in-flight count uniform at the end of the frame range)
  auto issue_chunk = [&](int b, int t0, int st, bool live) {
    const long long recs = live ? 1 : 0;
    const __amdgpu_buffer_rsrc_t r0 = make_rsrc(g0 + (long long)b*p.bsg0, g0 ? recs*T*p.ldg0*2 : 0);
    const __amdgpu_buffer_rsrc_t r1 = make_rsrc(g1 + (long long)b*p.bsg1, recs*T*p.ldg1*2);
    const __amdgpu_buffer_rsrc_t rh = make_rsrc(hsrc + (long long)b*p.bsh, recs*T*p.ldh*2);
    unsigned char* gimg = smem + st*W2_GBYTES;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ld = j < 2 ? p.ldg0 : p.ldg1;
      dma16(j < 2 ? r0 : r1, gimg + (wid + 8*j)*1024, gvoff[j] + (unsigned int)(t0*ld*2));
    }
    unsigned char* himg = smem + W2_OFF_H + st*W2_HBYTES;
    dma16(rh, himg + wid*1024, (unsigned int)((t0 + hrow)*p.ldh*2 + hch*2));
  };

  // bias gradient = column sums of G: the workgroups of a block take the chunks in turn;
  // each lane sums the slots its own DMAs filled (chunk gch of both halves)
  float bias0[8], bias1[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { bias0[j] = 0.f; bias1[j] = 0.f; }
  const bool want_bias = q.gbias0 != nullptr || q.gbias1 != nullptr;

  // own raw H (stage st) -> transformed, padded image; optional bias pass over own G.
  // Only reads what this lane's own DMAs wrote (covered by the caller's vmcnt wait).
  const unsigned int smem_a = lds_addr(smem);
  // `nvalid` = frames of the chunk inside the item (0 for the dead chunks past the end):
  // rows beyond it are written as ZEROS -- an out-of-range LDS-DMA leaves the previous
  // (finite) contents of its G rows in place, and 0 * finite adds nothing.
  // The transform of chunk c+1 is split so that its VALU work sits between the MFMAs of
  // chunk c: raw reads (own DMA'd slots) are issued first, the math floats, the writes
  // and the optional bias pass come after the last MFMA.
  auto h_read = [&](int st, u32x4& raw) {
    raw = lds_read16(smem_a + W2_OFF_H + st*W2_HBYTES + wid*1024 + lane*16);
  };
  auto h_math = [&](const u32x4& raw, int nvalid, uint4& packed) {
    float f[8]; unpack8(as_uint4(raw), f);
    f32x2 o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x2 zz = {f[2*k], f[2*k + 1]};
      const f32x2 az = {__builtin_fabsf(f[2*k]), __builtin_fabsf(f[2*k + 1])};
      o[k] = ca[k]*zz + (cb[k]*az + cc[k]);
    }
    packed = pack8v(o);
    if (hrow >= nvalid) packed = make_uint4(0, 0, 0, 0);
  };
  auto h_write = [&](int par, const uint4& packed) {
    lds_write16(smem_a + W2_OFF_HT + par*W2_HT_BYTES + (hrow*W2_LDH + hcc*8)*2, packed);
  };
  auto bias_pass = [&](int st, int nvalid) {
    const unsigned int gimg = smem_a + st*W2_GBYTES + wid*1024 + lane*16;
#pragma unroll
    for (int j = 0; j < 4; j += 2) {
      u32x4 ga = lds_read16(gimg + 8*j*1024), gb = lds_read16(gimg + 8*(j + 1)*1024);
      lds_wait16(ga, gb);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float f[8]; unpack8(as_uint4(h ? gb : ga), f);
        const float live = 4*((wid + 8*(j + h)) & 15) + grow < nvalid ? 1.f : 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) { if (j < 2) bias0[k] += live*f[k]; else bias1[k] += live*f[k]; }
      }
    }
  };
  auto transform_chunk = [&](int st, int par, bool bias_turn, int nvalid) {
    u32x4 raw, dummy = {0u, 0u, 0u, 0u}; uint4 packed;
    h_read(st, raw);
    lds_wait16(raw, dummy);
    h_math(raw, nvalid, packed);
    h_write(par, packed);
    if (bias_turn) bias_pass(st, nvalid);
  };

  // wave `wid` owns G channels [32 wid, 32 wid + 32) x the 64 H channels
  f32x16 acc[2];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;

  // chunk bookkeeping: (b1, t1) = chunk c+1, (b2, t2) = chunk c+2
  auto advance = [&](int& b, int& t) { t += W2_BT; if (t >= T) { t = 0; ++b; } };
  int b1 = b_lo, t1 = 0;
  item_coefs(b_lo);
  issue_chunk(b_lo, 0, 0, total > 0);
  advance(b1, t1);
  int b2 = b1, t2 = t1;
  issue_chunk(b1, t1, 1, total > 1);
  advance(b2, t2);
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(W2_DMA) : "memory");     // chunk 0 landed
  transform_chunk(0, 0, want_bias && htile == 0, T);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // One chunk of the pipeline with COMPILE-TIME stage numbers: with runtime LDS offsets
  // hipcc cannot tell the stage being filled from the stage being read and drains
  // vmcnt(0) before the first ds_read of every chunk. The loop is unrolled 6x (3 stages x
  // 2 transformed-H buffers); the chunk count is rounded up with dead chunks (zeros).
  const int total6 = ceil_div(total, 6)*6;
#ifdef BRV_DIAG
  const int dbg = p.dbg;
#else
  constexpr int dbg = 0;
#endif
  auto body = [&](auto tag, int c) {
    constexpr int S = decltype(tag)::value;
    constexpr int ST = S % 3, NX = (S + 1) % 3, FAR = (S + 2) % 3, PAR = S & 1;
    // stage FAR was last read by the MFMAs of chunk c-1: free after the barrier before
    if (!(dbg & 1)) issue_chunk(b2, t2, FAR, c + 2 < total);
    const unsigned int gt = smem_a + ST*W2_GBYTES;
    const unsigned int ht = smem_a + W2_OFF_HT + PAR*W2_HT_BYTES;
    // fragment reads run two k-steps ahead of the MFMAs (6 reads per k-step: 1 G + 2 H)
    Frag fa[4], fc[4], fd[4];
    auto issue = [&](int s) {
      fa[s] = frag_issue(tr_addr_g(gt, 16*s, 32*wid, lane));
      fc[s] = frag_issue(tr_addr_h(ht, 16*s, 0, lane));
      fd[s] = frag_issue(tr_addr_h(ht, 16*s, 32, lane));
    };
    auto mfma = [&](int s) {
      const bf16x8 ga = frag_value(fa[s]);
      const bf16x8 h0 = frag_value(fc[s]), h1 = frag_value(fd[s]);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga, h0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga, h1, acc[1], 0, 0, 0);
    };
    // chunk c+1 (issued one iteration ago): landed once only chunk c+2's DMAs are pending
    if (dbg & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(W2_DMA) : "memory");
    if (t1 == 0 && b1 < b_hi) item_coefs(b1);
    const int nvalid = b1 < b_hi ? T - t1 : 0;
    u32x4 raw, rdummy = {0u, 0u, 0u, 0u}; uint4 packed = make_uint4(0, 0, 0, 0);
    h_read(NX, raw);                                  // oldest in the LDS queue
    if (!(dbg & 2)) {
    issue(0); issue(1);
    lds_wait16_n<6>(raw, rdummy);                     // <= 6 pending: raw + k-step 0 are back
    frag_wait3<6>(fa[0], fc[0], fd[0]); mfma(0);
    if (!(dbg & 4)) h_math(raw, nvalid, packed);      // VALU between the MFMAs
    issue(2);
    frag_wait3<6>(fa[1], fc[1], fd[1]); mfma(1);
    issue(3);
    frag_wait3<6>(fa[2], fc[2], fd[2]); mfma(2);
    frag_wait3<0>(fa[3], fc[3], fd[3]); mfma(3);
    } else {
      lds_wait16_n<0>(raw, rdummy);
      if (!(dbg & 4)) h_math(raw, nvalid, packed);
    }
    if (!(dbg & 4)) h_write(PAR ^ 1, packed);
    if (want_bias && (c + 1) % p.n_htiles == htile) bias_pass(NX, nvalid);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    b1 = b2; t1 = t2;
    advance(b2, t2);
  };
#pragma unroll 1
  for (int c = 0; c < total6; c += 6) {
    body(std::integral_constant<int, 0>{}, c);
    body(std::integral_constant<int, 1>{}, c + 1);
    body(std::integral_constant<int, 2>{}, c + 2);
    body(std::integral_constant<int, 3>{}, c + 3);
    body(std::integral_constant<int, 4>{}, c + 4);
    body(std::integral_constant<int, 5>{}, c + 5);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- epilogue: this workgroup is the only writer of its tile ---------------------------
  const int fr = lane & 31, fh = lane >> 5;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int k = k0 + 32*c + fr;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int n = 32*wid + (i & 3) + 8*(i >> 2) + 4*fh;
      if (p.n_split > 1) {
        if (k < p.Kout) p.part[(((long long)split*p.nprob + z)*W2_G + n)*p.ldo + k] = acc[c][i];
        continue;
      }
      if (k < p.Kout) {
        if (n < 128) {
          if (q.out0 && n < p.N0) q.out0[(long long)n*p.ldo + k] += acc[c][i];
        } else if (q.out1 && n - 128 < p.N1) {
          q.out1[(long long)(n - 128)*p.ldo + k] += acc[c][i];
        }
      }
    }
  }
  if (want_bias) {
    // 32 lanes hold partial sums of the same channel chunk: (wid >> 1, row & 7) names them
    __syncthreads();
    float* sc = reinterpret_cast<float*>(smem);           // [32][256]
    const int part = (wid >> 1)*8 + (gsw >> 1);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      sc[part*W2_G + gch*8 + k] = bias0[k];
      sc[part*W2_G + 128 + gch*8 + k] = bias1[k];
    }
    __syncthreads();
    if (tid < W2_G) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 4*W2_NW; ++r) s += sc[r*W2_G + tid];
      // one partial per workgroup of the block: 8-way atomics only
      if (tid < 128) { if (q.gbias0 && tid < p.N0) atomic_add_f32(q.gbias0 + tid, s); }
      else if (q.gbias1 && tid - 128 < p.N1) atomic_add_f32(q.gbias1 + (tid - 128), s);
    }
  }
}

// out += sum over the splits of the partial tiles, in split order (deterministic)
__global__ __launch_bounds__(256) void wgrad_full_reduce_kernel(const WgradFullParams p) {
  const int z = blockIdx.y;
  const WgradProb& q = p.prob[z];
  const long long per = (long long)W2_G*p.ldo;
  for (long long e = (long long)blockIdx.x*256 + threadIdx.x; e < per; e += (long long)gridDim.x*256) {
    const int n = (int)(e / p.ldo), k = (int)(e % p.ldo);
    if (k >= p.Kout) continue;
    float* dst = nullptr;
    if (n < 128) { if (q.out0 && n < p.N0) dst = q.out0 + (long long)n*p.ldo + k; }
    else if (q.out1 && n - 128 < p.N1) dst = q.out1 + (long long)(n - 128)*p.ldo + k;
    if (!dst) continue;
    float s = 0.f;
    for (int sp = 0; sp < p.n_split; ++sp) s += p.part[((long long)sp*p.nprob + z)*per + e];
    *dst += s;
  }
}

}  // namespace brv
