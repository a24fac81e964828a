// [res | skip] weight gradient of all TCN blocks, second tiling (round 5): a workgroup owns the full G width (256)
// and a 128-channel slice of H (gemm_wgrad_full.cuh: 64). Why: the ablations of the 64-wide form
// (profiles/r05_dwpw2_ablation.txt, section "wgrad_full (pw2_wgrad) ablations") show its loop is not bound by what it loads -- without any LDS-DMA it
// still takes 589 of its 850 us: a chunk of 64 frames is 8 MFMAs per wave (256 matrix-pipe cycles) between two
// barriers, and the fragment-read latencies, the DMA issue and the barrier hand-over around them take three times
// that. Here a chunk is 16 MFMAs per wave on FIVE fragments per k-step (1 G + 4 H) instead of 8 on three, and G (the
// operand every H slice re-reads: 32 of a chunk's 40 KB) is fetched by four workgroups per block instead of eight.
//   * LDS: G ring 3 x 32 KB (LDS-DMA, two chunks in flight, as before) + transformed H 2 x 20 KB; the raw H rows no
//     longer pass through LDS: a thread loads its two 16-byte pieces of a chunk to registers two chunks ahead (the
//     staging ring would not fit beside a 128-wide image) and writes the transformed pieces;
//   * partial tiles: [split][block][256][512] as before; the host halves the split when two chains run.
// Reference: autograd of res_conv / skip_conv, brever/models/convtasnet/convtasnet.py:240-260.
#pragma once
#include "gemm_wgrad_full.cuh"

namespace brv {

constexpr int W3_BH = 128;                 // H channels per workgroup
constexpr int W3_LDH = W3_BH + 32;         // transformed H rows: 320 B = 16 banks per row, conflict-free tr reads
constexpr int W3_HT_BYTES = W2_BT*W3_LDH*2;          // 20 KiB
constexpr int W3_OFF_HT = W2_STAGES*W2_GBYTES;       // 96 KiB
constexpr int W3_SMEM = W3_OFF_HT + 2*W3_HT_BYTES;   // 136 KiB
constexpr int W3_SETS = 2;                 // k-steps of fragments in registers
constexpr int W3_VM = 6;                   // vector-memory operations per wave and chunk (4 G DMAs + 2 H loads)

__device__ __forceinline__ TrAddr tr_addr_h3(unsigned int img, int row0, int col0, int lane) {
  const int g4 = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
  const int col = col0 + 16*(g4 & 1) + 4*pp;
  const int row = row0 + 8*(g4 >> 1) + q;
  TrAddr t;
  t.a0 = img + (row*W3_LDH + col)*2;
  t.a1 = t.a0 + 4*W3_LDH*2;
  return t;
}
// transposing read with the stage / k-step / column-group part of the address as the instruction's 16-bit immediate:
// ONE address register per operand and lane instead of one per read (40 of them in a chunk: the kernel spilled)
template <int OFF>
__device__ __forceinline__ s16x4 lds_read_tr_o(unsigned int addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
  s16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
template <int N>
__device__ __forceinline__ void frag_wait5(Frag& a, Frag& b, Frag& c, Frag& d, Frag& e) {
  asm volatile("s_waitcnt lgkmcnt(%10)"
               : "+v"(a.lo), "+v"(a.hi), "+v"(b.lo), "+v"(b.hi), "+v"(c.lo), "+v"(c.hi),
                 "+v"(d.lo), "+v"(d.hi), "+v"(e.lo), "+v"(e.hi)
               : "n"(N) : "memory");
}

__global__ __launch_bounds__(64*W2_NW) void wgrad_full128_kernel(const WgradFullParams p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[W3_SMEM];

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
  const int k0 = htile*W3_BH;

  for (int o = tid*16; o < W3_SMEM; o += 64*W2_NW*16)
    *reinterpret_cast<uint4*>(smem + o) = make_uint4(0, 0, 0, 0);
  __syncthreads();

  // ---- G by LDS-DMA: exactly the geometry of wgrad_full_kernel ------------------------------------------
  const int grow = lane >> 4;
  const int gsw = ((4*(wid & 1) + grow) & 7) << 1;
  const int gch = (lane & 15) ^ gsw;
  const bf16_t* g0 = reinterpret_cast<const bf16_t*>(q.g0);
  const bf16_t* g1 = reinterpret_cast<const bf16_t*>(q.g1);
  unsigned int gvoff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int u = wid + 8*j, row = 4*(u & 15) + grow;
    const int ld = j < 2 ? p.ldg0 : p.ldg1;
    gvoff[j] = (unsigned int)(row*ld*2 + gch*16);
  }
  // ---- H to registers: the thread's pieces are rows hrow and hrow + 32 of a chunk, channel octet hcc ----
  const int hcc = lane & 15, hrow = 4*wid + (lane >> 4);
  const int hch = k0 + hcc*8;
  const bf16_t* hsrc = reinterpret_cast<const bf16_t*>(q.h);

  const float slope = q.slope ? *q.slope : 1.f;
  const float c1r = 0.5f*(1.f + slope), c2 = 0.5f*(1.f - slope);
  const float c1 = __builtin_fabsf(c1r) < 0x1p-40f ? 0x1p-40f : c1r;   // (a slope of exactly -1: nudged, as dwpw2_fused.cuh)
  const float rho = c2/c1;
  // per item: H = a (z + rho |z|) + c (PReLU_2 has ONE slope: a = c1 rstd gamma; gamma / beta are re-read from L2 when the
  // item changes instead of living in 16 registers: this kernel is at the edge of its register file)
  f32x2 ca[4], cc[4];
  auto item_coefs = [&](int b) {
    NormStat ns = {0.f, 1.f};
    if (q.stats) ns = norm_stat(q.stats, b, p.inv_n, p.eps);
    float hg[8], hb[8];
    load8_masked(q.gamma, hch, p.Kout, hg);
    load8_masked(q.beta, hch, p.Kout, hb);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float s0 = q.stats ? ns.rstd*hg[2*j] : 1.f, s1 = q.stats ? ns.rstd*hg[2*j + 1] : 1.f;
      ca[j] = f32x2{c1*s0, c1*s1};
      cc[j] = q.stats ? f32x2{hb[2*j] - ns.mean*s0, hb[2*j + 1] - ns.mean*s1} : f32x2{0.f, 0.f};
    }
  };

  // the 4 G DMAs of chunk (b, t0) into stage st and the 2 H loads into `hr`; `live` false -> zeros
  auto issue_chunk = [&](int b, int t0, int st, bool live, uint4 (&hr)[2]) {
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
    hr[0] = buf_load16(rh, (unsigned int)((t0 + hrow)*p.ldh*2 + hch*2));
    hr[1] = buf_load16(rh, (unsigned int)((t0 + hrow + 32)*p.ldh*2 + hch*2));
  };

  float bias0[8], bias1[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { bias0[j] = 0.f; bias1[j] = 0.f; }
  const bool want_bias = q.gbias0 != nullptr || q.gbias1 != nullptr;

  const unsigned int smem_a = lds_addr(smem);
  auto h_math = [&](const uint4& raw, int row, int nvalid, uint4& packed) {
    float f[8]; unpack8(raw, f);
    f32x2 o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x2 zz = {__builtin_fmaf(rho, __builtin_fabsf(f[2*k]), f[2*k]), __builtin_fmaf(rho, __builtin_fabsf(f[2*k + 1]), f[2*k + 1])};
      o[k] = ca[k]*zz + cc[k];
    }
    packed = pack8v(o);
    if (row >= nvalid) packed = make_uint4(0, 0, 0, 0);
  };
  auto h_write = [&](int par, int row, const uint4& packed) {
    lds_write16(smem_a + W3_OFF_HT + par*W3_HT_BYTES + (row*W3_LDH + hcc*8)*2, packed);
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

  // wave `wid` owns G channels [32 wid, 32 wid + 32) x the 128 H channels
  f32x16 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;

  auto advance = [&](int& b, int& t) { t += W2_BT; if (t >= T) { t = 0; ++b; } };
  uint4 hra[2], hrb[2];                     // H pieces of the chunks with even / odd index
  int b1 = b_lo, t1 = 0;
  item_coefs(b_lo);
  issue_chunk(b_lo, 0, 0, total > 0, hra);
  advance(b1, t1);
  int b2 = b1, t2 = t1;
  issue_chunk(b1, t1, 1, total > 1, hrb);
  advance(b2, t2);
  {
    // chunk 0: its H pieces are back once only chunk 1's six operations are pending (the compiler counts the
    // loads it can see; the asm wait below is for the DMA'd G image, which it cannot)
    uint4 pk0, pk1;
    h_math(hra[0], hrow, T, pk0); h_math(hra[1], hrow + 32, T, pk1);
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(W3_VM) : "memory");
    h_write(0, hrow, pk0); h_write(0, hrow + 32, pk1);
    if (want_bias && htile == 0) bias_pass(0, T);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // lane parts of the fragment addresses (stage, k-step and column group are immediates of the reads)
  const TrAddr ga0 = tr_addr_g(smem_a, 0, 32*wid, lane);            // a0 / a1: rows q and q + 4 (own swizzles)
  const unsigned int ha0 = tr_addr_h3(smem_a + W3_OFF_HT, 0, 0, lane).a0;
  const int total6 = ceil_div(total, 6)*6;
  auto body = [&](auto tag, int c) {
    constexpr int S = decltype(tag)::value;
    constexpr int ST = S % 3, NX = (S + 1) % 3, FAR = (S + 2) % 3, PAR = S & 1;
    // H pieces: chunk c + 1 (index parity PAR ^ 1) is transformed in this iteration; chunk c + 2 is requested FIRST,
    // into the pair chunk c left (transformed one iteration ago), so that two chunks stay in flight
    uint4 (&hnx)[2] = PAR ? hra : hrb;
    uint4 (&hfar)[2] = PAR ? hrb : hra;
    issue_chunk(b2, t2, FAR, c + 2 < total, hfar);
    const int nvalid = b1 < b_hi ? T - t1 : 0;
    if (t1 == 0 && b1 < b_hi) item_coefs(b1);
    // chunk c + 1 (G image and H pieces): landed once only chunk c + 2's operations are pending
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(W3_VM) : "memory");
    // (the other parity of the transformed image was last read by the MFMAs of chunk c - 1: free since the barrier)
    {
      uint4 pk;
      h_math(hnx[0], hrow, nvalid, pk); h_write(PAR ^ 1, hrow, pk);
      h_math(hnx[1], hrow + 32, nvalid, pk); h_write(PAR ^ 1, hrow + 32, pk);
    }
    Frag fa[W3_SETS], fh[W3_SETS][4];
    // (G image of stage ST: 32 KB apart -- beyond the offset field for ST = 2, so the stage is one add per operand)
    const unsigned int g0a = ga0.a0 + ST*W2_GBYTES, g1a = ga0.a1 + ST*W2_GBYTES;
    auto issue = [&](auto s_tag) {
      constexpr int s_ = decltype(s_tag)::value;
      fa[s_ % W3_SETS].lo = lds_read_tr_o<s_*16*256>(g0a);
      fa[s_ % W3_SETS].hi = lds_read_tr_o<s_*16*256>(g1a);
      constexpr int hb_ = PAR*W3_HT_BYTES + s_*16*W3_LDH*2;
      fh[s_ % W3_SETS][0].lo = lds_read_tr_o<hb_>(ha0);           fh[s_ % W3_SETS][0].hi = lds_read_tr_o<hb_ + 4*W3_LDH*2>(ha0);
      fh[s_ % W3_SETS][1].lo = lds_read_tr_o<hb_ + 64>(ha0);      fh[s_ % W3_SETS][1].hi = lds_read_tr_o<hb_ + 64 + 4*W3_LDH*2>(ha0);
      fh[s_ % W3_SETS][2].lo = lds_read_tr_o<hb_ + 128>(ha0);     fh[s_ % W3_SETS][2].hi = lds_read_tr_o<hb_ + 128 + 4*W3_LDH*2>(ha0);
      fh[s_ % W3_SETS][3].lo = lds_read_tr_o<hb_ + 192>(ha0);     fh[s_ % W3_SETS][3].hi = lds_read_tr_o<hb_ + 192 + 4*W3_LDH*2>(ha0);
    };
    auto mfma = [&](int s) {
      const bf16x8 ga = frag_value(fa[s % W3_SETS]);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga, frag_value(fh[s % W3_SETS][j]), acc[j], 0, 0, 0);
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    // two k-steps of fragments in registers, 20 reads issued ahead (lgkmcnt counts to 15: the last reads of the second
    // k-step wait at the issue port for the first to return -- a third set would gain nothing)
    issue(I0{}); issue(I1{});
    frag_wait5<10>(fa[0], fh[0][0], fh[0][1], fh[0][2], fh[0][3]); mfma(0);
    issue(I2{});
    frag_wait5<10>(fa[1], fh[1][0], fh[1][1], fh[1][2], fh[1][3]); mfma(1);
    issue(I3{});
    frag_wait5<10>(fa[0], fh[0][0], fh[0][1], fh[0][2], fh[0][3]); mfma(2);
    frag_wait5<0>(fa[1], fh[1][0], fh[1][1], fh[1][2], fh[1][3]); mfma(3);
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
  const int fr = lane & 31, fh_ = lane >> 5;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int k = k0 + 32*c + fr;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int n = 32*wid + (i & 3) + 8*(i >> 2) + 4*fh_;
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
      if (tid < 128) { if (q.gbias0 && tid < p.N0) atomic_add_f32(q.gbias0 + tid, s); }
      else if (q.gbias1 && tid - 128 < p.N1) atomic_add_f32(q.gbias1 + (tid - 128), s);
    }
  }
}

}  // namespace brv
