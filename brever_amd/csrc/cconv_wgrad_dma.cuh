// Weight gradient of the row convolutions on bf16 IMAGES with the staging done by 16-byte LDS-DMA (round 6): the tile,
// the LDS images and the MFMA loop of cconv_wgrad_kernel (cconv.hip), the loader of cconv_dma.cuh. Included by
// cconv.hip behind cconv_wgrad_kernel (inside its anonymous namespace).
//   * a stage = 64 frames of one (b, h) pair: `small` twice (as it is, and shifted by one frame = the j = 1 tap: the same
//     rows fetched from an address 2 bytes lower) and the 5 x 32 rows of `big`: 52 instructions of 64 lanes x 16 bytes
//     (8 image rows of 128 bytes each) per workgroup, against 112 8-byte register loads + as many LDS writes before;
//     the XOR swizzle of the images (wg_off) is applied on the source side;
//   * ring of 3 stages (3 x 52 KB of LDS), two in flight; one hand-placed vmcnt per stage; every LDS access of the
//     loop is inline asm (cconv_dma.cuh says why);
//   * the frame axis is the REDUCTION axis here, so frames past a row's end must not contribute: the 16-byte pieces of
//     `small` that lie entirely past the end are fetched from an out-of-range offset (zeros), the one piece that
//     straddles it is masked in LDS by the wave that staged it (`patch`), and so is frame -1 of the shifted image in
//     the first stage of a row; `big` is masked the same way: its columns past the end meet zeros of `small`, but they
//     hold whatever lies behind the row -- behind the LAST row that is the readable slack, arbitrary bits, and 0 x NaN
//     is NaN (found by the two-rank lock-step test: a NaN weight gradient on the second step of one rank).
// CONTRACT as brv_cconv_rows_bf16: 16 readable bytes in front of and behind every image (brever_hip.h).
#pragma once

constexpr int WD_STAGE = WG_BUFB;                       // 52 KB: [small | small shifted | big], as cconv_wgrad_kernel
constexpr int WD_R = 3;
constexpr int WD_U = 7;                                  // DMAs per wave and stage (52 units of 8 rows, padded to 56)
constexpr int WD_PAD = WD_R*WD_STAGE;                    // 1 KB the padding units write zeros to

__global__ __launch_bounds__(WG_THREADS) void cconv_wgrad_dma_kernel(const CWgradParams p) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[WD_R*WD_STAGE + 1024];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar for the compiler)
  const int j = wid & 1, afr = wid >> 1;
  unsigned bx = blockIdx.x, by = blockIdx.y;             // XCD-aware order: as cconv_wgrad_kernel
  {
    const unsigned total = gridDim.x*gridDim.y;
    if ((total & 7u) == 0) {
      const unsigned L = blockIdx.x + gridDim.x*blockIdx.y;
      const unsigned g = (L & 7u)*(total >> 3) + (L >> 3);
      bx = g % gridDim.x; by = g / gridDim.x;
    }
  }
  const int atile = bx % p.atiles, ctile = bx / p.atiles;
  const int pair0 = by*p.pairs_per;
  const int pair1 = pair0 + p.pairs_per < p.npairs ? pair0 + p.pairs_per : p.npairs;
  const int nitems = (pair1 - pair0)*p.nstage;
  const unsigned int lds0 = dma::lds_a(lds);
  constexpr unsigned int kFar = 0x80000000u;

  // ---- DMA units of this wave: u = wid + 8 i. Units 0..15: `small` rows 8 u .., 16..31: the same rows shifted,
  // 32..51: `big` rows (tap row i = (u - 32) / 4, channels 8 ((u - 32) & 3) ..), 52..55: padding. Lane l: row
  // 8 u' + (l >> 3), physical 16-byte piece l & 7 = logical piece (l & 7) ^ swz(row).
  const long long small_n = (long long)p.B*p.small_bs, big_n = (long long)p.B*p.big_bs;
  const __amdgpu_buffer_rsrc_t r_small = make_rsrc(static_cast<const bf16_t*>(p.small) - 8, small_n*2 + 32);
  const __amdgpu_buffer_rsrc_t r_small2 = make_rsrc(static_cast<const bf16_t*>(p.seg > 0 ? p.small2 : p.small) - 8, small_n*2 + 32);
  const __amdgpu_buffer_rsrc_t r_big = make_rsrc(static_cast<const bf16_t*>(p.big) - 8, big_n*2 + 32);
  unsigned int voff[WD_U];        // lane part of the source offset (bytes; kFar: nothing to fetch)
  int lpiece[WD_U];               // first frame of the lane's piece inside the stage (8 x logical piece)
  int kind[WD_U];                 // wave-uniform: 0 small, 1 small2 (second source), 2 big, 3 padding;  + 4: shifted
#pragma unroll
  for (int i = 0; i < WD_U; ++i) {
    const int u = wid + 8*i;
    voff[i] = kFar; lpiece[i] = 0; kind[i] = 3;
    if (u < 32) {
      const int row = 8*(u & 15) + (lane >> 3);                // row of the 128-row image = channel of the tile
      const int a = atile*WG_A + row;
      const int lp = (lane & 7) ^ wg_swz(row);
      lpiece[i] = 8*lp;
      int ch = a, k = 0;
      if (p.seg > 0) { const int sg = a / p.seg; ch = (sg >> 1)*p.seg + a % p.seg; k = sg & 1; }
      // (8 consecutive channels never straddle two segments: seg % 8 == 0 is a condition of this kernel)
      k = __builtin_amdgcn_readfirstlane(k);
      kind[i] = k + (u >= 16 ? 4 : 0);
      if (a < p.A) voff[i] = (unsigned int)(16 + ((long long)ch*p.Hs*p.Ws + 8*lp - (u >= 16 ? 1 : 0))*2);
    } else if (u < 52) {
      const int rs = 8*(u - 32) + (lane >> 3);                 // row of the 160-row image: (tap row, channel)
      const int c = ctile*WG_C + (rs & 31);
      const int lp = (lane & 7) ^ wg_swz(rs);
      lpiece[i] = 8*lp;
      kind[i] = 2;
      if (c < p.C) voff[i] = (unsigned int)(16 + ((long long)c*p.Hb*p.Wb + 8*lp)*2);
    }
  }
  auto issue = [&](int it) {
    const bool live = it < nitems;
    const int pr = pair0 + (live ? it : 0) / p.nstage, stg = (live ? it : 0) % p.nstage;
    const int b = pr / p.Hs, h = pr % p.Hs, f = stg*WG_F;
    const bool last = stg == p.nstage - 1;
    unsigned char* st = lds + (it % WD_R)*WD_STAGE;
#pragma unroll
    for (int i = 0; i < WD_U; ++i) {
      const int u = wid + 8*i;                                  // (wave-uniform)
      unsigned int v = voff[i];
      unsigned char* dst = lds + WD_PAD;
      __amdgpu_buffer_rsrc_t rs = r_big;
      if (u < 32) {
        dst = st + (u >= 16 ? WG_SMALLB : 0) + (u & 15)*1024;
        rs = (kind[i] & 1) ? r_small2 : r_small;
        v += (unsigned int)(((long long)b*p.small_bs + (long long)h*p.Ws + f)*2);
        // pieces entirely past the row's end (the shifted image holds frame Ws - 1 at column Ws): zeros
        if (last && f + lpiece[i] >= p.Ws + (u >= 16 ? 1 : 0)) v = kFar;
      } else if (u < 52) {
        dst = st + 2*WG_SMALLB + (u - 32)*1024;
        const int row = 2*h - 2 + (u - 32)/4;
        v += (unsigned int)(((long long)b*p.big_bs + (long long)row*p.Wb + f)*2);
        if (row < 0 || row >= p.Hb || (last && f + lpiece[i] >= p.Wb)) v = kFar;
      }
      if (!live) v = kFar;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (dma::lds_void_p)dst, 16, (int)v, 0, 0, 0);
    }
  };
  // ---- after a stage has landed: the frames of `small` that must not contribute. Lane l of the wave patches row
  // 8 u' + (l & 7) of its unit i = l >> 3 (units 0..3 of a wave are its `small` units, plain and shifted: rows it
  // staged itself).
  // masks the 16-byte piece at `dst` down to its first `keep` elements
  auto keep_first = [&](unsigned int dst, int keep) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(dst) : "memory");
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const int e0 = 2*d;
      unsigned int w = v[d];
      if (e0 >= keep) w = 0; else if (e0 + 1 >= keep) w &= 0xffffu;
      v[d] = w;
    }
    asm volatile("ds_write_b128 %0, %1" :: "v"(dst), "v"(v) : "memory");
  };
  auto patch = [&](int it) {
    if (it >= nitems) return;
    const int stg = it % p.nstage, f = stg*WG_F;
    const unsigned int st = lds0 + (it % WD_R)*WD_STAGE;
    const int i = lane >> 3;
    const int u = wid + 8*i;
    if (i >= 4 && u < 52) {                      // `big` rows: the piece that straddles the row's end
      const int rs = 8*(u - 32) + (lane & 7);
      const int lim = p.Wb - f;
      if (stg == p.nstage - 1 && lim > 0 && lim < WG_F && (lim & 7))
        keep_first(st + 2*WG_SMALLB + 128*rs + 16*((lim >> 3) ^ wg_swz(rs)), lim & 7);
    }
    if (i < 4 && u < 32) {
      const int row = 8*(u & 15) + (lane & 7);
      const unsigned int img = st + (u >= 16 ? WG_SMALLB : 0) + 128*row;
      if (u >= 16 && stg == 0) {                 // frame -1 of the row (the previous row's last frame arrived there)
        const unsigned int dst = img + 16*wg_swz(row), z = 0;
        asm volatile("ds_write_b16 %0, %1" :: "v"(dst), "v"(z) : "memory");
      }
      const int lim = p.Ws + (u >= 16 ? 1 : 0) - f;       // first column of the stage that must be zero
      if (stg == p.nstage - 1 && lim > 0 && lim < WG_F && (lim & 7))       // the piece that straddles the end
        keep_first(img + 16*((lim >> 3) ^ wg_swz(row)), lim & 7);
    }
  };

  f32x16 acc[CC_KH];
#pragma unroll
  for (int i = 0; i < CC_KH; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

  issue(0);
  issue(1);
  dma::wait_vm<WD_U>();
  patch(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  const int m = lane & 31, g = lane >> 5;
  // fragment addresses: lane parts (the 16-byte piece index 2 ks + g changes with the k step: XOR with the row's swizzle)
  const int arow = 32*afr + m;
  const unsigned int a_base = j*WG_SMALLB + 128*arow;
  unsigned int b_base[CC_KH];
#pragma unroll
  for (int i = 0; i < CC_KH; ++i) b_base[i] = 2*WG_SMALLB + 128*(32*i + m);
  const int a_swz = wg_swz(arow);
  int b_swz[CC_KH];
#pragma unroll
  for (int i = 0; i < CC_KH; ++i) b_swz[i] = wg_swz(32*i + m);

  // (plain macros, not lambdas: clang rejects an asm operand that names a captured array element inside a generic lambda)
  u32x4 af0, af1, bf0[CC_KH], bf1[CC_KH];
#define WD_FRAGS(KS, AF, BF)                                                                                  \
  {                                                                                                           \
    const int ch_ = 2*(KS) + g;                                                                               \
    const unsigned int aa_ = cur + a_base + 16*(ch_ ^ a_swz);                                                 \
    asm volatile("ds_read_b128 %0, %1" : "=v"(AF) : "v"(aa_) : "memory");                                     \
    _Pragma("unroll") for (int i = 0; i < CC_KH; ++i) {                                                       \
      const unsigned int ba_ = cur + b_base[i] + 16*(ch_ ^ b_swz[i]);                                         \
      asm volatile("ds_read_b128 %0, %1" : "=v"(BF[i]) : "v"(ba_) : "memory");                                \
    }                                                                                                         \
  }
  // the six reads of a k step are back once only PENDING younger ones are outstanding (DS operations return in order)
#define WD_MFMAS(AF, BF, PENDING)                                                                             \
  {                                                                                                           \
    asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(AF), "+v"(BF[0]), "+v"(BF[1]), "+v"(BF[2]), "+v"(BF[3]), "+v"(BF[4]) \
                 : "n"(PENDING) : "memory");                                                                  \
    _Pragma("unroll") for (int i = 0; i < CC_KH; ++i)                                                         \
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, AF), __builtin_bit_cast(bf16x8, BF[i]), \
                                                       acc[i], 0, 0, 0);                                      \
  }
  static_assert(WG_F/16 == 4, "four k steps per stage");
#pragma unroll 1
  for (int it = 0; it < nitems; ++it) {
    issue(it + 2);                                  // into the ring slot stage it - 1 was read from
    const unsigned int cur = lds0 + (it % WD_R)*WD_STAGE;
    WD_FRAGS(0, af0, bf0)
    WD_FRAGS(1, af1, bf1) WD_MFMAS(af0, bf0, 6)
    WD_FRAGS(2, af0, bf0) WD_MFMAS(af1, bf1, 6)
    WD_FRAGS(3, af1, bf1) WD_MFMAS(af0, bf0, 6)
    WD_MFMAS(af1, bf1, 0)
    dma::wait_vm<WD_U>();                           // stage it + 1 has landed; it + 2 stays in flight
    patch(it + 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
#undef WD_FRAGS
#undef WD_MFMAS
  dma::wait_vm<0>();

  // ---- D[a][c] of tap (i, j) -> part[split][2 i + j][a][c] (as cconv_wgrad_kernel)
  const int c = ctile*WG_C + m;
  if (c < p.C) {
    float* part = p.part + (long long)by*10*p.A*p.C;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = atile*WG_A + 32*afr + (e & 3) + 8*(e >> 2) + 4*g;
      if (row < p.A) {
#pragma unroll
        for (int i = 0; i < CC_KH; ++i) part[((long long)(2*i + j)*p.A + row)*p.C + c] = acc[i][e];
      }
    }
  }
}
