// The row convolutions of cconv.hip on bf16 IMAGES with the staging done by LDS-DMA (round 6). Included by cconv.hip
// (inside its anonymous namespace, after CConvParams / cc_swz / cc_off).
//
// Why (profiles/r06_cconv_ablation.txt): with the images read through registers (cconv_tile_lean) the image loads
// are half of the kernels' time -- 4 194 us of forward + data-gradient launches per DCCRN step, 2 020 without the
// image loads, 4 035 without the weight loads, 3 019 with every chunk re-reading cache-hot rows -- and neither their
// bytes (bf16 against fp32 images: -4 %) nor their alignment (-3.5 %) is what costs: it is the round trip of a chunk's
// loads, exposed once per chunk because a chunk is only 160 - 1 280 MFMA cycles long and the registers hold one or
// two chunks. bf16 images need no conversion on the way into LDS, so here they never pass through registers:
//   * a RING of R chunk images in LDS (R = 3 with five taps, 4 with three or two), filled by buffer_load_dwordx4 ... lds:
//     one instruction = 64 lanes x 16 bytes = four 256-byte image rows (128 frames of one channel and tap each); the
//     XOR swizzle of the image layout (cc_off) is applied on the SOURCE side (a lane fetches the 16-byte piece that
//     belongs at its LDS position); the image shifted by one frame (the j = 1 tap) is the same row fetched from an
//     address 2 bytes off -- unaligned 16-byte DMA is fine on gfx950 (profiles/r06_cconv_ablation.txt). A first version
//     with one dword per lane (160 instructions per chunk instead of 40) was SLOWER than the register loaders in the
//     five-tap form: what an LDS-DMA costs is its instruction, not its bytes (same file);
//   * no prologue zero-fill: rows outside the image are fetched from an out-of-range offset (the descriptor returns
//     zeros), so every LDS byte a fragment read touches is rewritten every chunk;
//   * the chunk's weight fragments for ALL its taps are requested in one burst one chunk ahead, BEFORE the chunk's
//     DMA burst: loads return in order, so a weight fragment requested behind a DMA burst would pull that burst's
//     round trip into the chunk that uses the fragment (the defect of the rotating ring of the register loaders);
//   * one hand-placed s_waitcnt vmcnt(U (R - 2)) per chunk (U = DMAs per wave and chunk): the image of the NEXT
//     chunk has landed, the R - 2 younger ones stay in flight; every LDS access of the loop is inline asm (hipcc puts
//     vmcnt(0) in front of any LDS access it can see while DMAs are pending).
// What a row's neighbours leak: a row is fetched as 128 (+1) consecutive frames of memory, so frames past the row's
// end arrive holding the NEXT row's first frames (and frame -1 the previous row's last). In the strided form those
// columns only feed output frames >= Wout, which are not stored. In the transposed form out[Win] reads in[Win] and
// out[0] reads in[-1], which must be zero: after a chunk has landed the wave that staged a row zeroes those two
// positions (`patch`). A 16-byte piece that lies partly outside the descriptor arrives as zeros entirely, so the
// descriptors reach 16 bytes in front of and behind the tensor: a bf16 image handed to these kernels must have 16
// readable bytes on BOTH sides (brever_hip.h; the Python side allocates its bf16 images that way, `_bf16_empty`).
#pragma once

namespace dma {

typedef __attribute__((address_space(3))) void* lds_void_p;

__device__ __forceinline__ unsigned int lds_a(const void* p) { return (unsigned int)(unsigned long long)p; }

template <int OFF>
__device__ __forceinline__ s16x4 read_tr(unsigned int addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
  s16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
// the fragment halves are valid once at most N younger LDS operations are pending (DS operations return in order)
template <int N>
__device__ __forceinline__ void wait_lgkm(s16x4& a, s16x4& b) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

}  // namespace dma

// NTAP weight taps (tap_i) reading NSLOT staged input rows (tap_row): NSLOT = NTAP, or 3 rows for the 5 taps of the
// pair form (tap t reads slot t >> 1 into accumulator set t & 1, as cconv_tile_lean)
template <int MF, int NF, int WM, int WN, int NTAP, bool SEG, bool PAIR, int R>
__device__ __forceinline__ void cconv_tile_dma(const CConvParams& p, unsigned char* lds, int b, int r, int ftile,
                                               int mtile, const int (&tap_i)[NTAP], const int (&tap_row)[PAIR ? 3 : NTAP],
                                               int shift) {
  constexpr int NSLOT = PAIR ? 3 : NTAP, NSET = PAIR ? 2 : 1;
  static_assert(!PAIR || NTAP == 5, "pair form: five taps");
  constexpr int NT = 32*NF*WN;               // output frames per workgroup
  constexpr int TILES = NT/128;              // 128-column images side by side
  constexpr int TAPB = TILES*4096, BUFB = NSLOT*TAPB;
  constexpr int U = NSLOT;                   // DMAs per wave and chunk
  static_assert(WM*WN == 8 && NT % 128 == 0, "8 waves, whole images");
  static_assert(U*(R - 1) < 64, "vmcnt field");
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar for the compiler)
  const int wm = wid / WN, wn = wid % WN;
  const int f0 = ftile*NT;
  const int plane = p.Hin*p.Win;
  const bf16_t* in_b = static_cast<const bf16_t*>(p.in) + (long long)b*p.in_bs;
  const bf16_t* in2_b = SEG && p.in_seg > 0 ? static_cast<const bf16_t*>(p.in2) + (long long)b*p.in_bs : nullptr;
  const unsigned int lds0 = dma::lds_a(lds);

  // ---- DMA sources. One instruction = 64 lanes x 16 bytes = four 256-byte image rows: wave w stages rows
  // 4 q .. 4 q + 3 (q = w & 3: image q >> 1, channels 4 (q & 1) ..) of tile k = w >> 2 for every tap. Lane l: row
  // rho = 4 q + (l >> 4), physical piece l & 15 of it = logical piece (l & 15) ^ swz(rho) (the XOR is its own inverse)
  static_assert(TILES == 2, "two 128-frame tiles: 8 waves = 2 tiles x 4 row quads");
  constexpr unsigned int kFar = 0x80000000u;     // out of range for any image
  const int dq = wid & 3, dk = wid >> 2;
  const int rho = 4*dq + (lane >> 4), dimg = dq >> 1, dch = rho & 7;
  unsigned int voff[NSLOT];
  {
    const int frame = f0 + 128*dk + shift*dimg + 8*((lane & 15) ^ cc_swz(rho));
#pragma unroll
    for (int t = 0; t < NSLOT; ++t) {
      const bool ok = tap_row[t] >= 0 && tap_row[t] < p.Hin;
      // (+16: the descriptor starts 16 bytes in front of the chunk, so that frame -1 of its first row is IN range --
      // a 16-byte piece that is partly out of range arrives as zeros entirely)
      voff[t] = ok ? (unsigned int)(16 + (dch*plane + tap_row[t]*p.Win + frame)*2) : kFar;
    }
  }
  // the descriptor of a chunk: from 16 bytes in front of its 8 channels to 16 bytes behind the batch item (the header
  // comment: the images are allocated with that much readable slack on both sides)
  auto chunk_rsrc = [&](int cc) {
    const bf16_t* base = in_b;
    int ch0 = 8*cc;
    if (SEG && p.in_seg > 0) {
      const int sg = (ch0 >= p.in_seg) + (ch0 >= 2*p.in_seg) + (ch0 >= 3*p.in_seg);
      ch0 -= ((sg + 1) >> 1)*p.in_seg;
      if (sg & 1) base = in2_b;
    }
    const long long off = (long long)ch0*plane;
    return make_rsrc(base + off - 8, cc < p.ncc ? (p.in_bs - off)*2 + 32 : 0);
  };
  auto burst = [&](int cc) {                 // chunk cc -> ring slot cc % R (chunks past the end: zeros, never read)
    if (CC_ABL & 1) return;
    const __amdgpu_buffer_rsrc_t rs = chunk_rsrc((CC_ABL & 32) ? (cc < p.ncc ? 0 : cc) : cc);
    unsigned char* dst = lds + (cc % R)*BUFB + dk*4096 + dq*1024;
#pragma unroll
    for (int t = 0; t < NSLOT; ++t)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (dma::lds_void_p)(dst + t*TAPB), 16, (int)voff[t], 0, 0, 0);
  };
  // ---- the two frames of the transposed form that must read as zero (shift = -1): in[Win] in the plain image and
  // in[-1] in the shifted one (they arrive holding the neighbouring rows' frames). Lane 4 t + j of a wave patches
  // row 4 q + j of slot t: only rows the wave staged itself, so its own vmcnt wait covers them.
  const int col_end = p.Win - f0;            // column of in[Win] in the plain image of this workgroup
  auto patch = [&](int cc) {
    if (shift >= 0) return;
    if (lane < 4*NSLOT) {
      const int t = lane >> 2, prow = 4*dq + (lane & 3);
      const unsigned int row = lds0 + (cc % R)*BUFB + t*TAPB + dk*4096 + 256*prow;
      const unsigned int z = 0;
      if (dimg == 0) {
        if (col_end >= 128*dk && col_end < 128*dk + 128) {
          const int cw = col_end & 127;
          const unsigned int dst = row + 16*((cw >> 3) ^ cc_swz(prow)) + 2*(cw & 7);
          asm volatile("ds_write_b16 %0, %1" :: "v"(dst), "v"(z) : "memory");
        }
      } else if (f0 == 0 && dk == 0) {
        const unsigned int dst = row + 16*cc_swz(prow);
        asm volatile("ds_write_b16 %0, %1" :: "v"(dst), "v"(z) : "memory");
      }
    }
  };

  // ---- weights: ALL fragments of a chunk (this wave's MF row groups x NTAP taps) in one burst, one chunk ahead
  const int mfrag0 = (mtile*WM + wm)*MF;
  const uint4* wq[MF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) {         // row groups past M: any packed group (their outputs are not stored)
    const int fr = mfrag0 + mf < p.mfrags ? mfrag0 + mf : p.mfrags - 1;
    wq[mf] = p.wp + (long long)fr*p.ncc*CC_KH*64 + lane;
  }
  uint4 aw[2][NTAP][MF];
  auto a_burst = [&](int cc, auto set_tag) {
    constexpr int set = decltype(set_tag)::value;
    const int c = cc < p.ncc ? cc : p.ncc - 1;   // (past the end: the last chunk once more instead of a branch)
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        if (CC_ABL & 2) aw[set][t][mf] = make_uint4(c, t, mf, 0); else
        aw[set][t][mf] = wq[mf][(c*CC_KH + tap_i[t])*64];
      }
  };

  // ---- B fragments: lane parts of the transposing reads (cc_frag), relative to (ring slot, tap slot)
  unsigned int fr_lo[NF], fr_hi[NF];
  {
    const int g4 = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
    const int row = 8*(g4 >> 1) + q;
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      const int col0 = 32*(wn*NF + nf);
      const int chunk = ((col0 & 127) >> 3) + 2*(g4 & 1) + (pp >> 1);
      fr_lo[nf] = (col0 >> 7)*4096 + cc_off(row, chunk) + 8*(pp & 1);
      fr_hi[nf] = (col0 >> 7)*4096 + cc_off(row + 4, chunk) + 8*(pp & 1);
    }
  }

  f32x16 acc[NSET][MF][NF];
#pragma unroll
  for (int st_ = 0; st_ < NSET; ++st_)
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[st_][mf][nf][i] = 0.f;

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  // ---- prologue: chunks 0 .. R - 2 on their way, chunk 0 landed and patched
  a_burst(0, S0{});
#pragma unroll
  for (int c = 0; c < R - 1; ++c) burst(c);
  dma::wait_vm<U*(R - 2)>();
  patch(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  auto body = [&](int cc, auto cur_tag, auto nxt_tag) {
    constexpr int cur = decltype(cur_tag)::value;
    a_burst(cc + 1, nxt_tag);                      // BEFORE the DMA burst: see the header comment
    burst(cc + R - 1);                             // into the ring slot chunk cc - 1 was read from
    const unsigned int img = lds0 + (cc % R)*BUFB;
    s16x4 lo[2][NF], hi[2][NF];
    auto issue = [&](auto slot_tag, auto par_tag) {
      constexpr int slot = decltype(slot_tag)::value, par = decltype(par_tag)::value;
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        lo[par][nf] = dma::read_tr<slot*TAPB>(img + fr_lo[nf]);
        hi[par][nf] = dma::read_tr<slot*TAPB>(img + fr_hi[nf]);
      }
    };
    auto step = [&](auto t_tag) {
      constexpr int t = decltype(t_tag)::value;
      // fragment reads per slot: the pair form's odd tap reuses the even tap's fragments
      constexpr int slot = PAIR ? (t >> 1) : t, set = PAIR ? (t & 1) : 0;
      constexpr bool reads = !PAIR || (t & 1) == 0;
      constexpr int par = slot & 1;
      constexpr int tn = PAIR ? t + 2 - (t & 1) : t + 1;          // the tap whose fragments are requested next
      constexpr bool more = tn < NTAP;
      if constexpr (reads) {
        if constexpr (more) issue(std::integral_constant<int, PAIR ? (tn >> 1) : tn>{}, std::integral_constant<int, par ^ 1>{});
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
          if constexpr (more) dma::wait_lgkm<2*NF>(lo[par][nf], hi[par][nf]);
          else dma::wait_lgkm<0>(lo[par][nf], hi[par][nf]);
        }
      }
      typedef __attribute__((ext_vector_type(8))) short s16x8;
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        const bf16x8 af = __builtin_bit_cast(bf16x8, aw[cur][t][mf]);
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
          const s16x8 v = __builtin_shufflevector(lo[par][nf], hi[par][nf], 0, 1, 2, 3, 4, 5, 6, 7);
          if (CC_ABL & 4) { acc[set][mf][nf][0] += __builtin_bit_cast(float, (int)af[0]) + (float)v[0]; continue; }
          acc[set][mf][nf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, v), acc[set][mf][nf], 0, 0, 0);
        }
      }
    };
    issue(S0{}, S0{});
    step(std::integral_constant<int, 0>{});
    if constexpr (NTAP > 1) step(std::integral_constant<int, 1>{});
    if constexpr (NTAP > 2) step(std::integral_constant<int, 2>{});
    if constexpr (NTAP > 3) step(std::integral_constant<int, 3>{});
    if constexpr (NTAP > 4) step(std::integral_constant<int, 4>{});
    // chunk cc + 1 has landed once only the R - 2 younger bursts are pending (the weights of chunk cc + 1 were
    // requested before the youngest burst: landed as well)
    dma::wait_vm<U*(R - 2)>();
    patch(cc + 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
#pragma unroll 1
  for (int cc = 0; cc < p.ncc; cc += 2) {
    body(cc, S0{}, S1{});
    if (cc + 1 < p.ncc) body(cc + 1, S1{}, S0{});
  }
  dma::wait_vm<0>();                               // (the bursts past the end)

  // ---- D[m][frame] -> out[b][m][row][frame] (+ bias)
  const int oes = p.out_bf16 ? 2 : 4;        // bytes per output element
  char* out_b = static_cast<char*>(p.out) + (long long)b*p.out_bs*oes;
  char* out2_b = SEG && p.out_seg > 0 ? static_cast<char*>(p.out2) + (long long)b*p.out_bs*oes : nullptr;
#pragma unroll
  for (int set = 0; set < NSET; ++set) {
    const int orow = PAIR ? 2*r + set : r;
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        const int w = f0 + 32*(wn*NF + nf) + (lane & 31);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int m = 32*(mfrag0 + mf) + (i & 3) + 8*(i >> 2) + 4*(lane >> 5);
          if (m < p.M && w < p.Wout && (!(CC_ABL & 8) || acc[set][mf][nf][i] == 1234.5f)) {
            float v = acc[set][mf][nf][i];
            if (p.bias) v += p.bias[m];
            char* dst = out_b;
            int mc = m;
            if (SEG && p.out_seg > 0) {
              const int sg = (m >= p.out_seg) + (m >= 2*p.out_seg) + (m >= 3*p.out_seg);
              mc -= ((sg + 1) >> 1)*p.out_seg;
              if (sg & 1) dst = out2_b;
            }
            const long long idx = ((long long)mc*p.Hout + orow)*p.Wout + w;
            if (p.out_bf16) *reinterpret_cast<bf16_t*>(dst + idx*2) = f2bf(v);
            else *reinterpret_cast<float*>(dst + idx*4) = v;
          }
        }
      }
  }
}

// MODE 0: strided form (five taps); 1: transposed form, one output row per workgroup (three / two taps); 2: the pair form
template <int MF, int NF, int WM, int WN, bool SEG, int MODE>
__global__ __launch_bounds__(CC_THREADS) void cconv_rows_dma_kernel(const CConvParams p) {
  constexpr int NT = 32*NF*WN;
#ifndef CC_RING1
#define CC_RING1 4       // ring slots of the three- / two-tap forms (diagnostic builds: 2 = 48 KB of LDS, several workgroups per CU)
#endif
#ifndef CC_RING0
#define CC_RING0 3       // ... of the five-tap form
#endif
  // (the 32-row tile keeps 74 - 106 registers: with two ring slots = 48 KB of LDS two or three of its workgroups share a
  // CU, which pays more than the deeper ring there -- 157 -> 127, 140 -> 118, 89 -> 76 us on its three launch shapes;
  // the larger tiles lose as much: profiles/r06_cconv_ablation.txt, section 8)
  constexpr int R = MODE == 0 ? CC_RING0 : (MF == 1 ? 2 : CC_RING1);
  constexpr int NSLOT = MODE == 0 ? 5 : 3;
  __shared__ __attribute__((aligned(256))) unsigned char lds[R*NSLOT*(NT/128)*4096];
  int ftile = blockIdx.x, r = blockIdx.y, bz = blockIdx.z;     // XCD-aware order: as cconv_rows_kernel
  {
    const unsigned total = gridDim.x*gridDim.y*gridDim.z;
    if ((total & 7u) == 0) {
      const unsigned L = blockIdx.x + gridDim.x*(blockIdx.y + gridDim.y*blockIdx.z);
      unsigned g = (L & 7u)*(total >> 3) + (L >> 3);
      ftile = (int)(g % gridDim.x); g /= gridDim.x;
      r = (int)(g % gridDim.y); bz = (int)(g / gridDim.y);
    }
  }
  const int b = bz / p.mtiles, mtile = bz % p.mtiles;
  if constexpr (MODE == 0) {
    const int ti[5] = {0, 1, 2, 3, 4};
    const int tr[5] = {2*r - 2, 2*r - 1, 2*r, 2*r + 1, 2*r + 2};
    cconv_tile_dma<MF, NF, WM, WN, 5, SEG, false, R>(p, lds, b, r, ftile, mtile, ti, tr, 1);
  } else if constexpr (MODE == 2) {
    const int ti[5] = {0, 1, 2, 3, 4};
    const int tr[3] = {r + 1, r, r - 1};
    cconv_tile_dma<MF, NF, WM, WN, 5, SEG, true, R>(p, lds, b, r, ftile, mtile, ti, tr, -1);
  } else if (r & 1) {
    const int ti[2] = {1, 3};
    const int tr[2] = {(r + 1) >> 1, (r - 1) >> 1};
    cconv_tile_dma<MF, NF, WM, WN, 2, SEG, false, R>(p, lds, b, r, ftile, mtile, ti, tr, -1);
  } else {
    const int ti[3] = {0, 2, 4};
    const int tr[3] = {(r >> 1) + 1, r >> 1, (r >> 1) - 1};
    cconv_tile_dma<MF, NF, WM, WN, 3, SEG, false, R>(p, lds, b, r, ftile, mtile, ti, tr, -1);
  }
}
