// Persistent weight-stationary pointwise GEMM (bf16 MFMA, fp32 accumulate).
//
//   D[b][t][n] = sum_k A[b][t][k] * W[n][k]      for the FULL n range of one layer
//
// Why: with 128x128 tiles the 1x1 convolutions of the TCN block were bound by
// L2 -> LDS traffic (every tile re-streams its weights: ~12 TB/s measured) instead of
// HBM. Here one workgroup (8 waves) per CU keeps the whole weight matrix of the layer
// in REGISTERS -- wave w owns NSL output channels for all Kp inputs (NSL*Kp <= 16384
// -> <= 128 VGPRs) -- and streams 32-frame A tiles through a double-buffered LDS
// ring: one barrier per tile, A prefetched one tile ahead into registers. Each wave
// stages its accumulator sub-tile in a private LDS region and runs the row-wise
// epilogue on it (no second barrier).
//
// The loop body is branch-free around memory operations: every global access is a
// buffer_load / buffer_store through a per-item descriptor, so frames past the end of
// an item (and the padding tile of an odd tile count) read zeros and drop their stores
// in hardware. That keeps the compiler's s_waitcnt counted (vmcnt(N), never a drain)
// and the prefetches in flight across the barrier.
//
// Instances: 128->512 (pw1_fwd, bottleneck_dgrad, last pw2_dgrad), 512->256
// (pw2_fwd), 256->512 (pw2_dgrad). Anything else goes through gemm_rows_kernel.
#pragma once
#include "gemm_rows.cuh"
#include "tcn_kernels.cuh"

namespace brv {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, long long bytes) {
  const unsigned int n = bytes < 0 ? 0u : (bytes > 0xffffffffLL ? 0xffffffffu : (unsigned int)bytes);
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)n, 0x00020000);
}
__device__ __forceinline__ uint4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned int off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void buf_store16(__amdgpu_buffer_rsrc_t r, unsigned int off, const uint4& q) {
  u32x4 v; v.x = q.x; v.y = q.y; v.z = q.z; v.w = q.w;
  __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)off, 0, 0);
}
constexpr unsigned int kOob = 0xfffffff0u;     // offset that is out of range for any descriptor

template <int KP, int NSL, int WM, int NW = 8>
struct GemmWsCfg {
  static constexpr int NTHR = 64*NW;           // threads per workgroup
  static constexpr int WN = NW/WM;
  static constexpr int NP = WN*NSL;            // full output width
  static constexpr int BMW = 32*WM;            // frames per workgroup tile
  static constexpr int NF = NSL/32;            // 32x32 MFMA tiles per wave
  static constexpr int KS = KP/16;             // MFMA k-steps
  static constexpr int LDA = KP + 8;           // padded LDS row (conflict-free b128)
  static constexpr int LDW = NSL + 4;          // fp32 staging row of one wave
  static constexpr int ACH = BMW*KP/8/NTHR;    // 16-byte A chunks per thread
  static constexpr int kA = 2*BMW*LDA*2;
  static constexpr int kC = NW*32*LDW*4;
  static constexpr int kSmem = kA + kC + 2*KP*4;
};

// AT: 0 = A used as stored, 1 = PReLU + gLN affine applied while staging.
// CAT: A is the concatenation [p0 | p1] along k (split at a.K0).
template <int KP, int NSL, int WM, int EM, int AT, bool CAT, int NW = 8>
__global__ __launch_bounds__(64*NW) void gemm_ws_kernel(const GemmRowsParams p) {
  using C = GemmWsCfg<KP, NSL, WM, NW>;
  static_assert(C::ACH >= 1, "tile too small for the workgroup");
  __shared__ __attribute__((aligned(16))) unsigned char smem[C::kSmem];
  bf16_t* As = reinterpret_cast<bf16_t*>(smem);
  float* Cw_all = reinterpret_cast<float*>(smem + C::kA);
  float* scs = reinterpret_cast<float*>(smem + C::kA + C::kC);      // [KP] scale
  float* shs = scs + KP;                                             // [KP] shift

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / C::WN, wn = wid % C::WN;
  const int fr = lane & 31, fh = lane >> 5;
  const int T = p.T;
  const ASpec& a = p.a;
  const EpiSpec& e = p.e;
  float* Cw = Cw_all + wid*32*C::LDW;

  // ---- weights of this wave's channel slice -> registers ------------------------
  bf16x8 wf[C::NF][C::KS];
  {
    // blockIdx.y selects a group of NP output channels when the layer is wider than
    // what one workgroup keeps in registers
    const bf16_t* wbase = p.W + (long long)(blockIdx.y*C::NP + wn*NSL)*KP;
#pragma unroll
    for (int f = 0; f < C::NF; ++f)
#pragma unroll
      for (int s = 0; s < C::KS; ++s)
        wf[f][s] = *reinterpret_cast<const bf16x8*>(wbase + (long long)(32*f + fr)*KP
                                                    + 16*s + 8*fh);
  }

  // ---- tile schedule: contiguous range per workgroup ------------------------------
  const int tpi = ceil_div(T, C::BMW);                 // tiles per item
  const int total = tpi*p.batch;
  const int per = ceil_div(total, (int)gridDim.x);
  const int t_begin = blockIdx.x*per;
  const int t_end = min(total, t_begin + per);

  // ---- A staging geometry (fixed k chunk per thread) -------------------------------
  constexpr int CPR = KP/8;                            // chunks per frame
  constexpr int RSTEP = C::NTHR/CPR;
  const int kc = tid % CPR;
  const int arow0 = tid / CPR;                         // + RSTEP*ci
  const int kbase = kc*8;
  const float slope = (AT == 1 && a.slope) ? *a.slope : 1.f;
  const bool first = !CAT || kbase < a.K0;
  const unsigned int akoff = (unsigned int)((first ? kbase : kbase - a.K0)*2);

  auto load_tile = [&](int tile, bool valid, uint4 (&araw)[C::ACH]) {
    const int b = tile / tpi, t0 = (tile % tpi)*C::BMW;
    const __amdgpu_buffer_rsrc_t r0 = make_rsrc(
        reinterpret_cast<const bf16_t*>(a.p0) + (long long)b*a.bs0,
        valid ? (long long)T*a.ld0*2 : 0);
#pragma unroll
    for (int ci = 0; ci < C::ACH; ++ci) {
      const unsigned int t = (unsigned int)(t0 + arow0 + RSTEP*ci);
      if (!CAT) {
        araw[ci] = buf_load16(r0, t*(unsigned int)(a.ld0*2) + akoff);
      } else {
        const __amdgpu_buffer_rsrc_t r1 = make_rsrc(
            reinterpret_cast<const bf16_t*>(a.p1) + (long long)b*a.bs1,
            valid ? (long long)T*a.ld1*2 : 0);
        const uint4 x0 = buf_load16(r0, first ? t*(unsigned int)(a.ld0*2) + akoff : kOob);
        const uint4 x1 = buf_load16(r1, first ? kOob : t*(unsigned int)(a.ld1*2) + akoff);
        araw[ci] = make_uint4(x0.x | x1.x, x0.y | x1.y, x0.z | x1.z, x0.w | x1.w);
      }
    }
  };
  auto store_tile = [&](int tile, int buf, const uint4 (&araw)[C::ACH]) {
    const int t0 = (tile % tpi)*C::BMW;
    bf16_t* dst = As + buf*C::BMW*C::LDA;
    float sc[8], sh[8];
    if (AT == 1) {
      const float4 s0 = *reinterpret_cast<const float4*>(scs + kbase);
      const float4 s1 = *reinterpret_cast<const float4*>(scs + kbase + 4);
      const float4 h0 = *reinterpret_cast<const float4*>(shs + kbase);
      const float4 h1 = *reinterpret_cast<const float4*>(shs + kbase + 4);
      sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w;
      sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
      sh[0] = h0.x; sh[1] = h0.y; sh[2] = h0.z; sh[3] = h0.w;
      sh[4] = h1.x; sh[5] = h1.y; sh[6] = h1.z; sh[7] = h1.w;
    }
#pragma unroll
    for (int ci = 0; ci < C::ACH; ++ci) {
      const int row = arow0 + RSTEP*ci;
      uint4 q = araw[ci];
      if (AT == 1) {
        float f[8];
        unpack8(q, f);
        // rows past the end were read as zeros, but the affine shift would make them
        // non-zero: scale the result by the row validity
        const float live = (t0 + row < T) ? 1.f : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = live*(prelu(f[j], slope)*sc[j] + sh[j]);
        q = pack8(f);
      }
      *reinterpret_cast<uint4*>(dst + row*C::LDA + kbase) = q;
    }
  };
  // per-item gLN scale/shift table in LDS (rebuilt when the item changes: rare)
  int cur_item = -1;
  auto update_affine = [&](int b) {
    if (AT != 1 || b == cur_item) return;
    __syncthreads();
    if (tid < KP) {
      const NormStat ns = norm_stat(a.stats, b, a.inv_n, a.eps);
      const float g = tid < a.C ? a.gamma[tid] : 0.f;
      const float be = tid < a.C ? a.beta[tid] : 0.f;
      scs[tid] = ns.rstd*g;
      shs[tid] = be - ns.mean*ns.rstd*g;
    }
    __syncthreads();
    cur_item = b;
  };

  // ---- epilogue state (wave-private) -------------------------------------------------
  constexpr int CH = NSL/8;                            // 8-column chunks per staged row
  constexpr int RPP = 64/CH;                           // rows per pass
  constexpr int NPASS = 32/RPP;
  const int ech = lane % CH, erow0 = lane / CH;
  const int ncol = blockIdx.y*C::NP + wn*NSL + ech*8;  // global output column of the chunk
  float biasv[8], gam[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { biasv[j] = 0.f; gam[j] = 0.f; }
  if (EM == E_STORE) load8_masked(e.bias, ncol, e.N, biasv);
  const bool is_res = EM == E_RES_SKIP && ncol < e.Nsplit;
  if (EM == E_RES_SKIP) {
    if (is_res) load8_masked(e.bias, ncol, e.N, biasv);
    else load8_masked(e.bias2, ncol - e.Nsplit, e.N2, biasv);
  }
  if (EM == E_GLN_BWD) load8_masked(e.gamma, ncol, e.N, gam);
  // slope 1 makes prelu() the identity: no "has a slope" branch inside the loop
  const float eslope = (EM == E_GLN_BWD && e.src_slope) ? *e.src_slope
                     : (EM == E_STORE && e.stats_slope ? *e.stats_slope : 1.f);
  const bool want_stats = EM == E_STORE && e.stats_out != nullptr;
  float cmask[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) cmask[j] = (want_stats && ncol + j < e.N) ? 1.f : 0.f;
  const float skip_keep = (EM == E_RES_SKIP && !e.skip_init) ? 1.f : 0.f;
  double st_sum = 0.0, st_sq = 0.0;
  int st_item = -1;
  float colA[8], colB[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { colA[j] = 0.f; colB[j] = 0.f; }

  auto flush_stats = [&]() {
    if (st_item < 0) return;
    if (want_stats || EM == E_GLN_BWD) {
      const double s0 = wave_sum(st_sum), s1 = wave_sum(st_sq);
      double* dst = (EM == E_STORE) ? e.stats_out : e.sums_out;
      if (lane == 0) {
        atomic_add_f64(dst + 2*st_item, s0);
        atomic_add_f64(dst + 2*st_item + 1, s1);
      }
    }
    st_sum = 0.0; st_sq = 0.0;
  };

  // compute + epilogue of one tile whose A operand sits in LDS buffer `buf`
  auto process_tile = [&](int tile, bool valid, int buf) {
    const int b = valid ? tile / tpi : (t_end - 1) / tpi;
    const int t0 = (tile % tpi)*C::BMW + 32*wm;
    const long long recs = valid ? 1 : 0;
    // per-item descriptors: frames >= T fall outside and are dropped / read as zero
    const __amdgpu_buffer_rsrc_t rout = make_rsrc(
        reinterpret_cast<const bf16_t*>(e.out) + (long long)b*T*e.ldo, recs*T*e.ldo*2);
    __amdgpu_buffer_rsrc_t rc0 = rout, rc1 = rout;
    if (EM == E_RES_SKIP) {
      rc0 = make_rsrc(e.res_in + (long long)b*T*e.ld_res, recs*T*e.ld_res*2);
      rc1 = make_rsrc(e.skip + (long long)b*T*e.ld_skip, recs*T*e.ld_skip*4);
    } else if (EM == E_GLN_BWD) {
      rc0 = make_rsrc(e.src + (long long)b*T*e.ld_src, recs*T*e.ld_src*2);
    } else if (EM == E_ADD) {
      rc0 = make_rsrc(e.add_in + (long long)b*T*e.ld_add, e.add_in ? recs*T*e.ld_add*2 : 0);
    }
    // companion tensors of the row-wise epilogue: loads issued before the MFMAs
    uint4 comp0[NPASS], comp1[(EM == E_RES_SKIP) ? NPASS : 1];
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
      const unsigned int t = (unsigned int)(t0 + erow0 + RPP*pass);
      if (EM == E_RES_SKIP) {
        const unsigned int so = (t*(unsigned int)e.ld_skip + (unsigned int)(ncol - e.Nsplit))*4u;
        comp0[pass] = is_res
            ? buf_load16(rc0, (t*(unsigned int)e.ld_res + (unsigned int)ncol)*2u)
            : buf_load16(rc1, so);
        comp1[pass] = buf_load16(rc1, is_res ? kOob : so + 16u);
      } else if (EM == E_GLN_BWD) {
        comp0[pass] = buf_load16(rc0, (t*(unsigned int)e.ld_src + (unsigned int)ncol)*2u);
      } else if (EM == E_ADD) {
        comp0[pass] = buf_load16(rc0, (t*(unsigned int)e.ld_add + (unsigned int)ncol)*2u);
      } else {
        comp0[pass] = make_uint4(0, 0, 0, 0);
      }
    }

    f32x16 acc[C::NF];
#pragma unroll
    for (int f = 0; f < C::NF; ++f)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[f][i] = 0.f;
    const bf16_t* abuf = As + buf*C::BMW*C::LDA + (32*wm + fr)*C::LDA + 8*fh;
#pragma unroll
    for (int s = 0; s < C::KS; ++s) {
      const bf16x8 af = *reinterpret_cast<const bf16x8*>(abuf + 16*s);
#pragma unroll
      for (int f = 0; f < C::NF; ++f)
        acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, wf[f][s], acc[f], 0, 0, 0);
    }
    // accumulators -> this wave's private fp32 staging tile [32 frames][NSL]
#pragma unroll
    for (int f = 0; f < C::NF; ++f)
#pragma unroll
      for (int i = 0; i < 16; ++i)
        Cw[((i & 3) + 8*(i >> 2) + 4*fh)*C::LDW + 32*f + fr] = acc[f][i];

    if (valid && b != st_item) { flush_stats(); st_item = b; }
    NormStat es = {0.f, 1.f};
    if (EM == E_GLN_BWD) es = norm_stat(e.src_stats, b, e.inv_n, e.eps);

#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
      const int row = erow0 + RPP*pass;
      const unsigned int t = (unsigned int)(t0 + row);
      float v[8];
      {
        const float4 lo = *reinterpret_cast<const float4*>(Cw + row*C::LDW + ech*8);
        const float4 hi = *reinterpret_cast<const float4*>(Cw + row*C::LDW + ech*8 + 4);
        v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w;
        v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
      }
      const unsigned int ooff = (t*(unsigned int)e.ldo + (unsigned int)ncol)*2u;
      if (EM == E_STORE) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += biasv[j];
        const uint4 q = pack8(v);
        buf_store16(rout, ooff, q);
        float r[8]; unpack8(q, r);
        const float rm = (valid && (int)t < T) ? 1.f : 0.f;
        float ls = 0.f, lq = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float pv = rm*cmask[j]*prelu(r[j], eslope);
          ls += pv; lq += pv*pv;
        }
        st_sum += ls; st_sq += lq;
      } else if (EM == E_RES_SKIP) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += biasv[j];
        if (is_res) {                               // wave-uniform
          float r[8];
          unpack8(comp0[pass], r);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += r[j];
          buf_store16(rout, ooff, pack8(v));
        } else {
          const uint4 c0 = comp0[pass];
          const uint4 c1 = comp1[(EM == E_RES_SKIP) ? pass : 0];
          v[0] += skip_keep*__uint_as_float(c0.x); v[1] += skip_keep*__uint_as_float(c0.y);
          v[2] += skip_keep*__uint_as_float(c0.z); v[3] += skip_keep*__uint_as_float(c0.w);
          v[4] += skip_keep*__uint_as_float(c1.x); v[5] += skip_keep*__uint_as_float(c1.y);
          v[6] += skip_keep*__uint_as_float(c1.z); v[7] += skip_keep*__uint_as_float(c1.w);
          const unsigned int so = (t*(unsigned int)e.ld_skip + (unsigned int)(ncol - e.Nsplit))*4u;
          buf_store16(rc1, so, make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]),
                                          __float_as_uint(v[2]), __float_as_uint(v[3])));
          buf_store16(rc1, so + 16u, make_uint4(__float_as_uint(v[4]), __float_as_uint(v[5]),
                                                __float_as_uint(v[6]), __float_as_uint(v[7])));
        }
      } else if (EM == E_GLN_BWD) {
        float s[8], o[8];
        unpack8(comp0[pass], s);
        // A rows past the end are zero, so v == 0 there: no row mask needed
        float l1 = 0.f, l2 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float pv = prelu(s[j], eslope);
          const float xh = (pv - es.mean)*es.rstd;
          const float ev = gam[j]*v[j];
          o[j] = ev;
          l1 += ev; l2 += ev*xh;
          colA[j] += v[j]*xh; colB[j] += v[j];
        }
        st_sum += l1; st_sq += l2;
        buf_store16(rout, ooff, pack8(o));
      } else if (EM == E_ADD) {
        float r[8];
        unpack8(comp0[pass], r);                    // zeros when add_in is null
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += r[j];
        buf_store16(rout, ooff, pack8(v));
      }
    }
  };

  // ---- main loop: ONE copy of the tile body (code size is a first-order cost: these
  // kernels run ~40 us and a cold instruction fetch costs ~1 us per KiB), A prefetched
  // one tile ahead behind counted waits ------------------------------------------------
  uint4 araw[C::ACH];
  if (t_begin < t_end) {
    load_tile(t_begin, true, araw);
    int buf = 0;
#pragma unroll 1
    for (int tile = t_begin; tile < t_end; ++tile, buf ^= 1) {
      update_affine(tile / tpi);
      store_tile(tile, buf, araw);
      __syncthreads();
      load_tile(tile + 1, tile + 1 < t_end, araw);
      process_tile(tile, true, buf);
    }
  }
  flush_stats();
  if (EM == E_GLN_BWD) {
    // column sums: reduce over the lanes that share a chunk (same lane % CH)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float x = colA[j], y = colB[j];
#pragma unroll
      for (int off = 32; off >= CH; off >>= 1) {
        x += __shfl_xor(x, off, 64);
        y += __shfl_xor(y, off, 64);
      }
      colA[j] = x; colB[j] = y;
    }
    if (lane < CH) {
      const long long ro = e.n_rep > 1 ? (long long)(blockIdx.x % e.n_rep)*e.rep_stride : 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int n = ncol + j;
        if (n < e.N) {
          atomic_add_f32(e.dgamma + ro + n, colA[j]);
          atomic_add_f32(e.dbeta + ro + n, colB[j]);
        }
      }
    }
  }
}

}  // namespace brv
