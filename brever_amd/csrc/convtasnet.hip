// Conv-TasNet forward / backward orchestration + C ABI (include/brever_hip.h).
//
// Reference being replaced: brever/models/convtasnet/convtasnet.py:66-72 (forward),
// :100-260 (Encoder / Decoder / TCN / Conv1DBlock) and their autograd. The kernel
// sequence and the saved-activation plan are described in DESIGN.md.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

#include "../../include/brever_hip.h"
#include "gemm_rows.cuh"
#include "gemm_wgrad.cuh"
#include "gemm_ws.cuh"
#include "dwpw2_fused.cuh"
// Two kernel organisations that were built, measured slower and rejected are NOT part of the
// default library (round 6): `tools/mkvariant.sh <tag> -DBRV_WITH_VARIANTS` builds a library that holds them, the
// default one answers BRV_OPT_DWPW2_V2 / BRV_OPT_BWD_PERSIST with an error.
#ifdef BRV_WITH_VARIANTS
#include "dwpw2_fused_v2.cuh"
#else
constexpr int D2_TT = 64;            // (tile of the whole-row form; only its launch arithmetic needs the name)
#endif
#include "gemm_wgrad_full.cuh"
#include "gemm_wgrad_full128.cuh"

constexpr int kWgSplit = 4;          // item splits of the [res | skip] weight-gradient launch
#include "prep.cuh"
#include "tcn_kernels.cuh"
#include "bwd_fused.cuh"
#ifdef BRV_WITH_VARIANTS
#include "bwd_fused_p.cuh"
#endif
#include "pw1_bwd.cuh"
#include "cln_kernels.cuh"

using namespace brv;

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIP_OK(expr)                                                        \
  do {                                                                      \
    hipError_t e_ = (expr);                                                 \
    if (e_ != hipSuccess)                                                   \
      return fail((int)e_, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

inline long long align_up(long long x, long long a) { return (x + a - 1)/a*a; }

// ---- per-call options (include/brever_hip.h: brv_launch_opts) -------------------------------------
// The options of the running API call, thread-local for its duration (OptsScope in every entry point):
// the launch helpers below read them without threading an argument through ~40 call sites. Nothing
// here outlives a call and nothing is shared between threads: no process-global state.
const brv_launch_opts kDefaultOpts = {sizeof(brv_launch_opts), 0u, 8, 0, nullptr};
thread_local const brv_launch_opts* t_opts = &kDefaultOpts;
struct OptsScope {
  const brv_launch_opts* prev;
  explicit OptsScope(const brv_launch_opts* o) : prev(t_opts) {
    t_opts = (o && o->size >= sizeof(brv_launch_opts)) ? o : &kDefaultOpts;
  }
  ~OptsScope() { t_opts = prev; }
};
inline bool opt(uint32_t flag) { return (t_opts->flags & flag) != 0; }
inline bool fwd_fuse_requested() { return !opt(BRV_OPT_NO_FWD_FUSE) && !opt(BRV_OPT_NO_WS); }
inline bool bwd_fuse_requested() { return !opt(BRV_OPT_NO_BWD_FUSE); }
inline bool pw1_rc_requested() { return !opt(BRV_OPT_NO_PW1_RC) && !opt(BRV_OPT_NO_DZ1_FUSE); }
// the weight gradient rebuilds dz1 too (nothing stored) -- else the data-gradient kernel stores dz1
inline bool pw1_rc_wgrad() { return opt(BRV_OPT_PW1_RC_WGRAD) || opt(BRV_OPT_PW1_RC_TILES); }

// ---- optional per-launch event timing (bench / profiling only) ---------------
// A profiler object the caller owns (brv_prof_create); launches of calls whose options carry it are
// bracketed by two events on the launch stream; brv_prof_collect() aggregates them per label
// together with the algorithmic FLOPs / bytes.
struct ProfEntry { const char* label; hipEvent_t a, b; double flops, bytes; };
struct Prof { std::vector<ProfEntry> entries; bool by_dil; };
struct ProfScope {
  hipStream_t st; Prof* prof;
  ProfScope(const char* label, double flops, double bytes, hipStream_t s)
      : st(s), prof(static_cast<Prof*>(t_opts->prof)) {
    if (!prof) return;
    ProfEntry e; e.label = label; e.flops = flops; e.bytes = bytes;
    (void)hipEventCreate(&e.a); (void)hipEventCreate(&e.b);
    (void)hipEventRecord(e.a, st);
    prof->entries.push_back(e);
  }
  ~ProfScope() { if (prof) (void)hipEventRecord(prof->entries.back().b, st); }
};
inline bool prof_by_dil() { return t_opts->prof && static_cast<Prof*>(t_opts->prof)->by_dil; }

struct BlockOff {
  long long conv_w, conv_b, dconv_w, dconv_b, res_w, res_b, skip_w, skip_b,
      n1_g, n1_b, n2_g, n2_b, prelu1, prelu2;
  long long p_c1_f, p_c1_b, p_rs_f, p_rs_b;     // prepared (bf16 elements)
  long long p_c1_fp, p_rs_fp, p_rs_bp;          // fragment-order copies (persistent GEMMs)
  // lazy second norm (fused forward): [res | skip] weights times gamma_2, always Bnp + Scp rows
  // (zero residual rows in the last block), plain and in fragment order; p_lazy: floats
  // v0[n] = bias[n] + sum_k W[n][k] beta_2[k], v1[n] = sum_k bf16(W[n][k] gamma_2[k]), n < Bnp + Scp
  long long p_rs_g, p_rs_gp, p_lazy;
};

struct Layout {
  int N, K, Bn, H, Sc, P, nb, S, hop, causal;
  int Np, Kfp, Bnp, Hp, Scp;
  long long enc_w, dec_w, ln_g, ln_b, bott_w, bott_b, tcn_prelu, out_w, out_b, n_params;
  long long p_enc, p_dec_f, p_dec_b, p_bott_f, p_bott_b, p_out_f, p_out_b, p_stamp, n_prepared;
  std::vector<BlockOff> blk;
  std::vector<long long> tensor_offsets;

  // the fused forward runs for the default widths, non-causal, kernel_size 3 (see brv_ctn_forward);
  // `fusable` is a property of the architecture (it sizes the workspace), `fused_fwd` of the call
  bool fusable() const {
    return !causal && P == 3 && H == 512 && Bn == 128 && Sc == 128 && nb <= 24 /* kWgMaxProb */;
  }
  bool fused_fwd() const { return fusable() && fwd_fuse_requested(); }
  // first-conv backward without the stored z1 / dz1 (pw1_bwd.cuh): default widths of that layer
  bool pw1_rc() const { return !causal && Hp == RC_H && Bnp == RC_N && pw1_rc_requested(); }

  int init(const brv_ctn_config* c) {
    if (!c) return fail(-1, "null config");
    if (c->filters < 1 || c->filter_length < 2 || c->bottleneck_channels < 1 ||
        c->hidden_channels < 1 || c->skip_channels < 1 || c->layers < 1 ||
        c->repeats < 1 || c->output_sources < 1)
      return fail(-1, "invalid Conv-TasNet hyper-parameters");
    if (c->kernel_size < 1 || c->kernel_size > 5)
      return fail(-2, "kernel_size must be in [1, 5] in the HIP path");
    N = c->filters; K = c->filter_length; Bn = c->bottleneck_channels;
    H = c->hidden_channels; Sc = c->skip_channels; P = c->kernel_size;
    nb = c->layers*c->repeats; S = c->output_sources; hop = K/2; causal = c->causal != 0;
    Np = round_up(N, 64); Kfp = round_up(K, 64); Bnp = round_up(Bn, 64);
    Hp = round_up(H, 64); Scp = round_up(Sc, 64);
    long long o = 0;
    auto take = [&](long long n) { tensor_offsets.push_back(o); long long r = o; o += n; return r; };
    enc_w = take((long long)N*K);
    dec_w = take((long long)N*K);
    ln_g = take(N); ln_b = take(N);
    bott_w = take((long long)Bn*N); bott_b = take(Bn);
    blk.resize(nb);
    for (int i = 0; i < nb; ++i) {
      BlockOff& b = blk[i];
      b.conv_w = take((long long)H*Bn); b.conv_b = take(H);
      b.dconv_w = take((long long)H*P); b.dconv_b = take(H);
      if (i < nb - 1) { b.res_w = take((long long)Bn*H); b.res_b = take(Bn); }
      else { b.res_w = -1; b.res_b = -1; }
      b.skip_w = take((long long)Sc*H); b.skip_b = take(Sc);
      b.n1_g = take(H); b.n1_b = take(H); b.n2_g = take(H); b.n2_b = take(H);
      b.prelu1 = take(1); b.prelu2 = take(1);
    }
    tcn_prelu = take(1);
    out_w = take((long long)S*N*Sc); out_b = take((long long)S*N);
    n_params = o;
    // prepared bf16 operands
    long long q = 0;
    auto ptake = [&](long long n) { long long r = q; q += align_up(n, 128); return r; };
    p_enc = ptake((long long)Np*Kfp);
    p_dec_f = ptake((long long)Kfp*Np);
    p_dec_b = ptake((long long)Np*Kfp);
    p_bott_f = ptake((long long)Bnp*Np);
    p_bott_b = ptake((long long)Np*Bnp);
    for (int i = 0; i < nb; ++i) {
      const int rs = (i < nb - 1 ? Bnp : 0) + Scp;
      blk[i].p_c1_f = ptake((long long)Hp*Bnp);
      blk[i].p_c1_b = ptake((long long)Bnp*Hp);
      blk[i].p_rs_f = ptake((long long)rs*Hp);
      blk[i].p_rs_b = ptake((long long)Hp*rs);
      blk[i].p_c1_fp = ptake((long long)Hp*Bnp);
      blk[i].p_rs_fp = ptake((long long)rs*Hp);
      blk[i].p_rs_bp = ptake((long long)Hp*rs);
      blk[i].p_rs_g = ptake((long long)(Bnp + Scp)*Hp);
      blk[i].p_rs_gp = ptake((long long)(Bnp + Scp)*Hp);
      blk[i].p_lazy = ptake(4LL*(Bnp + Scp));           // 2 x (Bnp + Scp) floats
    }
    p_out_f = ptake((long long)S*Np*Scp);
    p_out_b = ptake((long long)Scp*S*Np);
    p_stamp = ptake(128);                              // mode stamp (one int): see mode_stamp()
    n_prepared = q;
    return 0;
  }

  long long frames(long long L) const {
    long long pad = ((K - L) % hop + hop) % hop;      // Python modulo
    long long Lp = L + pad;
    if (Lp < K) return 0;
    return (Lp - K)/hop + 1;
  }
};

struct Workspace {
  long long w, x, z1, z2, skip, m, y, stats, sums, dpre, dw1, gskip, gout, eA, eB,
      e0, dwt, vg, gcopy, total;
  long long gcopy_stride, eB_stride;
  long long wgpart;                   // partial tiles of the split [res | skip] weight gradient
  // causal (cLN) model only: materialised normalised tensors, per-frame statistics tables,
  // identity operands that let the non-causal kernels run as plain convolutions
  long long h1, h2, wn, ctab, ctab_stride, cfs, cbt, ident, fake_stats, scratch_stats;
  long long vg_stride, vg_bytes;      // replicated vector-gradient block (floats / bytes)
  long long x_stride, z_stride;       // bytes between consecutive blocks' buffers
  long long u, u_stride;              // fused forward: unfinished [res | skip] products per block
  long long stamp;                    // mode of the forward call that filled the workspace (mode_stamp())
  long long stats_bytes;
  void init(const Layout& l, long long B, long long T) {
    long long o = 0;
    auto take = [&](long long bytes) { long long r = o; o += align_up(bytes, 256); return r; };
    const long long BT = B*T;
    w = take(BT*l.Np*2);
    x_stride = align_up(BT*l.Bnp*2, 256);
    x = take(x_stride*l.nb);
    z_stride = align_up(BT*l.Hp*2, 256);
    z1 = take(z_stride*l.nb);
    z2 = take(z_stride*l.nb);
    skip = take(BT*l.Scp*4);
    m = take(BT*l.S*l.Np*2);
    y = take(BT*l.S*l.Np*2);
    stats_bytes = (long long)(1 + 2*l.nb)*B*kStatStride*8;
    stats = take(stats_bytes);
    sums = take(stats_bytes);
    dpre = take(BT*l.S*l.Np*2);
    dw1 = take(BT*l.S*l.Np*2);
    // [g_out | g_skip] side by side (row stride Bnp + Scp): the concatenated operand of
    // the [res | skip] data/weight gradients is then ONE plain tensor
    gout = take(BT*(l.Bnp + l.Scp)*2);
    gskip = gout + (long long)l.Bnp*2;
    eA = take(BT*l.Hp*2);
    // kept per block until the grouped weight-gradient launches at the end of backward
    eB_stride = align_up(BT*l.Hp*2, 256);
    eB = take(eB_stride*l.nb);
    gcopy_stride = align_up(BT*l.Bnp*2, 256);
    gcopy = take(gcopy_stride*l.nb);
    e0 = take(BT*l.Np*2);
    dwt = take(BT*l.Np*2);
    // (+ 2H floats of scratch per replica for the outputs the causal path discards)
    vg_stride = align_up(2LL*l.N + (long long)l.nb*l.H*(5 + l.P) + 1 + 2*l.nb + 2LL*l.Hp, 64);
    vg_bytes = vg_stride*kReplicas*4;
    vg = take(vg_bytes);
    wgpart = take((long long)2*kWgSplit*l.nb*W2_G*l.H*4);      // (the 128-wide form splits the items 8 ways)
    u_stride = align_up(BT*(l.Bnp + l.Scp)*2, 256);
    u = l.fusable() ? take(u_stride*l.nb) : 0;
    stamp = take(256);
    h1 = h2 = wn = ctab = cfs = cbt = ident = fake_stats = scratch_stats = 0; ctab_stride = 0;
    if (l.causal) {
      h1 = take(z_stride*l.nb);
      h2 = take(z_stride*l.nb);
      wn = take(BT*l.Np*2);
      ctab_stride = align_up(BT*2*4, 256);
      ctab = take(ctab_stride*(1 + 2*l.nb));
      cfs = take(BT*2*4);
      cbt = take(BT*2*4);
      const long long cmax = l.Hp > l.Np ? l.Hp : l.Np;
      ident = take((2*cmax + 64)*4);              // ones | zeros | 1.0f
      fake_stats = take(B*kStatStride*8);
      scratch_stats = take(B*kStatStride*8);
    }
    total = o;
  }
};

// ---------------------------------------------------------------------------
template <int BN, int AK, int EM>
int launch_gemm_rows_t(const GemmRowsParams& p0, int batch, hipStream_t st) {
  GemmRowsParams p = p0;
  p.n_ttiles = ceil_div(p.T, GR_BM); p.n_ntiles = ceil_div(p.Np, BN); p.batch = batch;
  dim3 grid(p.n_ttiles*p.n_ntiles*batch);
  hipLaunchKernelGGL((gemm_rows_kernel<BN, AK, EM>), grid, dim3(256), 0, st, p);
  HIP_OK(hipGetLastError());
  return 0;
}

#ifdef BRV_DIAG
// records the 100 MHz real-time counter: slot 0 = end of the kernel before, slot 1 = start
// of the kernel after a stamped launch (slots 2 / 3 = first entry / last exit inside it)
__global__ void diag_stamp_kernel(long long* out, int slot) {
  long long t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  if (threadIdx.x == 0) {
    out[slot] = t;
    if (slot == 0) { out[2] = 0x7fffffffffffffffLL; out[3] = 0; }
  }
}
// diagnostic builds only: a leaked 1 MiB device buffer for cycle stamps
long long* debug_buffer() {
  static long long* buf = nullptr;
  if (!buf) {
    if (hipMalloc(&buf, 1 << 20) != hipSuccess) return nullptr;
    (void)hipMemset(buf, 0, 1 << 20);
  }
  return buf;
}
#endif

// ---- mode stamps -----------------------------------------------------------------------
// prepare / forward / backward decide between the fused and the three-launch kernels from the options of
// THEIR OWN call; operands prepared for one mode (gamma-folded or plain [res | skip] weights) and a workspace
// filled in one mode (u tensors or a finished skip sum) are garbage to the other. prepare stamps `prepared`,
// forward checks that stamp and stamps the workspace, backward checks the workspace: a mismatch (a C-ABI
// caller passing BRV_OPT_NO_FWD_FUSE to one call and not to the next) poisons the results with NaN instead
// of returning plausible numbers. One-thread kernels on the call's stream: no host synchronisation.
inline int mode_stamp(bool fused_fwd) { return fused_fwd ? 0x46555345 : 0x504c4149; }    // 'FUSE' / 'PLAI'
__global__ void stamp_write_kernel(int* dst, int value) { *dst = value; }
__global__ void stamp_check_f64_kernel(const int* stamp, int expect, double* poison, long long n) {
  if (*stamp == expect) return;
  const double nan = __longlong_as_double(0x7ff8000000000000LL);
  for (long long i = threadIdx.x; i < n; i += blockDim.x) poison[i] = nan;
}
// The start of a forward / backward call in ONE launch instead of two fills and two one-thread kernels in a row
// (each ~5 us of an otherwise idle chip at the step boundary: profiles/r05_step_boundary.txt): region `a` (doubles)
// <- 0, or NaN when the stamp at `check` is not `expect`; region `b` <- 0; `write` <- value.
struct CallInitParams {
  double* a; long long a_n; void* b; long long b_bytes;
  const int* check; int expect; int* write; int value;
};
__global__ __launch_bounds__(256) void call_init_kernel(const CallInitParams p) {
  const bool bad = *p.check != p.expect;
  const double fill = bad ? __longlong_as_double(0x7ff8000000000000LL) : 0.0;
  const long long tid = (long long)blockIdx.x*256 + threadIdx.x, nth = (long long)gridDim.x*256;
  for (long long i = tid; i < p.a_n; i += nth) p.a[i] = fill;
  if (p.b) {
    unsigned char* b = (unsigned char*)p.b;
    // (sizes are multiples of 4: fp32 / fp64 tensors) 16-byte stores over the aligned middle, words at the ends
    long long head = (16 - ((unsigned long long)b & 15)) & 15;
    if (head > p.b_bytes) head = p.b_bytes;
    const long long n16 = (p.b_bytes - head) >> 4, tail0 = head + (n16 << 4);
    uint4* mid = reinterpret_cast<uint4*>(b + head);
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (long long i = tid; i < n16; i += nth) mid[i] = z;
    for (long long i = tid*4; i < head; i += nth*4) *reinterpret_cast<unsigned int*>(b + i) = 0u;
    for (long long i = tail0 + tid*4; i < p.b_bytes; i += nth*4) *reinterpret_cast<unsigned int*>(b + i) = 0u;
  }
  if (tid == 0 && p.write) *p.write = p.value;
}
inline int launch_call_init(double* a, long long a_n, void* b, long long b_bytes, const int* check, int expect,
                            int* write, int value, hipStream_t st) {
  CallInitParams ip; ip.a = a; ip.a_n = a_n; ip.b = b; ip.b_bytes = b_bytes; ip.check = check; ip.expect = expect;
  ip.write = write; ip.value = value;
  const long long units = a_n + (b_bytes >> 4);
  int gx = (int)std::min<long long>(512, (units + 1023)/1024);
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(call_init_kernel, dim3(gx), dim3(256), 0, st, ip);
  return (int)hipGetLastError();
}

// Workgroups of a persistent launch: one per CU -- or opts.cu_eighths/8 of that while two kernel
// chains share the chip: with 8 items per chain a full-width launch has exactly one tile per workgroup
// and both chains' launches start and drain in lockstep; at 7/8 the chains interleave (2085 -> 2132
// utt/s). Worse for a single chain (2014 -> 1948), hence an option of the call.
int device_cus() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
      v = 256;
    return v;
  }();                                  // (a constant of the machine, not state)
  return n;
}
int num_cus() {
  const int n = device_cus();
  const int e = t_opts->cu_eighths >= 1 && t_opts->cu_eighths <= 8 ? t_opts->cu_eighths : 8;
  const int m = n*e/8/8*8;
  return m >= 8 ? m : n;
}

// Persistent weight-stationary fast path (gemm_ws.cuh) for the hot TCN shapes.
template <int KP, int NSL, int WM, int EM, int AT, bool CAT, int NW = 8>
int launch_gemm_ws(const GemmRowsParams& p0, int batch, hipStream_t st) {
  using C = GemmWsCfg<KP, NSL, WM, NW>;
  GemmRowsParams p = p0;
  p.batch = batch;
#ifdef BRV_DIAG
  if (const char* d = getenv("BRV_DBG")) p.dbg = atoi(d);
  if (p.dbg & (64 | 512)) p.dbg_out = debug_buffer();
#endif
  if (!CAT && p.a.K0 <= 0) {          // everything comes from the second source
    p.a.p0 = p.a.p1; p.a.ld0 = p.a.ld1; p.a.bs0 = p.a.bs1; p.a.K0 = 1 << 30;
  }
  const int total = ceil_div(p.T, C::BMW)*batch;
  const int groups = p.Np / C::NP;             // column groups (blockIdx.y)
  // 8-wave workgroups fill a CU alone; 4-wave ones are sized so that 2-3 of them share
  // a CU and progress independently between their own barriers
  int grid = (NW == 8 ? num_cus() : 2*num_cus())/groups;
  if (grid > total) grid = total;
  if (grid < 1) grid = 1;
#ifdef BRV_DIAG
  if (p.dbg & 512) hipLaunchKernelGGL(diag_stamp_kernel, dim3(1), dim3(64), 0, st, debug_buffer() + 131000, 0);
#endif
  hipLaunchKernelGGL((gemm_ws_kernel<KP, NSL, WM, EM, AT, CAT, NW>), dim3(grid, groups),
                     dim3(64*NW), 0, st, p);
#ifdef BRV_DIAG
  if (p.dbg & 512) hipLaunchKernelGGL(diag_stamp_kernel, dim3(1), dim3(64), 0, st, debug_buffer() + 131000, 1);
#endif
  HIP_OK(hipGetLastError());
  return 0;
}

template <int AK, int EM>
int launch_gemm_rows(const GemmRowsParams& p, int batch, hipStream_t st,
                     const char* label = "gemm_rows", double bytes = 0) {
  if (p.T <= 0 || batch <= 0) return 0;
  ProfScope prof(label, 2.0*batch*p.T*(double)p.Np*p.Kp, bytes, st);
  if (AK == A_BF16 && p.a.nsrc <= 1 && !opt(BRV_OPT_NO_WS)) {
    const bool plain = p.a.slope == nullptr && p.a.stats == nullptr;
    const bool full = p.a.slope != nullptr && p.a.stats != nullptr;
    const bool one_src = p.a.K0 >= p.Kp || p.a.K0 <= 0;
    if ((EM == E_STORE || EM == E_GLN_BWD) && p.Kp == 128 && p.Np == 512 && plain && one_src)
      return launch_gemm_ws<128, 64, 1, (EM == E_STORE ? E_STORE : E_GLN_BWD), 0, false, 8>(p, batch, st);
    if (EM == E_RES_SKIP && p.Kp == 512 && p.Np == 256 && full && one_src)
      return launch_gemm_ws<512, 32, 1, E_RES_SKIP, 1, false>(p, batch, st);
    if (EM == E_GLN_BWD && p.Kp == 256 && p.Np == 512 && plain && one_src)
      return launch_gemm_ws<256, 32, 1, E_GLN_BWD, 0, false, 8>(p, batch, st);
  }
  if (p.Kp % GR_BK != 0 || p.Np % 64 != 0) return fail(-1, "gemm_rows: unpadded dims");
  if (p.Np % 128 == 0) return launch_gemm_rows_t<128, AK, EM>(p, batch, st);
  return launch_gemm_rows_t<64, AK, EM>(p, batch, st);
}

// Workgroups a weight-gradient launch aims for (tiles x frame splits x problems). More splits fill
// the CUs but every split adds its partial tile to the output with float atomics, so the best count
// depends on the output size: measured per label on MI355X (BRV_WG_TARGET[_<label>] override).
inline int wgrad_target(const char* label, int dflt) {
  (void)label;
  return t_opts->wg_target != 0 ? t_opts->wg_target : dflt;
}

template <int BH, int HK>
int launch_wgrad_t(WgradGroupParams& gp, hipStream_t st, int target) {
  WgradParams& p = gp.base;
  const int nprob = gp.nprob > 0 ? gp.nprob : 1;
  const int tiles = ceil_div(p.Gp, WG_BG)*(p.Hp/BH);
  const int total = p.B*ceil_div(p.T, WG_BT);
  int ns = ceil_div(target > 0 ? target : 384, tiles*nprob);
  if (target < 0) {
    // item-aligned splits (one item per split measured best for the grouped first-conv gradient at
    // B = 16, both for 24 blocks in one launch and for 8 per gradient bucket), within 384 .. 2048
    // workgroups
    int nb = p.B;
    while (tiles*nprob*nb > 2048 && nb % 2 == 0) nb /= 2;
    if (nb > ns) ns = nb;
  }
  if (ns > total) ns = total;
  if (ns < 1) ns = 1;
  p.nsplit = ns;
  dim3 grid(tiles, ns, nprob);
  hipLaunchKernelGGL((gemm_wgrad_kernel<BH, HK>), grid, dim3(256), 0, st, gp);
  HIP_OK(hipGetLastError());
  return 0;
}
// `gp.base` carries the shared dimensions / strides; `gp.prob[0..nprob)` the tensors
// (nprob == 0: a single problem described entirely by `gp.base`).
template <int HK>
int launch_wgrad_group(WgradGroupParams& gp, hipStream_t st, const char* label,
                       double bytes, int target_dflt = 384) {
  const WgradParams& p = gp.base;
  if (p.T <= 0 || p.B <= 0) return 0;
  const int nprob = gp.nprob > 0 ? gp.nprob : 1;
  ProfScope prof(label, 2.0*nprob*p.B*p.T*(double)p.Gp*p.Hp, bytes*nprob, st);
  if (p.Hp % 64 != 0) return fail(-1, "wgrad: unpadded dims");
  const int target = wgrad_target(label, target_dflt);
  if (p.Hp % 128 == 0) return launch_wgrad_t<128, HK>(gp, st, target);
  return launch_wgrad_t<64, HK>(gp, st, target);
}
template <int HK>
int launch_wgrad(WgradParams& p, hipStream_t st, const char* label = "wgrad",
                 double bytes = 0, int target_dflt = 384) {
  WgradGroupParams gp;
  gp.base = p; gp.nprob = 0;
  return launch_wgrad_group<HK>(gp, st, label, bytes, target_dflt);
}

template <template <int> class F, typename... Args>
int dispatch_p(int P, Args&&... args) {
  switch (P) {
    case 1: return F<1>::run(args...);
    case 2: return F<2>::run(args...);
    case 3: return F<3>::run(args...);
    case 4: return F<4>::run(args...);
    case 5: return F<5>::run(args...);
  }
  return fail(-2, "unsupported kernel_size");
}
template <int P> struct DwFwd {
  static int run(const DwParams& p, hipStream_t st) {
    ProfScope prof("dwconv_fwd", 2.0*P*p.B*p.T*(double)p.Cp, 4.0*p.B*p.T*(double)p.Cp, st);
    dim3 grid(ceil_div(p.T, DW_TT_F)*p.B);
    hipLaunchKernelGGL((dwconv_fwd_kernel<P>), grid, dim3(256), 0, st, p);
    HIP_OK(hipGetLastError());
    return 0;
  }
};
template <int P> struct DwBwd {
  static int run(const DwParams& p, hipStream_t st) {
    if (p.z2in != nullptr) {      // gLN_2 backward fused: dz2 built once per element in LDS
      static const char* by_dil[9] = {"dwconv_bwd_d1", "dwconv_bwd_d2", "dwconv_bwd_d4", "dwconv_bwd_d8", "dwconv_bwd_d16",
                                      "dwconv_bwd_d32", "dwconv_bwd_d64", "dwconv_bwd_d128", "dwconv_bwd_dx"};
      int lg = 0; while ((1 << lg) < p.dil && lg < 8) ++lg;
      ProfScope prof(prof_by_dil() ? by_dil[lg] : "dwconv_bwd", 4.0*P*p.B*p.T*(double)p.Cp, 8.0*p.B*p.T*(double)p.Cp, st);
      const int R = hl_rows_per_tooth(p.dil), K = HL_TT/R;
      const int tiles = ceil_div(p.dil, R)*ceil_div((p.T - 1)/p.dil + 1, K);
      dim3 grid(tiles*p.B*(p.Cp/HL_CG));
      const size_t lds = (size_t)hl_window_rows(p.dil, P)*HL_CG*2;
      hipLaunchKernelGGL((dwconv_bwd_halo_kernel<P>), grid, dim3(256), lds, st, p);
      HIP_OK(hipGetLastError());
      return 0;
    }
    ProfScope prof("dwconv_bwd", 4.0*P*p.B*p.T*(double)p.Cp, 6.0*p.B*p.T*(double)p.Cp, st);
    dim3 grid(ceil_div(p.T, DW_TT_B)*p.B);
    hipLaunchKernelGGL((dwconv_bwd_kernel<P>), grid, dim3(256), 0, st, p);
    HIP_OK(hipGetLastError());
    return 0;
  }
};

// [res | skip] data gradient + gLN_2 / PReLU_2 backward + transposed depthwise stencil in one launch
// (bwd_fused.cuh); `p.d` as for DwBwd with z2in set
template <int P> struct DwBwdFused {
  static int run(const BwdFusedParams& p0, hipStream_t st) {
    BwdFusedParams p = p0;
    const DwParams& d = p.d;
#ifdef BF_STAMP
    p.dbg = debug_buffer();
#endif
    bf_tile_shape(d.T, d.dil, P, p.K, p.R);
    const int tiles = ceil_div(d.dil, p.R)*ceil_div((d.T - 1)/d.dil + 1, p.K);
    // algorithmic bytes: g (256-wide) + z2 + z1 read, e1 written
    static const char* by_dil[9] = {"dwpw2_bwd_d1", "dwpw2_bwd_d2", "dwpw2_bwd_d4", "dwpw2_bwd_d8", "dwpw2_bwd_d16",
                                    "dwpw2_bwd_d32", "dwpw2_bwd_d64", "dwpw2_bwd_d128", "dwpw2_bwd_dx"};
    int lg = 0; while ((1 << lg) < d.dil && lg < 8) ++lg;
    ProfScope prof(prof_by_dil() ? by_dil[lg] : "dwpw2_bwd", 2.0*d.B*d.T*(double)d.Cp*(p.Kg + 2*P),
                   2.0*d.B*d.T*((double)p.Kg + 3.0*d.Cp), st);
    dim3 grid(ceil_div(tiles*d.B, 8)*8*(d.Cp/HL_CG));     // whole runs of 8 tiles (XCD map of the kernel)
    if (d.C != d.Cp) return fail(-1, "fused backward: channel count must be a multiple of 64");
#ifndef BF_NT
#define BF_NT 2
#endif
    const int nt = opt(BRV_OPT_BWD_PERSIST) ? BF_NT : 1;       // (opt-in: 83 against 73 us per launch)
    if (nt > 1 && p.Kg == 256) {
#ifdef BRV_WITH_VARIANTS
      // persistent form (bwd_fused_p.cuh): a workgroup walks `nt` tiles of its channel group
      dim3 gridp(ceil_div(ceil_div(tiles*d.B, 8), nt)*8*(d.Cp/HL_CG));
      hipLaunchKernelGGL((dwconv_bwd_fused_p_kernel<P, 256>), gridp, dim3(256), BF_LDS, st, p, nt);
#else
      return fail(-1, "BRV_OPT_BWD_PERSIST: the persistent fused backward is not in this build (tools/mkvariant.sh <tag> -DBRV_WITH_VARIANTS)");
#endif
    } else
    if (p.Kg == 256) hipLaunchKernelGGL((dwconv_bwd_fused_kernel<P, 256>), grid, dim3(256), BF_LDS, st, p);
    else if (p.Kg == 128) hipLaunchKernelGGL((dwconv_bwd_fused_kernel<P, 128>), grid, dim3(256), BF_LDS, st, p);
    else return fail(-1, "fused backward: unexpected [res | skip] width");
    HIP_OK(hipGetLastError());
    return 0;
  }
};

int launch_dz(const DzParams& p, hipStream_t st) {
  ProfScope prof("gln_prelu_bwd", 0, 6.0*p.B*p.T*(double)p.Cp, st);
  const long long per_item = (long long)p.T*(p.Cp/8);
  // every workgroup ends with one atomic on the single PReLU-slope gradient word
  // (~12 ns each, serialised): keep the grid near 1024 workgroups
  int gx = (int)((per_item + 256*8 - 1)/(256*8));
  const int cap = 1024/p.B > 1 ? 1024/p.B : 1;
  if (gx > cap) gx = cap;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(dz_kernel, dim3(gx, p.B), dim3(256), 0, st, p);
  HIP_OK(hipGetLastError());
  return 0;
}

ASpec rows_bf16(const void* ptr, int ld, long long T) {
  ASpec a; memset(&a, 0, sizeof(a));
  a.p0 = ptr; a.ld0 = ld; a.bs0 = T*ld; a.K0 = 1 << 30; a.nsrc = 1;
  return a;
}
void set_affine(ASpec& a, const double* stats, const float* g, const float* b, int C,
                long long T) {
  a.stats = stats; a.gamma = g; a.beta = b; a.C = C;
  a.inv_n = 1.0/((double)T*(double)C); a.eps = 1e-8f;
}
ASpec frames_of(const float* wav, long long L, int hop, int K, long long stride = 0) {
  ASpec a; memset(&a, 0, sizeof(a));
  a.p0 = wav; a.hop = hop; a.Kf = K; a.wav_stride = stride > 0 ? stride : L; a.wav_len = (int)L;
  a.K0 = 1 << 30; a.nsrc = 1;
  return a;
}

// Jobs travel in the kernel arguments (4 KB): a compact form holds 168 of them, so the ~150 jobs of the
// default network are ONE launch per step instead of three dependent ones (the step boundary is a
// serial section: DESIGN.md section 8 item 1e).
struct PrepJobC { int src_off, dst_off, scale_off; unsigned short R, C, rows, cols, dst_ld, tr; };
constexpr int kPrepBatch = 168;
struct PrepBatch { PrepJobC jobs[kPrepBatch]; int n; };
static_assert(sizeof(PrepBatch) <= 4096, "kernel argument size");

// plain [N][K] bf16 -> fragment order of the persistent GEMMs: slices of nsl rows, then
// (f = 32-row chunk, s = 16-column step, lane = row % 32 + 32*((col % 16)/8), 8 values)
struct PackJob { long long src_off, dst_off; int N, K, nsl; };
struct PackBatch { PackJob jobs[96]; int n; };
__global__ __launch_bounds__(256) void pack_frag_kernel(bf16_t* prepared, const PackBatch pb) {
  const PackJob j = pb.jobs[blockIdx.y];
  const long long total = (long long)j.N*j.K/8;
  const int KS = j.K/16;
  for (long long i = (long long)blockIdx.x*256 + threadIdx.x; i < total; i += (long long)gridDim.x*256) {
    const int n = (int)(i / (j.K/8)), k = (int)(i % (j.K/8))*8;
    const int slice = n / j.nsl, f = (n % j.nsl)/32, fr = n % 32, s = k/16, fh = (k % 16)/8;
    const long long dst = (long long)slice*j.nsl*j.K + ((long long)(f*KS + s)*64 + fr + 32*fh)*8;
    *reinterpret_cast<uint4*>(prepared + j.dst_off + dst) =
        *reinterpret_cast<const uint4*>(prepared + j.src_off + (long long)n*j.K + k);
  }
}
__global__ __launch_bounds__(256) void prep_weights_kernel(const float* params,
                                                           bf16_t* prepped,
                                                           const PrepBatch pb) {
  const PrepJobC& c = pb.jobs[blockIdx.y];
  PrepJob j;
  j.src_off = c.src_off; j.dst_off = c.dst_off; j.scale_off = c.scale_off;
  j.R = c.R; j.C = c.C; j.rows = c.rows; j.cols = c.cols; j.dst_ld = c.dst_ld; j.tr = c.tr;
  prep_job_run(params, prepped, j, blockIdx.x, gridDim.x);
}

// Constants of the lazily applied second norm (fused forward, gemm_ws.cuh AT == 3): with
// u = (W gamma) p the finished convolution output is rstd u + v0 - mean rstd v1.
struct LazyPrepBlk { long long res_w, res_b, skip_w, skip_b, beta, wg, out; };
struct LazyPrepParams {
  int nb, H, Hp, Bn, Sc, Bnp, Scp;
  int* stamp; int stamp_value;             // the mode stamp of `prepared` (last launch of prepare: no launch of its own)
  LazyPrepBlk blk[kWgMaxProb];
};
__global__ __launch_bounds__(256) void lazy_prep_kernel(const float* params, bf16_t* prepped,
                                                        const LazyPrepParams p) {
  // one wave per output row: lanes stride over the H inputs, butterfly reduce
  const LazyPrepBlk& b = p.blk[blockIdx.y];
  float* out = reinterpret_cast<float*>(prepped + b.out);
  const int NP = p.Bnp + p.Scp;
  const int n = blockIdx.x*4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *p.stamp = p.stamp_value;
  if (n >= NP) return;
  const bool res = n < p.Bnp;
  const int r = res ? n : n - p.Bnp;
  const bool ok = res ? (r < p.Bn && b.res_w >= 0) : r < p.Sc;
  float v0 = 0.f, v1 = 0.f;
  if (ok) {
    const float* W = params + (res ? b.res_w : b.skip_w) + (long long)r*p.H;
    const bf16_t* wg = prepped + b.wg + (long long)n*p.Hp;
    for (int k = lane; k < p.H; k += 64) {
      v0 = __builtin_fmaf(W[k], params[b.beta + k], v0);
      v1 += bf2f(wg[k]);
    }
  }
  for (int o = 32; o; o >>= 1) { v0 += __shfl_xor(v0, o); v1 += __shfl_xor(v1, o); }
  if (lane == 0) {
    out[n] = ok ? v0 + params[(res ? b.res_b : b.skip_b) + r] : 0.f;
    out[NP + n] = v1;
  }
}

// skip_sum[b][t][n] = sum_i rstd_i[b] u_i[b][t][Bnp + n] + v0_i[n] - mean_i[b] rstd_i[b] v1_i[n]:
// the skip connections of all blocks finished in one pass (fp32 out, as the backward expects)
struct SkipCombineParams {
  const bf16_t* u; long long u_stride;     // elements between blocks
  const double* stats; long long stats_stride;   // stats of block i's second norm: stats + (2 + 2i)*stats_stride
  const bf16_t* prepared; long long lazy_off[kWgMaxProb];
  float* skip; int nb, B, T, Bnp, Scp; double inv_n; float eps;
};
__global__ __launch_bounds__(256) void skip_combine_kernel(const SkipCombineParams p) {
  __shared__ float rs[kWgMaxProb];
  __shared__ float cs[128];                 // sum over the blocks of the per-item constants
  const int b = blockIdx.y, tid = threadIdx.x;
  const int NP = p.Bnp + p.Scp;
  // per-item constants: the blocks' (rstd, mean rstd) once -- one memory round trip for nb threads -- then channel n
  // sums its nb table entries with the loads of a batch of blocks in flight together. (Round 6: every channel thread
  // recomputed every block's statistics -- nb dependent double loads and square roots in front of a workgroup that
  // then streams for ~10 us: 2 048 short-lived workgroups each paid ~20 us of prologue.)
  __shared__ float ms[kWgMaxProb];
  for (int i = tid; i < p.nb; i += 256) {
    const NormStat ns = norm_stat(p.stats + (2 + 2*i)*p.stats_stride, b, p.inv_n, p.eps);
    rs[i] = ns.rstd; ms[i] = ns.mean*ns.rstd;
  }
  __syncthreads();
  for (int n = tid; n < p.Scp; n += 256) {
    float c = 0.f;
    int i = 0;
    for (; i + 8 <= p.nb; i += 8) {
      float a[8], v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float* lz = reinterpret_cast<const float*>(p.prepared + p.lazy_off[i + k]);
        a[k] = lz[p.Bnp + n]; v[k] = lz[NP + p.Bnp + n];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) c += a[k] - ms[i + k]*v[k];
    }
    for (; i < p.nb; ++i) {
      const float* lz = reinterpret_cast<const float*>(p.prepared + p.lazy_off[i]);
      c += lz[p.Bnp + n] - ms[i]*lz[NP + p.Bnp + n];
    }
    cs[n] = c;
  }
  __syncthreads();
  const int cpr = p.Scp/8;
  const long long per_item = (long long)p.T*cpr;
  for (long long e = (long long)blockIdx.x*256 + tid; e < per_item; e += (long long)gridDim.x*256) {
    const int c0 = (int)(e % cpr)*8; const long long t = e / cpr;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = cs[c0 + j];
    const bf16_t* src = p.u + ((long long)b*p.T + t)*NP + p.Bnp + c0;
    int i = 0;
    for (; i + 8 <= p.nb; i += 8) {         // eight 16-byte loads in flight per thread
      uint4 q[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) q[k] = *reinterpret_cast<const uint4*>(src + (i + k)*p.u_stride);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float f[8];
        unpack8(q[k], f);
        const float r = rs[i + k];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_fmaf(r, f[j], acc[j]);
      }
    }
    for (; i < p.nb; ++i) {
      float f[8];
      unpack8(*reinterpret_cast<const uint4*>(src + i*p.u_stride), f);
      const float r = rs[i];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = __builtin_fmaf(r, f[j], acc[j]);
    }
    float* dst = p.skip + ((long long)b*p.T + t)*p.Scp + c0;
    *reinterpret_cast<float4*>(dst) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    *reinterpret_cast<float4*>(dst + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
  }
}

}  // namespace

// ===========================================================================
// fold the replicated per-channel gradients into the flat gradient
int reduce_vector_grads(const Layout& l, const Workspace& ws, const float* vg, float* grads,
                        hipStream_t st, int blk_lo = 0, int blk_hi = 1 << 30, bool do_ln = true,
                        bool do_tcn = true) {
  const long long vper = (long long)l.H*(5 + l.P);
  {
    VgradParams vp; memset(&vp, 0, sizeof(vp));
    vp.vg = vg; vp.grads = grads; vp.rep_stride = ws.vg_stride;
    vp.N = l.N; vp.H = l.H; vp.P = l.P; vp.nb = l.nb;
    vp.ln_g_off = l.ln_g; vp.blk0_off = l.blk[0].conv_w;
    vp.blk_full = l.nb > 1 ? l.blk[1].conv_w - l.blk[0].conv_w : 0;
    const BlockOff& b0 = l.blk[0];
    const BlockOff& bl = l.blk[l.nb - 1];
    vp.o_dconv_w = b0.dconv_w - b0.conv_w; vp.o_dconv_b = b0.dconv_b - b0.conv_w;
    vp.o_n1_g_full = b0.n1_g - b0.conv_w; vp.o_n1_g_last = bl.n1_g - bl.conv_w;
    vp.tcn_prelu_off = l.tcn_prelu;
    vp.blk_lo = blk_lo; vp.blk_hi = blk_hi; vp.do_ln = do_ln; vp.do_tcn = do_tcn;
    if (b0.prelu1 != b0.n1_g + 4LL*l.H || b0.prelu2 != b0.prelu1 + 1)
      return fail(-1, "vgrad_reduce: unexpected PReLU offsets");
    const long long total = 2LL*l.N + vper*l.nb + 1 + 2*l.nb;
    int gx = (int)((total + 255)/256);
    if (gx > 1024) gx = 1024;
    ProfScope prof("vgrad_reduce", 0, 4.0*total*(kReplicas + 2), st);
    hipLaunchKernelGGL(vgrad_reduce_kernel, dim3(gx), dim3(256), 0, st, vp);
    HIP_OK(hipGetLastError());
  }
  return 0;
}

// ===========================================================================================
// Causal Conv-TasNet (cumulative layer norm, all-left depthwise padding).
// Reference: convtasnet.py:240-268 with causal=True, CausalLayerNorm (normalization.py:5-62).
// Correctness-first variant: every cLN output is materialised (cln_kernels.cuh) and the
// convolution kernels of the non-causal path run as plain convolutions on it -- their
// "apply PReLU + gLN on load" hooks get identity operands (slope 1, mean 0, rstd 1, gain 1,
// bias 0) and the statistics / norm-gradient outputs they still produce go to scratch.
// ===========================================================================================
__global__ void fill_identity_kernel(float* ones, float* zeros, float* one, int n,
                                     double* fake_stats, int B, double sumsq) {
  const int i = blockIdx.x*256 + threadIdx.x;
  if (i < n) { ones[i] = 1.f; zeros[i] = 0.f; }
  if (i == 0) *one = 1.f;
  if (i < B) { fake_stats[stat_sum(i)] = 0.0; fake_stats[stat_sq(i)] = sumsq; }
}

struct CausalCtx {
  float* ones; float* zeros; float* one; double* fake; double* scratch_stats;
  float* cfs; float* cbt; char* ctab; long long ctab_stride;
  float* tab(int i) const { return (float*)(ctab + ctab_stride*i); }
};

int causal_ctx(const Layout& l, const Workspace& ws, char* base, int B, long long T,
               hipStream_t st, CausalCtx& c) {
  const long long cmax = l.Hp > l.Np ? l.Hp : l.Np;
  c.ones = (float*)(base + ws.ident); c.zeros = c.ones + cmax; c.one = c.zeros + cmax;
  c.fake = (double*)(base + ws.fake_stats); c.scratch_stats = (double*)(base + ws.scratch_stats);
  c.cfs = (float*)(base + ws.cfs); c.cbt = (float*)(base + ws.cbt);
  c.ctab = base + ws.ctab; c.ctab_stride = ws.ctab_stride;
  // fake statistics with mean 0 and rstd 1 for n = T*H elements (var + eps = 1)
  const double sumsq = (1.0 - (double)1e-8f)*(double)T*(double)l.H;
  const int n = (int)cmax > B ? (int)cmax : B;
  hipLaunchKernelGGL(fill_identity_kernel, dim3((n + 255)/256), dim3(256), 0, st, c.ones, c.zeros,
                     c.one, (int)cmax, c.fake, B, sumsq);
  HIP_OK(hipGetLastError());
  HIP_OK(hipMemsetAsync(c.scratch_stats, 0, (size_t)B*kStatStride*8, st));
  return 0;
}

int cln_forward(const CausalCtx& cx, const bf16_t* z, const float* slope, const float* gain,
                const float* bias, bf16_t* y, float* table, int B, long long T, int Cp, int C,
                hipStream_t st) {
  ClnParams p; memset(&p, 0, sizeof(p));
  p.z = z; p.slope = slope; p.y = y; p.fsum = cx.cfs; p.table = table; p.gain = gain;
  p.bias = bias; p.B = B; p.T = (int)T; p.Cp = Cp; p.C = C; p.eps = 1e-8f;
  ProfScope prof("cln_fwd", 0, 6.0*B*T*(double)Cp, st);
  hipLaunchKernelGGL(cln_frame_sums_kernel, dim3(ceil_div((int)T, CLN_FPB)*B), dim3(256), 0, st, p);
  hipLaunchKernelGGL(cln_scan_kernel, dim3(B), dim3(256), 0, st, p);
  int gx = (int)(((long long)B*T*(Cp/8) + 255)/256);
  if (gx > 4096) gx = 4096;
  hipLaunchKernelGGL(cln_apply_kernel, dim3(gx), dim3(256), 0, st, p);
  HIP_OK(hipGetLastError());
  return 0;
}

// dz = d(loss)/dz of y = cLN(PReLU(z)); norm-parameter gradients to the replicated block
int cln_backward(const CausalCtx& cx, const bf16_t* g, const bf16_t* z, const float* slope,
                 const float* gain, const float* fwd_table, bf16_t* dz, const bf16_t* add_in,
                 int n_add, float* dgain, float* dbias, float* dslope, long long rep_stride,
                 int B, long long T, int Cp, int C, hipStream_t st) {
  ClnParams p; memset(&p, 0, sizeof(p));
  p.z = z; p.slope = slope; p.g = g; p.fsum = cx.cfs; p.table = cx.cbt; p.fwd_table = fwd_table;
  p.gain = gain; p.B = B; p.T = (int)T; p.Cp = Cp; p.C = C; p.eps = 1e-8f;
  p.dz = dz; p.add_in = add_in; p.n_add = n_add;
  p.dgain = dgain; p.dbias = dbias; p.dslope = dslope; p.rep_stride = rep_stride;
  ProfScope prof("cln_bwd", 0, 10.0*B*T*(double)Cp, st);
  hipLaunchKernelGGL(cln_bwd_sums_kernel, dim3(ceil_div((int)T, CLN_FPB)*B), dim3(256), 0, st, p);
  hipLaunchKernelGGL(cln_bwd_scan_kernel, dim3(B), dim3(256), 0, st, p);
  int gx = (int)(((long long)B*T*(Cp/8) + 255)/256);
  if (gx > 2048) gx = 2048;
  hipLaunchKernelGGL(cln_bwd_apply_kernel, dim3(gx), dim3(256), 0, st, p);
  HIP_OK(hipGetLastError());
  return 0;
}

int forward_causal(const Layout& l, const brv_ctn_config* cfg, const float* params,
                   const void* prepared, void* workspace, const float* wave, long long wave_stride,
                   float* out, int B, long long L, long long T, hipStream_t st) {
  Workspace ws; ws.init(l, B, T);
  const double BT = (double)B*(double)T;
  char* base = (char*)workspace;
  const bf16_t* prep = (const bf16_t*)prepared;
  bf16_t* w = (bf16_t*)(base + ws.w);
  bf16_t* wn = (bf16_t*)(base + ws.wn);
  auto xbuf = [&](int i) { return (bf16_t*)(base + ws.x + ws.x_stride*i); };
  auto z1buf = [&](int i) { return (bf16_t*)(base + ws.z1 + ws.z_stride*i); };
  auto z2buf = [&](int i) { return (bf16_t*)(base + ws.z2 + ws.z_stride*i); };
  auto h1buf = [&](int i) { return (bf16_t*)(base + ws.h1 + ws.z_stride*i); };
  auto h2buf = [&](int i) { return (bf16_t*)(base + ws.h2 + ws.z_stride*i); };
  float* skip = (float*)(base + ws.skip);
  bf16_t* m = (bf16_t*)(base + ws.m);
  bf16_t* y = (bf16_t*)(base + ws.y);
  CausalCtx cx; if (int r = causal_ctx(l, ws, base, B, T, st, cx)) return r;
  HIP_OK(hipMemsetAsync(out, 0, (size_t)B*l.S*L*sizeof(float), st));

  GemmRowsParams g;
  memset(&g, 0, sizeof(g));                                // encoder
  g.a = frames_of(wave, L, l.hop, l.K, wave_stride);
  g.W = prep + l.p_enc; g.T = (int)T; g.Np = l.Np; g.Kp = l.Kfp;
  g.e.out = w; g.e.ldo = l.Np; g.e.N = l.N;
  if (int r = launch_gemm_rows<A_FRAMES, E_STORE>(g, B, st, "enc_fwd", 4.0*B*L + 2.0*BT*l.Np)) return r;
  // cLN of the encoder output, bottleneck conv
  if (int r = cln_forward(cx, w, nullptr, params + l.ln_g, params + l.ln_b, wn, cx.tab(0), B, T,
                          l.Np, l.N, st)) return r;
  memset(&g, 0, sizeof(g));
  g.a = rows_bf16(wn, l.Np, T);
  g.W = prep + l.p_bott_f; g.T = (int)T; g.Np = l.Bnp; g.Kp = l.Np;
  g.e.out = xbuf(0); g.e.ldo = l.Bnp; g.e.bias = params + l.bott_b; g.e.N = l.Bn;
  if (int r = launch_gemm_rows<A_BF16, E_STORE>(g, B, st, "bottleneck_fwd", 2.0*BT*(l.Np + l.Bnp))) return r;

  for (int i = 0; i < l.nb; ++i) {
    const BlockOff& b = l.blk[i];
    const bool has_res = i < l.nb - 1;
    const int dil = 1 << (i % cfg->layers);
    memset(&g, 0, sizeof(g));                              // 1x1 conv Bn -> H
    g.a = rows_bf16(xbuf(i), l.Bnp, T);
    g.W = prep + b.p_c1_f; g.T = (int)T; g.Np = l.Hp; g.Kp = l.Bnp;
    g.e.out = z1buf(i); g.e.ldo = l.Hp; g.e.bias = params + b.conv_b; g.e.N = l.H;
    if (int r = launch_gemm_rows<A_BF16, E_STORE>(g, B, st, "pw1_fwd", 2.0*BT*(l.Bnp + l.Hp))) return r;
    if (int r = cln_forward(cx, z1buf(i), params + b.prelu1, params + b.n1_g, params + b.n1_b,
                            h1buf(i), cx.tab(1 + 2*i), B, T, l.Hp, l.H, st)) return r;
    // depthwise dilated conv on h1, all padding on the left (convtasnet.py:244-247)
    DwParams d; memset(&d, 0, sizeof(d));
    d.z1 = h1buf(i); d.z2 = z2buf(i); d.B = B; d.T = (int)T; d.Cp = l.Hp; d.C = l.H;
    d.slope1 = cx.one; d.stats1 = cx.fake; d.gamma1 = cx.ones; d.beta1 = cx.zeros;
    d.inv_n = 1.0/((double)T*l.H); d.eps = 1e-8f;
    d.taps = params + b.dconv_w; d.bias = params + b.dconv_b;
    d.dil = dil; d.left = (l.P - 1)*dil;
    d.stats2 = cx.scratch_stats; d.slope2 = cx.one;
    if (int r = dispatch_p<DwFwd>(l.P, d, st)) return r;
    if (int r = cln_forward(cx, z2buf(i), params + b.prelu2, params + b.n2_g, params + b.n2_b,
                            h2buf(i), cx.tab(2 + 2*i), B, T, l.Hp, l.H, st)) return r;
    memset(&g, 0, sizeof(g));                              // [res | skip] 1x1 convs
    g.a = rows_bf16(h2buf(i), l.Hp, T);
    const int rs0 = has_res ? l.Bnp : 0;
    g.W = prep + b.p_rs_f; g.T = (int)T; g.Np = rs0 + l.Scp; g.Kp = l.Hp;
    g.e.out = has_res ? xbuf(i + 1) : nullptr; g.e.ldo = l.Bnp;
    g.e.bias = has_res ? params + b.res_b : nullptr; g.e.N = has_res ? l.Bn : 0;
    g.e.Nsplit = rs0; g.e.bias2 = params + b.skip_b; g.e.N2 = l.Sc;
    g.e.res_in = xbuf(i); g.e.ld_res = l.Bnp;
    g.e.skip = skip; g.e.ld_skip = l.Scp; g.e.skip_init = (i == 0);
    if (int r = launch_gemm_rows<A_BF16, E_RES_SKIP>(g, B, st, "pw2_fwd", 2.0*BT*(l.Hp + l.Bnp + rs0) + 4.0*BT*l.Scp*(i == 0 ? 1 : 2))) return r;
  }
  memset(&g, 0, sizeof(g));                                // prelu -> output conv -> mask
  g.a = rows_bf16(skip, l.Scp, T);
  g.a.slope = params + l.tcn_prelu;
  g.W = prep + l.p_out_f; g.T = (int)T; g.Np = l.S*l.Np; g.Kp = l.Scp;
  g.e.out = y; g.e.ldo = l.Np; g.e.bias = params + l.out_b; g.e.N = l.N;
  g.e.w_in = w; g.e.ld_w = l.Np; g.e.m_out = m; g.e.S = l.S; g.e.Np_src = l.Np;
  if (int r = launch_gemm_rows<A_F32, E_MASK>(g, B, st, "mask_fwd", 4.0*BT*l.Scp + 2.0*BT*l.Np*(1 + 2*l.S))) return r;
  memset(&g, 0, sizeof(g));                                // decoder
  g.a = rows_bf16(y, l.Np, T);
  g.W = prep + l.p_dec_f; g.T = (int)T; g.Np = l.Kfp; g.Kp = l.Np;
  g.e.wave_out = out; g.e.hop = l.hop; g.e.Kf = l.K; g.e.wave_stride = L;
  g.e.wave_len = (int)L;
  if (int r = launch_gemm_rows<A_BF16, E_OLA>(g, B*l.S, st, "dec_fwd", 2.0*BT*l.S*l.Np + 4.0*B*l.S*L)) return r;
  return 0;
}

int backward_causal(const Layout& l, const brv_ctn_config* cfg, const float* params,
                    const void* prepared, void* workspace, const float* wave, long long wave_stride,
                    const float* d_out, float* grads, int B, long long L, long long T, hipStream_t st) {
  Workspace ws; ws.init(l, B, T);
  const double BT = (double)B*(double)T;
  char* base = (char*)workspace;
  const bf16_t* prep = (const bf16_t*)prepared;
  bf16_t* w = (bf16_t*)(base + ws.w);
  bf16_t* wn = (bf16_t*)(base + ws.wn);
  auto xbuf = [&](int i) { return (bf16_t*)(base + ws.x + ws.x_stride*i); };
  auto z1buf = [&](int i) { return (bf16_t*)(base + ws.z1 + ws.z_stride*i); };
  auto z2buf = [&](int i) { return (bf16_t*)(base + ws.z2 + ws.z_stride*i); };
  auto h1buf = [&](int i) { return (bf16_t*)(base + ws.h1 + ws.z_stride*i); };
  auto h2buf = [&](int i) { return (bf16_t*)(base + ws.h2 + ws.z_stride*i); };
  float* skip = (float*)(base + ws.skip);
  bf16_t* m = (bf16_t*)(base + ws.m);
  bf16_t* y = (bf16_t*)(base + ws.y);
  bf16_t* dpre = (bf16_t*)(base + ws.dpre);
  bf16_t* dw1 = (bf16_t*)(base + ws.dw1);
  bf16_t* gskip = (bf16_t*)(base + ws.gskip);
  bf16_t* gout = (bf16_t*)(base + ws.gout);
  const int ldg = l.Bnp + l.Scp;
  bf16_t* eA = (bf16_t*)(base + ws.eA);
  bf16_t* eB = (bf16_t*)(base + ws.eB);
  bf16_t* e0 = (bf16_t*)(base + ws.e0);
  bf16_t* dwt = (bf16_t*)(base + ws.dwt);
  const int BS = B*l.S;
  float* vg = (float*)(base + ws.vg);
  const long long vper = (long long)l.H*(5 + l.P);
  auto vslot = [&](int i) { return vg + 2LL*l.N + vper*i; };
  float* vslope = vg + 2LL*l.N + vper*l.nb;
  float* vscratch = vslope + 1 + 2*l.nb;                   // 2*Hp floats per replica, discarded
  CausalCtx cx; if (int r = causal_ctx(l, ws, base, B, T, st, cx)) return r;
  HIP_OK(hipMemsetAsync(vg, 0, ws.vg_bytes, st));

  GemmRowsParams g; WgradParams wg;
  memset(&g, 0, sizeof(g));                                // decoder data gradient + mask backward
  g.a = frames_of(d_out, L, l.hop, l.K);
  g.W = prep + l.p_dec_b; g.T = (int)T; g.Np = l.Np; g.Kp = l.Kfp;
  g.e.out = dpre; g.e.ldo = l.Np; g.e.out2 = dw1; g.e.w_in = w; g.e.ld_w = l.Np;
  g.e.m_in = m; g.e.S = l.S;
  if (int r = launch_gemm_rows<A_FRAMES, E_MASK_BWD>(g, BS, st, "dec_bwd", 4.0*BS*L + 2.0*BT*l.Np*(1 + 3*l.S))) return r;
  memset(&wg, 0, sizeof(wg));                              // decoder weight gradient
  wg.g = rows_bf16(y, l.Np, T); wg.h = frames_of(d_out, L, l.hop, l.K);
  wg.B = BS; wg.T = (int)T; wg.Gp = l.Np; wg.Hp = l.Kfp;
  wg.out0 = grads + l.dec_w; wg.G0p = l.Np; wg.N0 = l.N; wg.Kout = l.K; wg.ldo = l.K;
  if (int r = launch_wgrad<A_FRAMES>(wg, st, "wgrad_dec", 2.0*BT*l.S*l.Np + 4.0*BS*L)) return r;
  memset(&g, 0, sizeof(g));                                // output conv dgrad + PReLU backward
  g.a = rows_bf16(dpre, l.Np, T); g.a.nsrc = l.S;
  g.W = prep + l.p_out_b; g.T = (int)T; g.Np = l.Scp; g.Kp = l.S*l.Np;
  g.e.out = gskip; g.e.ldo = ldg; g.e.src_f32 = skip; g.e.ld_srcf = l.Scp;
  g.e.src_slope = params + l.tcn_prelu; g.e.dslope = vslope;
  g.e.rep_stride = ws.vg_stride; g.e.n_rep = kReplicas;
  if (int r = launch_gemm_rows<A_BF16, E_PRELU_BWD>(g, B, st, "mask_bwd", 2.0*BT*l.S*l.Np + 6.0*BT*l.Scp)) return r;
  for (int s = 0; s < l.S; ++s) {                          // output conv weight / bias gradients
    memset(&wg, 0, sizeof(wg));
    wg.g = rows_bf16(dpre + (long long)s*T*l.Np, l.Np, T); wg.g.bs0 = (long long)l.S*T*l.Np;
    wg.h = rows_bf16(skip, l.Scp, T); wg.h.slope = params + l.tcn_prelu;
    wg.B = B; wg.T = (int)T; wg.Gp = l.Np; wg.Hp = l.Scp;
    wg.out0 = grads + l.out_w + (long long)s*l.N*l.Sc; wg.G0p = l.Np; wg.N0 = l.N;
    wg.Kout = l.Sc; wg.ldo = l.Sc; wg.gbias0 = grads + l.out_b + (long long)s*l.N;
    if (int r = launch_wgrad<A_F32>(wg, st, "wgrad_out", 2.0*BT*l.Np + 4.0*BT*l.Scp)) return r;
  }

  for (int i = l.nb - 1; i >= 0; --i) {
    const BlockOff& b = l.blk[i];
    const bool has_res = i < l.nb - 1;
    const int dil = 1 << (i % cfg->layers);
    const int rs0 = has_res ? l.Bnp : 0;
    // [res | skip] weight / bias gradients: G = [g_out | g_skip], H = h2 (materialised)
    memset(&wg, 0, sizeof(wg));
    wg.g = rows_bf16(has_res ? gout : gskip, ldg, T);
    wg.h = rows_bf16(h2buf(i), l.Hp, T);
    wg.B = B; wg.T = (int)T; wg.Gp = rs0 + l.Scp; wg.Hp = l.Hp;
    wg.out0 = has_res ? grads + b.res_w : nullptr; wg.out1 = grads + b.skip_w;
    wg.G0p = rs0; wg.N0 = has_res ? l.Bn : 0; wg.N1 = l.Sc; wg.Kout = l.H; wg.ldo = l.H;
    wg.gbias0 = has_res ? grads + b.res_b : nullptr; wg.gbias1 = grads + b.skip_b;
    if (int r = launch_wgrad<A_BF16>(wg, st, "pw2_wgrad", 2.0*BT*(rs0 + l.Scp + l.Hp))) return r;
    // [res | skip] data gradient -> gradient wrt h2
    memset(&g, 0, sizeof(g));
    g.a = rows_bf16(has_res ? gout : gskip, ldg, T);
    g.W = prep + b.p_rs_b; g.T = (int)T; g.Np = l.Hp; g.Kp = rs0 + l.Scp;
    g.e.out = eA; g.e.ldo = l.Hp; g.e.N = l.H;
    if (int r = launch_gemm_rows<A_BF16, E_STORE>(g, B, st, "pw2_dgrad", 2.0*BT*(rs0 + l.Scp + l.Hp))) return r;
    // cLN_2 + PReLU_2 backward -> dz2 (in place over eA is not possible: g is read twice)
    if (int r = cln_backward(cx, eA, z2buf(i), params + b.prelu2, params + b.n2_g,
                             cx.tab(2 + 2*i), eB, nullptr, 0, vslot(i) + 2*l.H, vslot(i) + 3*l.H,
                             vslope + 2 + 2*i, ws.vg_stride, B, T, l.Hp, l.H, st)) return r;
    // depthwise conv backward on h1 (identity operands: plain transposed convolution)
    DwParams d; memset(&d, 0, sizeof(d));
    d.z1 = h1buf(i); d.dz2 = eB; d.e1 = eA; d.B = B; d.T = (int)T; d.Cp = l.Hp; d.C = l.H;
    d.slope1 = cx.one; d.stats1 = cx.fake; d.gamma1 = cx.ones; d.beta1 = cx.zeros;
    d.inv_n = 1.0/((double)T*l.H); d.eps = 1e-8f;
    d.taps = params + b.dconv_w; d.dil = dil; d.left = (l.P - 1)*dil;
    d.dgamma1 = vscratch; d.dbeta1 = vscratch + l.Hp;
    d.dtaps = vslot(i) + 4*l.H; d.dbias = vslot(i) + 4*l.H + (long long)l.H*l.P;
    d.rep_stride = ws.vg_stride; d.sums1 = cx.scratch_stats;
    if (int r = dispatch_p<DwBwd>(l.P, d, st)) return r;
    // cLN_1 + PReLU_1 backward -> dz1
    if (int r = cln_backward(cx, eA, z1buf(i), params + b.prelu1, params + b.n1_g,
                             cx.tab(1 + 2*i), eB, nullptr, 0, vslot(i), vslot(i) + l.H,
                             vslope + 1 + 2*i, ws.vg_stride, B, T, l.Hp, l.H, st)) return r;
    // first 1x1 conv: weight / bias gradients, then data gradient + residual path
    memset(&wg, 0, sizeof(wg));
    wg.g = rows_bf16(eB, l.Hp, T); wg.h = rows_bf16(xbuf(i), l.Bnp, T);
    wg.B = B; wg.T = (int)T; wg.Gp = l.Hp; wg.Hp = l.Bnp;
    wg.out0 = grads + b.conv_w; wg.G0p = l.Hp; wg.N0 = l.H; wg.Kout = l.Bn; wg.ldo = l.Bn;
    wg.gbias0 = grads + b.conv_b;
    if (int r = launch_wgrad<A_BF16>(wg, st, "pw1_wgrad", 2.0*BT*(l.Hp + l.Bnp))) return r;
    memset(&g, 0, sizeof(g));
    g.a = rows_bf16(eB, l.Hp, T);
    g.W = prep + b.p_c1_b; g.T = (int)T; g.Np = l.Bnp; g.Kp = l.Hp;
    g.e.out = gout; g.e.ldo = ldg; g.e.add_in = has_res ? gout : nullptr; g.e.ld_add = ldg;
    if (int r = launch_gemm_rows<A_BF16, E_ADD>(g, B, st, "pw1_dgrad", 2.0*BT*(l.Hp + l.Bnp*(has_res ? 2 : 1)))) return r;
  }
  // bottleneck conv: weight gradient (H = cLN(w), materialised), data gradient, cLN backward
  memset(&wg, 0, sizeof(wg));
  wg.g = rows_bf16(gout, ldg, T); wg.h = rows_bf16(wn, l.Np, T);
  wg.B = B; wg.T = (int)T; wg.Gp = l.Bnp; wg.Hp = l.Np;
  wg.out0 = grads + l.bott_w; wg.G0p = l.Bnp; wg.N0 = l.Bn; wg.Kout = l.N; wg.ldo = l.N;
  wg.gbias0 = grads + l.bott_b;
  if (int r = launch_wgrad<A_BF16>(wg, st, "bottleneck_wgrad", 2.0*BT*(l.Bnp + l.Np))) return r;
  memset(&g, 0, sizeof(g));
  g.a = rows_bf16(gout, ldg, T);
  g.W = prep + l.p_bott_b; g.T = (int)T; g.Np = l.Np; g.Kp = l.Bnp;
  g.e.out = e0; g.e.ldo = l.Np; g.e.N = l.N;
  if (int r = launch_gemm_rows<A_BF16, E_STORE>(g, B, st, "bottleneck_dgrad", 2.0*BT*(l.Bnp + l.Np))) return r;
  // total gradient wrt the encoder output: cLN backward + the S mask-path terms
  if (int r = cln_backward(cx, e0, w, nullptr, params + l.ln_g, cx.tab(0), dwt, dw1, l.S, vg,
                           vg + l.N, nullptr, ws.vg_stride, B, T, l.Np, l.N, st)) return r;
  memset(&wg, 0, sizeof(wg));                              // encoder weight gradient
  wg.g = rows_bf16(dwt, l.Np, T); wg.h = frames_of(wave, L, l.hop, l.K, wave_stride);
  wg.B = B; wg.T = (int)T; wg.Gp = l.Np; wg.Hp = l.Kfp;
  wg.out0 = grads + l.enc_w; wg.G0p = l.Np; wg.N0 = l.N; wg.Kout = l.K; wg.ldo = l.K;
  if (int r = launch_wgrad<A_FRAMES>(wg, st, "wgrad_enc", 2.0*BT*l.Np + 4.0*B*L)) return r;
  return reduce_vector_grads(l, ws, vg, grads, st);
}

extern "C" {

int brv_version(void) { return 100; }

#ifdef BRV_DIAG
// Copies the cycle stamps of a BRV_DBG=64 run (tools/ablate.py) to the host.
int brv_debug_read(long long* out, int64_t n) {
  (void)hipDeviceSynchronize();
  const int r = (int)hipMemcpy(out, debug_buffer(), (size_t)n*8, hipMemcpyDeviceToHost);
  (void)hipMemset(debug_buffer(), 0, 1 << 20);           // next run starts clean
  return r;
}
#endif

void* brv_prof_create(int by_dilation) {
  Prof* p = new Prof();
  p->by_dil = by_dilation != 0;
  return p;
}
void brv_prof_destroy(void* prof) {
  Prof* p = static_cast<Prof*>(prof);
  if (!p) return;
  for (auto& e : p->entries) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  delete p;
}

// Writes one line per label: "label calls total_ms flops bytes\n" and clears the records. Returns the
// number of bytes needed (a call with a too small or null buffer keeps the records).
int64_t brv_prof_collect(void* prof, char* buf, int64_t buflen) {
  Prof* p = static_cast<Prof*>(prof);
  if (!p) return 0;
  struct Agg { std::string label; long long calls; double ms, flops, bytes; };
  std::vector<Agg> agg;
  for (auto& e : p->entries) {
    (void)hipEventSynchronize(e.b);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e.a, e.b);
    Agg* a = nullptr;
    for (auto& x : agg) if (x.label == e.label) { a = &x; break; }
    if (!a) { agg.push_back({e.label, 0, 0, 0, 0}); a = &agg.back(); }
    a->calls += 1; a->ms += ms; a->flops += e.flops; a->bytes += e.bytes;
  }
  std::string out;
  char line[256];
  for (auto& a : agg) {
    snprintf(line, sizeof(line), "%s %lld %.6f %.6e %.6e\n", a.label.c_str(), a.calls, a.ms,
             a.flops, a.bytes);
    out += line;
  }
  if ((int64_t)out.size() + 1 <= buflen && buf) {
    memcpy(buf, out.c_str(), out.size() + 1);
    for (auto& e : p->entries) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    p->entries.clear();
  }
  return (int64_t)out.size() + 1;
}
const char* brv_last_error(void) { return g_err.c_str(); }
// other translation units of the library report through the same thread-local message (not exported)
__attribute__((visibility("hidden"))) void brv_internal_set_error(const char* msg) { g_err = msg ? msg : ""; }

int64_t brv_ctn_param_count(const brv_ctn_config* cfg) {
  Layout l; if (l.init(cfg)) return -1; return l.n_params;
}
int64_t brv_ctn_param_tensors(const brv_ctn_config* cfg) {
  Layout l; if (l.init(cfg)) return -1; return (int64_t)l.tensor_offsets.size();
}
int64_t brv_ctn_param_offset(const brv_ctn_config* cfg, int64_t index) {
  Layout l; if (l.init(cfg)) return -1;
  if (index < 0 || index >= (int64_t)l.tensor_offsets.size()) return -1;
  return l.tensor_offsets[index];
}
int64_t brv_ctn_frames(const brv_ctn_config* cfg, int64_t length) {
  Layout l; if (l.init(cfg)) return -1; return l.frames(length);
}
int64_t brv_ctn_prepared_bytes(const brv_ctn_config* cfg) {
  Layout l; if (l.init(cfg)) return -1; return l.n_prepared*2;
}
int64_t brv_ctn_workspace_bytes(const brv_ctn_config* cfg, int64_t batch, int64_t length) {
  Layout l; if (l.init(cfg)) return -1;
  Workspace ws; ws.init(l, batch, l.frames(length));
  return ws.total;
}
int64_t brv_ctn_workspace_offset(const brv_ctn_config* cfg, int64_t batch, int64_t length,
                                 const char* name, int64_t index) {
  Layout l; if (l.init(cfg)) return -1;
  Workspace ws; ws.init(l, batch, l.frames(length));
  const std::string n(name);
  if (n == "w") return ws.w;
  if (n == "x") return ws.x + ws.x_stride*index;
  if (n == "z1") return ws.z1 + ws.z_stride*index;
  if (n == "z2") return ws.z2 + ws.z_stride*index;
  if (n == "skip") return ws.skip;
  if (n == "m") return ws.m;
  if (n == "y") return ws.y;
  if (n == "stats") return ws.stats + index*batch*kStatStride*8;
  if (n == "sums") return ws.sums + index*batch*kStatStride*8;
  if (n == "dpre") return ws.dpre;
  if (n == "gskip") return ws.gskip;
  if (n == "gout") return ws.gout;
  if (n == "dwt") return ws.dwt;
  if (n == "h1" && l.causal) return ws.h1 + ws.z_stride*index;
  if (n == "h2" && l.causal) return ws.h2 + ws.z_stride*index;
  fail(-1, "unknown workspace tensor " + n);
  return -1;
}

int brv_ctn_prepare(const brv_ctn_config* cfg, const float* params, void* prepared,
                    const brv_launch_opts* opts, brv_stream_t stream) {
  OptsScope scope(opts);
  Layout l; if (int r = l.init(cfg)) return r;
  hipStream_t st = (hipStream_t)stream;
  std::vector<PrepJob> jobs;
  const bool fused = l.fused_fwd();
  auto add = [&](long long src, long long dst, int R, int C, int rows, int cols, int ld,
                 int tr) {
    PrepJob j; j.src_off = src; j.dst_off = dst; j.R = R; j.C = C; j.rows = rows;
    j.cols = cols; j.dst_ld = ld; j.tr = tr; j.scale_off = -1; jobs.push_back(j);
  };
  add(l.enc_w, l.p_enc, l.N, l.K, l.Np, l.Kfp, l.Kfp, 0);
  add(l.dec_w, l.p_dec_f, l.N, l.K, l.Kfp, l.Np, l.Np, 1);
  add(l.dec_w, l.p_dec_b, l.N, l.K, l.Np, l.Kfp, l.Kfp, 0);
  add(l.bott_w, l.p_bott_f, l.Bn, l.N, l.Bnp, l.Np, l.Np, 0);
  add(l.bott_w, l.p_bott_b, l.Bn, l.N, l.Np, l.Bnp, l.Bnp, 1);
  for (int i = 0; i < l.nb; ++i) {
    const BlockOff& b = l.blk[i];
    const bool has_res = i < l.nb - 1;
    const int rs0 = has_res ? l.Bnp : 0, rs = rs0 + l.Scp;
    add(b.conv_w, b.p_c1_f, l.H, l.Bn, l.Hp, l.Bnp, l.Bnp, 0);
    add(b.conv_w, b.p_c1_b, l.H, l.Bn, l.Bnp, l.Hp, l.Hp, 1);
    if (has_res) add(b.res_w, b.p_rs_b, l.Bn, l.H, l.Hp, l.Bnp, rs, 1);
    add(b.skip_w, b.p_rs_b + rs0, l.Sc, l.H, l.Hp, l.Scp, rs, 1);
    if (fused) {
      // gamma_2-folded [res | skip] weights of the fused forward (residual rows zero in the last block)
      add(has_res ? b.res_w : 0, b.p_rs_g, has_res ? l.Bn : 0, l.H, l.Bnp, l.Hp, l.Hp, 0);
      jobs.back().scale_off = b.n2_g;
      add(b.skip_w, b.p_rs_g + (long long)l.Bnp*l.Hp, l.Sc, l.H, l.Scp, l.Hp, l.Hp, 0);
      jobs.back().scale_off = b.n2_g;
    } else {
      if (has_res) add(b.res_w, b.p_rs_f, l.Bn, l.H, l.Bnp, l.Hp, l.Hp, 0);
      add(b.skip_w, b.p_rs_f + (long long)rs0*l.Hp, l.Sc, l.H, l.Scp, l.Hp, l.Hp, 0);
    }
  }
  for (int s = 0; s < l.S; ++s) {
    add(l.out_w + (long long)s*l.N*l.Sc, l.p_out_f + (long long)s*l.Np*l.Scp,
        l.N, l.Sc, l.Np, l.Scp, l.Scp, 0);
    add(l.out_w + (long long)s*l.N*l.Sc, l.p_out_b + (long long)s*l.Np,
        l.N, l.Sc, l.Scp, l.Np, l.S*l.Np, 1);
  }
  for (size_t i = 0; i < jobs.size(); i += kPrepBatch) {
    PrepBatch pb;
    pb.n = (int)std::min<size_t>(kPrepBatch, jobs.size() - i);
    for (int k = 0; k < pb.n; ++k) {
      const PrepJob& j = jobs[i + k];
      if (j.src_off > 0x7fffffffLL || j.dst_off > 0x7fffffffLL || j.scale_off > 0x7fffffffLL ||
          (j.R | j.C | j.rows | j.cols | j.dst_ld) > 0xffff)
        return fail(-1, "prepare: layout too large for the compact job form");
      PrepJobC& c = pb.jobs[k];
      c.src_off = (int)j.src_off; c.dst_off = (int)j.dst_off; c.scale_off = (int)j.scale_off;
      c.R = (unsigned short)j.R; c.C = (unsigned short)j.C; c.rows = (unsigned short)j.rows;
      c.cols = (unsigned short)j.cols; c.dst_ld = (unsigned short)j.dst_ld; c.tr = (unsigned short)j.tr;
    }
    hipLaunchKernelGGL(prep_weights_kernel, dim3(8, pb.n), dim3(256), 0, st, params,
                       (bf16_t*)prepared, pb);
    HIP_OK(hipGetLastError());
  }
  // fragment-order copies for the persistent GEMMs (default widths only)
  std::vector<PackJob> packs;
  auto pack = [&](long long src, long long dst, int N, int K, int nsl) {
    if (N % nsl || K % 16) return;
    PackJob j; j.src_off = src; j.dst_off = dst; j.N = N; j.K = K; j.nsl = nsl; packs.push_back(j);
  };
  for (int i = 0; i < l.nb; ++i) {
    const BlockOff& b = l.blk[i];
    const int rs = (i < l.nb - 1 ? l.Bnp : 0) + l.Scp;
    pack(b.p_c1_f, b.p_c1_fp, l.Hp, l.Bnp, 64);
    pack(b.p_rs_b, b.p_rs_bp, l.Hp, rs, 32);
    if (fused) pack(b.p_rs_g, b.p_rs_gp, l.Bnp + l.Scp, l.Hp, 32);
    else pack(b.p_rs_f, b.p_rs_fp, rs, l.Hp, 32);
  }
  for (size_t i = 0; i < packs.size(); i += 96) {
    PackBatch pb;
    pb.n = (int)std::min<size_t>(96, packs.size() - i);
    for (int k = 0; k < pb.n; ++k) pb.jobs[k] = packs[i + k];
    hipLaunchKernelGGL(pack_frag_kernel, dim3(16, pb.n), dim3(256), 0, st, (bf16_t*)prepared, pb);
    HIP_OK(hipGetLastError());
  }
  if (fused) {
    LazyPrepParams lp; memset(&lp, 0, sizeof(lp));
    lp.nb = l.nb; lp.H = l.H; lp.Hp = l.Hp; lp.Bn = l.Bn; lp.Sc = l.Sc; lp.Bnp = l.Bnp; lp.Scp = l.Scp;
    for (int i = 0; i < l.nb; ++i) {
      const BlockOff& b = l.blk[i];
      lp.blk[i].res_w = b.res_w; lp.blk[i].res_b = b.res_b; lp.blk[i].skip_w = b.skip_w;
      lp.blk[i].skip_b = b.skip_b; lp.blk[i].beta = b.n2_b; lp.blk[i].wg = b.p_rs_g; lp.blk[i].out = b.p_lazy;
    }
    lp.stamp = reinterpret_cast<int*>((bf16_t*)prepared + l.p_stamp); lp.stamp_value = mode_stamp(fused);
    hipLaunchKernelGGL(lazy_prep_kernel, dim3((l.Bnp + l.Scp + 3)/4, l.nb), dim3(256), 0, st, params,
                       (bf16_t*)prepared, lp);
    HIP_OK(hipGetLastError());
  } else {
    hipLaunchKernelGGL(stamp_write_kernel, dim3(1), dim3(1), 0, st,
                       reinterpret_cast<int*>((bf16_t*)prepared + l.p_stamp), mode_stamp(fused));
    HIP_OK(hipGetLastError());
  }
  return 0;
}

int brv_ctn_forward(const brv_ctn_config* cfg, const float* params, const void* prepared,
                    void* workspace, const float* wave, int64_t wave_stride, float* out, int64_t batch,
                    int64_t length, const brv_launch_opts* opts, brv_stream_t stream) {
  OptsScope scope(opts);
  if (wave_stride != 0 && wave_stride < length) return fail(-1, "wave_stride shorter than the rows");
  Layout l; if (int r = l.init(cfg)) return r;
  hipStream_t st = (hipStream_t)stream;
  const int B = (int)batch; const long long L = length;
  const long long T = l.frames(L);
  if (B < 1 || T < 1) return fail(-1, "empty batch or input shorter than one frame");
  if (l.causal) return forward_causal(l, cfg, params, prepared, workspace, wave, wave_stride, out, B, L, T, st);
  Workspace ws; ws.init(l, B, T);
  const double BT = (double)B*(double)T;
  char* base = (char*)workspace;
  const bf16_t* prep = (const bf16_t*)prepared;
  double* stats = (double*)(base + ws.stats);
  auto stat = [&](int i) { return stats + (long long)i*B*kStatStride; };
  bf16_t* w = (bf16_t*)(base + ws.w);
  auto xbuf = [&](int i) { return (bf16_t*)(base + ws.x + ws.x_stride*i); };
  auto z1buf = [&](int i) { return (bf16_t*)(base + ws.z1 + ws.z_stride*i); };
  auto z2buf = [&](int i) { return (bf16_t*)(base + ws.z2 + ws.z_stride*i); };
  float* skip = (float*)(base + ws.skip);
  bf16_t* m = (bf16_t*)(base + ws.m);
  bf16_t* y = (bf16_t*)(base + ws.y);

  // statistics and output <- 0; operands prepared for the other mode -> NaN statistics (and with them a NaN
  // output); then this call's mode stamped on the workspace
  if (int r = launch_call_init(stats, (long long)(ws.stats_bytes/8), out, (long long)B*l.S*L*(long long)sizeof(float),
                               reinterpret_cast<const int*>(prep + l.p_stamp), mode_stamp(l.fused_fwd()),
                               reinterpret_cast<int*>(base + ws.stamp), mode_stamp(l.fused_fwd()), st))
    return fail(r, "forward: call_init launch");

  GemmRowsParams g;
  // encoder: framed filterbank analysis, statistics for the first gLN
  memset(&g, 0, sizeof(g));
  g.a = frames_of(wave, L, l.hop, l.K, wave_stride);
  g.W = prep + l.p_enc; g.T = (int)T; g.Np = l.Np; g.Kp = l.Kfp;
  g.e.out = w; g.e.ldo = l.Np; g.e.N = l.N; g.e.stats_out = stat(0);
  if (int r = launch_gemm_rows<A_FRAMES, E_STORE>(g, B, st, "enc_fwd", 4.0*B*L + 2.0*BT*l.Np)) return r;
  // bottleneck 1x1 conv on gLN(w)
  memset(&g, 0, sizeof(g));
  g.a = rows_bf16(w, l.Np, T);
  set_affine(g.a, stat(0), params + l.ln_g, params + l.ln_b, l.N, T);
  g.W = prep + l.p_bott_f; g.T = (int)T; g.Np = l.Bnp; g.Kp = l.Np;
  g.e.out = xbuf(0); g.e.ldo = l.Bnp; g.e.bias = params + l.bott_b; g.e.N = l.Bn;
  if (int r = launch_gemm_rows<A_BF16, E_STORE>(g, B, st, "bottleneck_fwd", 2.0*BT*(l.Np + l.Bnp))) return r;

  // Fused forward (default widths, non-causal, kernel_size 3; BRV_FWD_FUSE=0 selects the three-launch
  // sequence below): two launches per block instead of three. pw1_fwd
  // finishes the block input lazily (x_i = x_{i-1} + rstd u_{i-1} + c) while staging it; dwpw2_fwd
  // (dwpw2_fused.cuh) runs the depthwise stage in front of the [res | skip] product and leaves the
  // second norm to the consumers. z2 never comes back from HBM in the forward pass and the fp32
  // skip accumulation (read + write of 4 B x 128 channels per frame and block) becomes one bf16
  // write per block and one pass at the end.
  const bool fused_fwd = l.fused_fwd();
  auto ubuf = [&](int i) { return (bf16_t*)(base + ws.u + ws.u_stride*i); };
  const int NPu = l.Bnp + l.Scp;
  if (fused_fwd) {
    for (int i = 0; i < l.nb; ++i) {
      const BlockOff& b = l.blk[i];
      const int dil = 1 << (i % cfg->layers);
      memset(&g, 0, sizeof(g));
      g.W = prep + b.p_c1_f; g.T = (int)T; g.Np = l.Hp; g.Kp = l.Bnp;
      g.Wp = prep + b.p_c1_fp; g.wp_nsl = 64;
      g.e.out = z1buf(i); g.e.ldo = l.Hp; g.e.bias = params + b.conv_b; g.e.N = l.H;
      g.e.stats_out = stat(1 + 2*i); g.e.stats_slope = params + b.prelu1;
      if (i == 0) {
        g.a = rows_bf16(xbuf(0), l.Bnp, T);
        ProfScope prof("pw1_fwd", 2.0*BT*l.Hp*l.Bnp, 2.0*BT*(l.Bnp + l.Hp), st);
        if (int r = launch_gemm_ws<128, 64, 1, E_STORE, 0, false, 8>(g, B, st)) return r;
      } else {
        const BlockOff& pb = l.blk[i - 1];
        g.a = rows_bf16(xbuf(i - 1), l.Bnp, T);
        g.a.p1 = ubuf(i - 1); g.a.ld1 = NPu; g.a.bs1 = T*NPu;
        g.a.xout = xbuf(i);
        g.a.stats = stat(2 + 2*(i - 1)); g.a.inv_n = 1.0/((double)T*l.H); g.a.eps = 1e-8f;
        const float* lz = reinterpret_cast<const float*>(prep + pb.p_lazy);
        g.a.lazy_v0 = lz; g.a.lazy_v1 = lz + NPu;
        ProfScope prof("pw1_fwd", 2.0*BT*l.Hp*l.Bnp, 2.0*BT*(3*l.Bnp + l.Hp), st);
#ifndef PW1F_NSL
#define PW1F_NSL 64
#define PW1F_WM 1
#endif
        if (int r = launch_gemm_ws<128, PW1F_NSL, PW1F_WM, E_STORE, 2, false, 8>(g, B, st)) return r;
      }
      memset(&g, 0, sizeof(g));
      g.a = rows_bf16(z1buf(i), l.Hp, T);
      g.a.slope = params + b.prelu1; g.a.stats = stat(1 + 2*i);
      g.a.gamma = params + b.n1_g; g.a.beta = params + b.n1_b; g.a.C = l.H;
      g.a.inv_n = 1.0/((double)T*l.H); g.a.eps = 1e-8f;
      g.a.z2out = z2buf(i); g.a.taps = params + b.dconv_w; g.a.dbias = params + b.dconv_b;
      g.a.slope2 = params + b.prelu2; g.a.stats2_out = stat(2 + 2*i);
      g.a.dil = dil; g.a.left = ((l.P - 1)*dil)/2;
      g.W = prep + b.p_rs_g; g.Wp = prep + b.p_rs_gp; g.wp_nsl = 32;
      g.T = (int)T; g.Np = NPu; g.Kp = l.Hp;
      g.e.out = ubuf(i); g.e.ldo = NPu; g.e.N = NPu;
      {
        ProfScope prof("dwpw2_fwd", 2.0*BT*l.Hp*(NPu + l.P), 2.0*BT*(2*l.Hp + NPu), st);
        const bool old_stage = opt(BRV_OPT_DWPW2_WS);         // first version of the stage (gemm_ws.cuh AT == 3)
        if (old_stage || l.Hp != DP_H || NPu != DP_N) {
          if (int r = launch_gemm_ws<512, 32, 1, E_STORE, 3, false, 8>(g, B, st)) return r;
        } else {
          DwPw2Params dp; memset(&dp, 0, sizeof(dp));
          dp.z1 = z1buf(i); dp.z2 = z2buf(i); dp.u = ubuf(i);
          dp.Wp = prep + b.p_rs_gp;
          dp.slope1 = params + b.prelu1; dp.slope2 = params + b.prelu2;
          dp.stats1 = stat(1 + 2*i); dp.stats2 = stat(2 + 2*i);
          dp.gamma1 = params + b.n1_g; dp.beta1 = params + b.n1_b;
          dp.taps = params + b.dconv_w; dp.dbias = params + b.dconv_b;
          dp.B = B; dp.T = (int)T; dp.dil = dil; dp.left = ((l.P - 1)*dil)/2; dp.C = l.H;
          dp.inv_n = 1.0/((double)T*l.H); dp.eps = 1e-8f;
          const bool v2 = opt(BRV_OPT_DWPW2_V2);
          const int n_tiles = B*(int)((T + (v2 ? D2_TT : DP_TT) - 1)/(v2 ? D2_TT : DP_TT));
          // (a multiple of 8 workgroups: the kernel deals the tiles to the 8 XCDs in equal runs of slots)
          const int cap = num_cus() >= 8 ? num_cus()/8*8 : 8;
          // (one workgroup per tile for the half-batch launches of the two-chain step -- 256 tiles on 256 instead of 224
          // workgroups, so that none walks two tiles -- measured no change: 6.49 ms either way, profiles/r05_dwpw2_ablation.txt)
#ifndef D2V_SEQ
#define D2V_SEQ 0
#endif
          const int cap2 = (v2 && D2V_SEQ) ? 2*cap : cap;
          const int n_wg = n_tiles < cap2 ? (n_tiles + 7)/8*8 : cap2;
#ifndef D2V_NW
#define D2V_NW 4
#endif
#ifndef D2V_AHEAD
#define D2V_AHEAD 8
#endif
#ifdef BRV_WITH_VARIANTS
          if (v2) hipLaunchKernelGGL((dwpw2_v2_kernel<D2V_NW, D2V_AHEAD, (bool)D2V_SEQ>), dim3(n_wg), dim3(64*D2V_NW), 0, st, dp);
          else
#else
          if (v2) return fail(-1, "BRV_OPT_DWPW2_V2: the whole-row fused forward is not in this build (tools/mkvariant.sh <tag> -DBRV_WITH_VARIANTS)");
#endif
          hipLaunchKernelGGL(dwpw2_fused_kernel, dim3(n_wg), dim3(512), 0, st, dp);
          HIP_OK(hipGetLastError());
        }
      }
    }
    SkipCombineParams sc; memset(&sc, 0, sizeof(sc));
    sc.u = ubuf(0); sc.u_stride = ws.u_stride/2; sc.stats = stats; sc.stats_stride = (long long)B*kStatStride;
    sc.prepared = prep; sc.skip = skip; sc.nb = l.nb; sc.B = B; sc.T = (int)T; sc.Bnp = l.Bnp; sc.Scp = l.Scp;
    sc.inv_n = 1.0/((double)T*l.H); sc.eps = 1e-8f;
    for (int i = 0; i < l.nb; ++i) sc.lazy_off[i] = l.blk[i].p_lazy;
    {
      ProfScope prof("skip_combine", 0, 2.0*BT*l.Scp*l.nb + 4.0*BT*l.Scp, st);
      int gx = (int)((T*(l.Scp/8) + 255)/256);
      if (gx > 128) gx = 128;      // (64 / 32 / 16: 136 - 137 us as well, round 5: not the per-workgroup prologue)
      hipLaunchKernelGGL(skip_combine_kernel, dim3(gx, B), dim3(256), 0, st, sc);
      HIP_OK(hipGetLastError());
    }
  } else
  for (int i = 0; i < l.nb; ++i) {
    const BlockOff& b = l.blk[i];
    const bool has_res = i < l.nb - 1;
    const int dil = 1 << (i % cfg->layers);
    // 1x1 conv Bn -> H (+ statistics of prelu_1 output)
    memset(&g, 0, sizeof(g));
    g.a = rows_bf16(xbuf(i), l.Bnp, T);
    g.W = prep + b.p_c1_f; g.T = (int)T; g.Np = l.Hp; g.Kp = l.Bnp;
    g.Wp = prep + b.p_c1_fp; g.wp_nsl = 64;
    g.e.out = z1buf(i); g.e.ldo = l.Hp; g.e.bias = params + b.conv_b; g.e.N = l.H;
    g.e.stats_out = stat(1 + 2*i); g.e.stats_slope = params + b.prelu1;
    if (int r = launch_gemm_rows<A_BF16, E_STORE>(g, B, st, "pw1_fwd", 2.0*BT*(l.Bnp + l.Hp))) return r;
    // prelu_1 -> gLN -> depthwise dilated conv (+ statistics of prelu_2 output)
    DwParams d; memset(&d, 0, sizeof(d));
    d.z1 = z1buf(i); d.z2 = z2buf(i); d.B = B; d.T = (int)T; d.Cp = l.Hp; d.C = l.H;
    d.slope1 = params + b.prelu1; d.stats1 = stat(1 + 2*i);
    d.gamma1 = params + b.n1_g; d.beta1 = params + b.n1_b;
    d.inv_n = 1.0/((double)T*l.H); d.eps = 1e-8f;
    d.taps = params + b.dconv_w; d.bias = params + b.dconv_b;
    d.dil = dil; d.left = ((l.P - 1)*dil)/2;
    d.stats2 = stat(2 + 2*i); d.slope2 = params + b.prelu2;
    if (int r = dispatch_p<DwFwd>(l.P, d, st)) return r;
    // prelu_2 -> gLN -> [res | skip] 1x1 convs, residual add, skip accumulation
    memset(&g, 0, sizeof(g));
    g.a = rows_bf16(z2buf(i), l.Hp, T);
    g.a.slope = params + b.prelu2;
    set_affine(g.a, stat(2 + 2*i), params + b.n2_g, params + b.n2_b, l.H, T);
    const int rs0 = has_res ? l.Bnp : 0;
    g.W = prep + b.p_rs_f; g.T = (int)T; g.Np = rs0 + l.Scp; g.Kp = l.Hp;
    g.Wp = prep + b.p_rs_fp; g.wp_nsl = 32;
    g.e.out = has_res ? xbuf(i + 1) : nullptr; g.e.ldo = l.Bnp;
    g.e.bias = has_res ? params + b.res_b : nullptr; g.e.N = has_res ? l.Bn : 0;
    g.e.Nsplit = rs0; g.e.bias2 = params + b.skip_b; g.e.N2 = l.Sc;
    g.e.res_in = xbuf(i); g.e.ld_res = l.Bnp;
    g.e.skip = skip; g.e.ld_skip = l.Scp; g.e.skip_init = (i == 0);
    if (int r = launch_gemm_rows<A_BF16, E_RES_SKIP>(g, B, st, "pw2_fwd", 2.0*BT*(l.Hp + l.Bnp + rs0) + 4.0*BT*l.Scp*(i == 0 ? 1 : 2))) return r;
  }
  // prelu -> output 1x1 conv -> sigmoid -> mask * encoder output
  memset(&g, 0, sizeof(g));
  g.a = rows_bf16(skip, l.Scp, T);
  g.a.slope = params + l.tcn_prelu;
  g.W = prep + l.p_out_f; g.T = (int)T; g.Np = l.S*l.Np; g.Kp = l.Scp;
  g.e.out = y; g.e.ldo = l.Np; g.e.bias = params + l.out_b; g.e.N = l.N;
  g.e.w_in = w; g.e.ld_w = l.Np; g.e.m_out = m; g.e.S = l.S; g.e.Np_src = l.Np;
  if (int r = launch_gemm_rows<A_F32, E_MASK>(g, B, st, "mask_fwd", 4.0*BT*l.Scp + 2.0*BT*l.Np*(1 + 2*l.S))) return r;
  // decoder: synthesis filterbank + overlap-add, cropped to the input length
  memset(&g, 0, sizeof(g));
  g.a = rows_bf16(y, l.Np, T);
  g.W = prep + l.p_dec_f; g.T = (int)T; g.Np = l.Kfp; g.Kp = l.Np;
  g.e.wave_out = out; g.e.hop = l.hop; g.e.Kf = l.K; g.e.wave_stride = L;
  g.e.wave_len = (int)L;
  if (int r = launch_gemm_rows<A_BF16, E_OLA>(g, B*l.S, st, "dec_fwd", 2.0*BT*l.S*l.Np + 4.0*B*l.S*L)) return r;
  return 0;
}

// Block range of part `part` of `nparts`: the backward pass walks the blocks from the last to
// the first, so part 0 owns the LAST group of blocks.
static void part_range(int nb, int part, int nparts, int& lo, int& hi) {
  lo = (int)((long long)nb*(nparts - 1 - part)/nparts);
  hi = (int)((long long)nb*(nparts - part)/nparts) - 1;
}

int brv_ctn_grad_bucket(const brv_ctn_config* cfg, int32_t part, int32_t nparts, int64_t* offset,
                        int64_t* count) {
  Layout l; if (int r = l.init(cfg)) return r;
  if (nparts < 1 || part < 0 || part >= nparts || !offset || !count) return fail(-1, "bad part");
  if (l.causal) {                       // the causal path runs whole in its last part
    *offset = 0; *count = part == nparts - 1 ? l.n_params : 0;
    return 0;
  }
  int lo, hi; part_range(l.nb, part, nparts, lo, hi);
  // flat layout: [enc dec ln bottleneck | block 0 .. block nb-1 | tcn prelu, output conv]
  const long long begin = part == nparts - 1 ? 0 : (lo <= hi ? l.blk[lo].conv_w : -1);
  int plo, phi; long long end = l.n_params;
  if (part > 0) {                        // ends where the previous (later-block) part began
    int q = part - 1;
    for (;; --q) { part_range(l.nb, q, nparts, plo, phi); if (plo <= phi || q == 0) break; }
    end = plo <= phi ? l.blk[plo].conv_w : l.n_params;
    // empty earlier parts took nothing but the tail; the tail belongs to part 0 only
    if (plo > phi) end = l.tcn_prelu;
  }
  if (begin < 0) { *offset = 0; *count = 0; if (part == 0) { *offset = l.tcn_prelu; *count = l.n_params - l.tcn_prelu; } return 0; }
  *offset = begin; *count = end - begin;
  return 0;
}

}  // extern "C"

// Deferred weight gradients of blocks [blk_lo, blk_hi] (the data-gradient chain of these blocks has
// run on a stream this one is ordered after): two grouped launches.
static int deferred_wgrads(const Layout& l, const Workspace& ws, char* base, const bf16_t* prep,
                           const float* params, float* grads, double* stats, double* sums, int B, long long T,
                           int blk_lo, int blk_hi, hipStream_t st) {
  const double BT = (double)B*(double)T;
  auto stat = [&](int i) { return stats + (long long)i*B*kStatStride; };
  auto xbuf = [&](int i) { return (bf16_t*)(base + ws.x + ws.x_stride*i); };
  auto z2buf = [&](int i) { return (bf16_t*)(base + ws.z2 + ws.z_stride*i); };
  auto eBbuf = [&](int i) { return (bf16_t*)(base + ws.eB + ws.eB_stride*i); };
  auto gcopy = [&](int i) { return (bf16_t*)(base + ws.gcopy + ws.gcopy_stride*i); };
  bf16_t* gskip = (bf16_t*)(base + ws.gskip);
  const int ldg = l.Bnp + l.Scp;
  WgradParams wg;
  // deferred weight gradients of this part's blocks: two grouped launches
  // residual / skip convs of every block in ONE launch when the padded widths are the
  // default 128 | 128 (gemm_wgrad_full.cuh); other architectures use the generic path
  const bool full_rs = l.Bnp == 128 && l.Scp == 128 && l.Hp % W2_BH == 0 && l.nb <= kWgMaxProb &&
                       !opt(BRV_OPT_NO_WGRAD_FULL);
  if (full_rs && blk_lo <= blk_hi) {
    WgradFullParams fp; memset(&fp, 0, sizeof(fp));
    const int nblk = blk_hi - blk_lo + 1;
    fp.B = B; fp.T = (int)T; fp.nprob = nblk; fp.n_htiles = l.Hp/W2_BH;
    fp.ldg0 = l.Bnp; fp.bsg0 = T*l.Bnp; fp.ldg1 = ldg; fp.bsg1 = T*ldg;
    fp.ldh = l.Hp; fp.bsh = T*l.Hp;
    fp.N0 = l.Bn; fp.N1 = l.Sc; fp.Kout = l.H; fp.ldo = l.H;
    fp.inv_n = 1.0/((double)T*l.H); fp.eps = 1e-8f;
    for (int i = blk_lo; i <= blk_hi; ++i) {
      const BlockOff& b = l.blk[i];
      WgradProb& pr = fp.prob[i - blk_lo];
      const bool has_res = i < l.nb - 1;
      pr.g0 = has_res ? gcopy(i) : nullptr; pr.g1 = gskip; pr.h = z2buf(i);
      pr.out0 = has_res ? grads + b.res_w : nullptr; pr.out1 = grads + b.skip_w;
      pr.gbias0 = has_res ? grads + b.res_b : nullptr; pr.gbias1 = grads + b.skip_b;
      pr.slope = params + b.prelu2; pr.stats = stat(2 + 2*i);
      pr.gamma = params + b.n2_g; pr.beta = params + b.n2_b;
    }
    ProfScope prof("pw2_wgrad", 2.0*nblk*BT*(double)(l.Bnp + l.Scp)*l.Hp,
                   2.0*BT*(l.Bnp + l.Scp + l.Hp)*nblk, st);
#ifdef BRV_DIAG
    if (const char* d = getenv("BRV_DBG_WG")) fp.dbg = atoi(d);
#endif
    // 24 blocks x 8 H slices = 192 owners would leave a quarter of the CUs idle: the items are
    // divided over kWgSplit workgroups each (768 = 3 full rounds) when the batch allows
    fp.n_split = (B >= kWgSplit && !opt(BRV_OPT_NO_WGRAD_SPLIT)) ? kWgSplit : 1;
    fp.part = reinterpret_cast<float*>(base + ws.wgpart);
    const bool wide = !opt(BRV_OPT_NO_WGRAD_128) && l.Hp % W3_BH == 0;
    if (wide) {
      // 128 H channels per workgroup: 24 blocks x 4 slices; 8 item splits at the whole batch (768 workgroups), 4
      // when this call is one of two half-batch chains (384 each)
      fp.n_htiles = l.Hp/W3_BH;
      if (fp.n_split > 1 && B >= 4*kWgSplit) fp.n_split = 2*kWgSplit;
      const int grid = 8*ceil_div(nblk, 8)*fp.n_htiles*fp.n_split;
      hipLaunchKernelGGL(wgrad_full128_kernel, dim3(grid), dim3(64*W2_NW), 0, st, fp);
    } else {
      const int grid = 8*ceil_div(nblk, 8)*fp.n_htiles*fp.n_split;
      hipLaunchKernelGGL(wgrad_full_kernel, dim3(grid), dim3(64*W2_NW), 0, st, fp);
    }
    if (fp.n_split > 1)
      hipLaunchKernelGGL(wgrad_full_reduce_kernel, dim3(128, (unsigned)nblk), dim3(256), 0, st, fp);
    HIP_OK(hipGetLastError());
  }
  for (int i0 = blk_lo; i0 <= blk_hi; i0 += kWgMaxProb) {
    const int n = std::min(kWgMaxProb, blk_hi + 1 - i0);
    // [res | skip]: G = [g_out_i | g_skip], H = gLN_2(prelu_2(z2_i)); the last block has
    // no residual conv, so it goes in a launch of its own (different G width)
    WgradGroupParams gp; memset(&gp, 0, sizeof(gp));
    WgradParams& q = gp.base;
    q.g = rows_bf16(nullptr, l.Bnp, T); q.g.K0 = l.Bnp;
    q.g.ld1 = ldg; q.g.bs1 = T*ldg;
    q.h = rows_bf16(nullptr, l.Hp, T);
    set_affine(q.h, nullptr, nullptr, nullptr, l.H, T);
    q.B = B; q.T = (int)T; q.Gp = l.Bnp + l.Scp; q.Hp = l.Hp;
    q.G0p = l.Bnp; q.N0 = l.Bn; q.N1 = l.Sc; q.Kout = l.H; q.ldo = l.H;
    int np = 0;
    for (int i = i0; i < i0 + n; ++i) {
      if (i == l.nb - 1) continue;
      const BlockOff& b = l.blk[i];
      WgradProb& pr = gp.prob[np++];
      pr.g0 = gcopy(i); pr.g1 = gskip; pr.h = z2buf(i);
      pr.out0 = grads + b.res_w; pr.out1 = grads + b.skip_w;
      pr.gbias0 = grads + b.res_b; pr.gbias1 = grads + b.skip_b;
      pr.slope = params + b.prelu2; pr.stats = stat(2 + 2*i);
      pr.gamma = params + b.n2_g; pr.beta = params + b.n2_b;
    }
    gp.nprob = np;
    if (np > 0 && !full_rs)
      if (int r = launch_wgrad_group<A_BF16>(gp, st, "pw2_wgrad",
                                             2.0*BT*(l.Bnp + l.Scp + l.Hp))) return r;
    if (i0 + n == l.nb && !full_rs) {             // last block: skip conv only
      const BlockOff& b = l.blk[l.nb - 1];
      memset(&wg, 0, sizeof(wg));
      wg.g = rows_bf16(gskip, ldg, T);
      wg.h = rows_bf16(z2buf(l.nb - 1), l.Hp, T); wg.h.slope = params + b.prelu2;
      set_affine(wg.h, stat(2 + 2*(l.nb - 1)), params + b.n2_g, params + b.n2_b, l.H, T);
      wg.B = B; wg.T = (int)T; wg.Gp = l.Scp; wg.Hp = l.Hp;
      wg.out1 = grads + b.skip_w; wg.G0p = 0; wg.N0 = 0; wg.N1 = l.Sc;
      wg.Kout = l.H; wg.ldo = l.H; wg.gbias1 = grads + b.skip_b;
      if (int r = launch_wgrad<A_BF16>(wg, st, "pw2_wgrad", 2.0*BT*(l.Scp + l.Hp))) return r;
    }
    // first 1x1 conv: G = dz1_i, H = x_i
    memset(&gp, 0, sizeof(gp));
    WgradParams& q1 = gp.base;
    q1.g = rows_bf16(nullptr, l.Hp, T); q1.h = rows_bf16(nullptr, l.Bnp, T);
    q1.B = B; q1.T = (int)T; q1.Gp = l.Hp; q1.Hp = l.Bnp;
    q1.G0p = l.Hp; q1.N0 = l.H; q1.Kout = l.Bn; q1.ldo = l.Bn;
    for (int i = i0; i < i0 + n; ++i) {
      const BlockOff& b = l.blk[i];
      WgradProb& pr = gp.prob[i - i0];
      pr.g0 = eBbuf(i); pr.h = xbuf(i);
      pr.out0 = grads + b.conv_w; pr.gbias0 = grads + b.conv_b;
    }
    gp.nprob = n;
    if (l.pw1_rc() && pw1_rc_wgrad()) {
      // dz1 was never written: rebuilt from e1 (kept per block in eB) and x inside the product
      Pw1WgradRcParams rp; memset(&rp, 0, sizeof(rp));
      rp.B = B; rp.T = (int)T; rp.nprob = n; rp.C = l.H; rp.Kout = l.Bn; rp.ldo = l.Bn;
      rp.lde = l.Hp; rp.ldx = l.Bnp; rp.bse = T*l.Hp; rp.bsx = T*l.Bnp;
      rp.inv_n = 1.0/((double)T*l.H); rp.eps = 1e-8f;
      for (int i = i0; i < i0 + n; ++i) {
        const BlockOff& b = l.blk[i];
        Pw1WgradRcProb& pr = rp.prob[i - i0];
        pr.e1 = eBbuf(i); pr.x = xbuf(i); pr.Wfp = prep + b.p_c1_fp; pr.bias = params + b.conv_b;
        pr.out = grads + b.conv_w; pr.gbias = grads + b.conv_b;
        pr.stats = stat(1 + 2*i); pr.sums = sums + (long long)(1 + 2*i)*B*kStatStride; pr.slope = params + b.prelu1;
      }
      const int tiles = l.Hp/WG_BG;
      int ns = B;                                  // item-aligned splits (as the stored-dz1 launch)
      while (tiles*n*ns > 2048 && ns % 2 == 0) ns /= 2;
      if (t_opts->wg_target > 0) ns = ceil_div(t_opts->wg_target, tiles*n);
      const int total = B*ceil_div((int)T, WG_BT);
      if (ns > total) ns = total;
      if (ns < 1) ns = 1;
      rp.nsplit = ns;
      ProfScope prof("pw1_wgrad", 2.0*n*BT*(double)l.Hp*l.Bnp*2, 2.0*BT*(l.Hp + l.Bnp)*n, st);
      hipLaunchKernelGGL(pw1_wgrad_rc_kernel, dim3(tiles, ns, n), dim3(256), 0, st, rp);
      HIP_OK(hipGetLastError());
    }
    // (The owner-computes kernel of the [res | skip] gradient was tried for this product too -- 256 of dz1's channels
    // as its DMA'd operand, x as the 128-wide one, two launches of 24 problems: 580 us against the 429 us of the
    // grouped 128 x 128 tiles below: with one item per workgroup its 136 KB zero fill and 128 KB partial tile are
    // no longer amortised. profiles/r05_wgrad128.txt)
    else
    if (int r = launch_wgrad_group<A_BF16>(gp, st, "pw1_wgrad", 2.0*BT*(l.Hp + l.Bnp), -1))
      return r;
  }
  return 0;
}

extern "C" {

int brv_ctn_backward(const brv_ctn_config* cfg, const float* params, const void* prepared,
                     void* workspace, const float* wave, int64_t wave_stride, const float* d_out,
                     float* grads, int64_t batch, int64_t length, const brv_launch_opts* opts,
                     brv_stream_t stream) {
  return brv_ctn_backward_part(cfg, params, prepared, workspace, wave, wave_stride, d_out, grads, batch,
                               length, 0, 1, opts, stream);
}

int brv_ctn_backward_part(const brv_ctn_config* cfg, const float* params, const void* prepared,
                          void* workspace, const float* wave, int64_t wave_stride, const float* d_out,
                          float* grads,
                          int64_t batch, int64_t length, int32_t part, int32_t nparts,
                          const brv_launch_opts* opts, brv_stream_t stream) {
  OptsScope scope(opts);
  Layout l; if (int r = l.init(cfg)) return r;
  hipStream_t st = (hipStream_t)stream;
  const int B = (int)batch; const long long L = length;
  const long long T = l.frames(L);
  if (B < 1 || T < 1) return fail(-1, "empty batch or input shorter than one frame");
  if (nparts < 1 || part < 0 || part >= nparts) return fail(-1, "bad part");
  if (l.causal) {
    if (part != nparts - 1) return 0;
    return backward_causal(l, cfg, params, prepared, workspace, wave, wave_stride, d_out, grads, B, L, T, st);
  }
  int blk_lo, blk_hi; part_range(l.nb, part, nparts, blk_lo, blk_hi);
  const bool head = part == 0, tail = part == nparts - 1;
  Workspace ws; ws.init(l, B, T);
  const double BT = (double)B*(double)T;
  char* base = (char*)workspace;
  const bf16_t* prep = (const bf16_t*)prepared;
  double* stats = (double*)(base + ws.stats);
  double* sums = (double*)(base + ws.sums);
  auto stat = [&](int i) { return stats + (long long)i*B*kStatStride; };
  auto sum = [&](int i) { return sums + (long long)i*B*kStatStride; };
  bf16_t* w = (bf16_t*)(base + ws.w);
  auto xbuf = [&](int i) { return (bf16_t*)(base + ws.x + ws.x_stride*i); };
  auto z1buf = [&](int i) { return (bf16_t*)(base + ws.z1 + ws.z_stride*i); };
  auto z2buf = [&](int i) { return (bf16_t*)(base + ws.z2 + ws.z_stride*i); };
  float* skip = (float*)(base + ws.skip);
  bf16_t* m = (bf16_t*)(base + ws.m);
  bf16_t* y = (bf16_t*)(base + ws.y);
  bf16_t* dpre = (bf16_t*)(base + ws.dpre);
  bf16_t* dw1 = (bf16_t*)(base + ws.dw1);
  bf16_t* gskip = (bf16_t*)(base + ws.gskip);
  bf16_t* gout = (bf16_t*)(base + ws.gout);
  const int ldg = l.Bnp + l.Scp;           // row stride of [g_out | g_skip]
  bf16_t* eA = (bf16_t*)(base + ws.eA);
  auto eBbuf = [&](int i) { return (bf16_t*)(base + ws.eB + ws.eB_stride*i); };
  auto gcopy = [&](int i) { return (bf16_t*)(base + ws.gcopy + ws.gcopy_stride*i); };
  bf16_t* e0 = (bf16_t*)(base + ws.e0);
  bf16_t* dwt = (bf16_t*)(base + ws.dwt);
  const int BS = B*l.S;
  float* vg = (float*)(base + ws.vg);
  const long long vper = (long long)l.H*(5 + l.P);
  auto vslot = [&](int i) { return vg + 2LL*l.N + vper*i; };   // block i's vector grads
  float* vslope = vg + 2LL*l.N + vper*l.nb;    // PReLU slope grads: tcn, then (prelu1, prelu2) per block

  auto ubuf = [&](int i) { return (bf16_t*)(base + ws.u + ws.u_stride*i); };
  // Backward mirror of the fused forward (bwd_fused.cuh; BRV_BWD_FUSE=0: three launches per block):
  // needs the u tensors the fused forward stored and the default widths
  const bool bwd_fused = l.fused_fwd() && bwd_fuse_requested() && l.Bnp == 128 && l.Scp == 128 &&
                         l.Hp % HL_CG == 0 && !opt(BRV_OPT_NO_DZ_FUSE) && bf_frames_ok(T);

  GemmRowsParams g; WgradParams wg;
  if (head) {
  // sums and replica block <- 0; a workspace filled by a forward call in the other mode -> NaN sums (and with
  // them NaN gradients)
  if (int r = launch_call_init(sums, (long long)(ws.stats_bytes/8), vg, (long long)ws.vg_bytes,
                               reinterpret_cast<const int*>(base + ws.stamp), mode_stamp(l.fused_fwd()), nullptr, 0, st))
    return fail(r, "backward: call_init launch");
  // decoder data gradient (framing of d_out) fused with the mask backward
  memset(&g, 0, sizeof(g));
  g.a = frames_of(d_out, L, l.hop, l.K);
  g.W = prep + l.p_dec_b; g.T = (int)T; g.Np = l.Np; g.Kp = l.Kfp;
  g.e.out = dpre; g.e.ldo = l.Np; g.e.out2 = dw1; g.e.w_in = w; g.e.ld_w = l.Np;
  g.e.m_in = m; g.e.S = l.S;
  if (int r = launch_gemm_rows<A_FRAMES, E_MASK_BWD>(g, BS, st, "dec_bwd", 4.0*BS*L + 2.0*BT*l.Np*(1 + 3*l.S))) return r;
  // decoder weight gradient: y^T * frames(d_out)
  memset(&wg, 0, sizeof(wg));
  wg.g = rows_bf16(y, l.Np, T); wg.h = frames_of(d_out, L, l.hop, l.K);
  wg.B = BS; wg.T = (int)T; wg.Gp = l.Np; wg.Hp = l.Kfp;
  wg.out0 = grads + l.dec_w; wg.G0p = l.Np; wg.N0 = l.N; wg.Kout = l.K; wg.ldo = l.K;
  if (int r = launch_wgrad<A_FRAMES>(wg, st, "wgrad_dec", 2.0*BT*l.S*l.Np + 4.0*BS*L, 768)) return r;
  // output conv data gradient, PReLU backward -> gradient wrt skip_sum
  memset(&g, 0, sizeof(g));
  g.a = rows_bf16(dpre, l.Np, T); g.a.nsrc = l.S;
  g.W = prep + l.p_out_b; g.T = (int)T; g.Np = l.Scp; g.Kp = l.S*l.Np;
  g.e.out = gskip; g.e.ldo = ldg; g.e.src_f32 = skip; g.e.ld_srcf = l.Scp;
  g.e.src_slope = params + l.tcn_prelu; g.e.dslope = vslope;
  g.e.rep_stride = ws.vg_stride; g.e.n_rep = kReplicas;
  if (int r = launch_gemm_rows<A_BF16, E_PRELU_BWD>(g, B, st, "mask_bwd", 2.0*BT*l.S*l.Np + 6.0*BT*l.Scp)) return r;
  if (bwd_fused) {
    // fused backward: the layer-norm backward means of the LAST block from g_skip and its stored u
    // (every other block gets them from the kernel that produces its g_out, E_ADD epilogue below)
    GuDotsParams gu; memset(&gu, 0, sizeof(gu));
    gu.g = gskip; gu.ldg = ldg; gu.u = ubuf(l.nb - 1) + l.Bnp; gu.ldu = ldg;
    gu.v1 = reinterpret_cast<const float*>(prep + l.blk[l.nb - 1].p_lazy) + ldg + l.Bnp;
    gu.ncols = l.Scp; gu.B = B; gu.T = (int)T; gu.out = sum(2 + 2*(l.nb - 1));
    ProfScope prof("gu_dots", 0, 4.0*BT*l.Scp, st);
    int gx = (int)((T*(l.Scp/8) + 255)/256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(gu_dots_kernel, dim3(gx, B), dim3(256), 0, st, gu);
    HIP_OK(hipGetLastError());
  }
  // output conv weight / bias gradients, one source at a time
  for (int s = 0; s < l.S; ++s) {
    memset(&wg, 0, sizeof(wg));
    wg.g = rows_bf16(dpre + (long long)s*T*l.Np, l.Np, T); wg.g.bs0 = (long long)l.S*T*l.Np;
    wg.h = rows_bf16(skip, l.Scp, T); wg.h.slope = params + l.tcn_prelu;
    wg.B = B; wg.T = (int)T; wg.Gp = l.Np; wg.Hp = l.Scp;
    wg.out0 = grads + l.out_w + (long long)s*l.N*l.Sc; wg.G0p = l.Np; wg.N0 = l.N;
    wg.Kout = l.Sc; wg.ldo = l.Sc; wg.gbias0 = grads + l.out_b + (long long)s*l.N;
    if (int r = launch_wgrad<A_F32>(wg, st, "wgrad_out", 2.0*BT*l.Np + 4.0*BT*l.Scp, 256)) return r;
  }
  }   // head

  // (Running the deferred weight gradients of block chunks on a side stream next to the data-gradient
  // chain was measured slower at every chunk count -- DESIGN.md 5g -- and is gone.)
  const int n_chunks = 1;
  auto wgrads = [&](int blk_lo, int blk_hi, hipStream_t st) -> int {
    return deferred_wgrads(l, ws, base, prep, params, grads, stats, sums, B, T, blk_lo, blk_hi, st);
  };
  const int n_blk_call = blk_hi - blk_lo + 1;
  for (int ch = n_chunks - 1; ch >= 0; --ch) {
  const int c_lo = blk_lo + (int)((long long)n_blk_call*ch/n_chunks);
  const int c_hi = blk_lo + (int)((long long)n_blk_call*(ch + 1)/n_chunks) - 1;
  for (int i = c_hi; i >= c_lo; --i) {
    const BlockOff& b = l.blk[i];
    const bool has_res = i < l.nb - 1;
    const int dil = 1 << (i % cfg->layers);
    const int rs0 = has_res ? l.Bnp : 0;
    const bool blk_fused = bwd_fused && (((l.P - 1)*dil)/2) % dil == 0;
    if (!blk_fused) {
    // [res | skip] data gradient + gLN_2 backward partials
    memset(&g, 0, sizeof(g));
    g.a = rows_bf16(has_res ? gout : gskip, ldg, T);
    g.W = prep + b.p_rs_b; g.T = (int)T; g.Np = l.Hp; g.Kp = rs0 + l.Scp;
    g.Wp = prep + b.p_rs_bp; g.wp_nsl = 32;
    g.e.out = eA; g.e.ldo = l.Hp; g.e.N = l.H;
    g.e.src = z2buf(i); g.e.ld_src = l.Hp; g.e.src_slope = params + b.prelu2;
    g.e.src_stats = stat(2 + 2*i); g.e.inv_n = 1.0/((double)T*l.H); g.e.eps = 1e-8f;
    g.e.gamma = params + b.n2_g;
    g.e.dgamma = vslot(i) + 2*l.H; g.e.dbeta = vslot(i) + 3*l.H;
    g.e.rep_stride = ws.vg_stride; g.e.n_rep = kReplicas;
    g.e.sums_out = sum(2 + 2*i);
    if (int r = launch_gemm_rows<A_BF16, E_GLN_BWD>(g, B, st, "pw2_dgrad", 2.0*BT*(rs0 + l.Scp + 2*l.Hp))) return r;
    }
    // gLN_2 + prelu_2 backward: fused into the depthwise backward below when its LDS window
    // (tile + halo rows) fits -- dz2 is then built once per element in LDS and never written
    // (dwconv_bwd_halo_kernel). BRV_NO_DZ_FUSE keeps the separate pass.
    const bool fuse_dz2 = blk_fused || (!opt(BRV_OPT_NO_DZ_FUSE) && l.Hp % HL_CG == 0 &&
                          (((l.P - 1)*dil)/2) % dil == 0 && hl_window_rows(dil, l.P) <= HL_MAXROWS);
    DzParams dz; memset(&dz, 0, sizeof(dz));
    if (!fuse_dz2) {
      dz.e = eA; dz.z = z2buf(i); dz.B = B; dz.T = (int)T; dz.Cp = l.Hp; dz.C = l.H;
      dz.slope = params + b.prelu2; dz.stats = stat(2 + 2*i); dz.sums = sum(2 + 2*i);
      dz.inv_n = 1.0/((double)T*l.H); dz.eps = 1e-8f; dz.dslope = vslope + 2 + 2*i; dz.rep_stride = ws.vg_stride;
      if (int r = launch_dz(dz, st)) return r;
    }
    // depthwise conv backward (data, taps, bias) + gLN_1 partials
    DwParams d; memset(&d, 0, sizeof(d));
    bf16_t* eB = eBbuf(i);
    d.z1 = z1buf(i); d.dz2 = eA; d.e1 = eB; d.B = B; d.T = (int)T; d.Cp = l.Hp; d.C = l.H;
    d.slope1 = params + b.prelu1; d.stats1 = stat(1 + 2*i);
    d.gamma1 = params + b.n1_g; d.beta1 = params + b.n1_b;
    d.inv_n = 1.0/((double)T*l.H); d.eps = 1e-8f;
    d.taps = params + b.dconv_w; d.dil = dil; d.left = ((l.P - 1)*dil)/2;
    d.dgamma1 = vslot(i); d.dbeta1 = vslot(i) + l.H;
    d.dtaps = vslot(i) + 4*l.H; d.dbias = vslot(i) + 4*l.H + (long long)l.H*l.P;
    d.rep_stride = ws.vg_stride; d.sums1 = sum(1 + 2*i);
    if (fuse_dz2) {
      d.z2in = z2buf(i); d.stats2 = stat(2 + 2*i); d.sums2 = sum(2 + 2*i);
      d.slope2 = params + b.prelu2; d.dslope2 = vslope + 2 + 2*i;
    }
    if (blk_fused) {
      BwdFusedParams bf; memset(&bf, 0, sizeof(bf));
      bf.d = d; bf.d.dz2 = nullptr;
      bf.g = has_res ? gout : gskip; bf.ldg = ldg; bf.Kg = rs0 + l.Scp;
      bf.Wp = prep + b.p_rs_bp; bf.gamma2 = params + b.n2_g;
      bf.dgamma2 = vslot(i) + 2*l.H; bf.dbeta2 = vslot(i) + 3*l.H;
      if (int r = DwBwdFused<3>::run(bf, st)) return r;      // (fused_fwd() implies kernel_size 3)
    } else
    if (int r = dispatch_p<DwBwd>(l.P, d, st)) return r;
    // gLN_1 + prelu_1 backward -> dz1: fused into the A staging of the data-gradient GEMM
    // below (A_DZ: dz1 is computed from e1 and z1 on load and written back over e1 for the
    // deferred weight gradient); BRV_NO_DZ1_FUSE keeps the separate pass
    // (needs one n-tile: the workgroup that stages an A element must be its only reader)
    const bool fuse_dz1 = !opt(BRV_OPT_NO_DZ1_FUSE) && (l.Bnp == 128 || l.Bnp == 64);
    if (!fuse_dz1) {
      memset(&dz, 0, sizeof(dz));
      dz.e = eB; dz.z = z1buf(i); dz.B = B; dz.T = (int)T; dz.Cp = l.Hp; dz.C = l.H;
      dz.slope = params + b.prelu1; dz.stats = stat(1 + 2*i); dz.sums = sum(1 + 2*i);
      dz.inv_n = 1.0/((double)T*l.H); dz.eps = 1e-8f; dz.dslope = vslope + 1 + 2*i; dz.rep_stride = ws.vg_stride;
      if (int r = launch_dz(dz, st)) return r;
    }
    // first 1x1 conv: data gradient + residual path -> gradient wrt block input
    memset(&g, 0, sizeof(g));
    g.a = rows_bf16(eB, l.Hp, T);
    g.W = prep + b.p_c1_b; g.T = (int)T; g.Np = l.Bnp; g.Kp = l.Hp;
    g.e.out = gout; g.e.ldo = ldg; g.e.add_in = has_res ? gout : nullptr; g.e.ld_add = ldg;
    // block i-1's deferred [res | skip] weight gradient needs this g_out after gout is
    // overwritten again: keep a copy (16 MB per block at the BASELINE size)
    g.e.out2 = i > 0 ? gcopy(i - 1) : nullptr; g.e.ld_srcf = l.Bnp;
    if (bwd_fused && i > 0) {        // block i - 1's layer-norm backward means, from its g and u
      g.e.gu_u = ubuf(i - 1); g.e.ld_gu = ldg; g.e.gu_gskip = gskip; g.e.ld_gs = ldg;
      g.e.gu_v1 = reinterpret_cast<const float*>(prep + l.blk[i - 1].p_lazy) + ldg;
      g.e.gu_out = sum(2 + 2*(i - 1));
    }
    if (fuse_dz1 && l.pw1_rc()) {
      Pw1DgradRcParams rp; memset(&rp, 0, sizeof(rp));
      rp.e1 = eB; rp.lde = l.Hp; rp.bse = T*l.Hp;
      rp.x = xbuf(i); rp.ldx = l.Bnp; rp.bsx = T*l.Bnp;
      rp.Wfp = prep + b.p_c1_fp; rp.bias = params + b.conv_b; rp.Wb = prep + b.p_c1_b;
      rp.stats = stat(1 + 2*i); rp.sums = sum(1 + 2*i); rp.slope = params + b.prelu1;
      rp.inv_n = 1.0/((double)T*l.H); rp.eps = 1e-8f; rp.C = l.H;
      rp.T = (int)T; rp.n_ttiles = ceil_div((int)T, RC_BM); rp.batch = B;
      rp.e = g.e;
      rp.e.dslope = vslope + 1 + 2*i; rp.e.rep_stride = ws.vg_stride; rp.e.n_rep = kReplicas;
      if (opt(BRV_OPT_PW1_RC_TILES)) {        // first form: one tile per workgroup, weights re-read per tile
        ProfScope prof("pw1_dgrad", 2.0*BT*(double)l.Hp*l.Bnp*2,
                       2.0*BT*(l.Hp + l.Bnp*(has_res ? 3 : 2)), st);
        hipLaunchKernelGGL(pw1_dgrad_rc_kernel, dim3(rp.n_ttiles*B), dim3(256), 0, st, rp);
      } else {
        Pw1DgradWsParams wp; wp.r = rp;
        wp.dz_out = pw1_rc_wgrad() ? nullptr : eB;      // stored-dz1 weight gradient: dz1 over e1, in place
        wp.dbg = nullptr;
#ifdef WSD_STAMP
        wp.dbg = debug_buffer();
#endif
        // algorithmic bytes: e1 and x in (+ the residual g_out), dz1 and the new g_out out -- and the companions the
        // row-wise epilogue streams: the copy of g_out kept for block i - 1's weight gradient and, for that block's
        // layer-norm backward means, its u ([res | skip] wide) and g_skip rows (the line understated
        // the kernel by 2 BT 384; PMC 244.8 MB per launch = this count)
        const double companions = (i > 0 ? l.Bnp : 0) + (bwd_fused && i > 0 ? (double)l.Bnp + 2.0*l.Scp : 0.0);
        ProfScope prof("pw1_dgrad", 2.0*BT*(double)l.Hp*l.Bnp*2,
                       2.0*BT*(l.Hp*(wp.dz_out ? 2 : 1) + l.Bnp*(has_res ? 3 : 2) + companions), st);
        int grid = num_cus();
        const int total = ceil_div((int)T, WSD_TF)*B;
        if (grid > total) grid = total;
        if (l.H == RC_H) hipLaunchKernelGGL(pw1_dgrad_ws_kernel<true>, dim3(grid), dim3(512), 0, st, wp);
        else hipLaunchKernelGGL(pw1_dgrad_ws_kernel<false>, dim3(grid), dim3(512), 0, st, wp);
      }
      HIP_OK(hipGetLastError());
      continue;
    }
    if (fuse_dz1) {
      g.a.p1 = z1buf(i); g.a.ld1 = l.Hp; g.a.bs1 = T*l.Hp;
      g.a.slope = params + b.prelu1; g.a.stats = stat(1 + 2*i); g.a.sums = sum(1 + 2*i);
      g.a.inv_n = 1.0/((double)T*l.H); g.a.eps = 1e-8f; g.a.C = l.H;
      g.e.dslope = vslope + 1 + 2*i; g.e.rep_stride = ws.vg_stride; g.e.n_rep = kReplicas;
      if (int r = launch_gemm_rows<A_DZ, E_ADD>(g, B, st, "pw1_dgrad", 2.0*BT*(3*l.Hp + l.Bnp*(has_res ? 2 : 1)))) return r;
      continue;
    }
    if (int r = launch_gemm_rows<A_BF16, E_ADD>(g, B, st, "pw1_dgrad", 2.0*BT*(l.Hp + l.Bnp*(has_res ? 2 : 1)))) return r;
  }
    if (int r = wgrads(c_lo, c_hi, st)) return r;
  }   // chunks
  if (tail) {
  // bottleneck conv: data gradient + first gLN backward partials
  memset(&g, 0, sizeof(g));
  g.a = rows_bf16(gout, ldg, T);
  g.W = prep + l.p_bott_b; g.T = (int)T; g.Np = l.Np; g.Kp = l.Bnp;
  g.e.out = e0; g.e.ldo = l.Np; g.e.N = l.N;
  g.e.src = w; g.e.ld_src = l.Np; g.e.src_stats = stat(0);
  g.e.inv_n = 1.0/((double)T*l.N); g.e.eps = 1e-8f;
  g.e.gamma = params + l.ln_g; g.e.dgamma = vg; g.e.dbeta = vg + l.N;
  g.e.rep_stride = ws.vg_stride; g.e.n_rep = kReplicas;
  g.e.sums_out = sum(0);
  if (int r = launch_gemm_rows<A_BF16, E_GLN_BWD>(g, B, st, "bottleneck_dgrad", 2.0*BT*(l.Bnp + 2*l.Np))) return r;
  // bottleneck conv: weight / bias gradients against gLN(w)
  memset(&wg, 0, sizeof(wg));
  wg.g = rows_bf16(gout, ldg, T); wg.h = rows_bf16(w, l.Np, T);
  set_affine(wg.h, stat(0), params + l.ln_g, params + l.ln_b, l.N, T);
  wg.B = B; wg.T = (int)T; wg.Gp = l.Bnp; wg.Hp = l.Np;
  wg.out0 = grads + l.bott_w; wg.G0p = l.Bnp; wg.N0 = l.Bn; wg.Kout = l.N; wg.ldo = l.N;
  wg.gbias0 = grads + l.bott_b;
  if (int r = launch_wgrad<A_BF16>(wg, st, "bottleneck_wgrad", 2.0*BT*(l.Bnp + l.Np), 256)) return r;
  // total gradient wrt the encoder output
  CombineParams cb; memset(&cb, 0, sizeof(cb));
  cb.e0 = e0; cb.w = w; cb.dw1 = dw1; cb.dw = dwt; cb.B = B; cb.T = (int)T;
  cb.Cp = l.Np; cb.C = l.N; cb.S = l.S; cb.stats = stat(0); cb.sums = sum(0);
  cb.inv_n = 1.0/((double)T*l.N); cb.eps = 1e-8f;
  {
    const long long per_item = T*(l.Np/8);
    int gx = (int)((per_item + 1023)/1024);
    if (gx > 2048) gx = 2048;
    if (gx < 1) gx = 1;
    ProfScope prof("gln0_bwd", 0, 2.0*BT*l.Np*(3 + l.S), st);
    hipLaunchKernelGGL(gln0_bwd_combine_kernel, dim3(gx, B), dim3(256), 0, st, cb);
    HIP_OK(hipGetLastError());
  }
  // encoder weight gradient: dw^T * frames(wave)
  memset(&wg, 0, sizeof(wg));
  wg.g = rows_bf16(dwt, l.Np, T); wg.h = frames_of(wave, L, l.hop, l.K, wave_stride);
  wg.B = B; wg.T = (int)T; wg.Gp = l.Np; wg.Hp = l.Kfp;
  wg.out0 = grads + l.enc_w; wg.G0p = l.Np; wg.N0 = l.N; wg.Kout = l.K; wg.ldo = l.K;
  if (int r = launch_wgrad<A_FRAMES>(wg, st, "wgrad_enc", 2.0*BT*l.Np + 4.0*B*L, 768)) return r;
  }   // tail
  // per-channel / slope gradients of this part: fold the replicas into the flat gradient
  return reduce_vector_grads(l, ws, vg, grads, st, blk_lo, blk_hi, tail, head);
}

}  // extern "C"
