// Shared device helpers for the brever_amd HIP kernels (gfx950 / CDNA4 only).
//
// Conventions used by every kernel in this directory:
//  * activations are channels-last: [batch][frame t][channel], bf16, channel
//    count padded to a multiple of 64 (padding columns hold exact zeros);
//  * one wavefront = 64 lanes; workgroups are 256 threads (4 waves);
//  * reductions that feed normalisation statistics accumulate in fp64 so the
//    result does not depend on the (unordered) arrival of the partial sums.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace brv {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

typedef uint16_t bf16_t;   // storage type of a bf16 element

constexpr int kWave = 64;
constexpr int kThreads = 256;

// ---- bf16 <-> f32 ---------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) {
  return __uint_as_float(((uint32_t)v) << 16);
}
// round-to-nearest-even; the plain cast lowers to v_cvt_pk_bf16_f32 and keeps NaNs
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(bf16_t, h);
}
// two values -> one dword: the vector conversion is ONE v_cvt_pk_bf16_f32 (two scalar casts
// compile to two of them plus a shift and an or)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ void unpack8(const uint4& q, float (&f)[8]) {
  f[0] = __uint_as_float(q.x << 16); f[1] = __uint_as_float(q.x & 0xffff0000u);
  f[2] = __uint_as_float(q.y << 16); f[3] = __uint_as_float(q.y & 0xffff0000u);
  f[4] = __uint_as_float(q.z << 16); f[5] = __uint_as_float(q.z & 0xffff0000u);
  f[6] = __uint_as_float(q.w << 16); f[7] = __uint_as_float(q.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  uint4 q;
  q.x = pack2(f[0], f[1]); q.y = pack2(f[2], f[3]);
  q.z = pack2(f[4], f[5]); q.w = pack2(f[6], f[7]);
  return q;
}
// value a bf16 store would hold, as f32
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }

__device__ __forceinline__ float prelu(float v, float a) {
  return v > 0.f ? v : a*v;
}

// ---- buffer (descriptor) loads/stores, packed fp32 -------------------------
// Out-of-range offsets read zeros and drop stores in hardware: no branches around memory
// operations, so the compiler keeps its waits counted.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;     // v_pk_{add,mul,fma}_f32 operands

__device__ __forceinline__ uint4 pack8v(const f32x2 (&v)[4]) {
  uint4 q;
  q.x = pack2(v[0].x, v[0].y); q.y = pack2(v[1].x, v[1].y);
  q.z = pack2(v[2].x, v[2].y); q.w = pack2(v[3].x, v[3].y);
  return q;
}

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
#ifndef BRV_STORE_AUX
#define BRV_STORE_AUX 0
#endif
  __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)off, 0, BRV_STORE_AUX);
}
constexpr unsigned int kOob = 0xfffffff0u;     // offset that is out of range for any descriptor

// ---- reductions -----------------------------------------------------------
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Sum over the 256 threads of a workgroup; result valid in thread 0.
// `scratch` must hold >= 4 elements of T and is reusable after the call.
template <typename T>
__device__ __forceinline__ T block_sum(T v, T* scratch) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  T r = T(0);
  if (threadIdx.x == 0) r = scratch[0] + scratch[1] + scratch[2] + scratch[3];
  return r;
}

// BRV_NO_ATOMICS (timing experiments only, results are wrong): bit 0 drops the f64 adds,
// bit 1 the f32 adds -- the time that disappears is what the atomics cost.
#ifndef BRV_NO_ATOMICS
#define BRV_NO_ATOMICS 0
#endif
__device__ __forceinline__ void atomic_add_f64(double* p, double v) {
  if (BRV_NO_ATOMICS & 1) return;
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void atomic_add_f32(float* p, float v) {
  if (BRV_NO_ATOMICS & 2) return;
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- global layer norm statistics ----------------------------------------
// stats[b] = {sum, sum of squares} over the (frames x true channels) of item b.
// Layout of the per-item statistics: item b owns kStatStride doubles, the sum at +0 and the
// sum of squares at +kStatStride/2, i.e. two PRIVATE 128-byte lines per item. Atomics
// that land in the same 128-byte line serialise at ~12 ns each on MI355X whatever the
// address inside the line (tools/atomicbench.hip): with the packed [B][2] layout the
// 4000 wave-level adds of one 1x1 convolution cost 21 us after the last wave had
// finished; spread over 2B lines they cost < 2 us.
constexpr int kStatStride = 32;
__host__ __device__ __forceinline__ constexpr long long stat_sum(long long b) { return b*kStatStride; }
__host__ __device__ __forceinline__ constexpr long long stat_sq(long long b) { return b*kStatStride + kStatStride/2; }
struct NormStat { float mean, rstd; };
// 1/sqrt(v) to fp32 accuracy: hardware v_rsq_f32 (1 ulp) + one Newton step. The fp64 square root and division
// this replaces (round 6) are ~45 half-rate VALU instructions that EVERY thread of every normalising kernel
// executed for a per-item constant; the variance itself (a difference of two nearly equal sums) stays in fp64.
__device__ __forceinline__ float rstd_f32(double v) {
  const float x = (float)v;
  const float r = __builtin_amdgcn_rsqf(x);
  return r*__builtin_fmaf(-0.5f*x*r, r, 1.5f);
}
__device__ __forceinline__ NormStat norm_stat(const double* stats, int b,
                                              double inv_n, float eps) {
  const double s = stats[stat_sum(b)], ss = stats[stat_sq(b)];
  const double mean = s*inv_n;
  double var = ss*inv_n - mean*mean;      // biased variance, as nn.GroupNorm
  if (var < 0.0) var = 0.0;
  NormStat r;
  r.mean = (float)mean;
  r.rstd = rstd_f32(var + (double)eps);
  return r;
}

// v[j] = p[c0 + j] for c0 + j < C, else 0 (p may be null -> zeros). The common case
// (whole chunk inside the tensor) issues 8 unconditional loads the compiler can batch
// and merge; the guarded form compiles to serialised load/wait pairs.
__device__ __forceinline__ void load8_masked(const float* p, int c0, int C, float (&v)[8]) {
  if (p != nullptr && c0 + 8 <= C) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = p[c0 + j];
  } else if (p != nullptr) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = c0 + j;
      const float x = p[c < C ? c : (C > 0 ? C - 1 : 0)];
      v[j] = c < C ? x : 0.f;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
  }
}

__host__ __device__ inline int round_up(int x, int m) { return (x + m - 1)/m*m; }
__host__ __device__ inline int ceil_div(int x, int m) { return (x + m - 1)/m; }

}  // namespace brv
