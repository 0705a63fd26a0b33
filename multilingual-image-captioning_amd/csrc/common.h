// common.h — shared device helpers for libmic_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/mic_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

void mic_set_error(const char* fmt, ...);
#define MIC_CHECK(cond, ...) do { if (!(cond)) { mic_set_error(__VA_ARGS__); return MIC_EINVAL; } } while (0)
#define MIC_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { \
  mic_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); return MIC_ELAUNCH; } } while (0)

// ---- bf16 <-> f32 (round-to-nearest-even; NaN preserved)
__device__ __forceinline__ float bf2f(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
// gfx950 converts in hardware (v_cvt_pk_bf16_f32, RNE, NaN -> quiet NaN): one instruction per PAIR instead of a compare,
// two exec-mask flips and three ALU ops per element (that software sequence was ~60 % of the GEMM epilogue's instructions).
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t f2bf_pk(float lo, float hi) {  // lo in bits [15:0]
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ uint16_t f2bf(float f) { return (uint16_t)(f2bf_pk(f, 0.0f) & 0xffffu); }

template <typename T> struct ElemT;
template <> struct ElemT<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct ElemT<uint16_t> {
  static __device__ __forceinline__ float ld(const uint16_t* p) { return bf2f(*p); }
  static __device__ __forceinline__ void st(uint16_t* p, float v) { *p = f2bf(v); }
};
// rounding through the storage type (what a store+load would give)
template <typename T> __device__ __forceinline__ float round_to(float v);
template <> __device__ __forceinline__ float round_to<float>(float v) { return v; }
template <> __device__ __forceinline__ float round_to<uint16_t>(float v) { return bf2f(f2bf(v)); }

// ---- vector loads of 8 (bf16: one 16-B load) / 4 (f32: one 16-B load) contiguous elements
__device__ __forceinline__ void ld8(const uint16_t* p, float* o) {
  uint4 u = *reinterpret_cast<const uint4*>(p);
  o[0] = __uint_as_float(u.x << 16); o[1] = __uint_as_float(u.x & 0xffff0000u);
  o[2] = __uint_as_float(u.y << 16); o[3] = __uint_as_float(u.y & 0xffff0000u);
  o[4] = __uint_as_float(u.z << 16); o[5] = __uint_as_float(u.z & 0xffff0000u);
  o[6] = __uint_as_float(u.w << 16); o[7] = __uint_as_float(u.w & 0xffff0000u);
}
__device__ __forceinline__ void ld8(const float* p, float* o) {
  float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
__device__ __forceinline__ void st8(uint16_t* p, const float* v) {
  uint4 u;
  u.x = f2bf_pk(v[0], v[1]); u.y = f2bf_pk(v[2], v[3]);
  u.z = f2bf_pk(v[4], v[5]); u.w = f2bf_pk(v[6], v[7]);
  *reinterpret_cast<uint4*>(p) = u;
}
__device__ __forceinline__ void st8(float* p, const float* v) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

// ---- wave (64-lane) reductions without the LDS crossbar.  __shfl_xor compiles to ds_bpermute_b32 (an LDS-pipe instruction,
// ~8+ cycles per wave and a round trip of latency each); a 64-lane butterfly is six of them.  gfx950 can do every stage in the
// VALU: lanes^1, ^2 by quad_perm DPP, the 8-lane group by row_half_mirror (i <-> 7-i) first, lanes^8 by row_ror:8,
// lanes^16 by v_permlane16_swap and lanes^32 by v_permlane32_swap (both new on gfx950).  All lanes end with the result.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
#define DPP_HALF_MIRROR 0x141
#define DPP_XOR1 0xB1   /* quad_perm [1,0,3,2] */
#define DPP_XOR2 0x4E   /* quad_perm [2,3,0,1] */
#define DPP_ROR8 0x128  /* row_ror:8 = lane ^ 8 inside a 16-lane row */
__device__ __forceinline__ float lane_xor16(float v) {  // v[lane ^ 16]
  const unsigned u = __builtin_bit_cast(unsigned, v);
  auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);  // r[0] = rows {0,0,2,2}, r[1] = rows {1,1,3,3} of v
  const bool odd_row = (threadIdx.x >> 4) & 1;
  return __builtin_bit_cast(float, odd_row ? r[0] : r[1]);
}
__device__ __forceinline__ float lane_xor32(float v) {  // v[lane ^ 32]
  const unsigned u = __builtin_bit_cast(unsigned, v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);  // r[0] = {lo, lo}, r[1] = {hi, hi}
  const bool hi = (threadIdx.x >> 5) & 1;
  return __builtin_bit_cast(float, hi ? r[0] : r[1]);
}
// sum / max over every aligned group of 8 lanes (all 8 lanes receive it)
__device__ __forceinline__ float group8_sum(float v) {
  v += dpp_f<DPP_HALF_MIRROR>(v);
  v += dpp_f<DPP_XOR1>(v);
  v += dpp_f<DPP_XOR2>(v);
  return v;
}
__device__ __forceinline__ float group8_max(float v) {
  v = fmaxf(v, dpp_f<DPP_HALF_MIRROR>(v));
  v = fmaxf(v, dpp_f<DPP_XOR1>(v));
  v = fmaxf(v, dpp_f<DPP_XOR2>(v));
  return v;
}
// sum over every aligned group of G lanes, G in {4, 8, 16}
template <int G>
__device__ __forceinline__ float group_sum(float v) {
  static_assert(G == 4 || G == 8 || G == 16, "group of 4, 8 or 16 lanes");
  if (G >= 8) v += dpp_f<DPP_HALF_MIRROR>(v);
  v += dpp_f<DPP_XOR1>(v);
  v += dpp_f<DPP_XOR2>(v);
  if (G == 16) v += dpp_f<DPP_ROR8>(v);
  return v;
}
// sum / max over each 32-lane half of the wave (all 32 lanes receive it)
__device__ __forceinline__ float half_sum(float v) {
  v = group8_sum(v);
  v += dpp_f<DPP_ROR8>(v);
  v += lane_xor16(v);
  return v;
}
__device__ __forceinline__ float half_max(float v) {
  v = group8_max(v);
  v = fmaxf(v, dpp_f<DPP_ROR8>(v));
  v = fmaxf(v, lane_xor16(v));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  v = group8_sum(v);
  v += dpp_f<DPP_ROR8>(v);
  v += lane_xor16(v);
  v += lane_xor32(v);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  v = group8_max(v);
  v = fmaxf(v, dpp_f<DPP_ROR8>(v));
  v = fmaxf(v, lane_xor16(v));
  v = fmaxf(v, lane_xor32(v));
  return v;
}

// ---- counter-based dropout: keep iff hash(seed, idx) >= p * 2^32  (same function in every fused epilogue and in
// mic_dropout_mask, so a mask can be materialised for the oracle)
__device__ __forceinline__ uint32_t mic_hash(uint32_t seed, uint32_t idx) {
  uint32_t x = idx * 0x9E3779B1u ^ seed;
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  x += seed * 0x27D4EB2Fu; x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12;
  return x;
}
__device__ __forceinline__ uint32_t dropout_threshold(float p) { return (uint32_t)fminf(p * 4294967296.0f, 4294967295.0f); }
__device__ __forceinline__ bool dropout_keep(uint32_t seed, uint32_t idx, uint32_t thr) { return mic_hash(seed, idx) >= thr; }

// ---- fused fp8 emission under DELAYED per-tensor scaling (BASELINE configs[4]): a producer kernel (LayerNorm forward / backward,
// the GELU / dGELU GEMM epilogues, attention backward) writes its result as OCP fp8 bytes itself, with the scale derived from the
// amax this tensor had in the PREVIOUS pass (state[0], rolled by mic_fp8_roll_amax), records this pass's max |x| in the tensor's
// table of partial maxima and leaves the dequantisation factor in state[1] — what mic_fp8_quantize did in a launch of its own
// (181 launches per train step).  Values are rounded to the storage type (bf16) first, so the bytes equal those of the two-kernel
// path on the same scale.
#define FP8_AMAX_PARTIALS 1024  // per-tensor partial maxima: one atomic per wave, spread over the table (1024 atomics on ONE address cost ~12 us)
struct Q8Out { uint8_t* q; int ldq; float* state; float* amax_next; int fmt; };
struct Q8Ctx { float scale, fmax, amax; };
__device__ __forceinline__ uint32_t cvt4_fp8(const float* v, int fmt) {  // 4 floats -> 4 fp8 bytes (RNE, OCP encodings on gfx950)
  int w = 0;
  if (fmt == MIC_E4M3) {
    w = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], w, true);
  } else {
    w = __builtin_amdgcn_cvt_pk_bf8_f32(v[0], v[1], w, false);
    w = __builtin_amdgcn_cvt_pk_bf8_f32(v[2], v[3], w, true);
  }
  return (uint32_t)w;
}
__device__ __forceinline__ Q8Ctx q8_begin(const Q8Out& o, bool first_thread) {
  Q8Ctx c;
  c.fmax = o.fmt == MIC_E4M3 ? 448.0f : 57344.0f;
  const float amax = o.state[0];
  c.scale = amax > 0.f ? c.fmax / amax : 1.0f;
  // wave-uniform values: keep them in scalar registers (the LayerNorm backward runs at its 128-register bound)
  c.scale = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(c.scale)));
  c.fmax = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(c.fmax)));
  c.amax = 0.f;
  if (first_thread) o.state[1] = amax > 0.f ? amax / c.fmax : 1.0f;
  return c;
}
// 8 values (already rounded to the storage type) -> 8 fp8 bytes; the context keeps the running max |x|
__device__ __forceinline__ uint2 q8_pack8(Q8Ctx& c, const float* v, int fmt) {
  float t[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    c.amax = fmaxf(c.amax, fabsf(v[i]));
    t[i] = fminf(fmaxf(v[i] * c.scale, -c.fmax), c.fmax);
  }
  return make_uint2(cvt4_fp8(t, fmt), cvt4_fp8(t + 4, fmt));
}
__device__ __forceinline__ uint8_t q8_pack1(Q8Ctx& c, float v, int fmt) {
  c.amax = fmaxf(c.amax, fabsf(v));
  const float t = fminf(fmaxf(v * c.scale, -c.fmax), c.fmax);
  int w = 0;
  w = fmt == MIC_E4M3 ? __builtin_amdgcn_cvt_pk_fp8_f32(t, 0.f, w, false) : __builtin_amdgcn_cvt_pk_bf8_f32(t, 0.f, w, false);
  return (uint8_t)(w & 0xff);
}
// every lane of the wave calls this once, after its last pack: one atomic max per wave (non-negative floats order as ints)
__device__ __forceinline__ void q8_end_wave(const Q8Out& o, const Q8Ctx& c, int wave_id) {
  const float m = wave_max(c.amax);
  if ((threadIdx.x & 63) == 0 && m > 0.f && o.amax_next)
    atomicMax(reinterpret_cast<int*>(o.amax_next + (wave_id & (FP8_AMAX_PARTIALS - 1))), __float_as_int(m));
}

// ---- activations
// The FFN activations run once per element inside GEMM epilogues, where the VALU is the bottleneck: each is written for the
// fewest instructions — one v_exp_f32 (base 2, constants pre-multiplied by -log2 e) and one v_rcp_f32 each way.
//   gelu_tanh(x) = 0.5 x (1 + tanh(c (x + a x^3))) = x * sigmoid(z),  z = x (2c + 2ca x^2)     (tanh u = 2 sigmoid(2u) - 1)
//   gelu_tanh'(x) = s + x s (1 - s) dz/dx,  dz/dx = 2c + 6ca x^2
//   quick_gelu(x) = x * sigmoid(1.702 x)
#define MIC_LOG2E 1.4426950408889634f
// ---- softmax partials of the LM-head GEMM epilogues (mic_gemm_args.rowstat): (max, sum exp(x - max)) over one 64-column granule of a
// logits row = an aligned group of 8 lanes x 8 packed bf16 values AS STORED.  nv: valid columns of this lane's 8 (>= 8: all; the
// granule that holds the end of the vocabulary masks the rest to -inf).  Written for the instruction count — this runs once per
// element of a [rows][250 112] matrix on SIMDs that hold ONE wave (63 cycles per element and lane before: tools/probe_head_timeline.hip):
// max3 trees, the subtraction and the base change as one packed FMA per pair, v_exp_f32 (base 2) straight, packed adds.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void granule_stat8(const uint32_t (&w)[4], int nv, float& gm, float& sm) {
  float x[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    x[2 * i] = __uint_as_float(w[i] << 16);
    x[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
  if (nv < 8) {
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = i < nv ? x[i] : -INFINITY;
  }
  const float mx = fmaxf(fmaxf(fmaxf(x[0], fmaxf(x[1], x[2])), fmaxf(x[3], fmaxf(x[4], x[5]))), fmaxf(x[6], x[7]));
  gm = group8_max(mx);
  // (a masked column is -inf: exp2(-inf) = 0 as long as the reference is finite; a granule with no valid column stores (-inf, 0))
  const float nb = gm > -INFINITY ? -gm * MIC_LOG2E : 0.0f;
  const f32x2 L = {MIC_LOG2E, MIC_LOG2E}, NB = {nb, nb};
  f32x2 e[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x2 v = {x[2 * i], x[2 * i + 1]};
    const f32x2 t = __builtin_elementwise_fma(v, L, NB);
    e[i] = f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
  }
  const f32x2 s2 = (e[0] + e[1]) + (e[2] + e[3]);
  sm = group8_sum(s2[0] + s2[1]);
}
#define GELU_K1 1.5957691216057308f   /* 2 * sqrt(2/pi) */
#define GELU_K2 0.07135481627261745f  /* 2 * sqrt(2/pi) * 0.044715 */
__device__ __forceinline__ float sigmoid_neg_log2(float zl) {  // sigmoid(z) given zl = -z * log2(e)
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(zl));
}
__device__ __forceinline__ float act_fwd(int act, float x) {
  switch (act) {
    case MIC_ACT_GELU_ERF: return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
    case MIC_ACT_GELU_TANH: {
      const float p = fmaf(-GELU_K2 * MIC_LOG2E, x * x, -GELU_K1 * MIC_LOG2E);
      return x * sigmoid_neg_log2(x * p);
    }
    case MIC_ACT_QUICK_GELU: return x * sigmoid_neg_log2(x * (-1.702f * MIC_LOG2E));
    default: return x;
  }
}
__device__ __forceinline__ float act_bwd(int act, float x) {
  switch (act) {
    case MIC_ACT_GELU_ERF: {
      float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
      return cdf + x * 0.3989422804014327f * __expf(-0.5f * x * x);
    }
    case MIC_ACT_GELU_TANH: {
      const float x2 = x * x;
      const float p = fmaf(-GELU_K2 * MIC_LOG2E, x2, -GELU_K1 * MIC_LOG2E);
      const float s = sigmoid_neg_log2(x * p);
      const float dz = fmaf(3.0f * GELU_K2, x2, GELU_K1);
      return fmaf(x * s * (1.0f - s), dz, s);
    }
    case MIC_ACT_QUICK_GELU: {
      const float s = sigmoid_neg_log2(x * (-1.702f * MIC_LOG2E));
      return fmaf(1.702f * x * s, 1.0f - s, s);
    }
    default: return 1.0f;
  }
}

// ---- GEMM epilogue (shared by the bf16-MFMA and f32-MFMA kernels)
struct EpiArgs {
  void* C; int ldc; int c_f32;
  const float* bias; int act;
  void* Zout; int ldz;
  const void* Zin; int dact;
  const void* R; int ldr;
  int accumulate;
  uint32_t drop_thr; uint32_t drop_seed; float drop_scale;
  float alpha;
  int N;
  float* rowstat; int stat_ld; int stat_nvalid;  // bf16 GEMM only: per (row, column tile) softmax partials, see mic_gemm_args
  // LayerNorm folded around the GEMM (bf16 GEMM only, see mic_gemm_args): ln_stats = (sum, sum of squares) of every A row,
  // ln_g[n] = sum_k gamma_k W[n][k]; the epilogue turns acc = x . (gamma o W)^T into LN(x) . W^T.  rowsum2: by-product of a
  // producer GEMM, (sum, sum of squares) of every output row as stored — 2^20 fixed point in int64, so that the atomics of
  // the column tiles add up to the same bits in any order (run-to-run and batch-permutation determinism of generate).
  const long long* ln_stats; const float* ln_g; const float* ln_bias; float ln_inv_d, ln_eps;
  long long* rowsum2;
  // fp8 GEMMs only: C leaves as fp8 bytes under the output tensor's delayed scale (c_q8 = 1 + MIC_E4M3 / MIC_E5M2; ldc in bytes),
  // see Q8Out above — the GELU output of the FFN-in projection (next: the fp8 FFN-out projection) and the dGELU-scaled dX of
  // FFN-out (next: FFN-in's backward GEMMs)
  int c_q8; float* q8_state; float* q8_amax;
};
template <typename T>
__device__ __forceinline__ void epilogue_store(const EpiArgs& e, int m, int n, float v) {
  v *= e.alpha;
  if (e.bias) v += e.bias[n];
  if (e.Zout) ElemT<T>::st((T*)e.Zout + (size_t)m * e.ldz + n, v);
  if (e.act) v = act_fwd(e.act, round_to<T>(v));  // act sees the stored (rounded) pre-activation, as backward will
  if (e.dact) v *= act_bwd(e.dact, ElemT<T>::ld((const T*)e.Zin + (size_t)m * e.ldz + n));
  if (e.drop_thr) v = dropout_keep(e.drop_seed, (uint32_t)m * (uint32_t)e.N + (uint32_t)n, e.drop_thr) ? v * e.drop_scale : 0.0f;
  if (e.R) v += ElemT<T>::ld((const T*)e.R + (size_t)m * e.ldr + n);
  if (e.c_f32) {
    float* c = (float*)e.C + (size_t)m * e.ldc + n;
    if (e.accumulate) v += *c;
    *c = v;
  } else {
    T* c = (T*)e.C + (size_t)m * e.ldc + n;
    if (e.accumulate) v += ElemT<T>::ld(c);
    ElemT<T>::st(c, v);
  }
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));  // native vector: stays in VGPRs (HIP's uint4 class may not)
__device__ __forceinline__ void unpack8(u32x4 u, float* o) {
  o[0] = __uint_as_float(u.x << 16); o[1] = __uint_as_float(u.x & 0xffff0000u);
  o[2] = __uint_as_float(u.y << 16); o[3] = __uint_as_float(u.y & 0xffff0000u);
  o[4] = __uint_as_float(u.z << 16); o[5] = __uint_as_float(u.z & 0xffff0000u);
  o[6] = __uint_as_float(u.w << 16); o[7] = __uint_as_float(u.w & 0xffff0000u);
}
__device__ __forceinline__ bool epilogue_vec_ok(const EpiArgs& e, int cnt) {
  return cnt == 8 && (e.ldc & 7) == 0 && (!e.Zout || (e.ldz & 7) == 0) && (!e.Zin || (e.ldz & 7) == 0) && (!e.R || (e.ldr & 7) == 0);
}
#define MIC_ROWSUM_SCALE 1048576.0f            /* 2^20: |sum of squares| up to 8.8e12 fits an int64 */
#define MIC_ROWSUM_INV_SCALE 9.5367431640625e-7f /* 2^-20 */
// LayerNorm folded around the GEMM: v[i] = acc of x . (gamma o W)^T for row m, columns n .. n+7  ->  LN(x) . W^T + bias'
//   LN(x) . W^T = rstd (x . (gamma o W)^T - mu g) + beta . W^T,  g[n] = sum_k gamma_k W[n][k];  bias' = bias + beta . W^T
__device__ __forceinline__ void ln_fold_apply8(const EpiArgs& e, int m, int n, float* v) {
  const longlong2 s = reinterpret_cast<const longlong2*>(e.ln_stats)[m];
  const float mu = (float)s.x * (MIC_ROWSUM_INV_SCALE * e.ln_inv_d);
  const float rstd = rsqrtf(fmaxf((float)s.y * (MIC_ROWSUM_INV_SCALE * e.ln_inv_d) - mu * mu, 0.0f) + e.ln_eps);
  float g[8], b[8];
  ld8(e.ln_g + n, g);
  ld8(e.ln_bias + n, b);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = fmaf(rstd, v[i] - mu * g[i], b[i]);
}
// bf16 epilogue split in two so a thread can issue the side loads (Zin / residual / accumulate-into-C) of ALL its 8-column
// groups before it starts computing and storing: inside the store loop every such load would cost a full memory latency.
// Two register slots per group: zc = Zin (dact) or the old C (bf16 accumulate) — a launch using both takes the plain path.
__device__ __forceinline__ bool epilogue_pre_ok(const EpiArgs& e) { return !(e.dact && e.accumulate && !e.c_f32); }
__device__ __forceinline__ void epilogue_prefetch8(const EpiArgs& e, int m, int n, u32x4& zc, u32x4& r) {
  if (e.dact) zc = *reinterpret_cast<const u32x4*>((const uint16_t*)e.Zin + (size_t)m * e.ldz + n);
  else if (e.accumulate && !e.c_f32) zc = *reinterpret_cast<const u32x4*>((const uint16_t*)e.C + (size_t)m * e.ldc + n);
  if (e.R) r = *reinterpret_cast<const u32x4*>((const uint16_t*)e.R + (size_t)m * e.ldr + n);
}
template <bool Q8 = false>
__device__ __forceinline__ void epilogue_store8_pre(const EpiArgs& e, int m, int n, float* v, u32x4 zc, u32x4 rq, Q8Ctx* qc = nullptr) {
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] *= e.alpha;
  if (e.bias) {
    float b[8];
    ld8(e.bias + n, b);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += b[i];
  }
  if (e.Zout) st8((uint16_t*)e.Zout + (size_t)m * e.ldz + n, v);
  // the activation kind is tested once per group, not once per element (the compiler does not unswitch these loops itself)
  if (e.act == MIC_ACT_GELU_TANH) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = act_fwd(MIC_ACT_GELU_TANH, round_to<uint16_t>(v[i]));
  } else if (e.act == MIC_ACT_QUICK_GELU) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = act_fwd(MIC_ACT_QUICK_GELU, round_to<uint16_t>(v[i]));
  } else if (e.act) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = act_fwd(e.act, round_to<uint16_t>(v[i]));
  }
  if (e.dact) {
    float z[8];
    unpack8(zc, z);
    if (e.dact == MIC_ACT_GELU_TANH) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] *= act_bwd(MIC_ACT_GELU_TANH, z[i]);
    } else if (e.dact == MIC_ACT_QUICK_GELU) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] *= act_bwd(MIC_ACT_QUICK_GELU, z[i]);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] *= act_bwd(e.dact, z[i]);
    }
  }
  if (e.drop_thr) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      v[i] = dropout_keep(e.drop_seed, (uint32_t)m * (uint32_t)e.N + (uint32_t)(n + i), e.drop_thr) ? v[i] * e.drop_scale : 0.0f;
  }
  if (e.R) {
    float r[8];
    unpack8(rq, r);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += r[i];
  }
  if constexpr (Q8) {
    if (e.c_q8) {  // (host side: no accumulate, not fp32)
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = round_to<uint16_t>(v[i]);
      *reinterpret_cast<uint2*>((uint8_t*)e.C + (size_t)m * e.ldc + n) = q8_pack8(*qc, v, e.c_q8 - 1);
      return;
    }
  }
  if (e.c_f32) {
    float* c = (float*)e.C + (size_t)m * e.ldc + n;
    if (e.accumulate) {
      float o[8];
      ld8(c, o);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] += o[i];
    }
    st8(c, v);
  } else {
    if (e.accumulate) {
      float o[8];
      unpack8(zc, o);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] += o[i];
    }
    st8((uint16_t*)e.C + (size_t)m * e.ldc + n, v);
  }
}

// 8 consecutive columns [n, n+cnt) of row m.  Vector (16-B) path when all 8 are in range and every touched pointer is
// 16-B aligned; scalar fallback otherwise.
template <typename T>
__device__ __forceinline__ void epilogue_store8(const EpiArgs& e, int m, int n, float* v, int cnt) {
  const bool vec = cnt == 8 && (e.ldc & 7) == 0 && (!e.Zout || (e.ldz & 7) == 0) && (!e.Zin || (e.ldz & 7) == 0) &&
                   (!e.R || (e.ldr & 7) == 0);
  if (!vec) {
    for (int i = 0; i < cnt; ++i) epilogue_store<T>(e, m, n + i, v[i]);
    return;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] *= e.alpha;
  if (e.bias) {
    float b[8];
    ld8(e.bias + n, b);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += b[i];
  }
  if (e.Zout) st8((T*)e.Zout + (size_t)m * e.ldz + n, v);
  if (e.act) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = act_fwd(e.act, round_to<T>(v[i]));
  }
  if (e.dact) {
    float z[8];
    ld8((const T*)e.Zin + (size_t)m * e.ldz + n, z);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] *= act_bwd(e.dact, z[i]);
  }
  if (e.drop_thr) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      v[i] = dropout_keep(e.drop_seed, (uint32_t)m * (uint32_t)e.N + (uint32_t)(n + i), e.drop_thr) ? v[i] * e.drop_scale : 0.0f;
  }
  if (e.R) {
    float r[8];
    ld8((const T*)e.R + (size_t)m * e.ldr + n, r);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += r[i];
  }
  if (e.c_f32) {
    float* c = (float*)e.C + (size_t)m * e.ldc + n;
    if (e.accumulate) {
      float o[8];
      ld8(c, o);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] += o[i];
    }
    st8(c, v);
  } else {
    T* c = (T*)e.C + (size_t)m * e.ldc + n;
    if (e.accumulate) {
      float o[8];
      ld8(c, o);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] += o[i];
    }
    st8(c, v);
  }
}
