// attention.hip — attention cores for the three tiny tile shapes of this model (ViT 50x50, decoder causal 64x64,
// cross 64x50; head_dim 64).  Occupancy comes from batching (B*H workgroups), not from tiling: one wave owns one
// (batch, head) problem, stages Q/K/V (and dO) as 64x64 tiles in LDS (zero padded), and runs every contraction on the
// matrix cores:  S^T = K Q^T  (lane <-> query, so the softmax row reduction is in-register + one cross-half
// shuffle),  O = P V,  and in backward  dP^T = V dO^T, dQ = dS K, dK = dS^T Q, dV = P^T dO  — transposed operands
// come from the same LDS tiles through ds_read_b64_tr_b16.  bf16: v_mfma_f32_32x32x16_bf16; f32 (parity mode):
// v_mfma_f32_32x32x2_f32.
#include "common.h"
#include <type_traits>

#define SCALE 0.125f  // 1/sqrt(64): q is scaled before q.k^T (power of two: exact either side of the product)

// ------------------------------------------------------------------ 64x64 LDS tiles
template <typename T> struct Tile;

template <> struct Tile<uint16_t> {
  static constexpr int BYTES = 64 * 64 * 2;
  // 16-B chunk index XOR-swizzled by row.  The same tile is read two ways: ds_read_b128 along k (lane <-> row: the 16-lane
  // groups {0-3,12-15,20-27}/... cover 8 distinct row PAIRS, so any bijection of the pair index p = (row>>1)&7 is conflict-
  // free) and ds_read_b64_tr_b16 across 4 consecutive rows x 4 chunks (rows k0,k0+1 vs k0+2,k0+3 must land in different
  // aligned blocks of 4 chunks: bit 2 of the swizzle = bit 0 of p).  sw(row) = ((p & 1) << 2) | (p >> 1) satisfies both;
  // row & 7 (the first version) measured 28 % / 42 % of LDS cycles as bank conflicts in the fwd / bwd kernels.
  static __device__ __forceinline__ int sw(int row) { const int p = (row >> 1) & 7; return ((p & 1) << 2) | (p >> 1); }
  // element (row, col) -> byte offset
  static __device__ __forceinline__ int off(int row, int col) { return row * 128 + ((((col >> 3) ^ sw(row)) << 4) | ((col & 7) << 1)); }
  // stage rows [0,nrows) x 64 cols from src (row stride ld); rows >= nrows are zero
  template <int NT = 64>
  static __device__ __forceinline__ void stage(char* t, const uint16_t* src, int ld, int nrows, int lane) {
#pragma unroll
    for (int it = 0; it < 512 / NT; ++it) {
      const int idx = it * NT + lane, row = idx >> 3, c = idx & 7;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (row < nrows) v = *reinterpret_cast<const uint4*>(src + (size_t)row * ld + c * 8);
      *reinterpret_cast<uint4*>(t + row * 128 + ((c ^ sw(row)) << 4)) = v;
    }
  }
  template <bool KM>
  static __device__ __forceinline__ bf16x8 frag(const char* t, int xb, int kk, int lane) {
    if (!KM) {
      const int row = xb + (lane & 31), kc = kk * 2 + (lane >> 5);
      return *reinterpret_cast<const bf16x8*>(t + row * 128 + ((kc ^ sw(row)) << 4));
    } else {
      const int g = lane >> 4, p = lane & 15;
      const int x = xb + 16 * (g & 1) + (p & 3) * 4;
      const int k = kk * 16 + 8 * (g >> 1) + (p >> 2);
      s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, t + off(k, x)));
      s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, t + off(k + 4, x)));
      s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      return __builtin_bit_cast(bf16x8, r);
    }
  }
  // acc(32x32) += sum_k A(a0+i, k) * B(k, b0+j) over k in [0,64).  AKM: A tile stored [k][i]; BKM: B tile stored [k][j]
  // (non-KM B tile is stored [j][k]).
  template <bool AKM, bool BKM>
  static __device__ __forceinline__ void mma(f32x16& acc, const char* At, int a0, const char* Bt, int b0, int lane) {
    // the four k-steps accumulate into ONE block: all fragment reads go out first, pinned in front of the MFMAs (the scheduler
    // otherwise sinks each pair of reads next to its MFMA and waits an LDS round trip per k-step)
    bf16x8 a[4], b[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      a[kk] = frag<AKM>(At, a0, kk, lane);
      b[kk] = frag<BKM>(Bt, b0, kk, lane);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kk], b[kk], acc, 0, 0, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, ((AKM ? 2 : 1) + (BKM ? 2 : 1)) * 4, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
  }
  static __device__ __forceinline__ void store4(char* t, int row, int col0, const float* v) {
    uint2 o;
    o.x = f2bf_pk(v[0], v[1]);
    o.y = f2bf_pk(v[2], v[3]);
    *reinterpret_cast<uint2*>(t + off(row, col0)) = o;
  }
};

template <> struct Tile<float> {
  static constexpr int BYTES = 64 * 64 * 4;
  template <int NT = 64>
  static __device__ __forceinline__ void stage(char* t, const float* src, int ld, int nrows, int lane) {
    float* f = reinterpret_cast<float*>(t);
#pragma unroll
    for (int it = 0; it < 1024 / NT; ++it) {
      const int idx = it * NT + lane, row = idx >> 4, c = idx & 15;
      float4 v = make_float4(0, 0, 0, 0);
      if (row < nrows) v = *reinterpret_cast<const float4*>(src + (size_t)row * ld + c * 4);
      *reinterpret_cast<float4*>(f + row * 64 + c * 4) = v;
    }
  }
  template <bool AKM, bool BKM>
  static __device__ __forceinline__ void mma(f32x16& acc, const char* At, int a0, const char* Bt, int b0, int lane) {
    const float* A = reinterpret_cast<const float*>(At);
    const float* B = reinterpret_cast<const float*>(Bt);
    const int i = lane & 31, h = lane >> 5;
#pragma unroll 8
    for (int s = 0; s < 32; ++s) {
      const int k = 2 * s + h;
      const float a = AKM ? A[k * 64 + a0 + i] : A[(a0 + i) * 64 + k];
      const float b = BKM ? B[k * 64 + b0 + i] : B[(b0 + i) * 64 + k];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
  }
  static __device__ __forceinline__ void store4(char* t, int row, int col0, const float* v) {
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(t) + row * 64 + col0) = make_float4(v[0], v[1], v[2], v[3]);
  }
};

__device__ __forceinline__ void zero16(f32x16& a) {
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = 0.f;
}
// accumulator element r of a 32x32 block: row (r&3) + 8*(r>>2) + 4*(lane>>5), col lane&31
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

__device__ __forceinline__ unsigned long long load_key_mask(const int32_t* key_mask, int b, int Tk, int lane) {
  if (!key_mask) return ~0ull;
  const int mv = lane < Tk ? key_mask[b * Tk + lane] : 0;
  return __ballot(mv != 0);
}

// Variable-length ("packed") rows: sequence b owns the q rows [q_off[b], q_off[b] + q_len[b]) of a packed [sum q_len][ld] matrix
// (no rows for padded positions at all).  kv_packed: keys / values are the same packed rows (self-attention) — otherwise every
// sequence has its Tk dense rows [b*Tk, (b+1)*Tk) (cross-attention over the encoder states).  q_off == nullptr: dense rows.
struct PackedRows { const int32_t* q_off; const int32_t* q_len; int kv_packed; };

// ------------------------------------------------------------------ forward
// 256 threads = 4 waves share one (batch, head) problem.  Phase 1: wave w owns the (ib, jb) = (w>>1, w&1) quadrant of
// S^T = K Q^T (lane <-> query): quadrant max -> LDS, barrier, p = exp(s - rowmax) written UNnormalised to the P tile,
// quadrant row sums -> LDS.  Phase 2: wave w owns output block (ib, db) of O = P V and scales each row by 1/rowsum.
template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_kernel(int H, int Tq_max, int Tk_max, const T* __restrict__ q, int ldq,
                                                       const T* __restrict__ k, int ldk, const T* __restrict__ v, int ldv,
                                                       T* __restrict__ out, int ldo, const int32_t* __restrict__ key_mask,
                                                       int causal, float* __restrict__ lse_out, PackedRows pk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TB = Tile<T>::BYTES;
  char* Qt = smem;
  char* Kt = smem + TB;
  char* Vt = smem + 2 * TB;
  char* Pt = smem + 3 * TB;
  float* pmax = reinterpret_cast<float*>(smem + 4 * TB);  // [2 jb][64 queries]
  float* psum = pmax + 128;                               // [2 jb][64 queries]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  int Tq = Tq_max, Tk = Tk_max;
  size_t q0 = (size_t)b * Tq_max, k0 = (size_t)b * Tk_max;
  if (pk.q_off) {
    q0 = pk.q_off[b]; Tq = pk.q_len[b];
    if (pk.kv_packed) { k0 = q0; Tk = Tq; }
  }
  Tile<T>::template stage<256>(Qt, q + q0 * ldq + h * 64, ldq, Tq, tid);
  Tile<T>::template stage<256>(Kt, k + k0 * ldk + h * 64, ldk, Tk, tid);
  Tile<T>::template stage<256>(Vt, v + k0 * ldv + h * 64, ldv, Tk, tid);
  const unsigned long long km = load_key_mask(key_mask, b, Tk, lane);
  __syncthreads();
  const int nib = (Tq + 31) >> 5, njb = (Tk + 31) >> 5;
  const int ib = wave >> 1, jb = wave & 1;
  const int i = ib * 32 + (lane & 31);
  f32x16 s;
  zero16(s);
  const bool live = ib < nib && jb < njb;
  if (live) Tile<T>::template mma<false, false>(s, Kt, jb * 32, Qt, ib * 32, lane);
  float m = -INFINITY;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int j = jb * 32 + acc_row(r, lane);
    const bool ok = live && j < Tk && (!causal || j <= i) && ((km >> j) & 1ull);
    const float x = ok ? s[r] * SCALE : -INFINITY;
    s[r] = x;
    m = fmaxf(m, x);
  }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  if (lane < 32) pmax[jb * 64 + i] = m;
  __syncthreads();
  const float mrow = fmaxf(pmax[i], pmax[64 + i]);
  float l = 0.f;
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    float pv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float p = __expf(s[g4 * 4 + e] - mrow);
      pv[e] = p;
      l += p;
    }
    Tile<T>::store4(Pt, i, jb * 32 + 8 * g4 + 4 * (lane >> 5), pv);
  }
  l += __shfl_xor(l, 32, 64);
  if (lane < 32) psum[jb * 64 + i] = l;
  __syncthreads();
  if (jb == 0 && lane < 32 && i < Tq && lse_out) lse_out[((size_t)b * H + h) * Tq_max + i] = mrow + logf(psum[i] + psum[64 + i]);
  {
    const int db = wave & 1;  // output block (ib, db)
    if (ib < nib) {
      f32x16 o;
      zero16(o);
      Tile<T>::template mma<false, true>(o, Pt, ib * 32, Vt, db * 32, lane);
      const int d = db * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qi = ib * 32 + acc_row(r, lane);
        if (qi < Tq) ElemT<T>::st(out + (q0 + qi) * ldo + h * 64 + d, o[r] / (psum[qi] + psum[64 + qi]));
      }
    }
  }
}

// ------------------------------------------------------------------ backward
// 256 threads = 4 waves (one per SIMD) share one (batch, head) problem: same LDS footprint as a single-wave design but
// 4x the waves per CU and 1/4 of the serial MFMA/LDS chain per wave.  Phase 1: wave w owns the (ib, jb) = (w>>1, w&1)
// quadrant of S^T / dP^T and writes its quadrant of the P and dS tiles.  Phase 2: wave w owns output block
// (w>>1, w&1) of dQ, dK and dV.
// Q8 (bf16 storage): dQ / dK / dV leave as e5m2 BYTES under their tensors' delayed scales (common.h: Q8Out; dq / dk / dv then point
// at byte matrices, lddq / lddk / lddv count bytes) — the dy operands of the fp8 q/k/v projections' backward GEMMs.  q8q: dQ's
// tensor; q8kv: the tensor dK and dV belong to (self-attention: the same fused [rows][3d] gradient as dQ).
template <typename T, bool Q8 = false>
__global__ __launch_bounds__(256) void attn_bwd_kernel(int H, int Tq_max, int Tk_max, const T* __restrict__ q, int ldq,
                                                       const T* __restrict__ k, int ldk, const T* __restrict__ v, int ldv,
                                                       const T* __restrict__ out, int ldo, const T* __restrict__ dout, int lddo,
                                                       const float* __restrict__ lse_in, const int32_t* __restrict__ key_mask,
                                                       int causal, T* __restrict__ dq, int lddq, T* __restrict__ dk, int lddk,
                                                       T* __restrict__ dv, int lddv, PackedRows pk, Q8Out q8q, Q8Out q8kv) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TB = Tile<T>::BYTES;
  // five tiles (40 KiB in bf16: FOUR workgroups per CU, so the 1024 (batch, head) problems of a decoder layer at batch 64 are one
  // residency round instead of 768 + 256): V is dead once dP^T = V dO^T is in the accumulators, so the P tile takes its place
  // (one extra barrier), and the 512 B of per-query lse / delta live in the head of the dS tile until dS itself is written
  char* Qt = smem;
  char* Kt = smem + TB;
  char* Vt = smem + 2 * TB;
  char* dOt = smem + 3 * TB;
  char* dSt = smem + 4 * TB;
  char* Pt = Vt;
  float* lse_s = reinterpret_cast<float*>(dSt);
  float* delta_s = lse_s + 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  int Tq = Tq_max, Tk = Tk_max;
  size_t q0 = (size_t)b * Tq_max, k0 = (size_t)b * Tk_max;
  if (pk.q_off) {
    q0 = pk.q_off[b]; Tq = pk.q_len[b];
    if (pk.kv_packed) { k0 = q0; Tk = Tq; }
  }
  Tile<T>::template stage<256>(Qt, q + q0 * ldq + h * 64, ldq, Tq, tid);
  Tile<T>::template stage<256>(Kt, k + k0 * ldk + h * 64, ldk, Tk, tid);
  Tile<T>::template stage<256>(Vt, v + k0 * ldv + h * 64, ldv, Tk, tid);
  Tile<T>::template stage<256>(dOt, dout + q0 * lddo + h * 64, lddo, Tq, tid);
  {
    // delta_i = dO_i . O_i : thread t covers 16 of the 64 dims of row t>>2
    const int row = tid >> 2, part = tid & 3;
    float dl = 0.f;
    if (row < Tq) {
      const T* orow = out + (q0 + row) * ldo + h * 64 + part * 16;
      const T* drow = dout + (q0 + row) * lddo + h * 64 + part * 16;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        float a[8], g[8];
        ld8(orow + c * 8, a);
        ld8(drow + c * 8, g);
#pragma unroll
        for (int e = 0; e < 8; ++e) dl += a[e] * g[e];
      }
    }
    dl += __shfl_xor(dl, 1, 64);
    dl += __shfl_xor(dl, 2, 64);
    if (part == 0) delta_s[row] = dl;
    if (tid < 64) lse_s[tid] = tid < Tq ? lse_in[((size_t)b * H + h) * Tq_max + tid] : 0.f;
  }
  const unsigned long long km = load_key_mask(key_mask, b, Tk, lane);
  __syncthreads();
  const int nib = (Tq + 31) >> 5, njb = (Tk + 31) >> 5;
  {
    // P and dS tiles (rows = queries); rows >= Tq and keys >= Tk are exact zeros
    const int ib = wave >> 1, jb = wave & 1;
    const int i = ib * 32 + (lane & 31);
    const float lse_i = lse_s[i], delta_i = delta_s[i];
    f32x16 sacc, dp;
    zero16(sacc); zero16(dp);
    const bool live = ib < nib && jb < njb;
    if (live) {
      Tile<T>::template mma<false, false>(sacc, Kt, jb * 32, Qt, ib * 32, lane);
      Tile<T>::template mma<false, false>(dp, Vt, jb * 32, dOt, ib * 32, lane);
    }
    __syncthreads();  // every wave has read V (-> P) and lse / delta (-> dS) before those regions are overwritten
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      float pv[4], dsv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = g4 * 4 + e;
        const int j = jb * 32 + acc_row(r, lane);
        const bool ok = live && i < Tq && j < Tk && (!causal || j <= i) && ((km >> j) & 1ull);
        const float p = ok ? __expf(sacc[r] * SCALE - lse_i) : 0.f;
        pv[e] = p;
        dsv[e] = p * (dp[r] - delta_i) * SCALE;
      }
      Tile<T>::store4(Pt, i, jb * 32 + 8 * g4 + 4 * (lane >> 5), pv);
      Tile<T>::store4(dSt, i, jb * 32 + 8 * g4 + 4 * (lane >> 5), dsv);
    }
  }
  __syncthreads();
  {
    const int xb = wave >> 1, db = wave & 1;
    const int d = db * 32 + (lane & 31);
    Q8Ctx cq, ckv;
    if constexpr (Q8) {
      cq = q8_begin(q8q, blockIdx.x == 0 && tid == 0);
      ckv = q8_begin(q8kv, blockIdx.x == 0 && tid == 0);
    }
    if (xb < nib) {
      f32x16 a;
      zero16(a);
      Tile<T>::template mma<false, true>(a, dSt, xb * 32, Kt, db * 32, lane);  // dQ = dS K
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = xb * 32 + acc_row(r, lane);
        if (i < Tq) {
          if constexpr (Q8) reinterpret_cast<uint8_t*>(dq)[(q0 + i) * lddq + h * 64 + d] = q8_pack1(cq, round_to<T>(a[r]), q8q.fmt);
          else ElemT<T>::st(dq + (q0 + i) * lddq + h * 64 + d, a[r]);
        }
      }
    }
    if (xb < njb) {
      f32x16 a, c;
      zero16(a); zero16(c);
      Tile<T>::template mma<true, true>(a, dSt, xb * 32, Qt, db * 32, lane);   // dK = dS^T Q
      Tile<T>::template mma<true, true>(c, Pt, xb * 32, dOt, db * 32, lane);   // dV = P^T dO
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = xb * 32 + acc_row(r, lane);
        if (j < Tk) {
          if constexpr (Q8) {
            reinterpret_cast<uint8_t*>(dk)[(k0 + j) * lddk + h * 64 + d] = q8_pack1(ckv, round_to<T>(a[r]), q8kv.fmt);
            reinterpret_cast<uint8_t*>(dv)[(k0 + j) * lddv + h * 64 + d] = q8_pack1(ckv, round_to<T>(c[r]), q8kv.fmt);
          } else {
            ElemT<T>::st(dk + (k0 + j) * lddk + h * 64 + d, a[r]);
            ElemT<T>::st(dv + (k0 + j) * lddv + h * 64 + d, c[r]);
          }
        }
      }
    }
    if constexpr (Q8) {
      // (when dQ and dK / dV are one tensor both contexts carry the same scale: two atomics into the same table)
      q8_end_wave(q8q, cq, blockIdx.x * 4 + wave);
      q8_end_wave(q8kv, ckv, blockIdx.x * 4 + wave);
    }
  }
}

// ------------------------------------------------------------------ sequences longer than one 64x64 tile
// (the reference's README names seq_len 128 as its next step; a ViT with more than 64 patch tokens lands here too)
// Forward: one workgroup per (batch, head, 64-query block) walks the 64-key blocks with the online-softmax recurrence:
// running row max / row sum in LDS, the four 32x32 output blocks in the waves' accumulators, rescaled by
// alpha = exp(m_old - m_new) before each block's P V is added.  Per block the arithmetic is the single-tile kernel's.
__device__ __forceinline__ unsigned long long load_key_mask_blk(const int32_t* key_mask, int b, int Tk, int k0, int lane) {
  if (!key_mask) return ~0ull;
  const int mv = k0 + lane < Tk ? key_mask[b * Tk + k0 + lane] : 0;
  return __ballot(mv != 0);
}

template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_tiled_kernel(int H, int Tq, int Tk, const T* __restrict__ q, int ldq,
                                                             const T* __restrict__ k, int ldk, const T* __restrict__ v, int ldv,
                                                             T* __restrict__ out, int ldo, const int32_t* __restrict__ key_mask,
                                                             int causal, float* __restrict__ lse_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TB = Tile<T>::BYTES;
  char* Qt = smem;
  char* Kt = smem + TB;
  char* Vt = smem + 2 * TB;
  char* Pt = smem + 3 * TB;
  float* pmax = reinterpret_cast<float*>(smem + 4 * TB);  // [2 jb][64 queries]
  float* psum = pmax + 128;                               // [2 jb][64 queries]
  float* m_run = psum + 128;                              // [64]
  float* l_run = m_run + 64;                              // [64]
  float* alpha_s = l_run + 64;                            // [64]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nqb = (Tq + 63) >> 6;
  const int qb = blockIdx.x % nqb, bh = blockIdx.x / nqb;
  const int b = bh / H, h = bh % H;
  const int q0 = qb * 64, nq = min(64, Tq - q0);
  Tile<T>::template stage<256>(Qt, q + ((size_t)b * Tq + q0) * ldq + h * 64, ldq, nq, tid);
  if (tid < 64) { m_run[tid] = -INFINITY; l_run[tid] = 0.f; }
  const int ib = wave >> 1, jb = wave & 1, db = wave & 1;
  const int i = ib * 32 + (lane & 31);  // this lane's query (local) in the score phase
  f32x16 o;
  zero16(o);
  int nkb = (Tk + 63) >> 6;
  if (causal) nkb = min(nkb, qb + 1);
  for (int kb = 0; kb < nkb; ++kb) {
    const int k0 = kb * 64, nk = min(64, Tk - k0);
    __syncthreads();  // the previous block's P V has read Kt / Vt / Pt
    Tile<T>::template stage<256>(Kt, k + ((size_t)b * Tk + k0) * ldk + h * 64, ldk, nk, tid);
    Tile<T>::template stage<256>(Vt, v + ((size_t)b * Tk + k0) * ldv + h * 64, ldv, nk, tid);
    const unsigned long long km = load_key_mask_blk(key_mask, b, Tk, k0, lane);
    __syncthreads();
    f32x16 s;
    zero16(s);
    Tile<T>::template mma<false, false>(s, Kt, jb * 32, Qt, ib * 32, lane);
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = jb * 32 + acc_row(r, lane);
      const bool ok = j < nk && (!causal || k0 + j <= q0 + i) && ((km >> j) & 1ull);
      const float x = ok ? s[r] * SCALE : -INFINITY;
      s[r] = x;
      m = fmaxf(m, x);
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    if (lane < 32) pmax[jb * 64 + i] = m;
    __syncthreads();
    const float m_new = fmaxf(m_run[i], fmaxf(pmax[i], pmax[64 + i]));
    const float m_safe = m_new == -INFINITY ? 0.f : m_new;  // a row with no admissible key so far: p = 0, not NaN
    float l = 0.f;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      float pv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float p = __expf(s[g4 * 4 + e] - m_safe);
        pv[e] = p;
        l += p;
      }
      Tile<T>::store4(Pt, i, jb * 32 + 8 * g4 + 4 * (lane >> 5), pv);
    }
    l += __shfl_xor(l, 32, 64);
    if (lane < 32) psum[jb * 64 + i] = l;
    __syncthreads();
    if (tid < 64) {
      const float mo = m_run[tid], mn = fmaxf(mo, fmaxf(pmax[tid], pmax[64 + tid]));
      const float al = mo == -INFINITY ? 0.f : __expf(mo - mn);
      alpha_s[tid] = al;
      l_run[tid] = l_run[tid] * al + psum[tid] + psum[64 + tid];
      m_run[tid] = mn;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] *= alpha_s[ib * 32 + acc_row(r, lane)];
    Tile<T>::template mma<false, true>(o, Pt, ib * 32, Vt, db * 32, lane);
  }
  __syncthreads();
  if (tid < 64 && tid < nq && lse_out) lse_out[((size_t)b * H + h) * Tq + q0 + tid] = m_run[tid] + logf(l_run[tid]);
  const int d = db * 32 + (lane & 31);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int qi = ib * 32 + acc_row(r, lane);
    if (qi < nq) ElemT<T>::st(out + ((size_t)b * Tq + q0 + qi) * ldo + h * 64 + d, o[r] / l_run[qi]);
  }
}

// Backward: with the saved log-sum-exp every P block is exp(s - lse) directly, so the two reductions are independent:
//   MODE 0: one workgroup per (batch, head, 64-query block) walks the key blocks and accumulates dQ = sum_j dS_ij K_j;
//   MODE 1: one workgroup per (batch, head, 64-key block) walks the query blocks and accumulates dK = sum_i dS_ij^T Q_i,
//           dV = sum_i P_ij^T dO_i.
// No atomics, deterministic; the score blocks are computed twice (the contraction over the other index is the expensive part).
template <typename T, int MODE>
__global__ __launch_bounds__(256) void attn_bwd_tiled_kernel(int H, int Tq, int Tk, const T* __restrict__ q, int ldq,
                                                             const T* __restrict__ k, int ldk, const T* __restrict__ v, int ldv,
                                                             const T* __restrict__ out, int ldo, const T* __restrict__ dout, int lddo,
                                                             const float* __restrict__ lse_in, const int32_t* __restrict__ key_mask,
                                                             int causal, T* __restrict__ dq, int lddq, T* __restrict__ dk, int lddk,
                                                             T* __restrict__ dv, int lddv) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TB = Tile<T>::BYTES;
  char* Qt = smem;
  char* Kt = smem + TB;
  char* Vt = smem + 2 * TB;
  char* dOt = smem + 3 * TB;
  char* Pt = smem + 4 * TB;
  char* dSt = smem + 5 * TB;
  float* lse_s = reinterpret_cast<float*>(smem + 6 * TB);
  float* delta_s = lse_s + 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nqb = (Tq + 63) >> 6, nkb = (Tk + 63) >> 6;
  const int nfix = MODE == 0 ? nqb : nkb;
  const int fix = blockIdx.x % nfix, bh = blockIdx.x / nfix;
  const int b = bh / H, h = bh % H;

  auto stage_q = [&](int qb) {
    const int q0 = qb * 64, nq = min(64, Tq - q0);
    Tile<T>::template stage<256>(Qt, q + ((size_t)b * Tq + q0) * ldq + h * 64, ldq, nq, tid);
    Tile<T>::template stage<256>(dOt, dout + ((size_t)b * Tq + q0) * lddo + h * 64, lddo, nq, tid);
    const int row = tid >> 2, part = tid & 3;  // delta_i = dO_i . O_i : thread t covers 16 of the 64 dims of row t>>2
    float dl = 0.f;
    if (row < nq) {
      const T* orow = out + ((size_t)b * Tq + q0 + row) * ldo + h * 64 + part * 16;
      const T* drow = dout + ((size_t)b * Tq + q0 + row) * lddo + h * 64 + part * 16;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        float a[8], g[8];
        ld8(orow + c * 8, a);
        ld8(drow + c * 8, g);
#pragma unroll
        for (int e = 0; e < 8; ++e) dl += a[e] * g[e];
      }
    }
    dl += __shfl_xor(dl, 1, 64);
    dl += __shfl_xor(dl, 2, 64);
    if (part == 0) delta_s[row] = dl;
    if (tid < 64) lse_s[tid] = tid < nq ? lse_in[((size_t)b * H + h) * Tq + q0 + tid] : 0.f;
  };
  auto stage_k = [&](int kb) {
    const int k0 = kb * 64, nk = min(64, Tk - k0);
    Tile<T>::template stage<256>(Kt, k + ((size_t)b * Tk + k0) * ldk + h * 64, ldk, nk, tid);
    Tile<T>::template stage<256>(Vt, v + ((size_t)b * Tk + k0) * ldv + h * 64, ldv, nk, tid);
  };
  // P and dS blocks of the (qb, kb) pair into LDS (rows = queries); inadmissible pairs are exact zeros
  auto scores = [&](int qb, int kb, unsigned long long km) {
    const int q0 = qb * 64, k0 = kb * 64, nq = min(64, Tq - q0), nk = min(64, Tk - k0);
    const int ib = wave >> 1, jb = wave & 1;
    const int i = ib * 32 + (lane & 31);
    const float lse_i = lse_s[i], delta_i = delta_s[i];
    f32x16 sacc, dp;
    zero16(sacc); zero16(dp);
    Tile<T>::template mma<false, false>(sacc, Kt, jb * 32, Qt, ib * 32, lane);
    Tile<T>::template mma<false, false>(dp, Vt, jb * 32, dOt, ib * 32, lane);
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      float pv[4], dsv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = g4 * 4 + e;
        const int j = jb * 32 + acc_row(r, lane);
        const bool ok = i < nq && j < nk && (!causal || k0 + j <= q0 + i) && ((km >> j) & 1ull);
        const float p = ok ? __expf(sacc[r] * SCALE - lse_i) : 0.f;
        pv[e] = p;
        dsv[e] = p * (dp[r] - delta_i) * SCALE;
      }
      if (MODE == 1) Tile<T>::store4(Pt, i, jb * 32 + 8 * g4 + 4 * (lane >> 5), pv);
      Tile<T>::store4(dSt, i, jb * 32 + 8 * g4 + 4 * (lane >> 5), dsv);
    }
  };
  const int xb = wave >> 1, db = wave & 1;
  const int d = db * 32 + (lane & 31);
  if (MODE == 0) {
    const int qb = fix, q0 = qb * 64, nq = min(64, Tq - q0);
    stage_q(qb);
    f32x16 a;
    zero16(a);
    const int jend = causal ? min(nkb, qb + 1) : nkb;
    for (int kb = 0; kb < jend; ++kb) {
      __syncthreads();
      stage_k(kb);
      const unsigned long long km = load_key_mask_blk(key_mask, b, Tk, kb * 64, lane);
      __syncthreads();
      scores(qb, kb, km);
      __syncthreads();
      Tile<T>::template mma<false, true>(a, dSt, xb * 32, Kt, db * 32, lane);  // dQ += dS K
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = xb * 32 + acc_row(r, lane);
      if (i < nq) ElemT<T>::st(dq + ((size_t)b * Tq + q0 + i) * lddq + h * 64 + d, a[r]);
    }
  } else {
    const int kb = fix, k0 = kb * 64, nk = min(64, Tk - k0);
    stage_k(kb);
    const unsigned long long km = load_key_mask_blk(key_mask, b, Tk, k0, lane);
    f32x16 a, c;
    zero16(a); zero16(c);
    for (int qb = causal ? kb : 0; qb < nqb; ++qb) {
      __syncthreads();
      stage_q(qb);
      __syncthreads();
      scores(qb, kb, km);
      __syncthreads();
      Tile<T>::template mma<true, true>(a, dSt, xb * 32, Qt, db * 32, lane);   // dK += dS^T Q
      Tile<T>::template mma<true, true>(c, Pt, xb * 32, dOt, db * 32, lane);   // dV += P^T dO
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = xb * 32 + acc_row(r, lane);
      if (j < nk) {
        ElemT<T>::st(dk + ((size_t)b * Tk + k0 + j) * lddk + h * 64 + d, a[r]);
        ElemT<T>::st(dv + ((size_t)b * Tk + k0 + j) * lddv + h * 64 + d, c[r]);
      }
    }
  }
}

template <typename K>
static int set_lds(K kernel, size_t bytes) {
  if (bytes > 65536) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) { mic_set_error("hipFuncSetAttribute(%zu): %s", bytes, hipGetErrorString(e)); return MIC_ELAUNCH; }
  }
  return MIC_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// The attention WEIGHTS themselves (`output_attentions=True`, modeling:499-510): the fused cores above never write them, this
// diagnostic kernel does — one wave per (sequence, head, query): softmax over the keys of q.k / sqrt(64) with the same masks,
// fp32 arithmetic on the stored q / k.  Not on any hot path.
template <typename T>
__global__ __launch_bounds__(256) void attn_probs_kernel(int n_rows, int H, int Tq, int Tk, const T* __restrict__ q, int ldq,
                                                         const T* __restrict__ k, int ldk, const int32_t* __restrict__ key_mask,
                                                         int causal, float* __restrict__ out) {
  __shared__ float qs[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + w;
  if (r >= n_rows) return;  // (no block-wide barrier below: the waves are independent)
  const int t = r % Tq, h = (r / Tq) % H, b = r / (Tq * H);
  qs[w][lane] = ElemT<T>::ld(q + (size_t)(b * Tq + t) * ldq + h * 64 + lane) * 0.125f;
  __builtin_amdgcn_wave_barrier();
  float sc[16];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int sidx = i * 64 + lane;
    sc[i] = -INFINITY;
    if (sidx < Tk) {
      const bool ok = !(causal && sidx > t) && !(key_mask && key_mask[b * Tk + sidx] == 0);
      if (ok) {
        const T* kr = k + (size_t)(b * Tk + sidx) * ldk + h * 64;
        float a = 0.0f;
        for (int d = 0; d < 64; ++d) a = fmaf(qs[w][d], ElemT<T>::ld(kr + d), a);
        sc[i] = a;
      }
    }
    mx = fmaxf(mx, sc[i]);
  }
  mx = wave_max(mx);
  float sum = 0.0f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    sc[i] = sc[i] > -INFINITY ? __expf(sc[i] - mx) : 0.0f;
    sum += sc[i];
  }
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int sidx = i * 64 + lane;
    if (sidx < Tk) out[(size_t)r * Tk + sidx] = sc[i] * inv;
  }
}

extern "C" int mic_attn_probs(int dtype, int B, int H, int Tq, int Tk, const void* q, int ldq, const void* k, int ldk,
                              const int32_t* key_mask, int causal, float* out, void* stream) {
  MIC_CHECK(B > 0 && H > 0 && Tq > 0 && Tk > 0 && Tk <= 1024, "mic_attn_probs: bad shape B=%d H=%d Tq=%d Tk=%d (Tk <= 1024)", B, H, Tq, Tk);
  MIC_CHECK(q && k && out, "mic_attn_probs: null pointer");
  const int n_rows = B * H * Tq;
  dim3 grid((n_rows + 3) / 4), block(256);
  if (dtype == MIC_BF16)
    hipLaunchKernelGGL(attn_probs_kernel<uint16_t>, grid, block, 0, (hipStream_t)stream, n_rows, H, Tq, Tk, (const uint16_t*)q, ldq, (const uint16_t*)k, ldk, key_mask, causal, out);
  else if (dtype == MIC_F32)
    hipLaunchKernelGGL(attn_probs_kernel<float>, grid, block, 0, (hipStream_t)stream, n_rows, H, Tq, Tk, (const float*)q, ldq, (const float*)k, ldk, key_mask, causal, out);
  else MIC_CHECK(false, "mic_attn_probs: bad dtype");
  MIC_LAUNCH_CHECK();
  return MIC_OK;
}

extern "C" int mic_attn_fwd(int dtype, int B, int H, int Tq, int Tk, const void* q, int ldq, const void* k, int ldk,
                            const void* v, int ldv, void* out, int ldo, const int32_t* key_mask, int causal, float* lse,
                            void* stream) {
  MIC_CHECK(B > 0 && H > 0 && Tq > 0 && Tk > 0, "mic_attn_fwd: bad shape B=%d H=%d Tq=%d Tk=%d", B, H, Tq, Tk);
  MIC_CHECK(q && k && v && out, "mic_attn_fwd: null pointer");
  const int align = dtype == MIC_BF16 ? 8 : 4;
  MIC_CHECK(ldq % align == 0 && ldk % align == 0 && ldv % align == 0, "mic_attn_fwd: row strides must keep 16-B alignment");
  if (Tq > 64 || Tk > 64) {  // online-softmax walk over 64-key blocks
    dim3 grid(B * H * ((Tq + 63) / 64)), block(256);
    if (dtype == MIC_BF16) {
      hipLaunchKernelGGL(attn_fwd_tiled_kernel<uint16_t>, grid, block, 4 * Tile<uint16_t>::BYTES + 2048, (hipStream_t)stream, H, Tq, Tk, (const uint16_t*)q, ldq, (const uint16_t*)k, ldk, (const uint16_t*)v, ldv, (uint16_t*)out, ldo, key_mask, causal, lse);
    } else if (dtype == MIC_F32) {
      const size_t lds = 4 * Tile<float>::BYTES + 2048;
      if (int rc = set_lds(attn_fwd_tiled_kernel<float>, lds)) return rc;
      hipLaunchKernelGGL(attn_fwd_tiled_kernel<float>, grid, block, lds, (hipStream_t)stream, H, Tq, Tk, (const float*)q, ldq, (const float*)k, ldk, (const float*)v, ldv, (float*)out, ldo, key_mask, causal, lse);
    } else MIC_CHECK(false, "mic_attn_fwd: bad dtype");
    MIC_LAUNCH_CHECK();
    return MIC_OK;
  }
  dim3 grid(B * H), block(256);
  if (dtype == MIC_BF16) {
    hipLaunchKernelGGL(attn_fwd_kernel<uint16_t>, grid, block, 4 * Tile<uint16_t>::BYTES + 1024, (hipStream_t)stream, H, Tq, Tk, (const uint16_t*)q, ldq, (const uint16_t*)k, ldk, (const uint16_t*)v, ldv, (uint16_t*)out, ldo, key_mask, causal, lse, PackedRows{nullptr, nullptr, 0});
  } else if (dtype == MIC_F32) {
    const size_t lds = 4 * Tile<float>::BYTES + 1024;
    if (int rc = set_lds(attn_fwd_kernel<float>, lds)) return rc;
    hipLaunchKernelGGL(attn_fwd_kernel<float>, grid, block, lds, (hipStream_t)stream, H, Tq, Tk, (const float*)q, ldq, (const float*)k, ldk, (const float*)v, ldv, (float*)out, ldo, key_mask, causal, lse, PackedRows{nullptr, nullptr, 0});
  } else MIC_CHECK(false, "mic_attn_fwd: bad dtype");
  MIC_LAUNCH_CHECK();
  return MIC_OK;
}

extern "C" int mic_attn_bwd(int dtype, int B, int H, int Tq, int Tk, const void* q, int ldq, const void* k, int ldk,
                            const void* v, int ldv, const void* out, int ldo, const void* dout, int lddo, const float* lse,
                            const int32_t* key_mask, int causal, void* dq, int lddq, void* dk, int lddk, void* dv, int lddv,
                            void* stream) {
  MIC_CHECK(B > 0 && H > 0 && Tq > 0 && Tk > 0, "mic_attn_bwd: bad shape");
  MIC_CHECK(q && k && v && out && dout && lse && dq && dk && dv, "mic_attn_bwd: null pointer");
  const int align = dtype == MIC_BF16 ? 8 : 4;
  MIC_CHECK(ldq % align == 0 && ldk % align == 0 && ldv % align == 0 && ldo % align == 0 && lddo % align == 0, "mic_attn_bwd: row strides must keep 16-B alignment");
  if (Tq > 64 || Tk > 64) {
    dim3 gq(B * H * ((Tq + 63) / 64)), gk(B * H * ((Tk + 63) / 64)), block(256);
#define BWD_TILED(TT)                                                                                                              \
    do {                                                                                                                           \
      const size_t lds = 6 * Tile<TT>::BYTES + 512;                                                                                \
      if (int rc = set_lds(attn_bwd_tiled_kernel<TT, 0>, lds)) return rc;                                                          \
      if (int rc = set_lds(attn_bwd_tiled_kernel<TT, 1>, lds)) return rc;                                                          \
      hipLaunchKernelGGL((attn_bwd_tiled_kernel<TT, 0>), gq, block, lds, (hipStream_t)stream, H, Tq, Tk, (const TT*)q, ldq, (const TT*)k, ldk, (const TT*)v, ldv, (const TT*)out, ldo, (const TT*)dout, lddo, lse, key_mask, causal, (TT*)dq, lddq, (TT*)dk, lddk, (TT*)dv, lddv); \
      hipLaunchKernelGGL((attn_bwd_tiled_kernel<TT, 1>), gk, block, lds, (hipStream_t)stream, H, Tq, Tk, (const TT*)q, ldq, (const TT*)k, ldk, (const TT*)v, ldv, (const TT*)out, ldo, (const TT*)dout, lddo, lse, key_mask, causal, (TT*)dq, lddq, (TT*)dk, lddk, (TT*)dv, lddv); \
    } while (0)
    if (dtype == MIC_BF16) BWD_TILED(uint16_t);
    else if (dtype == MIC_F32) BWD_TILED(float);
    else MIC_CHECK(false, "mic_attn_bwd: bad dtype");
#undef BWD_TILED
    MIC_LAUNCH_CHECK();
    return MIC_OK;
  }
  dim3 grid(B * H), block(256);
  if (dtype == MIC_BF16) {
    const size_t lds = 5 * Tile<uint16_t>::BYTES;
    hipLaunchKernelGGL(attn_bwd_kernel<uint16_t>, grid, block, lds, (hipStream_t)stream, H, Tq, Tk, (const uint16_t*)q, ldq, (const uint16_t*)k, ldk, (const uint16_t*)v, ldv, (const uint16_t*)out, ldo, (const uint16_t*)dout, lddo, lse, key_mask, causal, (uint16_t*)dq, lddq, (uint16_t*)dk, lddk, (uint16_t*)dv, lddv, PackedRows{nullptr, nullptr, 0}, Q8Out{}, Q8Out{});
  } else if (dtype == MIC_F32) {
    const size_t lds = 5 * Tile<float>::BYTES;
    if (int rc = set_lds(attn_bwd_kernel<float>, lds)) return rc;
    hipLaunchKernelGGL(attn_bwd_kernel<float>, grid, block, lds, (hipStream_t)stream, H, Tq, Tk, (const float*)q, ldq, (const float*)k, ldk, (const float*)v, ldv, (const float*)out, ldo, (const float*)dout, lddo, lse, key_mask, causal, (float*)dq, lddq, (float*)dk, lddk, (float*)dv, lddv, PackedRows{nullptr, nullptr, 0}, Q8Out{}, Q8Out{});
  } else MIC_CHECK(false, "mic_attn_bwd: bad dtype");
  MIC_LAUNCH_CHECK();
  return MIC_OK;
}

// ------------------------------------------------------------------ packed (variable-length) rows, one 64x64 tile per sequence
extern "C" int mic_attn_fwd_packed(int dtype, int B, int H, int Tq_max, int Tk, const int32_t* q_off, const int32_t* q_len, int kv_packed,
                                   const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* out, int ldo,
                                   int causal, float* lse, void* stream) {
  MIC_CHECK(B > 0 && H > 0 && Tq_max > 0 && Tq_max <= 64 && Tk > 0 && Tk <= 64, "mic_attn_fwd_packed: one 64x64 tile per sequence (Tq_max=%d Tk=%d)", Tq_max, Tk);
  MIC_CHECK(q && k && v && out && q_off && q_len, "mic_attn_fwd_packed: null pointer");
  const int align = dtype == MIC_BF16 ? 8 : 4;
  MIC_CHECK(ldq % align == 0 && ldk % align == 0 && ldv % align == 0, "mic_attn_fwd_packed: row strides must keep 16-B alignment");
  const PackedRows pk{q_off, q_len, kv_packed};
  dim3 grid(B * H), block(256);
  if (dtype == MIC_BF16) {
    hipLaunchKernelGGL(attn_fwd_kernel<uint16_t>, grid, block, 4 * Tile<uint16_t>::BYTES + 1024, (hipStream_t)stream, H, Tq_max, Tk, (const uint16_t*)q, ldq, (const uint16_t*)k, ldk, (const uint16_t*)v, ldv, (uint16_t*)out, ldo, nullptr, causal, lse, pk);
  } else if (dtype == MIC_F32) {
    const size_t lds = 4 * Tile<float>::BYTES + 1024;
    if (int rc = set_lds(attn_fwd_kernel<float>, lds)) return rc;
    hipLaunchKernelGGL(attn_fwd_kernel<float>, grid, block, lds, (hipStream_t)stream, H, Tq_max, Tk, (const float*)q, ldq, (const float*)k, ldk, (const float*)v, ldv, (float*)out, ldo, nullptr, causal, lse, pk);
  } else MIC_CHECK(false, "mic_attn_fwd_packed: bad dtype");
  MIC_LAUNCH_CHECK();
  return MIC_OK;
}
extern "C" int mic_attn_bwd_packed(int dtype, int B, int H, int Tq_max, int Tk, const int32_t* q_off, const int32_t* q_len, int kv_packed,
                                   const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* out, int ldo,
                                   const void* dout, int lddo, const float* lse, int causal, void* dq, int lddq, void* dk, int lddk,
                                   void* dv, int lddv, void* stream) {
  MIC_CHECK(B > 0 && H > 0 && Tq_max > 0 && Tq_max <= 64 && Tk > 0 && Tk <= 64, "mic_attn_bwd_packed: one 64x64 tile per sequence (Tq_max=%d Tk=%d)", Tq_max, Tk);
  MIC_CHECK(q && k && v && out && dout && lse && dq && dk && dv && q_off && q_len, "mic_attn_bwd_packed: null pointer");
  const int align = dtype == MIC_BF16 ? 8 : 4;
  MIC_CHECK(ldq % align == 0 && ldk % align == 0 && ldv % align == 0 && ldo % align == 0 && lddo % align == 0, "mic_attn_bwd_packed: row strides must keep 16-B alignment");
  const PackedRows pk{q_off, q_len, kv_packed};
  dim3 grid(B * H), block(256);
  if (dtype == MIC_BF16) {
    const size_t lds = 5 * Tile<uint16_t>::BYTES;
    hipLaunchKernelGGL(attn_bwd_kernel<uint16_t>, grid, block, lds, (hipStream_t)stream, H, Tq_max, Tk, (const uint16_t*)q, ldq, (const uint16_t*)k, ldk, (const uint16_t*)v, ldv, (const uint16_t*)out, ldo, (const uint16_t*)dout, lddo, lse, nullptr, causal, (uint16_t*)dq, lddq, (uint16_t*)dk, lddk, (uint16_t*)dv, lddv, pk, Q8Out{}, Q8Out{});
  } else if (dtype == MIC_F32) {
    const size_t lds = 5 * Tile<float>::BYTES;
    if (int rc = set_lds(attn_bwd_kernel<float>, lds)) return rc;
    hipLaunchKernelGGL(attn_bwd_kernel<float>, grid, block, lds, (hipStream_t)stream, H, Tq_max, Tk, (const float*)q, ldq, (const float*)k, ldk, (const float*)v, ldv, (const float*)out, ldo, (const float*)dout, lddo, lse, nullptr, causal, (float*)dq, lddq, (float*)dk, lddk, (float*)dv, lddv, pk, Q8Out{}, Q8Out{});
  } else MIC_CHECK(false, "mic_attn_bwd_packed: bad dtype");
  MIC_LAUNCH_CHECK();
  return MIC_OK;
}

// fused fp8 emission of dQ / dK / dV (bf16 storage, one 64x64 tile per sequence; dense rows with q_off == NULL, packed rows otherwise)
extern "C" int mic_attn_bwd_q8(int B, int H, int Tq, int Tk, const int32_t* q_off, const int32_t* q_len, int kv_packed, const void* q,
                               int ldq, const void* k, int ldk, const void* v, int ldv, const void* out, int ldo, const void* dout, int lddo,
                               const float* lse, const int32_t* key_mask, int causal, const mic_fp8_out* dq8, const mic_fp8_out* dk8,
                               void* dv8, void* stream) {
  MIC_CHECK(B > 0 && H > 0 && Tq > 0 && Tq <= 64 && Tk > 0 && Tk <= 64, "mic_attn_bwd_q8: one 64x64 tile per sequence (Tq=%d Tk=%d)", Tq, Tk);
  MIC_CHECK(q && k && v && out && dout && lse && dq8 && dk8 && dv8 && (!q_off == !q_len), "mic_attn_bwd_q8: null pointer");
  MIC_CHECK(dq8->q && dq8->state && dk8->q && dk8->state && dq8->fmt == dk8->fmt && (dq8->fmt == MIC_E4M3 || dq8->fmt == MIC_E5M2),
            "mic_attn_bwd_q8: bad fp8 outputs");
  MIC_CHECK(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0 && lddo % 8 == 0, "mic_attn_bwd_q8: row strides must keep 16-B alignment");
  const PackedRows pk{q_off, q_len, kv_packed};
  const Q8Out oq{(uint8_t*)dq8->q, dq8->ldq, dq8->state, dq8->amax_next, dq8->fmt}, okv{(uint8_t*)dk8->q, dk8->ldq, dk8->state, dk8->amax_next, dk8->fmt};
  dim3 grid(B * H), block(256);
  const size_t lds = 5 * Tile<uint16_t>::BYTES;
  hipLaunchKernelGGL((attn_bwd_kernel<uint16_t, true>), grid, block, lds, (hipStream_t)stream, H, Tq, Tk, (const uint16_t*)q, ldq, (const uint16_t*)k, ldk,
                     (const uint16_t*)v, ldv, (const uint16_t*)out, ldo, (const uint16_t*)dout, lddo, lse, key_mask, causal, (uint16_t*)dq8->q, dq8->ldq,
                     (uint16_t*)dk8->q, dk8->ldq, (uint16_t*)dv8, dk8->ldq, pk, oq, okv);
  MIC_LAUNCH_CHECK();
  return MIC_OK;
}

// 8 cache elements as they arrive from memory (converted at use: a load in flight costs 4 registers in bf16)
template <typename T> struct Raw8;
template <> struct Raw8<uint16_t> {
  u32x4 v;
  __device__ __forceinline__ void load(const uint16_t* p) { v = *reinterpret_cast<const u32x4*>(p); }
  __device__ __forceinline__ void get(float* o) const { unpack8(v, o); }
};
template <> struct Raw8<float> {
  float4 a, b;
  __device__ __forceinline__ void load(const float* p) { a = *reinterpret_cast<const float4*>(p); b = *reinterpret_cast<const float4*>(p + 4); }
  __device__ __forceinline__ void get(float* o) const { o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w; }
};
// ------------------------------------------------------------------ decode-time attention (K9d): one wave per (row, head)
// Latency-bound (measured ~15 us at one slot, 46 us at 63: dependent round trips x two residency rounds of the 16k waves;
// re-mapping blocks so an image's beams share a CU, issuing V together with K, and LDS-free reductions all measured +-0).
// 8 lanes cover one cache slot's 64-element row (16 B per lane, a full 128-B line in bf16), so one wave
// instruction streams 8 slots fully coalesced for both K and V.  Scores: 8-dim partial dots reduced over the 8 lanes of a
// slot group; softmax: wave reductions over slots; PV: each lane accumulates its 8 dims over its slots, then the 8 slot
// groups are summed by xor-shuffles.  Slot ownership (beam-parent indirection) is looked up per slot.
// Caches longer than 64 slots (generate()'s config default max_length is 200, gen:205-209) are walked in chunks of 64 slots
// with a running (max, sum, output) triple — the flash-decoding recurrence; a single chunk reduces to the plain softmax.
template <typename T, bool CHUNKED>
__global__ __launch_bounds__(256) void attn_decode_kernel(int R, int H, int max_len, int cur, const T* __restrict__ q, int ldq,
                                                          const T* __restrict__ kc, const T* __restrict__ vc, int ldc,
                                                          const int32_t* __restrict__ src_row, int row_div,
                                                          T* __restrict__ out, int ldo) {
  const int lane = threadIdx.x & 63;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= R * H) return;
  const int r = wid / H, h = wid % H;
  const int sub = lane & 7, grp = lane >> 3;  // 8 dims [sub*8, sub*8+8) of slot (c0 + it*8 + grp)
  float qv[8];
  ld8(q + (size_t)r * ldq + h * 64 + sub * 8, qv);
#pragma unroll
  for (int e = 0; e < 8; ++e) qv[e] *= SCALE;
  const int n = min(cur + 1, max_len);
  float m_run = -INFINITY, l_run = 0.f;
  float o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int c0 = 0; c0 < (CHUNKED ? n : 1); c0 += 64) {  // !CHUNKED (caches of at most 64 slots): exactly one pass, no rescaling
    float sc[8];
    int srow[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {  // slot ownership first: every K / V address of the chunk depends on it
      const int slot = c0 + it * 8 + grp;
      srow[it] = 0;
      if (slot < n) srow[it] = src_row ? src_row[(size_t)r * max_len + slot] : r / row_div;
    }
    Raw8<T> kraw[8], vraw[8];  // K and V of the chunk requested together: one round trip instead of K -> softmax -> V
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int slot = c0 + it * 8 + grp;
      if (slot < n) {
        kraw[it].load(kc + ((size_t)srow[it] * max_len + slot) * ldc + h * 64 + sub * 8);
        vraw[it].load(vc + ((size_t)srow[it] * max_len + slot) * ldc + h * 64 + sub * 8);
      }
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int slot = c0 + it * 8 + grp;
      float acc = 0.f;
      if (slot < n) {
        float kv[8];
        kraw[it].get(kv);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += qv[e] * kv[e];
      }
      acc = group8_sum(acc);  // the slot's 8 lanes
      sc[it] = slot < n ? acc : -INFINITY;
    }
    float m = sc[0];
#pragma unroll
    for (int it = 1; it < 8; ++it) m = fmaxf(m, sc[it]);
    m = wave_max(m);  // slot c0 is always valid, so m is finite
    if (CHUNKED) m = fmaxf(m, m_run);
    float l = 0.f;
#pragma unroll
    for (int it = 0; it < 8; ++it) { sc[it] = __expf(sc[it] - m); l += sc[it]; }
    l = wave_sum(l) * 0.125f;  // every slot's probability is replicated on its 8 lanes
    if (CHUNKED) {
      const float alpha = __expf(m_run - m);  // 0 for the first chunk (m_run = -inf)
      l_run = l_run * alpha + l;
      m_run = m;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] *= alpha;
    } else {
      l_run = l;
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int slot = c0 + it * 8 + grp;
      if (slot < n) {
        float vv[8];
        vraw[it].get(vv);
        const float p = sc[it];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += p * vv[e];
      }
    }
  }
  const float inv = 1.0f / l_run;
#pragma unroll
  for (int e = 0; e < 8; ++e) {  // sum the 8 slot groups (lanes ^8, ^16, ^32)
    o[e] += dpp_f<DPP_ROR8>(o[e]);
    o[e] += lane_xor16(o[e]);
    o[e] += lane_xor32(o[e]);
    o[e] *= inv;
  }
  if (grp == 0) st8(out + (size_t)r * ldo + h * 64 + sub * 8, o);
}
// Cross-attention at decode time: the G = row_div beams of an image attend the SAME keys and values (projected once per
// image).  One wave per (image, head) loads every K / V slot once and serves all G query rows from it: 1/G of the L2 traffic
// and of the waves of the row-per-wave kernel (at batch 256 x 4 beams: 52 MB instead of 205 MB per layer and step).
// No slot indirection here (src_row == nullptr), n = cur + 1 <= 64 slots.
template <typename T, int G>
__global__ __launch_bounds__(256) void attn_decode_group_kernel(int NI, int H, int max_len, int cur, const T* __restrict__ q, int ldq,
                                                                const T* __restrict__ kc, const T* __restrict__ vc, int ldc,
                                                                T* __restrict__ out, int ldo) {
  const int lane = threadIdx.x & 63;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= NI * H) return;
  const int img = wid / H, h = wid % H;
  const int sub = lane & 7, grp = lane >> 3;
  float qv[G][8];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    ld8(q + (size_t)(img * G + g) * ldq + h * 64 + sub * 8, qv[g]);
#pragma unroll
    for (int e = 0; e < 8; ++e) qv[g][e] *= SCALE;
  }
  const int n = min(cur + 1, max_len);
  // K and V of all this lane's slots are requested up front (raw 16-B registers): one memory round trip for the wave instead
  // of K -> softmax -> V
  Raw8<T> kraw[8], vraw[8];
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int slot = it * 8 + grp;
    if (slot < n) {
      kraw[it].load(kc + ((size_t)img * max_len + slot) * ldc + h * 64 + sub * 8);
      vraw[it].load(vc + ((size_t)img * max_len + slot) * ldc + h * 64 + sub * 8);
    }
  }
  float sc[G][8];
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int slot = it * 8 + grp;
    float kv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (slot < n) kraw[it].get(kv);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float acc = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc += qv[g][e] * kv[e];
      acc = group8_sum(acc);
      sc[g][it] = slot < n ? acc : -INFINITY;
    }
  }
  float inv[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    float m = sc[g][0];
#pragma unroll
    for (int it = 1; it < 8; ++it) m = fmaxf(m, sc[g][it]);
    m = wave_max(m);
    float l = 0.f;
#pragma unroll
    for (int it = 0; it < 8; ++it) { sc[g][it] = __expf(sc[g][it] - m); l += sc[g][it]; }
    inv[g] = 1.0f / (wave_sum(l) * 0.125f);
  }
  float o[G][8];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int e = 0; e < 8; ++e) o[g][e] = 0.f;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int slot = it * 8 + grp;
    if (slot < n) {
      float vv[8];
      vraw[it].get(vv);
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float p = sc[g][it] * inv[g];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[g][e] += p * vv[e];
      }
    }
  }
#pragma unroll
  for (int g = 0; g < G; ++g) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      o[g][e] += dpp_f<DPP_ROR8>(o[g][e]);
      o[g][e] += lane_xor16(o[g][e]);
      o[g][e] += lane_xor32(o[g][e]);
    }
    if (grp == 0) st8(out + (size_t)(img * G + g) * ldo + h * 64 + sub * 8, o[g]);
  }
}

extern "C" int mic_attn_decode(int dtype, int R, int H, int max_len, int cur, const void* q, int ldq, const void* kc,
                               const void* vc, int ldc, const int32_t* src_row, int row_div, void* out, int ldo, void* stream) {
  MIC_CHECK(R > 0 && H > 0 && max_len > 0 && cur >= 0 && row_div >= 1, "mic_attn_decode: bad shape R=%d H=%d max_len=%d cur=%d", R, H, max_len, cur);
  MIC_CHECK(q && kc && vc && out, "mic_attn_decode: null pointer");
  const bool chunked = (cur + 1 < max_len ? cur + 1 : max_len) > 64;
  if (!src_row && !chunked && (row_div == 2 || row_div == 4 || row_div == 8) && R % row_div == 0) {
    // the beams of an image share keys and values: one wave per (image, head) serves all of them
    const int NI = R / row_div;
    dim3 ggrid((NI * H + 3) / 4), gblock(256);
#define DECG_LAUNCH(TT, GG) hipLaunchKernelGGL((attn_decode_group_kernel<TT, GG>), ggrid, gblock, 0, (hipStream_t)stream, NI, H, max_len, cur, (const TT*)q, ldq, (const TT*)kc, (const TT*)vc, ldc, (TT*)out, ldo)
#define DECG_DISPATCH(TT) do { if (row_div == 2) DECG_LAUNCH(TT, 2); else if (row_div == 4) DECG_LAUNCH(TT, 4); else DECG_LAUNCH(TT, 8); } while (0)
    if (dtype == MIC_BF16) DECG_DISPATCH(uint16_t);
    else if (dtype == MIC_F32) DECG_DISPATCH(float);
    else MIC_CHECK(false, "mic_attn_decode: bad dtype");
#undef DECG_DISPATCH
#undef DECG_LAUNCH
    MIC_LAUNCH_CHECK();
    return MIC_OK;
  }
  dim3 grid((R * H + 3) / 4), block(256);
#define DEC_LAUNCH(TT, CH) hipLaunchKernelGGL((attn_decode_kernel<TT, CH>), grid, block, 0, (hipStream_t)stream, R, H, max_len, cur, (const TT*)q, ldq, (const TT*)kc, (const TT*)vc, ldc, src_row, row_div, (TT*)out, ldo)
  if (dtype == MIC_BF16) { if (chunked) DEC_LAUNCH(uint16_t, true); else DEC_LAUNCH(uint16_t, false); }
  else if (dtype == MIC_F32) { if (chunked) DEC_LAUNCH(float, true); else DEC_LAUNCH(float, false); }
  else MIC_CHECK(false, "mic_attn_decode: bad dtype");
#undef DEC_LAUNCH
  MIC_LAUNCH_CHECK();
  return MIC_OK;
}

template <typename T>
__global__ void kv_append_kernel(int R, int HD, int max_len, int cur, const T* __restrict__ k, int ldk, const T* __restrict__ v,
                                 int ldv, T* __restrict__ kc, T* __restrict__ vc) {
  const long total = (long)R * HD;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int c = (int)(e % HD);
    const int r = (int)(e / HD);
    kc[((size_t)r * max_len + cur) * HD + c] = k[(size_t)r * ldk + c];
    vc[((size_t)r * max_len + cur) * HD + c] = v[(size_t)r * ldv + c];
  }
}
extern "C" int mic_kv_append(int dtype, int R, int HD, int max_len, int cur, const void* k, int ldk, const void* v, int ldv,
                             void* kc, void* vc, void* stream) {
  MIC_CHECK(R > 0 && HD > 0 && cur >= 0 && cur < max_len && k && v && kc && vc, "mic_kv_append: bad args");
  const long total = (long)R * HD;
  int nb = (int)((total + 255) / 256); if (nb > 4096) nb = 4096;
  dim3 grid(nb), block(256);
  if (dtype == MIC_BF16)
    hipLaunchKernelGGL(kv_append_kernel<uint16_t>, grid, block, 0, (hipStream_t)stream, R, HD, max_len, cur, (const uint16_t*)k, ldk, (const uint16_t*)v, ldv, (uint16_t*)kc, (uint16_t*)vc);
  else if (dtype == MIC_F32)
    hipLaunchKernelGGL(kv_append_kernel<float>, grid, block, 0, (hipStream_t)stream, R, HD, max_len, cur, (const float*)k, ldk, (const float*)v, ldv, (float*)kc, (float*)vc);
  else MIC_CHECK(false, "mic_kv_append: bad dtype");
  MIC_LAUNCH_CHECK();
  return MIC_OK;
}
