// norm.hip — LayerNorm forward/backward (K2).  HBM-bound: one wave per row, 16-B vector loads, the row lives in
// registers, statistics by wave xor-shuffle reductions in fp32 (flax nn.LayerNorm semantics: biased variance).
#include "common.h"
#include <stdlib.h>

#define LN_MAXC 4  // 16-B chunks of 8 elements per lane: width <= 64*8*4 = 2048

// Q8 (bf16 storage): the normalised row ALSO (or, y == nullptr, only) goes out as fp8 bytes under the tensor's delayed scale
// (common.h: Q8Out) — the operand of an fp8 projection; the quantiser launch behind this kernel disappears.
template <typename T, bool Q8 = false>
__global__ __launch_bounds__(256) void ln_fwd_kernel(int rows, int width, const T* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float eps, T* __restrict__ y, float* __restrict__ mean_out,
                                                     float* __restrict__ rstd_out, uint32_t thr, uint32_t seed,
                                                     float dscale, Q8Out q8) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= rows) return;
  Q8Ctx qc;
  if constexpr (Q8) qc = q8_begin(q8, blockIdx.x == 0 && threadIdx.x == 0);
  const int nchunk = width >> 3;
  const T* xr = x + (size_t)row * width;
  float v[LN_MAXC][8];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < LN_MAXC; ++c) {
    const int ch = lane + c * 64;
    if (ch < nchunk) {
      ld8(xr + ch * 8, v[c]);
#pragma unroll
      for (int i = 0; i < 8; ++i) s += v[c][i];
    }
  }
  const float mean = wave_sum(s) / (float)width;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < LN_MAXC; ++c) {
    const int ch = lane + c * 64;
    if (ch < nchunk) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { const float d = v[c][i] - mean; q += d * d; }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)width + eps);
  if (lane == 0) {
    if (mean_out) mean_out[row] = mean;
    if (rstd_out) rstd_out[row] = rstd;
  }
  T* yr = y + (size_t)row * width;
#pragma unroll
  for (int c = 0; c < LN_MAXC; ++c) {
    const int ch = lane + c * 64;
    if (ch < nchunk) {
      float g[8], b[8], o[8];
      ld8(gamma + ch * 8, g);
      ld8(beta + ch * 8, b);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        o[i] = (v[c][i] - mean) * rstd * g[i] + b[i];
        if (thr) o[i] = dropout_keep(seed, (uint32_t)row * (uint32_t)width + (uint32_t)(ch * 8 + i), thr) ? o[i] * dscale : 0.f;
      }
      if (!Q8 || y) st8(yr + ch * 8, o);
      if constexpr (Q8) {
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = round_to<T>(o[i]);
        *reinterpret_cast<uint2*>(q8.q + (size_t)row * q8.ldq + ch * 8) = q8_pack8(qc, o, q8.fmt);
      }
    }
  }
  if constexpr (Q8) q8_end_wave(q8, qc, row);
}

// dx = (dres) + rstd * (dy*g - mean(dy*g) - xhat * mean(dy*g*xhat));  dgamma += dy*xhat, dbeta += dy.
// Each wave walks rows with a grid stride keeping per-column partial dgamma/dbeta in registers; the block combines
// its 4 waves through LDS and issues one fp32 atomic per column.
#define LNB_WAVES 8  // waves per backward block: atomics scale with the BLOCK count, latency hiding with the WAVE count
// NCH = 16-B chunks per lane (width <= 512*NCH): width <= 1024 runs with NCH = 2, half the registers of the generic
// NCH = 4 build, so twice the waves fit a SIMD — this kernel lives on memory-level parallelism.
// Q8: 0 = off; 1 = dxm (the dropout-masked gradient entering the residual branch) also / only (dxm == nullptr) as fp8 bytes;
// 2 = dx itself also as fp8 bytes (the ViT's residual-stream gradient is the FFN-out projection's dy) — e5m2 under the tensor's
// delayed scale (common.h: Q8Out).
template <typename T, int NCH, int NW, int Q8 = 0>
__global__ __launch_bounds__(64 * NW, NCH <= 2 ? 4 : 2) void ln_bwd_kernel(int rows, int width, const T* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const T* __restrict__ dy,
                                                     const T* __restrict__ dres, T* __restrict__ dx,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                     T* __restrict__ dxm, uint32_t thr_m, uint32_t seed_m, float scale_m,
                                                     uint32_t thr_in, uint32_t seed_in, float scale_in, float* __restrict__ partials,
                                                     Q8Out q8) {
  __shared__ float red[NW][64 * 8 + 8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nchunk = width >> 3;
  Q8Ctx qc;
  if constexpr (Q8 != 0) qc = q8_begin(q8, blockIdx.x == 0 && threadIdx.x == 0);
  float dg[NCH][8], db[NCH][8];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int i = 0; i < 8; ++i) { dg[c][i] = 0.f; db[c][i] = 0.f; }

  for (int row = blockIdx.x * NW + wave; row < rows; row += gridDim.x * NW) {
    const T* xr = x + (size_t)row * width;
    const T* dyr = dy + (size_t)row * width;
    const float mu = mean[row], rs = rstd[row];
    float xh[NCH][8], gdy[NCH][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + c * 64;
      if (ch < nchunk) {
        float xv[8], dv[8], g[8];
        ld8(xr + ch * 8, xv);
        ld8(dyr + ch * 8, dv);
        ld8(gamma + ch * 8, g);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          float d = dv[i];
          if (thr_in) d = dropout_keep(seed_in, (uint32_t)row * (uint32_t)width + (uint32_t)(ch * 8 + i), thr_in) ? d * scale_in : 0.f;
          const float h = (xv[i] - mu) * rs;
          xh[c][i] = h;
          dg[c][i] += d * h;
          db[c][i] += d;
          const float gd = d * g[i];
          gdy[c][i] = gd;
          s1 += gd;
          s2 += gd * h;
        }
      }
    }
    const float c1 = wave_sum(s1) / (float)width, c2 = wave_sum(s2) / (float)width;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + c * 64;
      if (ch < nchunk) {
        float o[8], r[8];
        if (dres) ld8(dres + (size_t)row * width + ch * 8, r);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          o[i] = rs * (gdy[c][i] - c1 - xh[c][i] * c2);
          if (dres) o[i] += r[i];
        }
        st8(dx + (size_t)row * width + ch * 8, o);
        if constexpr (Q8 == 2) {
          float t[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) t[i] = round_to<T>(o[i]);
          *reinterpret_cast<uint2*>(q8.q + (size_t)row * q8.ldq + ch * 8) = q8_pack8(qc, t, q8.fmt);
        }
        if (dxm || Q8 == 1) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float ov = round_to<T>(o[i]);
            o[i] = dropout_keep(seed_m, (uint32_t)row * (uint32_t)width + (uint32_t)(ch * 8 + i), thr_m) ? ov * scale_m : 0.f;
          }
          if (dxm) st8(dxm + (size_t)row * width + ch * 8, o);
          if constexpr (Q8 == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = round_to<T>(o[i]);
            *reinterpret_cast<uint2*>(q8.q + (size_t)row * q8.ldq + ch * 8) = q8_pack8(qc, o, q8.fmt);
          }
        }
      }
    }
  }
  if constexpr (Q8 != 0) q8_end_wave(q8, qc, blockIdx.x * NW + wave);
  // combine the waves' column partials (dgamma, then dbeta, through the same LDS buffer)
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ch = lane + c * 64;
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      __syncthreads();
      if (ch < nchunk) {
#pragma unroll
        for (int i = 0; i < 8; ++i) red[wave][lane * 8 + i] = which ? db[c][i] : dg[c][i];
      }
      __syncthreads();
      float* dst = which ? dbeta : dgamma;
      for (int col = threadIdx.x; col < 512; col += 64 * NW) {
        const int gcol = c * 512 + col;  // = (lane' + c*64)*8 + i with lane'*8+i = col
        if (gcol < width && (dst || partials)) {
          float a = 0.f;
#pragma unroll
          for (int w = 0; w < NW; ++w) a += red[w][col];
          // partials: this block's column sums as plain stores into [2][gridDim.x][width] (summed later, off the critical path, in a
          // fixed order: mic_ln_param_grads) instead of 2 * width atomics per block — a third of this kernel's time at 2.4 k rows
          if (partials) partials[((size_t)which * gridDim.x + blockIdx.x) * width + gcol] = a;
          else atomicAdd(dst + gcol, a);
        }
      }
    }
  }
}

static inline uint32_t thr_of(float p) { return p > 0.f ? (uint32_t)fminf(p * 4294967296.0f, 4294967295.0f) : 0u; }

static int q8_check(const mic_fp8_out* q8, int width, const char* who) {
  MIC_CHECK(q8 && q8->q && q8->state && q8->ldq >= width && q8->ldq % 8 == 0 && ((uintptr_t)q8->q & 7) == 0, "%s: bad fp8 output (q, state, ldq %% 8 == 0)", who);
  MIC_CHECK(q8->fmt == MIC_E4M3 || q8->fmt == MIC_E5M2, "%s: bad fp8 format", who);
  return MIC_OK;
}
static inline Q8Out q8_of(const mic_fp8_out* q8) {
  Q8Out o{};
  if (q8) { o.q = (uint8_t*)q8->q; o.ldq = q8->ldq; o.state = q8->state; o.amax_next = q8->amax_next; o.fmt = q8->fmt; }
  return o;
}
static int ln_fwd_impl(int dtype, int rows, int width, const void* x, const float* gamma, const float* beta, float eps, void* y,
                       float* mean, float* rstd, float dropout_p, uint32_t dropout_seed, const mic_fp8_out* q8, void* stream) {
  MIC_CHECK(rows > 0 && width > 0 && width % 8 == 0 && width <= 64 * 8 * LN_MAXC, "mic_layernorm_fwd: bad shape rows=%d width=%d", rows, width);
  MIC_CHECK(x && gamma && beta && (y || q8), "mic_layernorm_fwd: null pointer");
  dim3 grid((rows + 3) / 4), block(256);
  const uint32_t thr = thr_of(dropout_p);
  const float sc = 1.0f / (1.0f - dropout_p);
  if (q8) {
    MIC_CHECK(dtype == MIC_BF16, "mic_layernorm_fwd_q8: bf16 storage only");
    if (int rc = q8_check(q8, width, "mic_layernorm_fwd_q8")) return rc;
    hipLaunchKernelGGL((ln_fwd_kernel<uint16_t, true>), grid, block, 0, (hipStream_t)stream, rows, width, (const uint16_t*)x, gamma, beta, eps, (uint16_t*)y, mean, rstd, thr, dropout_seed, sc, q8_of(q8));
  } else if (dtype == MIC_BF16)
    hipLaunchKernelGGL((ln_fwd_kernel<uint16_t, false>), grid, block, 0, (hipStream_t)stream, rows, width, (const uint16_t*)x, gamma, beta, eps, (uint16_t*)y, mean, rstd, thr, dropout_seed, sc, Q8Out{});
  else if (dtype == MIC_F32)
    hipLaunchKernelGGL((ln_fwd_kernel<float, false>), grid, block, 0, (hipStream_t)stream, rows, width, (const float*)x, gamma, beta, eps, (float*)y, mean, rstd, thr, dropout_seed, sc, Q8Out{});
  else MIC_CHECK(false, "mic_layernorm_fwd: bad dtype");
  MIC_LAUNCH_CHECK();
  return MIC_OK;
}
extern "C" int mic_layernorm_fwd(int dtype, int rows, int width, const void* x, const float* gamma, const float* beta,
                                 float eps, void* y, float* mean, float* rstd, float dropout_p, uint32_t dropout_seed,
                                 void* stream) {
  return ln_fwd_impl(dtype, rows, width, x, gamma, beta, eps, y, mean, rstd, dropout_p, dropout_seed, nullptr, stream);
}
extern "C" int mic_layernorm_fwd_q8(int rows, int width, const void* x, const float* gamma, const float* beta, float eps, void* y,
                                    float* mean, float* rstd, float dropout_p, uint32_t dropout_seed, const mic_fp8_out* q8, void* stream) {
  MIC_CHECK(q8 != nullptr, "mic_layernorm_fwd_q8: null fp8 output");
  return ln_fwd_impl(MIC_BF16, rows, width, x, gamma, beta, eps, y, mean, rstd, dropout_p, dropout_seed, q8, stream);
}

// block cap: 512 (MIC_LNB_BLOCKS, A/B).  With the gamma / beta gradients as block partials (no atomics) a block per 8 rows up to 512
// blocks measured 12.3 -> 11.1 us at 2432 x 1024, 15.7 -> 14.2 at 4096 x 1024 against the cap of 256 the atomics form was tuned for
// (every wave gets one row instead of some getting two); 128: 15.1 / 22.3 us.
static int ln_bwd_blocks(int rows) {
  static const int cap = [] { const char* e = getenv("MIC_LNB_BLOCKS"); const int v = e ? atoi(e) : 512; return v < 1 ? 1 : (v > 4096 ? 4096 : v); }();
  const int nblk = (rows + LNB_WAVES - 1) / LNB_WAVES;
  return nblk > cap ? cap : nblk;
}
extern "C" int mic_layernorm_bwd_blocks(int rows) { return ln_bwd_blocks(rows > 0 ? rows : 1); }
static int ln_bwd_impl(int dtype, int rows, int width, const void* x, const float* gamma, const float* mean, const float* rstd,
                       const void* dy, const void* dres, void* dx, float* dgamma, float* dbeta, void* dxm, float dropout_p,
                       uint32_t dropout_seed, float in_dropout_p, uint32_t in_dropout_seed, float* partials, void* stream,
                       const mic_fp8_out* q8 = nullptr, int q8_of_dx = 0);
extern "C" int mic_layernorm_bwd(int dtype, int rows, int width, const void* x, const float* gamma, const float* mean,
                                 const float* rstd, const void* dy, const void* dres, void* dx, float* dgamma,
                                 float* dbeta, void* dxm, float dropout_p, uint32_t dropout_seed, float in_dropout_p,
                                 uint32_t in_dropout_seed, void* stream) {
  return ln_bwd_impl(dtype, rows, width, x, gamma, mean, rstd, dy, dres, dx, dgamma, dbeta, dxm, dropout_p, dropout_seed, in_dropout_p,
                     in_dropout_seed, nullptr, stream);
}
extern "C" int mic_layernorm_bwd_partials(int dtype, int rows, int width, const void* x, const float* gamma, const float* mean,
                                          const float* rstd, const void* dy, const void* dres, void* dx, float* partials, void* dxm,
                                          float dropout_p, uint32_t dropout_seed, float in_dropout_p, uint32_t in_dropout_seed, void* stream) {
  MIC_CHECK(partials != nullptr, "mic_layernorm_bwd_partials: null partials");
  return ln_bwd_impl(dtype, rows, width, x, gamma, mean, rstd, dy, dres, dx, nullptr, nullptr, dxm, dropout_p, dropout_seed, in_dropout_p,
                     in_dropout_seed, partials, stream);
}
extern "C" int mic_layernorm_bwd_partials_q8(int rows, int width, const void* x, const float* gamma, const float* mean, const float* rstd,
                                             const void* dy, const void* dres, void* dx, float* partials, void* dxm, float dropout_p,
                                             uint32_t dropout_seed, float in_dropout_p, uint32_t in_dropout_seed, const mic_fp8_out* q8,
                                             int q8_of_dx, void* stream) {
  MIC_CHECK(partials != nullptr && q8 != nullptr, "mic_layernorm_bwd_partials_q8: null partials / fp8 output");
  if (int rc = q8_check(q8, width, "mic_layernorm_bwd_partials_q8")) return rc;
  return ln_bwd_impl(MIC_BF16, rows, width, x, gamma, mean, rstd, dy, dres, dx, nullptr, nullptr, dxm, dropout_p, dropout_seed, in_dropout_p,
                     in_dropout_seed, partials, stream, q8, q8_of_dx);
}
// dgamma / dbeta from the block partials of mic_layernorm_bwd_partials: out[which][col] (+)= sum over blocks, in block order
struct LnParamItem { const float* partials; float* dgamma; float* dbeta; int nblk, width, accumulate; };
struct LnParamTable { int count; LnParamItem it[8]; };
// one block = 32 columns x 8 row lanes: every lane sums nblk / 8 partial rows (sixteen independent loads in flight per trip: with
// one thread per column walking all 256 rows the kernel was a 24-us latency chain), the eight lanes meet in LDS in a fixed order
__global__ __launch_bounds__(256) void ln_param_grads_kernel(LnParamTable tab) {
  __shared__ float red[8][33];
  const LnParamItem& I = tab.it[blockIdx.z];
  const int which = blockIdx.y, c = threadIdx.x & 31, r = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + c;
  float* dst = which ? I.dbeta : I.dgamma;
  if (blockIdx.x * 32 >= I.width || !dst) return;  // (block-uniform)
  float a = 0.f;
  if (col < I.width) {
    const float* src = I.partials + (size_t)which * I.nblk * I.width + col;
    for (int b0 = r; b0 < I.nblk; b0 += 8 * 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int b = b0 + 8 * u;
        v[u] = b < I.nblk ? src[(size_t)b * I.width] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) a += v[u];
    }
  }
  red[r][c] = a;
  __syncthreads();
  if (r == 0 && col < I.width) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][c];
    dst[col] = I.accumulate ? dst[col] + t : t;
  }
}
extern "C" int mic_ln_param_grads(const mic_ln_param_item* items, int count, void* stream) {
  MIC_CHECK(items && count >= 1, "mic_ln_param_grads: bad args");
  for (int i0 = 0; i0 < count; i0 += 8) {
    LnParamTable tab;
    tab.count = count - i0 < 8 ? count - i0 : 8;
    int wmax = 0;
    for (int i = 0; i < tab.count; ++i) {
      const mic_ln_param_item& m = items[i0 + i];
      MIC_CHECK(m.partials && m.nblk > 0 && m.width > 0, "mic_ln_param_grads: bad item %d", i0 + i);
      tab.it[i] = LnParamItem{m.partials, m.dgamma, m.dbeta, m.nblk, m.width, m.accumulate};
      wmax = m.width > wmax ? m.width : wmax;
    }
    hipLaunchKernelGGL(ln_param_grads_kernel, dim3((wmax + 31) / 32, 2, tab.count), dim3(256), 0, (hipStream_t)stream, tab);
    MIC_LAUNCH_CHECK();
  }
  return MIC_OK;
}
static int ln_bwd_impl(int dtype, int rows, int width, const void* x, const float* gamma, const float* mean, const float* rstd,
                       const void* dy, const void* dres, void* dx, float* dgamma, float* dbeta, void* dxm, float dropout_p,
                       uint32_t dropout_seed, float in_dropout_p, uint32_t in_dropout_seed, float* partials, void* stream,
                       const mic_fp8_out* q8, int q8_of_dx) {
  MIC_CHECK(rows > 0 && width > 0 && width % 8 == 0 && width <= 64 * 8 * LN_MAXC, "mic_layernorm_bwd: bad shape rows=%d width=%d", rows, width);
  MIC_CHECK(x && gamma && mean && rstd && dy && dx, "mic_layernorm_bwd: null pointer");
  // measured: 16-wave blocks 21.9 -> 20.9 us, more than 256 blocks slower (each block ends with 2*width fp32 atomics,
  // ~4 us of the ~20 us at 4096x1024) -- the kernel is latency-bound at two rows per wave, not bandwidth-bound
  const int NWS = LNB_WAVES;
  const int nblk = ln_bwd_blocks(rows);
  dim3 grid(nblk), block(64 * NWS);
  const uint32_t thr_m = (dxm || (q8 && !q8_of_dx)) ? thr_of(dropout_p) : 0u, thr_in = thr_of(in_dropout_p);
  const Q8Out q8o = q8_of(q8);
  const float sm = 1.0f / (1.0f - dropout_p), si = 1.0f / (1.0f - in_dropout_p);
#define LNB_LAUNCH(TT, NC) LNB_LAUNCH2(TT, NC, LNB_WAVES)
#define LNB_LAUNCH2(TT, NC, NWW) LNB_LAUNCH3(TT, NC, NWW, 0)
#define LNB_LAUNCH3(TT, NC, NWW, QQ) hipLaunchKernelGGL((ln_bwd_kernel<TT, NC, NWW, QQ>), grid, block, 0, (hipStream_t)stream, rows, width, (const TT*)x, gamma, mean, rstd, (const TT*)dy, (const TT*)dres, (TT*)dx, dgamma, dbeta, (TT*)dxm, thr_m, dropout_seed, sm, thr_in, in_dropout_seed, si, partials, q8o)
  if (q8) {
    MIC_CHECK(dtype == MIC_BF16, "mic_layernorm_bwd_partials_q8: bf16 storage only");
    if (q8_of_dx) { if (width <= 1024) LNB_LAUNCH3(uint16_t, 2, LNB_WAVES, 2); else LNB_LAUNCH3(uint16_t, LN_MAXC, LNB_WAVES, 2); }
    else { if (width <= 1024) LNB_LAUNCH3(uint16_t, 2, LNB_WAVES, 1); else LNB_LAUNCH3(uint16_t, LN_MAXC, LNB_WAVES, 1); }
  } else
  if (dtype == MIC_BF16) { if (width <= 1024) LNB_LAUNCH(uint16_t, 2); else LNB_LAUNCH(uint16_t, LN_MAXC); }
  else if (dtype == MIC_F32) { if (width <= 1024) LNB_LAUNCH(float, 2); else LNB_LAUNCH(float, LN_MAXC); }
#undef LNB_LAUNCH
#undef LNB_LAUNCH2
#undef LNB_LAUNCH3
  else MIC_CHECK(false, "mic_layernorm_bwd: bad dtype");
  MIC_LAUNCH_CHECK();
  return MIC_OK;
}

// ------------------------------------------------------------------ LayerNorm folded into the consumer Linear (decode path)
// one wave per output row n of w [N][K]:  w_fold = round(w * gamma),  colsum = sum of the ROUNDED products,  bias_fold = bias + w . beta
template <typename T>
__global__ __launch_bounds__(256) void ln_fold_weight_kernel(int N, int K, const T* __restrict__ w, int ldw, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ bias,
                                                             T* __restrict__ wf, int ldwf, float* __restrict__ colsum,
                                                             float* __restrict__ bias_fold) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= N) return;
  float g = 0.f, b = 0.f;
  for (int k = lane * 8; k < K; k += 64 * 8) {
    float v[8], ga[8], be[8], o[8];
    ld8(w + (size_t)n * ldw + k, v);
    ld8(gamma + k, ga);
    ld8(beta + k, be);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      o[i] = round_to<T>(v[i] * ga[i]);
      g += o[i];
      b = fmaf(v[i], be[i], b);
    }
    st8(wf + (size_t)n * ldwf + k, o);
  }
  g = wave_sum(g);
  b = wave_sum(b);
  if (lane == 0) {
    colsum[n] = g;
    bias_fold[n] = b + (bias ? bias[n] : 0.f);
  }
}
extern "C" int mic_ln_fold_weight(int dtype, int N, int K, const void* w, int ldw, const float* gamma, const float* beta,
                                  const float* bias, void* w_fold, int ldwf, float* colsum, float* bias_fold, void* stream) {
  MIC_CHECK(N > 0 && K > 0 && K % 8 == 0 && ldw % 8 == 0 && ldwf % 8 == 0 && w && gamma && beta && w_fold && colsum && bias_fold,
            "mic_ln_fold_weight: bad args (K, ldw, ldwf multiples of 8)");
  dim3 grid((N + 3) / 4), block(256);
  if (dtype == MIC_BF16)
    hipLaunchKernelGGL(ln_fold_weight_kernel<uint16_t>, grid, block, 0, (hipStream_t)stream, N, K, (const uint16_t*)w, ldw, gamma, beta, bias,
                       (uint16_t*)w_fold, ldwf, colsum, bias_fold);
  else if (dtype == MIC_F32)
    hipLaunchKernelGGL(ln_fold_weight_kernel<float>, grid, block, 0, (hipStream_t)stream, N, K, (const float*)w, ldw, gamma, beta, bias,
                       (float*)w_fold, ldwf, colsum, bias_fold);
  else MIC_CHECK(false, "mic_ln_fold_weight: bad dtype");
  MIC_LAUNCH_CHECK();
  return MIC_OK;
}
