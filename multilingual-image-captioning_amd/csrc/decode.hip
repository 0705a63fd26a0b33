// decode.hip — generation epilogues: log-softmax + index-stable top-k over the 250 054-wide head output (K16/K19) and
// the beam-search bookkeeping of gen:857-966 (K17) as ONE small kernel per step; the KV cache is never gathered
// (K18) — a [rows][max_len] slot-ownership table is updated instead (see attention.hip / mic_attn_decode).
#include "common.h"

#define TOPK_MAX 16   // widest top-k of the partials-based kernel (row_topk_tiles: k = 2 * num_beams <= 16, its candidate set lives in registers)
#define TOPK_WIDE 64  // widest top-k of the streaming kernel and of the beam bookkeeping: num_beams <= 32
#define NEG_BIG (-1.0e7f)

__device__ __forceinline__ bool better(float av, int ai, float bv, int bi) { return av > bv || (av == bv && ai < bi); }

// ------------------------------------------------------------------ per-row lse + top-k
// One 256-thread block per row of the [R][ld] logits.  Pass 1 streams the row once: online (max, sum-exp) and each
// thread's own maximum.  The 8th largest of the 256 thread maxima is a provable lower bound tau on the row's k-th best
// value (8 distinct elements are >= it), so pass 2 only has to run the (divergent, 8-deep) insertion on the rare
// elements >= tau; everything else costs two subtractions and a compare.  Ordering is exactly lax.top_k's
// (value desc, index asc) on the PROCESSED value (log-softmax, processors, + running score) — ties in the rounded fp32
// value are broken by index, so thresholds are compared in that same processed domain.
template <typename T, int KMAX>
__global__ __launch_bounds__(256) void row_lse_topk_kernel(int V, const T* __restrict__ logits, int ld, int k, int forced,
                                                           int suppress_eos, int eos, int raw, const float* __restrict__ row_bias,
                                                           float* __restrict__ top_val, int32_t* __restrict__ top_idx) {
  __shared__ float sm[256], ss[256];
  __shared__ int si[256];
  const int row = blockIdx.x, tid = threadIdx.x;
  const T* lr = logits + (size_t)row * ld;
  const int nchunk = (V + 7) >> 3;
  const float bias = row_bias ? row_bias[row] : 0.f;
  if (forced >= 0) {
    // ForcedBOS / ForcedEOS: everything -inf except the forced token := 0 (+ running score); lax.top_k then lists the
    // lowest indices among the -inf ties.  No scan needed.
    if (tid < k) {
      int idx = forced;
      if (tid > 0) { idx = tid - 1; if (idx >= forced) ++idx; }
      top_val[(size_t)row * k + tid] = tid == 0 ? 0.f + bias : -INFINITY;
      top_idx[(size_t)row * k + tid] = idx;
    }
    return;
  }
  // ---- pass 1: online log-sum-exp + per-thread maximum (of the values that stay eligible)
  float m = -INFINITY, s = 0.f, tmax = -INFINITY;
  int targ = 0x7fffffff;
  for (int ch0 = tid; ch0 < nchunk; ch0 += 4 * 256) {
    float v[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ch = ch0 + u * 256;
      if (ch < nchunk) ld8(lr + ch * 8, v[u]);
    }
    float cm = -INFINITY;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ch = ch0 + u * 256;
      if (ch < nchunk) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int c = ch * 8 + i;
          if (c < V) {
            cm = fmaxf(cm, v[u][i]);
            if (!(suppress_eos && c == eos) && v[u][i] > tmax) { tmax = v[u][i]; targ = c; }
          }
        }
      }
    }
    if (!raw) {
      const float mn = fmaxf(m, cm);
      float add = 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ch = ch0 + u * 256;
        if (ch < nchunk) {
#pragma unroll
          for (int i = 0; i < 8; ++i) if (ch * 8 + i < V) add += __expf(v[u][i] - mn);
        }
      }
      s = (mn == -INFINITY) ? 0.f : s * __expf(m - mn) + add;
      m = mn;
    }
  }
  float mx = 0.f, logsum = 0.f;
  if (!raw) {
    sm[tid] = m; ss[tid] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) {
        const float m1 = sm[tid], m2 = sm[tid + o], mn = fmaxf(m1, m2);
        // threads that saw no column carry (-inf, 0): exp(-inf - -inf) would be NaN
        ss[tid] = mn == -INFINITY ? 0.f : ss[tid] * __expf(m1 - mn) + ss[tid + o] * __expf(m2 - mn);
        sm[tid] = mn;
      }
      __syncthreads();
    }
    mx = sm[0];
    logsum = logf(ss[0]);
    __syncthreads();
  }
  if (raw && k == 1) {
    // greedy: first-max argmax (gen:499)
    sm[tid] = tmax; si[tid] = targ;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o && better(sm[tid + o], si[tid + o], sm[tid], si[tid])) { sm[tid] = sm[tid + o]; si[tid] = si[tid + o]; }
      __syncthreads();
    }
    if (tid == 0) { top_val[row] = sm[0] + bias; top_idx[row] = si[0]; }
    return;
  }
  // ---- tau: the KMAX-th largest thread maximum (processed domain): at least KMAX >= k distinct elements are >= tau
  float tau;
  {
    float mine = tmax == -INFINITY ? -INFINITY : ((raw ? tmax : (tmax - mx) - logsum) + bias);
    float last = INFINITY;
    for (int round = 0; round < KMAX; ++round) {
      sm[tid] = mine; si[tid] = tid;
      __syncthreads();
      for (int o = 128; o > 0; o >>= 1) {
        if (tid < o && better(sm[tid + o], si[tid + o], sm[tid], si[tid])) { sm[tid] = sm[tid + o]; si[tid] = si[tid + o]; }
        __syncthreads();
      }
      last = sm[0];
      if (tid == si[0]) mine = -INFINITY;
      __syncthreads();
    }
    tau = last;  // -inf when fewer than 8 threads saw an eligible value: then everything is a candidate
  }
  // ---- pass 2: exact per-thread top-k of the candidates >= tau
  float t_safe = -INFINITY;
  if (tau > -INFINITY) {
    const float inv = raw ? (tau - bias) : ((tau - bias) + logsum) + mx;  // raw logit whose processed value is ~tau
    t_safe = inv - 1e-5f * (fabsf(tau) + fabsf(bias) + fabsf(logsum) + fabsf(mx) + 1.0f);  // fp32 rounding of 3-4 ops is ~1e-7 relative
  }
  float bv[KMAX];
  int bi[KMAX];
#pragma unroll
  for (int i = 0; i < KMAX; ++i) { bv[i] = -INFINITY; bi[i] = 0x7fffffff; }
  for (int ch0 = tid; ch0 < nchunk; ch0 += 4 * 256) {
    float v[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ch = ch0 + u * 256;
      if (ch < nchunk) ld8(lr + ch * 8, v[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ch = ch0 + u * 256;
      if (ch >= nchunk) continue;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        // one compare per element in the RAW domain (the processed value is monotone in the logit; t_safe sits a safe
        // margin below the raw image of tau), the exact processed-domain test only for the few that pass
        if (!(v[u][i] >= t_safe)) continue;
        const int c = ch * 8 + i;
        if (c >= V) continue;
        float x = raw ? v[u][i] : (v[u][i] - mx) - logsum;   // log_softmax (gen:850)
        if (suppress_eos && c == eos) x = -INFINITY;           // MinLength
        x += bias;                                             // + running score (gen:857)
        if (x >= tau && better(x, c, bv[KMAX - 1], bi[KMAX - 1])) {
          bv[KMAX - 1] = x; bi[KMAX - 1] = c;
#pragma unroll
          for (int p = KMAX - 1; p > 0; --p) {
            if (better(bv[p], bi[p], bv[p - 1], bi[p - 1])) {
              const float tv = bv[p]; bv[p] = bv[p - 1]; bv[p - 1] = tv;
              const int ti = bi[p]; bi[p] = bi[p - 1]; bi[p - 1] = ti;
            }
          }
        }
      }
    }
  }
  // k rounds of block-wide arg-best over the threads' list heads
  for (int round = 0; round < k; ++round) {
    sm[tid] = bv[0]; si[tid] = bi[0]; ss[tid] = __int_as_float(tid);
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o && better(sm[tid + o], si[tid + o], sm[tid], si[tid])) { sm[tid] = sm[tid + o]; si[tid] = si[tid + o]; ss[tid] = ss[tid + o]; }
      __syncthreads();
    }
    const int winner = __float_as_int(ss[0]);
    if (tid == 0) { top_val[(size_t)row * k + round] = sm[0]; top_idx[(size_t)row * k + round] = si[0]; }
    if (tid == winner) {
#pragma unroll
      for (int p = 0; p < KMAX - 1; ++p) { bv[p] = bv[p + 1]; bi[p] = bi[p + 1]; }
      bv[KMAX - 1] = -INFINITY; bi[KMAX - 1] = 0x7fffffff;
    }
    __syncthreads();
  }
}
extern "C" int mic_row_lse_topk(int dtype, int R, int V, const void* logits, int ld, int k, int forced_token,
                                int suppress_eos, int eos_token_id, int raw_logits, const float* row_bias, float* top_val,
                                int32_t* top_idx, void* stream) {
  MIC_CHECK(R > 0 && V > 0 && ld >= V && ld % 8 == 0 && k >= 1 && k <= TOPK_WIDE && logits && top_val && top_idx, "mic_row_lse_topk: bad args (k <= 64)");
  dim3 grid(R), block(256);
#define TOPK_LAUNCH(TT, KM) hipLaunchKernelGGL((row_lse_topk_kernel<TT, KM>), grid, block, 0, (hipStream_t)stream, V, (const TT*)logits, ld, k, forced_token, suppress_eos, eos_token_id, raw_logits, row_bias, top_val, top_idx)
  if (dtype == MIC_BF16) { if (k <= 8) TOPK_LAUNCH(uint16_t, 8); else if (k <= 16) TOPK_LAUNCH(uint16_t, 16); else if (k <= 32) TOPK_LAUNCH(uint16_t, 32); else TOPK_LAUNCH(uint16_t, 64); }
  else if (dtype == MIC_F32) { if (k <= 8) TOPK_LAUNCH(float, 8); else if (k <= 16) TOPK_LAUNCH(float, 16); else if (k <= 32) TOPK_LAUNCH(float, 32); else TOPK_LAUNCH(float, 64); }
  else MIC_CHECK(false, "mic_row_lse_topk: bad dtype");
#undef TOPK_LAUNCH
  MIC_LAUNCH_CHECK();
  return MIC_OK;
}

// ------------------------------------------------------------------ per-row lse + top-k from the head GEMM's tile partials
// The LM-head GEMM can emit, per row and 64-column granule, (max, sum exp(x - max)) of the logits it stores
// (mic_gemm_args.rowstat).  Merging the ceil(V / 64) partials of a row gives its log-sum-exp without touching the logits.
// For the top-k let f(x) = ((x - max) - log sum) + bias be the processed value (monotone in x, the arithmetic of
// row_lse_topk_kernel) and tau the kk-th largest granule maximum (kk = k, + 1 when the EOS column — possibly a granule's
// maximum — is not eligible).  At least k eligible elements have f >= f(tau), so the row's top-k (value desc, index asc) is
//   * every element with f > f(tau): all of them sit in the granules A = {f(granule max) > f(tau)}, |A| < kk, and then
//   * elements with f == f(tau) in index order: they sit in A or in E = {f(granule max) == f(tau)}, and since every E granule
//     (but the EOS one) holds at least one, the FIRST kk granules of E by index hold the lowest-indexed k of them.
// So at most 2 kk - 1 granules (<= 33 x 64 logits) are read per row however many granule maxima tie — a randomly
// initialised model's bf16 logits tie in hundreds of granules, which made the "every granule >= tau" scan of the first
// version re-read most of the row k times (113 us in situ against 37 us on spread-out logits).
__device__ __forceinline__ void wave_best(float& v, int& i) {  // all lanes end with the wave's best (value, index)
#define MIC_BEST_STEP(X)                                                          \
  {                                                                               \
    const float ov = X(v);                                                        \
    const int oi = __builtin_bit_cast(int, X(__builtin_bit_cast(float, i)));      \
    if (better(ov, oi, v, i)) { v = ov; i = oi; }                                 \
  }
  MIC_BEST_STEP(dpp_f<DPP_HALF_MIRROR>)
  MIC_BEST_STEP(dpp_f<DPP_XOR1>)
  MIC_BEST_STEP(dpp_f<DPP_XOR2>)
  MIC_BEST_STEP(dpp_f<DPP_ROR8>)
  MIC_BEST_STEP(lane_xor16)
  MIC_BEST_STEP(lane_xor32)
#undef MIC_BEST_STEP
}
template <typename T>
__global__ __launch_bounds__(256) void row_topk_tiles_kernel(int V, const T* __restrict__ logits, int ld, const float2* __restrict__ stat,
                                                            int stat_ld, int ntiles, int k, int suppress_eos, int eos, int raw,
                                                            const float* __restrict__ row_bias, float* __restrict__ top_val,
                                                            int32_t* __restrict__ top_idx) {
  constexpr int TPT = 16;                     // granules per thread: up to 4096 granules of 64 columns (V <= 262 144)
  constexpr int NCAND = 2 * (TOPK_MAX + 1);   // >= 2 kk - 1 candidate granules
  constexpr int CPT = (NCAND + 3) / 4;        // candidate granules per wave
  __shared__ float rv[2][4];
  __shared__ int ri[2][4];
  __shared__ int ecnt[TPT * 4], ebase[TPT * 4 + 1];
  __shared__ int cand[NCAND];
  __shared__ int nA;
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float2* sr = stat + (size_t)row * stat_ld;
  const T* lr = logits + (size_t)row * ld;
  const float bias = row_bias ? row_bias[row] : 0.f;
  if (tid == 0) nA = 0;
  // one barrier per block-wide reduction: partials of the four waves in a slot that alternates between reductions
  int par = 0;
  auto block_best = [&](float& v, int& i) __attribute__((always_inline)) {
    wave_best(v, i);
    if (lane == 0) { rv[par][wv] = v; ri[par][wv] = i; }
    __syncthreads();
    v = rv[par][0]; i = ri[par][0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
      if (better(rv[par][w], ri[par][w], v, i)) { v = rv[par][w]; i = ri[par][w]; }
    par ^= 1;
  };
  float tm[TPT], ts[TPT];
#pragma unroll
  for (int u = 0; u < TPT; ++u) {  // 16 independent 8-byte loads in flight per thread
    const int t = tid + u * 256;
    float2 p = make_float2(-INFINITY, 0.f);
    if (t < ntiles) p = sr[t];
    tm[u] = p.x; ts[u] = p.y;
  }
  float mx = 0.f, logsum = 0.f;
  if (!raw) {
    float m = tm[0];
#pragma unroll
    for (int u = 1; u < TPT; ++u) m = fmaxf(m, tm[u]);
    m = wave_max(m);
    if (lane == 0) rv[par][wv] = m;
    __syncthreads();
    m = fmaxf(fmaxf(rv[par][0], rv[par][1]), fmaxf(rv[par][2], rv[par][3]));
    par ^= 1;
    float sum = 0.f;
    if (m > -INFINITY) {
#pragma unroll
      for (int u = 0; u < TPT; ++u) sum += ts[u] * __expf(tm[u] - m);
    }
    sum = wave_sum(sum);
    if (lane == 0) rv[par][wv] = sum;
    __syncthreads();
    sum = (rv[par][0] + rv[par][1]) + (rv[par][2] + rv[par][3]);
    par ^= 1;
    mx = m; logsum = logf(sum);
  }
  auto proc = [&](float x) __attribute__((always_inline)) -> float {  // row_lse_topk_kernel's arithmetic, in its order
    x = raw ? x : (x - mx) - logsum;
    return x + bias;
  };
  // tau = the kk-th largest granule maximum
  const int kk = k + (suppress_eos ? 1 : 0);
  float tau = INFINITY;
  {
    float mine[TPT];
#pragma unroll
    for (int u = 0; u < TPT; ++u) mine[u] = tm[u];
    for (int round = 0; round < kk; ++round) {
      float bv = mine[0];
      int bi = tid;
#pragma unroll
      for (int u = 1; u < TPT; ++u)
        if (mine[u] > bv) { bv = mine[u]; bi = tid + u * 256; }
      block_best(bv, bi);
      tau = bv;
      if ((bi & 255) == tid) {
#pragma unroll
        for (int u = 0; u < TPT; ++u) if (u == (bi >> 8)) mine[u] = -INFINITY;
      }
    }
  }
  const float ftau = proc(tau);
  // E granules ranked by index = (u, thread) order: counts per (u, wave), exclusive prefix by wave 0, rank by ballot
#pragma unroll
  for (int u = 0; u < TPT; ++u) {
    const bool isE = tid + u * 256 < ntiles && proc(tm[u]) == ftau;
    const unsigned long long em = __ballot(isE);
    if (lane == 0) ecnt[u * 4 + wv] = __popcll(em);
  }
  __syncthreads();
  if (wv == 0) {
    const int c = ecnt[lane];
    int inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int n = __shfl_up(inc, o);
      if (lane >= o) inc += n;
    }
    ebase[lane] = inc - c;
    if (lane == 63) ebase[64] = inc;
  }
  __syncthreads();
  const int nE = min(ebase[64], kk);
#pragma unroll
  for (int u = 0; u < TPT; ++u) {
    const int t = tid + u * 256;
    const float fm = proc(tm[u]);
    const bool isE = t < ntiles && fm == ftau;
    const unsigned long long em = __ballot(isE);
    if (isE) {
      const int rank = ebase[u * 4 + wv] + __popcll(em & ((1ull << lane) - 1ull));
      if (rank < kk) cand[rank] = t;
    }
    if (t < ntiles && fm > ftau) {  // fewer than kk of these
      const int slot = nE + atomicAdd(&nA, 1);
      if (slot < NCAND) cand[slot] = t;
    }
  }
  __syncthreads();
  const int nc = min(nE + nA, NCAND);
  // candidate values: granule q of the list belongs to wave q % 4; every load of a thread is issued before the first use
  float xv[CPT];
  int xc[CPT];
#pragma unroll
  for (int j = 0; j < CPT; ++j) {
    const int q = wv + 4 * j;
    const int c = q < nc ? cand[q] * 64 + lane : V;
    xc[j] = c;
    float x = -INFINITY;
    if (c < V) {
      x = proc(ElemT<T>::ld(lr + c));
      if (suppress_eos && c == eos) x = -INFINITY + bias;
    }
    xv[j] = x;
  }
  // k rounds: the best (processed value, index) that comes after the previous winner in the order (value desc, index asc)
  float pv = INFINITY;
  int pidx = -1;
  for (int round = 0; round < k; ++round) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      const bool after = xv[j] < pv || (xv[j] == pv && xc[j] > pidx);  // strictly behind the previous winner
      if (xc[j] < V && after && better(xv[j], xc[j], bv, bi)) { bv = xv[j]; bi = xc[j]; }
    }
    block_best(bv, bi);
    pv = bv; pidx = bi;
    if (tid == 0) { top_val[(size_t)row * k + round] = pv; top_idx[(size_t)row * k + round] = pidx; }
  }
}
extern "C" int mic_row_topk_tiles(int dtype, int R, int V, const void* logits, int ld, const float* rowstat, int stat_ld, int k,
                                  int suppress_eos, int eos_token_id, int raw_logits, const float* row_bias, float* top_val,
                                  int32_t* top_idx, void* stream) {
  const int ntiles = (V + 63) / 64;
  MIC_CHECK(R > 0 && V > 0 && ld >= V && k >= 1 && k <= TOPK_MAX && logits && rowstat && top_val && top_idx && stat_ld >= ntiles && ntiles <= 4096,
            "mic_row_topk_tiles: bad args (V <= 262144)");
  dim3 grid(R), block(256);
  if (dtype == MIC_BF16)
    hipLaunchKernelGGL(row_topk_tiles_kernel<uint16_t>, grid, block, 0, (hipStream_t)stream, V, (const uint16_t*)logits, ld, (const float2*)rowstat, stat_ld, ntiles, k, suppress_eos, eos_token_id, raw_logits, row_bias, top_val, top_idx);
  else if (dtype == MIC_F32)
    hipLaunchKernelGGL(row_topk_tiles_kernel<float>, grid, block, 0, (hipStream_t)stream, V, (const float*)logits, ld, (const float2*)rowstat, stat_ld, ntiles, k, suppress_eos, eos_token_id, raw_logits, row_bias, top_val, top_idx);
  else MIC_CHECK(false, "mic_row_topk_tiles: bad dtype");
  MIC_LAUNCH_CHECK();
  return MIC_OK;
}

// ------------------------------------------------------------------ one beam_search_body_fn iteration (gen:857-966)
// One block per batch item, the (beam, candidate) pairs over its threads: 128 threads up to K = 8 beams (2K*K <= 128 candidates), 512
// beyond (one pair per thread up to K = 16, four at K = 32).  All arithmetic is fp32 in the reference's operation order so scores are bit-identical to the oracle.
__global__ __launch_bounds__(512) void beam_step_kernel(mic_beam_step_args a) {
  extern __shared__ int32_t lds_i[];
  const int K = a.K, C = 2 * K, L = a.max_len, V = a.V;
  const int b = blockIdx.x, tid = threadIdx.x;
  // loop state on the device (gen:798-820 evaluated by the last block of every step): once the search has ended, further
  // launches of this kernel leave the state alone, so the host may enqueue steps ahead of reading the flag
  if (a.gstate && __hip_atomic_load(a.gstate + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
  int32_t* old_run = lds_i;                 // [K][L]
  int32_t* old_seq = old_run + K * L;       // [K][L]
  int32_t* old_src = old_seq + K * L;       // [K][L]
  int32_t* cand_seq_parent = old_src + K * L;  // [C] parent beam of candidate
  int32_t* cand_tok = cand_seq_parent + C;     // [C]
  float* cand_lp = reinterpret_cast<float*>(cand_tok + C);  // [C] (after the just-finished penalty, gen:890)
  int32_t* cand_fin = reinterpret_cast<int32_t*>(cand_lp + C);  // [C]
  float* fin_score = reinterpret_cast<float*>(cand_fin + C);    // [C] finished-candidate scores (gen:910-919)
  int32_t* run_pick = reinterpret_cast<int32_t*>(fin_score + C);  // [K] candidate index of new running beam nb
  int32_t* mrg_pick = run_pick + K;                               // [K] merged index of new finished slot
  float* old_scores = reinterpret_cast<float*>(mrg_pick + K);     // [K]
  int32_t* old_fin = reinterpret_cast<int32_t*>(old_scores + K);  // [K]
  float* new_run_scores = reinterpret_cast<float*>(old_fin + K);  // [K]

  for (int e = tid; e < K * L; e += blockDim.x) {
    old_run[e] = a.running_seq[(size_t)b * K * L + e];
    old_seq[e] = a.seq[(size_t)b * K * L + e];
    old_src[e] = a.src_row[(size_t)b * K * L + e];
  }
  if (tid < K) { old_scores[tid] = a.scores[b * K + tid]; old_fin[tid] = a.finished[b * K + tid]; }
  __syncthreads();
  // 1. top-2K of the K*2K per-row candidates, ordered (value desc, flat index asc)  (gen:872-874)
  const int N = K * C;
  for (int pr = tid; pr < N; pr += blockDim.x) {  // (512 threads: one pair per thread up to K = 16, four at K = 32)
    const int mybeam = pr / C;
    const float myv = a.cand_val[(size_t)(b * K + mybeam) * C + pr % C];
    const int mytok = a.cand_idx[(size_t)(b * K + mybeam) * C + pr % C];
    const long myflat = (long)mybeam * V + mytok;
    int rank = 0;
    for (int o = 0; o < N; ++o) {
      const int ob = o / C;
      const float ov = a.cand_val[(size_t)(b * K + ob) * C + o % C];
      const long of = (long)ob * V + a.cand_idx[(size_t)(b * K + ob) * C + o % C];
      if (ov > myv || (ov == myv && of < myflat)) ++rank;
    }
    if (rank < C) {
      cand_seq_parent[rank] = mybeam;
      cand_tok[rank] = mytok;
      const int jf = (mytok == a.eos_token_id);
      cand_fin[rank] = jf;                                   // gen:889
      cand_lp[rank] = myv + (float)jf * NEG_BIG;             // gen:890
    }
  }
  __syncthreads();
  // all beams of this item finished (old state) & early stopping  (gen:911-917)
  int all_fin_old = 1;
  for (int kx = 0; kx < K; ++kx) all_fin_old &= (old_fin[kx] != 0);
  const int full = all_fin_old && a.early_stopping;
  // 5. next running beams: top-K of cand_lp (value desc, candidate index asc), stored ascending (gen:895-903)
  // 6. finished-candidate scores (gen:910-919)
  if (tid < C) {
    const float v = cand_lp[tid];
    int rank = 0;
    for (int o = 0; o < C; ++o) { const float ov = cand_lp[o]; if (ov > v || (ov == v && o < tid)) ++rank; }
    if (rank < K) { run_pick[K - 1 - rank] = tid; new_run_scores[K - 1 - rank] = v; }
    float fs = v / powf((float)a.cur_len, a.length_penalty);
    const int add_pen = (!cand_fin[tid]) || full;
    fs += (float)add_pen * NEG_BIG;
    fin_score[tid] = fs;
  }
  __syncthreads();
  // 7. merge [old finished (K) | candidates (2K)], top-K (value desc, merged index asc), stored ascending (gen:925-940)
  if (tid < K + C) {
    const float v = tid < K ? old_scores[tid] : fin_score[tid - K];
    int rank = 0;
    for (int o = 0; o < K + C; ++o) { const float ov = o < K ? old_scores[o] : fin_score[o - K]; if (ov > v || (ov == v && o < tid)) ++rank; }
    if (rank < K) mrg_pick[K - 1 - rank] = tid;
  }
  __syncthreads();
  // write back: running beams
  for (int e = tid; e < K * L; e += blockDim.x) {
    const int nb = e / L, pos = e % L;
    const int c = run_pick[nb];
    const int pb = cand_seq_parent[c];
    a.running_seq[(size_t)b * K * L + e] = (pos == a.cur_len) ? cand_tok[c] : old_run[pb * L + pos];
    // slot ownership: inherit the parent's history, own the next slot (slot index = position of the fed token)
    a.src_row[(size_t)b * K * L + e] = (pos == a.cur_len) ? (b * K + nb) : old_src[pb * L + pos];
    const int mi = mrg_pick[nb];
    int sv;
    if (mi < K) sv = old_seq[mi * L + pos];
    else { const int cc = mi - K; sv = (pos == a.cur_len) ? cand_tok[cc] : old_run[cand_seq_parent[cc] * L + pos]; }
    a.seq[(size_t)b * K * L + e] = sv;
  }
  if (tid < K) {
    const int c = run_pick[tid];
    a.running_scores[b * K + tid] = new_run_scores[tid];
    a.next_token[b * K + tid] = cand_tok[c];
    const int mi = mrg_pick[tid];
    a.scores[b * K + tid] = mi < K ? old_scores[mi] : fin_score[mi - K];
    a.finished[b * K + tid] = mi < K ? old_fin[mi] : cand_fin[mi - K];
  }
  __syncthreads();
  // loop-condition inputs on the NEW state (gen:798-820), per item
  if (tid == 0) {
    int all_fin = 1; float mn = INFINITY;
    float ns[TOPK_WIDE / 2]; int nf[TOPK_WIDE / 2];
    for (int kx = 0; kx < K; ++kx) {
      const int mi = mrg_pick[kx];
      ns[kx] = mi < K ? old_scores[mi] : fin_score[mi - K];
      nf[kx] = mi < K ? old_fin[mi] : cand_fin[mi - K];
      all_fin &= (nf[kx] != 0);
      mn = fminf(mn, ns[kx]);
    }
    const float best_running = new_run_scores[K - 1] / powf((float)L, a.length_penalty);
    int improve = 1;
    for (int kx = 0; kx < K; ++kx) { const float worst = nf[kx] ? mn : NEG_BIG; improve &= (worst < best_running); }
    a.flags[b * 2 + 0] = all_fin;
    a.flags[b * 2 + 1] = improve;
    if (a.gstate) {
      // gstate: [0] items with every beam finished, [1] items that can still improve, [2] arrival ticket, [3] done, [4] steps taken
      if (all_fin) atomicAdd(a.gstate + 0, 1);
      if (improve) atomicAdd(a.gstate + 1, 1);
      __threadfence();
      if (atomicAdd(a.gstate + 2, 1) == a.B - 1) {  // last item of this step
        __threadfence();
        const int nfin = __hip_atomic_load(a.gstate + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int nimp = __hip_atomic_load(a.gstate + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int stop = (nfin == a.B && a.early_stopping) || nimp != a.B || a.cur_len + 1 >= L;
        a.gstate[0] = 0; a.gstate[1] = 0; a.gstate[2] = 0;
        a.gstate[4] += 1;
        __threadfence();
        if (stop) __hip_atomic_store(a.gstate + 3, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}
extern "C" int mic_beam_step(const mic_beam_step_args* a, void* stream) {
  MIC_CHECK(a && a->B > 0 && a->K >= 1 && 2 * a->K <= TOPK_WIDE && a->max_len > 1 && a->cur_len >= 1 && a->cur_len < a->max_len,
            "mic_beam_step: bad shape (K <= 32 supported: per-row candidates come from mic_row_lse_topk with k = 2K <= 64)");
  MIC_CHECK(a->cand_val && a->cand_idx && a->running_seq && a->running_scores && a->seq && a->scores && a->finished && a->src_row && a->next_token && a->flags, "mic_beam_step: null pointer");
  const size_t lds = (size_t)(3 * a->K * a->max_len + 8 * 2 * a->K + 8 * a->K) * 4;
  MIC_CHECK(lds <= 65536, "mic_beam_step: max_len too large for the LDS staging");
  const int threads = 2 * a->K * a->K <= 128 ? 128 : 512;  // one thread per (beam, candidate) pair
  hipLaunchKernelGGL(beam_step_kernel, dim3(a->B), dim3(threads), lds, (hipStream_t)stream, *a);
  MIC_LAUNCH_CHECK();
  return MIC_OK;
}

// ------------------------------------------------------------------ greedy step (gen:499-512)
__global__ void greedy_step_kernel(int B, int max_len, int cur_len, int eos, int pad, const int32_t* __restrict__ top_idx,
                                   int ld_top, int32_t* sequences, int32_t* finished, int32_t* next_token) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int tok = top_idx[(size_t)b * ld_top];
  const int fin = finished[b] | (tok == eos);   // gen:501-503
  tok = fin ? pad : tok;                        // gen:504-507: the EOS step itself writes PAD
  finished[b] = fin;
  sequences[(size_t)b * max_len + cur_len] = tok;
  next_token[b] = tok;
}
extern "C" int mic_greedy_step(int B, int max_len, int cur_len, int eos_token_id, int pad_token_id, const int32_t* top_idx,
                               int ld_top, int32_t* sequences, int32_t* finished, int32_t* next_token, void* stream) {
  MIC_CHECK(B > 0 && cur_len >= 1 && cur_len < max_len && top_idx && sequences && finished && next_token, "mic_greedy_step: bad args");
  hipLaunchKernelGGL(greedy_step_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, B, max_len, cur_len, eos_token_id, pad_token_id, top_idx, ld_top, sequences, finished, next_token);
  MIC_LAUNCH_CHECK();
  return MIC_OK;
}

__global__ void fill_i32_kernel(int32_t* p, int n, int v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// ------------------------------------------------------------------ sampling (gen:537-663)
// next_token = jax.random.categorical(prng_key, logits) (gen:625-627) = argmax(logits + Gumbel noise), with the noise
// generated exactly like jax 0.2.16 does it [restated from the published algorithm; pinned on the Random123 / JAX
// known-answer vectors in tests/test_oracle_cpu.py]:
//   bits    = threefry2x32(key, counters): the [R*V] counter array (padded to even) is split in halves x0 | x1, the two
//             output words are concatenated -> element e < h uses word 0 of (e, e+h), element e >= h word 1 of (e-h, e);
//   uniform = max(tiny, (bitcast(bits >> 9 | 0x3f800000) - 1) * (1 - tiny) + tiny)        (random.uniform, minval=tiny)
//   gumbel  = -log(-log(uniform)).
// One block per row; ties resolve to the lowest index (jnp.argmax).
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
__device__ __forceinline__ void threefry2x32(uint32_t k0, uint32_t k1, uint32_t& x0, uint32_t& x1) {
  const uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
  const int R[2][4] = {{13, 15, 26, 6}, {17, 29, 16, 24}};
  x0 += ks[0]; x1 += ks[1];
#pragma unroll
  for (int g = 0; g < 5; ++g) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { x0 += x1; x1 = rotl32(x1, R[g & 1][i]); x1 ^= x0; }
    x0 += ks[(g + 1) % 3];
    x1 += ks[(g + 2) % 3] + (uint32_t)(g + 1);
  }
}

template <typename T>
__global__ __launch_bounds__(1024) void sample_rows_kernel(int R, int V, const T* __restrict__ logits, int ld, uint32_t k0,
                                                           uint32_t k1, float temperature, int suppress_eos, int eos,
                                                           const float* __restrict__ min_keep,
                                                           const int32_t* __restrict__ tie_limit, int32_t* __restrict__ out) {
  const int row = blockIdx.x;
  const uint32_t n = (uint32_t)R * (uint32_t)V, h = (n + 1u) >> 1;
  const T* x = logits + (size_t)row * ld;
  const float thr = min_keep ? min_keep[row] : -INFINITY;
  const int lim = tie_limit ? tie_limit[row] : 0x7fffffff;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = threadIdx.x; i < V; i += blockDim.x) {
    const uint32_t e = (uint32_t)row * (uint32_t)V + (uint32_t)i;
    uint32_t c0, c1;
    if (e < h) { c0 = e; c1 = e + h < n ? e + h : 0u; } else { c0 = e - h; c1 = e; }
    threefry2x32(k0, k1, c0, c1);
    const uint32_t bits = e < h ? c0 : c1;
    const float f = __uint_as_float((bits >> 9) | 0x3f800000u) - 1.0f;
    const float tiny = 1.17549435e-38f;
    const float u = fmaxf(tiny, f * (1.0f - tiny) + tiny);
    const float g = -logf(-logf(u));
    float v = ElemT<T>::ld(x + i);
    if (temperature != 1.0f) v = v / temperature;
    if ((suppress_eos && i == eos) || v < thr || (v == thr && i >= lim)) v = -INFINITY;
    v += g;
    if (v > best || (v == best && i < bi)) { best = v; bi = i; }
  }
  // block argmax, lowest index on ties
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  __shared__ float sv[16];
  __shared__ int si[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { sv[wave] = best; si[wave] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w)
      if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
    out[row] = bi;
  }
}

extern "C" int mic_sample_rows(int dtype, int R, int V, const void* logits, int ld, uint32_t key0, uint32_t key1,
                               float temperature, int forced_token, int suppress_eos, int eos_token_id,
                               const float* min_keep, const int32_t* tie_limit, int32_t* out_idx, void* stream) {
  MIC_CHECK(R > 0 && V > 0 && logits && out_idx && temperature > 0.f, "mic_sample_rows: bad args");
  MIC_CHECK((uint64_t)R * (uint64_t)V < (1ull << 32), "mic_sample_rows: R*V must fit the 32-bit threefry counter");
  if (forced_token >= 0) {  // ForcedBOS / ForcedEOS leave one finite logit: the categorical draw is that token
    hipLaunchKernelGGL(fill_i32_kernel, dim3((R + 255) / 256), dim3(256), 0, (hipStream_t)stream, out_idx, R, forced_token);
    MIC_LAUNCH_CHECK();
    return MIC_OK;
  }
  dim3 grid(R), block(1024);
  if (dtype == MIC_BF16)
    hipLaunchKernelGGL(sample_rows_kernel<uint16_t>, grid, block, 0, (hipStream_t)stream, R, V, (const uint16_t*)logits, ld, key0, key1, temperature, suppress_eos, eos_token_id, min_keep, tie_limit, out_idx);
  else if (dtype == MIC_F32)
    hipLaunchKernelGGL(sample_rows_kernel<float>, grid, block, 0, (hipStream_t)stream, R, V, (const float*)logits, ld, key0, key1, temperature, suppress_eos, eos_token_id, min_keep, tie_limit, out_idx);
  else MIC_CHECK(false, "mic_sample_rows: bad dtype");
  MIC_LAUNCH_CHECK();
  return MIC_OK;
}

// ------------------------------------------------------------------ top-k / top-p warpers (gen:338-366)
// FlaxTopKLogitsWarper keeps the k largest scores (lax.top_k: ties go to the lower index); FlaxTopPLogitsWarper sorts the
// (already top-k-filtered) scores in descending order and keeps position j iff j == 0 or the softmax mass of the
// positions before it is < top_p.  Neither needs a sort: both are "everything above a threshold value, plus the first
// few (by index) of the entries equal to it" — found per row by a 3-pass radix descent over order-preserving 32-bit keys
// (11 + 11 + 10 bits; LDS histograms of counts for top-k, of fixed-point masses for top-p so the sums are exact and
// order-independent).  Output per row: thr (fp32 value) and tie_limit: keep v > thr, or v == thr and index < tie_limit.
__device__ __forceinline__ uint32_t okey(float v) {  // ascending order-preserving key
  const uint32_t u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float okey_inv(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

#define WARP_T 1024
template <typename T>
struct RowView {
  const T* x; int V; float temperature; int suppress_eos, eos;
  float thr_k; int lim_k;  // result of the top-k stage (thr_k = -inf: no filter yet)
  __device__ __forceinline__ float at(int i) const {
    float v = ElemT<T>::ld(x + i);
    if (temperature != 1.0f) v = v / temperature;
    if (suppress_eos && i == eos) v = -INFINITY;
    if (v < thr_k || (v == thr_k && i >= lim_k)) v = -INFINITY;
    return v;
  }
};

// block-wide inclusive scan over WARP_T threads of a 64-bit value (wave shuffles + one LDS hop); returns inclusive sum
__device__ __forceinline__ unsigned long long block_scan_u64(unsigned long long v, unsigned long long* wsum /*[16]*/) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned long long n = __shfl_up(v, o, 64);
    if (lane >= o) v += n;
  }
  __syncthreads();
  if (lane == 63) wsum[wave] = v;
  __syncthreads();
  unsigned long long base = 0;
  for (int w = 0; w < wave; ++w) base += wsum[w];
  return v + base;
}

// One radix-descent stage: hist[2048] (64-bit) holds the weight of every bin among the entries matching (key & pmask) ==
// prefix; walking bins from the highest key down, finds the bin where the running weight first reaches `target`
// (running_before < target <= running_before + hist[bin]); returns bin and the weight still needed inside it.
__device__ __forceinline__ void pick_bin(const unsigned long long* hist, int nbins, unsigned long long target,
                                         unsigned long long* wsum, int* out_bin, unsigned long long* out_need) {
  // thread t covers descending bins nbins-1-2t and nbins-2-2t
  const int b0 = nbins - 1 - 2 * (int)threadIdx.x, b1 = b0 - 1;
  const unsigned long long h0 = b0 >= 0 ? hist[b0] : 0ull, h1 = b1 >= 0 ? hist[b1] : 0ull;
  const unsigned long long incl = block_scan_u64(h0 + h1, wsum);
  const unsigned long long before = incl - (h0 + h1);
  if (before < target && target <= incl) {
    if (target <= before + h0) { *out_bin = b0; *out_need = target - before; }
    else { *out_bin = b1; *out_need = target - before - h0; }
  }
  __syncthreads();
}

template <typename T>
__global__ __launch_bounds__(WARP_T) void warp_threshold_kernel(int V, const T* __restrict__ logits, int ld, float temperature,
                                                                int suppress_eos, int eos, int top_k, float top_p,
                                                                float* __restrict__ thr_out, int32_t* __restrict__ lim_out) {
  __shared__ unsigned long long hist[2048];
  __shared__ unsigned long long wsum[16];
  __shared__ int s_bin;
  __shared__ unsigned long long s_need;
  __shared__ float s_red[16];
  __shared__ int s_lim;
  const int row = blockIdx.x, tid = threadIdx.x;
  RowView<T> rv{logits + (size_t)row * ld, V, temperature, suppress_eos, eos, -INFINITY, 0x7fffffff};
  const int shifts[3] = {21, 10, 0}, nb[3] = {2048, 2048, 1024};

  for (int stage = 0; stage < 2; ++stage) {  // 0: top-k (weights = counts), 1: top-p (weights = fixed-point masses)
    if (stage == 0 && (top_k <= 0 || top_k >= V)) continue;
    if (stage == 1 && !(top_p < 1.0f)) continue;
    float mx = -INFINITY;
    unsigned long long target;
    if (stage == 0) {
      target = (unsigned long long)top_k;
    } else {
      // max and partition function over the surviving entries
      for (int i = tid; i < V; i += WARP_T) mx = fmaxf(mx, rv.at(i));
      mx = wave_max(mx);
      if ((tid & 63) == 0) s_red[tid >> 6] = mx;
      __syncthreads();
      mx = s_red[0];
      for (int w = 1; w < WARP_T / 64; ++w) mx = fmaxf(mx, s_red[w]);
      unsigned long long z = 0;
      for (int i = tid; i < V; i += WARP_T) z += (unsigned long long)(expf(rv.at(i) - mx) * 1099511627776.0f);  // 2^40
      const unsigned long long tot = block_scan_u64(z, wsum);
      __syncthreads();
      if (tid == WARP_T - 1) s_need = tot;
      __syncthreads();
      const double t = (double)top_p * (double)s_need;
      target = (unsigned long long)t;
      if ((double)target < t) target += 1;   // keep position j iff mass_before(j) < top_p * Z  <=>  first reach of ceil(target)
      if (target == 0) target = 1;           // min_tokens_to_keep = 1
      __syncthreads();
    }
    uint32_t prefix = 0, pmask = 0;
    unsigned long long need = target;
    for (int ps = 0; ps < 3; ++ps) {
      for (int b = tid; b < 2048; b += WARP_T) hist[b] = 0ull;
      __syncthreads();
      const uint32_t bm = (uint32_t)nb[ps] - 1u;
      for (int i = tid; i < V; i += WARP_T) {
        const float v = rv.at(i);
        const uint32_t k = okey(v);
        if ((k & pmask) == prefix && v > -INFINITY) {
          const unsigned long long w = stage == 0 ? 1ull : (unsigned long long)(expf(v - mx) * 1099511627776.0f);
          atomicAdd(&hist[(k >> shifts[ps]) & bm], w);
        }
      }
      __syncthreads();
      if (tid == 0) { s_bin = -1; s_need = 0; }
      __syncthreads();
      pick_bin(hist, nb[ps], need, wsum, &s_bin, &s_need);
      const int bin = s_bin;
      if (bin < 0) break;  // fewer surviving entries than requested: keep them all
      need = s_need;
      prefix |= (uint32_t)bin << shifts[ps];
      pmask |= bm << shifts[ps];
      __syncthreads();
    }
    if (pmask != 0xffffffffu) continue;  // nothing to cut
    const float thr = okey_inv(prefix);
    // entries equal to thr: keep the first n_keep by index
    unsigned long long n_keep;
    if (stage == 0) n_keep = need;
    else {
      const unsigned long long m_eq = (unsigned long long)(expf(thr - mx) * 1099511627776.0f);
      n_keep = m_eq ? (need + m_eq - 1) / m_eq : 0x7fffffffull;
    }
    // index of the n_keep-th equal entry: per-thread contiguous segments, block scan of the counts, then a local walk
    const int S = (V + WARP_T - 1) / WARP_T, i0 = tid * S, i1 = min(V, i0 + S);
    unsigned long long cnt = 0;
    for (int i = i0; i < i1; ++i) cnt += rv.at(i) == thr ? 1ull : 0ull;
    const unsigned long long incl = block_scan_u64(cnt, wsum);
    if (tid == 0) s_lim = 0x7fffffff;
    __syncthreads();
    const unsigned long long before = incl - cnt;
    if (before < n_keep && n_keep <= incl) {
      unsigned long long c = before;
      for (int i = i0; i < i1; ++i)
        if (rv.at(i) == thr && ++c == n_keep) { s_lim = i + 1; break; }
    }
    __syncthreads();
    rv.thr_k = thr;
    rv.lim_k = s_lim;
    __syncthreads();
  }
  if (tid == 0) { thr_out[row] = rv.thr_k; lim_out[row] = rv.lim_k; }
}

extern "C" int mic_warp_thresholds(int dtype, int R, int V, const void* logits, int ld, float temperature, int suppress_eos,
                                   int eos_token_id, int top_k, float top_p, float* thr, int32_t* tie_limit, void* stream) {
  MIC_CHECK(R > 0 && V > 0 && logits && thr && tie_limit && temperature > 0.f && top_p > 0.f, "mic_warp_thresholds: bad args");
  dim3 grid(R), block(WARP_T);
  if (dtype == MIC_BF16)
    hipLaunchKernelGGL(warp_threshold_kernel<uint16_t>, grid, block, 0, (hipStream_t)stream, V, (const uint16_t*)logits, ld, temperature, suppress_eos, eos_token_id, top_k, top_p, thr, tie_limit);
  else if (dtype == MIC_F32)
    hipLaunchKernelGGL(warp_threshold_kernel<float>, grid, block, 0, (hipStream_t)stream, V, (const float*)logits, ld, temperature, suppress_eos, eos_token_id, top_k, top_p, thr, tie_limit);
  else MIC_CHECK(false, "mic_warp_thresholds: bad dtype");
  MIC_LAUNCH_CHECK();
  return MIC_OK;
}
