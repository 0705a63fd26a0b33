// fp8.hip — per-tensor-scaled OCP fp8 quantisation for the fp8 GEMM path (BASELINE configs[4]: QKV / FFN projections of the
// ViT and the mBART decoder in fp8, fp32 accumulate).  The reference has no counterpart (its compute dtypes are fp32 / fp16 /
// bf16, main.py:96-101, applied at main.py:425); the recipe is the usual one for fp8 training: e4m3 for activations and
// weights, e5m2 for gradients, one scale per tensor taken from that tensor's CURRENT absolute maximum.
//
//   mic_fp8_amax      state[0] = max |x| over the valid region (atomic max on the float bits; caller zeroes state)
//   mic_fp8_quantize  scale = FMAX / amax;  q[r][c] = fp8(x[r][c] * scale)  (row-major, k-contiguous for  x W^T  /  dy W)
//                     qT[c][r] = the same bytes transposed, rows zero-padded to rows_pad (k-contiguous for  dy^T x : both
//                     operands of the weight-gradient GEMM reduce over the row index);  state[1] = amax / FMAX, the
//                     dequantisation factor the GEMM epilogue multiplies back (mic_gemm_args.a_scale_inv / b_scale_inv).
// Both are table-driven (up to 8 tensors per launch): the 84 weight matrices are re-quantised once per optimizer step in a
// handful of launches.  HBM-bound: 2 B read twice, 1 B written twice per element.
//
// Delayed scaling (the usual production recipe: the scale of step t comes from the amax observed at step t-1): an item with
// `amax_next` set is quantised with the scale already in state[0] — no mic_fp8_amax pass, the tensor is read ONCE — while the
// quantiser records this pass's max |x| in the tensor's table of partial maxima amax_next[0..1023] (one atomic per 64x64 tile,
// spread over the table); mic_fp8_roll_amax reduces the table into state[0] at the start of the next pass.  Values beyond the
// old amax saturate at +-FMAX (the clamp below).
#include "common.h"

#define QMAX_ITEMS 8
struct QItem {
  const uint16_t* src; int ld;       // bf16 [rows][ld]
  int rows, cols, rows_pad;
  uint8_t* q; int ldq;               // [rows][ldq] or null
  uint8_t* qT; int ldqT;             // [cols][ldqT >= rows_pad] or null
  float* state;                      // [0] amax, [1] 1/scale
  float* amax_next;                  // delayed scaling: [FP8_AMAX_PARTIALS] partial maxima of THIS pass (atomic max) or null
  int fmt;                           // MIC_E4M3 / MIC_E5M2
  int tiles_c, block_begin;
};
struct QTable { int count; int total_blocks; QItem it[QMAX_ITEMS]; };

__device__ __forceinline__ const QItem& pick_item(const QTable& t, int bid, int& local) {
  int pi = 0;
#pragma unroll
  for (int i = 1; i < QMAX_ITEMS; ++i)
    if (i < t.count && bid >= t.it[i].block_begin) pi = i;
  local = bid - t.it[pi].block_begin;
  return t.it[pi];
}

// 64 x 64 tiles, thread t covers 8 columns (t & 7) of rows (t >> 3) and (t >> 3) + 32.  A block walks a contiguous run of tiles
// and issues ONE atomic per tensor it touched (block-wide maximum through LDS): a first version with one atomic per wave and
// tile put 4096 atomics on one address for a 4096 x 1024 tensor and took 90 us (L2 serialises them at ~12 ns each).
__global__ __launch_bounds__(256) void fp8_amax_kernel(QTable tab, int tiles_per_block) {
  __shared__ float red[4];
  const int tid = threadIdx.x;
  const int b0 = blockIdx.x * tiles_per_block, b1 = min(tab.total_blocks, b0 + tiles_per_block);
  float m = 0.f;
  float* cur_state = nullptr;
  auto flush = [&]() {
    m = wave_max(m);
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) {
      const float v = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
      if (v > 0.f && cur_state) atomicMax(reinterpret_cast<int*>(cur_state), __float_as_int(v));  // non-negative floats order as ints
    }
    __syncthreads();
    m = 0.f;
  };
  for (int bid = b0; bid < b1; ++bid) {
    int local;
    const QItem& I = pick_item(tab, bid, local);
    if (I.state != cur_state) {
      if (cur_state) flush();
      cur_state = I.state;
    }
    const int tr = local / I.tiles_c, tc = local % I.tiles_c;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = tr * 64 + h * 32 + (tid >> 3), c = tc * 64 + (tid & 7) * 8;
      if (r < I.rows && c < I.cols) {
        float v[8];
        ld8(I.src + (size_t)r * I.ld + c, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) m = fmaxf(m, (c + e < I.cols) ? fabsf(v[e]) : 0.f);
      }
    }
  }
  if (cur_state) flush();
}

// 128 (rows) x 64 (cols) tiles: thread t covers 8 columns (t & 7) of rows (t >> 3) + 32 h, h < 4 — four 16-B loads in flight per
// thread, 128-B row segments in, 64-B segments of q and full 128-B lines of qT out (the 64 x 64 version measured 11 us for a
// 4096 x 1024 tensor, 1.5 TB/s: too little work per block).
__global__ __launch_bounds__(256) void fp8_quantize_kernel(QTable tab) {
  __shared__ __attribute__((aligned(16))) uint8_t tile[64][144];  // [col][row], 144-byte pitch: 16-B aligned rows of 128 bytes
  __shared__ float red[4];
  int local;
  const QItem& I = pick_item(tab, blockIdx.x, local);
  const int tr = local / I.tiles_c, tc = local % I.tiles_c;
  const int tid = threadIdx.x;
  const float fmax = I.fmt == MIC_E4M3 ? 448.0f : 57344.0f;
  const float amax = I.state[0];
  const float scale = amax > 0.f ? fmax / amax : 1.0f;
  if (local == 0 && tid == 0) I.state[1] = amax > 0.f ? amax / fmax : 1.0f;
  float m = 0.f;
  const int cl = (tid & 7) * 8, c = tc * 64 + cl;
  float v[4][8];
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    const int r = tr * 128 + h * 32 + (tid >> 3);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[h][e] = 0.f;
    if (r < I.rows && c < I.cols) ld8(I.src + (size_t)r * I.ld + c, v[h]);
  }
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    const int rl = h * 32 + (tid >> 3), r = tr * 128 + rl;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = (c + e < I.cols) ? v[h][e] : 0.f;
      m = fmaxf(m, fabsf(x));
      v[h][e] = fminf(fmaxf(x * scale, -fmax), fmax);
    }
    const uint32_t lo = cvt4_fp8(v[h], I.fmt), hi = cvt4_fp8(v[h] + 4, I.fmt);
    if (I.q && r < I.rows && c < I.cols) *reinterpret_cast<uint2*>(I.q + (size_t)r * I.ldq + c) = make_uint2(lo, hi);
    if (I.qT) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        tile[cl + e][rl] = (uint8_t)(lo >> (8 * e));
        tile[cl + 4 + e][rl] = (uint8_t)(hi >> (8 * e));
      }
    }
  }
  if (I.amax_next) {  // delayed scaling: this tile's max |x| into the tensor's partial-maximum table
    m = wave_max(m);
    if ((tid & 63) == 0) red[tid >> 6] = m;
  }
  if (I.qT || I.amax_next) __syncthreads();
  if (I.amax_next && tid == 0) {
    const float x = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (x > 0.f) atomicMax(reinterpret_cast<int*>(I.amax_next + (local & (FP8_AMAX_PARTIALS - 1))), __float_as_int(x));
  }
  if (!I.qT) return;
  // transposed store: thread t writes rows [32 (t & 3), +32) of column t >> 2 as two 16-B pieces
  const int ct = tid >> 2, r0 = (tid & 3) * 32;
  const int cg = tc * 64 + ct;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int r = tr * 128 + r0 + u * 16;
    if (cg < I.cols && r < I.rows_pad)
      *reinterpret_cast<uint4*>(I.qT + (size_t)cg * I.ldqT + r) = *reinterpret_cast<const uint4*>(&tile[ct][r0 + u * 16]);
  }
}

// start of a pass under delayed scaling: slot i's amax becomes the maximum of the partials recorded during the previous pass
// (if any were), and the partials are cleared.  One wave per slot.
__global__ __launch_bounds__(64) void fp8_roll_kernel(float* state, int stride, float* partials, int count) {
  const int i = blockIdx.x, lane = threadIdx.x;
  if (i >= count) return;
  float* p = partials + (size_t)i * FP8_AMAX_PARTIALS;
  float m = 0.f;
  for (int j = lane; j < FP8_AMAX_PARTIALS; j += 64) {
    m = fmaxf(m, p[j]);
    p[j] = 0.f;
  }
  m = wave_max(m);
  if (lane == 0 && m > 0.f) state[(size_t)i * stride] = m;
}

static int build_table(const mic_fp8_item* items, int n, QTable& t, bool need_out) {
  const int tile_rows = need_out ? 128 : 64;  // the quantiser walks 128 x 64 tiles, the amax pass 64 x 64
  t.count = n;
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    const mic_fp8_item& s = items[i];
    MIC_CHECK(s.src && s.state && s.rows > 0 && s.cols > 0 && s.ld >= s.cols, "mic_fp8: bad item %d", i);
    MIC_CHECK(s.cols % 8 == 0 && s.ld % 8 == 0 && ((uintptr_t)s.src & 15) == 0, "mic_fp8: cols / ld must be multiples of 8, src 16-B aligned");
    MIC_CHECK(s.fmt == MIC_E4M3 || s.fmt == MIC_E5M2, "mic_fp8: bad format");
    const int rows_pad = s.rows_pad > s.rows ? s.rows_pad : s.rows;
    if (need_out) {
      MIC_CHECK(s.q || s.qT, "mic_fp8_quantize: item %d has no output", i);
      MIC_CHECK(!s.q || (s.ldq >= s.cols && s.ldq % 8 == 0 && ((uintptr_t)s.q & 7) == 0), "mic_fp8_quantize: bad q layout");
      MIC_CHECK(!s.qT || (rows_pad % 16 == 0 && s.ldqT >= rows_pad && s.ldqT % 16 == 0 && ((uintptr_t)s.qT & 15) == 0),
                "mic_fp8_quantize: qT needs rows_pad %% 16 == 0, ldqT %% 16 == 0");
    }
    QItem& d = t.it[i];
    d.src = (const uint16_t*)s.src; d.ld = s.ld; d.rows = s.rows; d.cols = s.cols; d.rows_pad = rows_pad;
    d.q = (uint8_t*)s.q; d.ldq = s.ldq; d.qT = (uint8_t*)s.qT; d.ldqT = s.ldqT; d.state = s.state; d.amax_next = s.amax_next; d.fmt = s.fmt;
    d.tiles_c = (s.cols + 63) / 64;
    d.block_begin = blocks;
    blocks += ((need_out && s.qT ? rows_pad : s.rows) + tile_rows - 1) / tile_rows * d.tiles_c;
  }
  t.total_blocks = blocks;
  return MIC_OK;
}

extern "C" int mic_fp8_amax(const mic_fp8_item* items, int count, void* stream) {
  MIC_CHECK(items && count >= 1, "mic_fp8_amax: bad args");
  for (int i = 0; i < count; i += QMAX_ITEMS) {
    QTable t;
    const int n = count - i < QMAX_ITEMS ? count - i : QMAX_ITEMS;
    if (int rc = build_table(items + i, n, t, false)) return rc;
    const int per = (t.total_blocks + 511) / 512;  // at most 512 blocks (two per CU), each a contiguous run of tiles
    hipLaunchKernelGGL(fp8_amax_kernel, dim3((t.total_blocks + per - 1) / per), dim3(256), 0, (hipStream_t)stream, t, per);
  }
  MIC_LAUNCH_CHECK();
  return MIC_OK;
}

extern "C" int mic_fp8_quantize(const mic_fp8_item* items, int count, void* stream) {
  MIC_CHECK(items && count >= 1, "mic_fp8_quantize: bad args");
  for (int i = 0; i < count; i += QMAX_ITEMS) {
    QTable t;
    const int n = count - i < QMAX_ITEMS ? count - i : QMAX_ITEMS;
    if (int rc = build_table(items + i, n, t, true)) return rc;
    hipLaunchKernelGGL(fp8_quantize_kernel, dim3(t.total_blocks), dim3(256), 0, (hipStream_t)stream, t);
  }
  MIC_LAUNCH_CHECK();
  return MIC_OK;
}

extern "C" int mic_fp8_amax_partials(void) { return FP8_AMAX_PARTIALS; }

extern "C" int mic_fp8_roll_amax(float* state, int stride_floats, float* partials, int count, void* stream) {
  MIC_CHECK(state && partials && stride_floats >= 1 && count >= 1, "mic_fp8_roll_amax: bad args");
  hipLaunchKernelGGL(fp8_roll_kernel, dim3(count), dim3(64), 0, (hipStream_t)stream, state, stride_floats, partials, count);
  MIC_LAUNCH_CHECK();
  return MIC_OK;
}
