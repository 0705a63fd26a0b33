// gemm_w4.hip — the 256x256 bf16 tile on FOUR waves, one per SIMD, each with the whole 512-register file: wave tile 128 x 128
// (16 accumulator blocks of 32x32 = 256 AGPRs), ONE barrier per 64-k tile.  Where gemm_phased.hip lets the two waves of a SIMD take
// turns at the matrix pipe (eight barrier hand-offs per 64-k tile), here every wave feeds its own pipe from one instruction stream
// with the side work pinned into the gaps between its 64 MFMAs per tile (the structure hipBLASLt's MT256x256x64 kernels have, read
// off their disassembly: four waves, 128 x 128 per wave, operands direct-to-LDS, one request every few MFMAs).  Takes the
// single-problem NT launches with a bare epilogue by default (LM head, all-layer cross k/v projection; MIC_GEMM_W4=0: four-phase
// kernel): 1305 TF/s at 4096^3 (1145 there), 202 k cycles against 252 k; 6-10 % faster on the LM-head shapes with their epilogues.
//
//   * LDS: two slots of 64 KiB, tile T in slot T%2 = [A rows 0-127 | A rows 128-255 | B cols 0-127 | B cols 128-255], each a
//     k-contiguous image of 128 rows x 64 k (128-B rows, 16-B chunk position p of row R holds source chunk p ^ ((R>>1)&7):
//     conflict-free for the 16-lane groups of ds_read_b128, see read_frag; SQ_LDS_BANK_CONFLICT reads 0).
//   * Operands by LDS-DMA: buffer_load_dwordx4 ... lds with a scalar resource, a 32-bit lane offset and a scalar k offset (nothing
//     to update per tile in vector registers), 1-KiB pieces of 8 rows x 128 B = whole lines; LDS-DMA writes lane-linear, so the
//     swizzle sits on the SOURCE address.  An LDS-DMA piece costs 60-100 cycles of ISSUE time and nobody else feeds this SIMD's
//     matrix pipe meanwhile: ONE piece in every other MFMA gap, never two in a row (eight in a row per 32-k step was this file's
//     first build: 930-1020 TF/s at deep K).  Register-staged operands (global_load + ds_write_b128, two builds of this file)
//     ran the K loop in 12 % fewer cycles than the four-phase kernel and at a 10 % lower clock — equal wall time; the DMA path
//     needs no vector registers for data in flight and is the one that is faster on the clock that power allows (DESIGN.md
//     section 3, "Round 4").
//   * A whole tile's fragments live in registers (128 VGPRs), re-read one quarter (16 k) behind their last use: quarter 0 of tile T
//     reads T's last quarter, then {lgkmcnt(0), vmcnt(0)} s_barrier — tile T+1 has landed and T's slot is dead; quarters 1-3 read
//     tile T+1's first three quarters; the 16 pieces of tile T+2 go out in quarters 1 and 2 into the dead slot.  RAW on DMA data:
//     the issuing wave's vmcnt, then a barrier the reader has passed.  WAR: the slot is refilled behind the barrier that follows
//     its last reads.  Past the end of K the requests go through a resource of zero records (dropped by the hardware).
//   * One block per tile, dispatched by the hardware.  Persistent blocks drawing tiles from per-XCD counters were built and measured
//     level with this form (-2 % on the LM head at 2432 rows, +1.5 % at 1024 rows and on the cross-k/v launch) once one trap was out
//     of the way: a tile index read back from LDS counts as DIVERGENT, and every buffer_load ... lds of the next tile then sits in a
//     waterfall loop over its resource descriptor (+10 us per tile) — __builtin_amdgcn_readfirstlane on the index fixes it.
//   * Same launch table, tile order and k ranges as the other kernels; its own bare per-wave epilogue (below).  Single-problem NT
//     launches without split, K a multiple of 128 and >= 256.
#include "gemm_common.h"
// tools/probe_head_timeline.hip builds this file with -DMIC_TRACE_BLOCKS: every block stamps kernel entry, first operands landed, end
// of the K loop and end of the epilogue (wall_clock64: 100 MHz, one counter for the whole chip) and where it ran; nothing in the product build
#ifdef MIC_TRACE_BLOCKS
__device__ unsigned long long* mic_trace_buf;
#define MIC_TRACE(SLOT) do { if (threadIdx.x == 0) mic_trace_buf[5 * blockIdx.x + (SLOT)] = wall_clock64(); } while (0)
#define MIC_TRACE_ID() do { if (threadIdx.x == 0) { unsigned hw_, xcc_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_)); \
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_)); mic_trace_buf[5 * blockIdx.x + 4] = ((unsigned long long)(xcc_ & 0xf) << 16) | (hw_ & 0xffff); } } while (0)
#else
#define MIC_TRACE(SLOT) do { } while (0)
#define MIC_TRACE_ID() do { } while (0)
#endif

namespace {

constexpr int W4_BK = 64, W4_HIMG = 128 * W4_BK * 2, W4_SLOT = 4 * W4_HIMG;  // 64-k tiles: half image 16 KiB, slot 64 KiB

// byte offset of this lane's source (k = 0 of the split) for 1-KiB piece q (8 rows x 128 B = whole lines) of a half image whose row 0
// is x0.  LDS-DMA writes lane-linear, so the image's swizzle (chunk position p of row R holds source chunk p ^ ((R>>1)&7), see
// read_frag) sits on the SOURCE address: every lane fetches the chunk that belongs at its LDS position.
__device__ __forceinline__ uint32_t w4_source(int ld, int x0, int lim, int q, int lane) {
  const int R = q * 8 + (lane >> 3), c = (lane & 7) ^ ((R >> 1) & 7);
  int gx = x0 + R;
  gx = gx < lim ? gx : lim - 1;
  return ((uint32_t)gx * (uint32_t)ld + (uint32_t)(c * 8)) * 2u;
}

// The bare epilogue of a four-wave block: C = alpha acc + bias as bf16, optionally the folded LayerNorm and the LM head's softmax
// partials.  The shared epilogue (gemm_common.h) is unrolled over every feature of mic_gemm_args; with one wave per SIMD nothing
// overlaps its latencies and it measured ~30 us per tile.  Here every wave drains its own 128 x 128 block on its own, no block
// barrier: four passes of 32 rows through a private fp32 LDS image (pitch 132 floats: the ds_read_b128 lane groups of two adjacent
// rows land on disjoint banks), read back as (row, 8 columns) units = 16-B stores, 256 B contiguous per row and instruction.
constexpr int W4_EP = 132;  // floats per restaged row
template <bool STATS, bool LNF>
__device__ __forceinline__ void w4_epilogue_lean(f32x16 (&acc)[4][4], const Problem& P, char* smem, int mw, int nw, int wave, int lane) {
  const EpiArgs& E = P.epi;
  const int M = P.M, N = P.N;
  {
    const float alpha = E.alpha;
    float bj[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = nw + j * 32 + (lane & 31);
      bj[j] = (E.bias && n < N) ? E.bias[n] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = acc[i][j][r] * alpha + bj[j];
  }
  float* Cw = reinterpret_cast<float*>(smem) + wave * (32 * W4_EP);
  uint16_t* C = (uint16_t*)E.C;
  const int ldc = E.ldc;
  const int urow = lane >> 4, c8 = (lane & 15) * 8;
  const int n = nw + c8;
  const bool nfull = n + 8 <= N;
  // folded LayerNorm: (mean, rstd) of the wave's 128 rows once, through LDS (the shared helper's per-unit int64 loads and conversions
  // would sit 32 times in every lane's path); the column terms of this lane's 8 columns once, in registers
  float2* lnr = reinterpret_cast<float2*>(smem + 4 * 32 * W4_EP * 4) + wave * 128;
  float g8[8], b8[8];
  if constexpr (LNF) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int m = mw + lane + 64 * k;
      float mu = 0.0f, rstd = 0.0f;
      if (m < M) {
        const longlong2 st = reinterpret_cast<const longlong2*>(E.ln_stats)[m];
        mu = (float)st.x * (MIC_ROWSUM_INV_SCALE * E.ln_inv_d);
        rstd = rsqrtf(fmaxf((float)st.y * (MIC_ROWSUM_INV_SCALE * E.ln_inv_d) - mu * mu, 0.0f) + E.ln_eps);
      }
      lnr[lane + 64 * k] = make_float2(mu, rstd);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) g8[i] = b8[i] = 0.0f;
    if (nfull) { ld8(E.ln_g + n, g8); ld8(E.ln_bias + n, b8); }
  }
  // softmax partials: only the column tile that holds the end of the vocabulary needs the per-column validity test
  const bool stat_all = STATS && nw + 128 <= E.stat_nvalid;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) Cw[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * W4_EP + j * 32 + (lane & 31)] = acc[p][j][r];
    // (one wave: its LDS operations complete in order, the reads below see the writes above)
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = it * 4 + urow;
      const int m = mw + p * 32 + row;
      const float* src = Cw + row * W4_EP + c8;
      const float4 lo = *reinterpret_cast<const float4*>(src);
      const float4 hi = *reinterpret_cast<const float4*>(src + 4);
      float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      const bool ok = m < M && n < N;
      if constexpr (LNF) {
        const float2 mr = lnr[p * 32 + row];
        if (nfull) {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = fmaf(mr.y, v[i] - mr.x * g8[i], b8[i]);
        }
      }
      uint4 u;  // the values as stored
      u.x = f2bf_pk(v[0], v[1]); u.y = f2bf_pk(v[2], v[3]);
      u.z = f2bf_pk(v[4], v[5]); u.w = f2bf_pk(v[6], v[7]);
      if (ok) {
        if (nfull) *reinterpret_cast<uint4*>(C + (size_t)m * ldc + n) = u;
        else
          for (int i = 0; i < N - n; ++i) C[(size_t)m * ldc + n + i] = f2bf(v[i]);
      }
      if constexpr (STATS) {
        // (max, sum exp(x - max)) of the values AS STORED over this row's 64-column granule = 8 consecutive lanes
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
        float gm, sm;
        granule_stat8(w, stat_all ? 8 : (ok ? E.stat_nvalid - n : 0), gm, sm);
        if ((lane & 7) == 0 && ok) reinterpret_cast<float2*>(E.rowstat)[(size_t)m * E.stat_ld + n / 64] = make_float2(gm, sm);
      }
    }
  }
}

// The fp32 epilogue (weight-gradient-style launches: the LM head's dE, and the split-K slabs of its dX): C32[split][m][n] = alpha acc
// (+ bias), four passes of 32 rows through the wave's private fp32 LDS image as above, read back as (row, 4 columns) units = 16-B
// stores, 512 B contiguous per row and instruction (two rows per instruction).
__device__ __forceinline__ void w4_epilogue_f32(f32x16 (&acc)[4][4], const Problem& P, char* smem, int mw, int nw, int split, int wave, int lane) {
  const EpiArgs& E = P.epi;
  const int M = P.M, N = P.N;
  {
    const float alpha = E.alpha;
    float bj[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = nw + j * 32 + (lane & 31);
      bj[j] = (E.bias && n < N) ? E.bias[n] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = acc[i][j][r] * alpha + bj[j];
  }
  float* Cw = reinterpret_cast<float*>(smem) + wave * (32 * W4_EP);
  float* C = (float*)E.C + (size_t)split * (size_t)P.split_stride;
  const int ldc = E.ldc;
  const int urow = lane >> 5, c4 = (lane & 31) * 4;
  const int n = nw + c4;
  const bool nfull = n + 4 <= N && (ldc & 3) == 0;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) Cw[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * W4_EP + j * 32 + (lane & 31)] = acc[p][j][r];
    // (one wave: its LDS operations complete in order, the reads below see the writes above)
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int row = it * 2 + urow;
      const int m = mw + p * 32 + row;
      const float4 v = *reinterpret_cast<const float4*>(Cw + row * W4_EP + c4);
      if (m < M && n < N) {
        float* dst = C + (size_t)m * ldc + n;
        if (nfull) *reinterpret_cast<float4*>(dst) = v;
        else {
          const float x[4] = {v.x, v.y, v.z, v.w};
          for (int i = 0; i < N - n && i < 4; ++i) dst[i] = x[i];
        }
      }
    }
  }
}

template <int EPI>
__global__ __launch_bounds__(256) void gemm_w4_kernel(LaunchTable tab) {
  constexpr int WM = 128, WN = 128, WNW = 2, AI = 4, NJ = 4, BM = 256, BN = 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int lid;
  {
    const int bid = blockIdx.x, nwg = tab.total_blocks;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const Problem& P = tab.p[0];
  MIC_TRACE(0);
  MIC_TRACE_ID();
  // split-K (fp32 slabs): split-major — an XCD's contiguous run of logical blocks is a run of TILES of one K range (two at most), so
  // the blocks that share its L2 read the same k-tiles of A and B at about the same time (tile-major order would give every block
  // of an XCD a K range of its own: every operand byte from the fabric once per block); with a multiple of 8 splits, K-range <->
  // XCD affinity as in gemm.hip
  int tile = lid, split = 0;
  if (P.nsplit > 1) {
    const int T = P.tiles_m * P.tiles_n;
    if ((P.nsplit & 7) == 0) {
      const int S = P.nsplit >> 3, j = blockIdx.x >> 3;
      split = (blockIdx.x & 7) * S + j / T;
      tile = j % T;
    } else {
      split = lid / T;
      tile = lid - split * T;
    }
  }
  int tm, tn;
  tile_coords(tile, P.tiles_m, P.tiles_n, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int wr = wave >> 1, wc = wave & 1;

  f32x16 acc[AI][NJ];
#pragma unroll
  for (int i = 0; i < AI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int nk_total = P.K / 64;  // even (K % 128 == 0)
  const int nk_per = ((nk_total + P.nsplit - 1) / P.nsplit + 1) & ~1;  // ... and every split's range too: the K loop runs tile pairs
  const int kt0 = min(nk_total, split * nk_per), kt1 = min(nk_total, kt0 + nk_per);
  const int nsteps = 2 * max(kt1 - kt0, 0);  // even, 32 k each

  // the sixteen 1-KiB pieces this wave brings per 64-k tile: pieces 4w .. 4w+3 of each of the four half images, by LDS-DMA
  // (buffer_load ... lds: scalar resource + 32-bit lane offset + scalar k offset — nothing to update per tile in vector registers;
  // see W4_DMA for tiles past the end of K)
  uint32_t go[16];
#pragma unroll
  for (int h = 0; h < 4; ++h)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      go[h * 4 + i] = h < 2 ? w4_source(P.lda, m0 + h * 128, P.M, wave * 4 + i, lane) : w4_source(P.ldb, n0 + (h - 2) * 128, P.N, wave * 4 + i, lane);
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(P.A) + (size_t)kt0 * 128), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(P.B) + (size_t)kt0 * 128), 0, 0x7fffffff, 0x00020000);
  const int nk = nsteps / 2;  // 64-k tiles of this block
  bf16x8 fa[AI][4], fb[NJ][4];  // one tile's fragments: [32-row block][k 16 kk .. 16 kk + 15]
  const char* fa_ = smem + wr * W4_HIMG;        // + slot: this wave's A half image
  const char* fb_ = smem + (2 + wc) * W4_HIMG;  // ... B half image
  // (a tile past the end of K is requested through a resource of zero records: every lane is out of range, the hardware drops the
  // request — the scalar k offset takes no part in the range check on this architecture, so it cannot do the masking)
  const __amdgpu_buffer_rsrc_t rZ = __builtin_amdgcn_make_buffer_rsrc((void*)P.A, 0, 0, 0x00020000);
#define W4_DMA(TILE, SLOT, Q)                                                                                                  \
  __builtin_amdgcn_raw_ptr_buffer_load_lds((TILE) < nk ? ((Q) < 8 ? rA : rB) : rZ,                                             \
                                           LDS_PTR(void, smem + (SLOT) * W4_SLOT + ((Q) >> 2) * W4_HIMG + (wave * 4 + ((Q) & 3)) * 1024), \
                                           16, go[Q], (TILE) * 128, 0, 0)
#define W4_BARRIER()                       \
  do {                                     \
    __builtin_amdgcn_sched_barrier(0);     \
    __builtin_amdgcn_s_barrier();          \
    __builtin_amdgcn_sched_barrier(0);     \
  } while (0)
// tile T = t + C in LDS slot C (t even, C = 0 / 1): 64 MFMAs in four quarters of 16 (quarter = kk).  The order is pinned gap by gap.
//   quarter 0: read this tile's kk = 3 fragments (gaps 0-7); then {lgkmcnt(0): every read of this slot is done; vmcnt(0): this
//              wave's pieces of tile T+1 have landed} s_barrier — tile T+1 is readable, this tile's slot is dead
//   quarters 1, 2: read tile T+1's kk = 0 / kk = 1 fragments (first 8 gaps of the quarter) into the registers the finished quarters
//              freed; one LDS-DMA piece of tile T+2 into the dead slot in every other gap (an LDS-DMA piece costs 60-100 cycles of
//              issue time: never two in a row; spreading the sixteen pieces over quarters 1-3 or every third gap measured the same
//              107.5-108.0 us at 4096^3, profiles/NOTES_r5.md section 4)
//   quarter 3: read tile T+1's kk = 2 fragments
#define W4_TILE(C)                                                                                                             \
  do {                                                                                                                         \
    _Pragma("unroll") for (int g_ = 0; g_ < 64; ++g_) {                                                                        \
      if (g_ == 16) {                                                                                                          \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                            \
        W4_BARRIER();                                                                                                          \
      }                                                                                                                        \
      acc[(g_ & 15) >> 2][g_ & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[(g_ & 15) >> 2][g_ >> 4], fb[g_ & 3][g_ >> 4],  \
                                                                           acc[(g_ & 15) >> 2][g_ & 3], 0, 0, 0);             \
      const int q_ = g_ >> 4, x_ = g_ & 15;                                                                                    \
      const int rk_ = (q_ + 3) & 3, rs_ = q_ == 0 ? (C) : 1 - (C);  /* kk and slot of this quarter's fragment reads */          \
      if (x_ < 4) fa[x_][rk_] = read_frag<false, W4_BK, 128>(fa_ + rs_ * W4_SLOT, x_ * 32, rk_, lane);                         \
      else if (x_ < 8) fb[x_ - 4][rk_] = read_frag<false, W4_BK, 128>(fb_ + rs_ * W4_SLOT, (x_ - 4) * 32, rk_, lane);          \
      if ((q_ == 1 || q_ == 2) && (x_ & 1) == 1) W4_DMA(t + (C) + 2, C, (q_ - 1) * 8 + (x_ >> 1));                             \
      __builtin_amdgcn_sched_barrier(0);                                                                                       \
    }                                                                                                                          \
  } while (0)

  const bool live = m0 + wr * WM < P.M;  // (wave-uniform)
  if (nk >= 2) {
#pragma unroll
    for (int q = 0; q < 16; ++q) W4_DMA(0, 0, q);
#pragma unroll
    for (int q = 0; q < 16; ++q) W4_DMA(1, 1, q);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // tile 0 has landed (tile 1 may still fly)
    W4_BARRIER();
    MIC_TRACE(1);
#pragma unroll
    for (int kk = 0; kk < 3; ++kk)
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        fa[x][kk] = read_frag<false, W4_BK, 128>(fa_, x * 32, kk, lane);
        fb[x][kk] = read_frag<false, W4_BK, 128>(fb_, x * 32, kk, lane);
      }
    __builtin_amdgcn_sched_barrier(0);
    // one uniform loop, no peeled tail: past the end of K the DMA requests are dropped (k offset out of the resource's range) and
    // the fragment reads of the tile that does not exist fetch values nobody uses
    if (live) {
      for (int t = 0; t < nk; t += 2) {
        W4_TILE(0);
        W4_TILE(1);
      }
    } else {
      // this wave's 128 rows lie past M (the last row tile of 2432 packed rows is half empty): it keeps bringing its DMA pieces and
      // meeting the barriers, without MFMAs and fragment reads — on a power-clocked kernel idle matrix pipes are clock for the others
      for (int t = 0; t < nk; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        W4_BARRIER();
#pragma unroll
        for (int q = 0; q < 16; ++q) W4_DMA(t + 2, t & 1, q);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (dropped requests still count)
  }
  __syncthreads();
  MIC_TRACE(2);
  if (live) {
    if constexpr (EPI == 8) w4_epilogue_f32(acc, P, smem, m0 + wr * WM, n0 + wc * WN, split, wave, lane);
    else w4_epilogue_lean<(EPI & 1) != 0, (EPI & 4) != 0>(acc, P, smem, m0 + wr * WM, n0 + wc * WN, wave, lane);
  }
#ifdef MIC_TRACE_BLOCKS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the stamp below = this wave's stores acknowledged)
#endif
  MIC_TRACE(3);
}

template <int EPI>
void launch_w4(const LaunchTable& tab, hipStream_t s) {
  constexpr int lds = 2 * W4_SLOT;  // two 64-KiB tile slots (the bare epilogue restages 4 x 16.5 KiB through the same memory)
  static bool attr_set_dev[64] = {};  // per instantiation and device
  int dev_ = 0;
  (void)hipGetDevice(&dev_);
  bool& attr_set = attr_set_dev[dev_ & 63];
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_w4_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_w4_kernel<EPI>), dim3(tab.total_blocks), dim3(256), lds, s, tab);
}

}  // namespace

// the launches this kernel takes: one problem, NT, no split, K a multiple of 128 and >= 256, bf16 C through the bare epilogue
// (alpha, bias, folded LayerNorm, softmax partials — the LM head and the all-layer cross k/v projection)
bool gemm_w4_takes(const LaunchTable& tab) {
  if (tab.count != 1) return false;
  const Problem& p = tab.p[0];
  const EpiArgs& e = p.epi;
  // the DMA addresses an operand through a buffer resource of 2^31 - 1 bytes with 32-bit lane offsets: larger operands stay on the
  // four-phase kernel (64-bit addresses)
  const long long lim = 0x7fffffffLL;
  if ((long long)p.M * p.lda * 2 >= lim || (long long)p.N * p.ldb * 2 >= lim) return false;
  if (p.K < 256 || p.K % 128 != 0 || e.R || e.drop_thr || e.rowsum2 || e.act || e.Zout || e.dact || e.accumulate) return false;
  if (((uintptr_t)e.C & 15) != 0) return false;
  if (e.c_f32)  // fp32 C (one slab per split with split-K): alpha and bias only
    return !e.rowstat && !e.ln_stats && (p.nsplit == 1 || p.split_stride > 0) && p.K / 64 / p.nsplit >= 2;
  // bf16 C: bias, folded LayerNorm (whole 8-column groups only: its column terms are loaded 8 at a time), softmax partials
  return p.nsplit == 1 && (e.ldc & 7) == 0 && (!e.ln_stats || p.N % 8 == 0);
}
void launch_gemm_w4(const LaunchTable& tab, hipStream_t s) {
  const EpiArgs& e = tab.p[0].epi;
  if (e.c_f32) { launch_w4<8>(tab, s); return; }
  const int epi = 2 + (e.rowstat ? 1 : 0) + (e.ln_stats ? 4 : 0);
  switch (epi) {
    case 2: launch_w4<2>(tab, s); break;
    case 3: launch_w4<3>(tab, s); break;
    case 6: launch_w4<6>(tab, s); break;
    default: launch_w4<7>(tab, s); break;
  }
}
