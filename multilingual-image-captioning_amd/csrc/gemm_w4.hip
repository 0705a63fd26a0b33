// gemm_w4.hip — the 256x256 bf16 tile on FOUR waves, one per SIMD, each with the whole 512-register file: wave tile 128 x 128
// (16 accumulator blocks of 32x32 = 256 AGPRs), ONE barrier per 32-k step.  Where gemm_phased.hip lets the two waves of a SIMD take
// turns at the matrix pipe (eight barrier hand-offs per 64-k tile), here every wave feeds its own pipe from one instruction stream
// with the side work pinned into the gaps between its 32 MFMAs per step.  OPT-IN (MIC_GEMM_W4=1): it needs 12 % fewer cycles than
// the four-phase kernel at 4096^3 (222 k against 251 k, rocprofv3 GRBM_GUI_ACTIVE) and takes about the same wall time, because the chip
// clocks these kernels by power (1.80 GHz against 1.99; DESIGN.md section 3, "Round 4"): 0-3 % ahead in isolation, level in situ.
//
//   * LDS: four slots of 32 KiB, step X in slot X%4 = [A rows 0-127 | A rows 128-255 | B cols 0-127 | B cols 128-255], each a
//     k-contiguous image of 128 rows x 32 k (64-B rows, 16-B chunk position p of row R holds source chunk p ^ ((R>>2)&3):
//     conflict-free for the 16-lane groups of ds_read_b128, see read_frag).  The swizzle sits on the LDS destination of the
//     register-staged pieces (per-lane write offsets `wl`).
//   * Operands travel global -> registers -> LDS: an LDS-DMA piece costs 60-100 cycles of ISSUE time and with one wave per SIMD
//     nobody else feeds the matrix pipe meanwhile (first build of this file: 930-1020 TF/s at 4096^3 / 8192^3);
//     global_load_dwordx4 (scalar base + 32-bit offset) + ds_write_b128 are ~20.  Two register sets of sixteen 16-B pieces per
//     64-k tile (8 rows x 128 B each: whole lines); tile U is requested in step 2U-6, written to the LDS slots of its two steps in
//     step 2U-2, read into fragments one half step ahead of the MFMAs that use them.
//   * step S: s_barrier {gaps 0-7: read this step's k 16-31 fragments} {gaps 16-23: read step S+1's k 0-15 fragments}; even steps
//     also {gaps 8-15 and 24-31: write the tile of steps S+2, S+3 from its register set, request the tile of steps S+6, S+7 into
//     the same registers}; s_waitcnt lgkmcnt(0).  RAW: a wave's own lgkmcnt(0), then the barrier, order its LDS writes before the
//     other waves' reads one step later.  WAR: a slot is rewritten at least one barrier after its last read.  vmcnt is left to the
//     compiler (counted waits).
//   * Same launch table, tile order and k ranges as the other kernels; its own bare epilogue (below).  Single-problem NT launches
//     without split, K a multiple of 128.
#include "gemm_common.h"

namespace {

constexpr int W4_BK = 32, W4_HIMG = 128 * W4_BK * 2, W4_SLOT = 4 * W4_HIMG;

// byte offset of this lane's source (k = 0) for 1-KiB piece `piece` of a 256-row operand tile whose row 0 is x0: 8 rows x 128 B (a
// whole line per row and 64-k tile: half the L2 requests of 64-B segments per 32-k step — the 16-rows-x-64-B form of the first build
// measured 5-7 % slower on the LM-head shapes); the operand base stays a scalar (global_load saddr + 32-bit voffset)
__device__ __forceinline__ uint32_t w4_source(int ld, int x0, int lim, int piece, int lane) {
  int gx = x0 + piece * 8 + (lane >> 3);
  gx = gx < lim ? gx : lim - 1;
  return ((uint32_t)gx * (uint32_t)ld + (uint32_t)((lane & 7) * 8)) * 2u;
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
      const int m = mw + p * 32 + row, n = nw + c8;
      const float* src = Cw + row * W4_EP + c8;
      const float4 lo = *reinterpret_cast<const float4*>(src);
      const float4 hi = *reinterpret_cast<const float4*>(src + 4);
      float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      const bool ok = m < M && n < N;
      if constexpr (LNF) {
        if (ok && n + 8 <= N) ln_fold_apply8(E, m, n, v);
      }
      if (ok) {
        if (n + 8 <= N) st8(C + (size_t)m * ldc + n, v);
        else
          for (int i = 0; i < N - n; ++i) C[(size_t)m * ldc + n + i] = f2bf(v[i]);
      }
      if constexpr (STATS) {
        // (max, sum exp(x - max)) of the values AS STORED over this row's 64-column granule = 8 consecutive lanes
        float mx = -INFINITY, x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          x[i] = (ok && n + i < E.stat_nvalid) ? bf2f(f2bf(v[i])) : -INFINITY;
          mx = fmaxf(mx, x[i]);
        }
        const float gm = group8_max(mx);
        float sm = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) sm += x[i] > -INFINITY ? __expf(x[i] - gm) : 0.0f;
        sm = group8_sum(sm);
        if ((lane & 7) == 0 && ok) reinterpret_cast<float2*>(E.rowstat)[(size_t)m * E.stat_ld + n / 64] = make_float2(gm, sm);
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
  int tile = lid / P.nsplit, split = lid - tile * P.nsplit;
  if (P.nsplit > 1 && (P.nsplit & 7) == 0) {  // split-K with K-range <-> XCD affinity (see gemm.hip)
    const int T = P.tiles_m * P.tiles_n, S = P.nsplit >> 3, j = blockIdx.x >> 3;
    split = (blockIdx.x & 7) * S + j / T;
    tile = j % T;
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

  const int nk_total = P.K / 64;
  const int nk_per = (nk_total + P.nsplit - 1) / P.nsplit;
  const int kt0 = split * nk_per, kt1 = min(nk_total, kt0 + nk_per);
  const int nsteps = 2 * max(kt1 - kt0, 0);  // even, 32 k each

  // the sixteen 1-KiB pieces this wave brings per 64-k TILE (= two steps): pieces 8w .. 8w+7 of the A tile and of the B tile.  Lane
  // (row r = lane>>3, chunk c = lane&7) of piece q holds k 8c..8c+7 of tile row 64w + 8q + r: chunks 0-3 belong to the tile's even
  // step, 4-7 to its odd step, so one ds_write_b128 scatters a piece over the two steps' LDS slots
  uint32_t go[16];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    go[q] = w4_source(P.lda, m0, P.M, wave * 8 + q, lane);
    go[8 + q] = w4_source(P.ldb, n0, P.N, wave * 8 + q, lane);
  }
  const char* gA = reinterpret_cast<const char*>(P.A) + (size_t)kt0 * 128;  // scalar bases, advanced by the loop
  const char* gB = reinterpret_cast<const char*>(P.B) + (size_t)kt0 * 128;
  // this lane's LDS write offset inside a pair of slots, for even / odd pieces: row 8q + r of the wave's 64 rows sits in half image
  // w>>1 at row 64(w&1) + 8q + r, chunk position (c&3) ^ ((row>>2)&3) with (row>>2)&3 = (2(q&1) + (lane>>5)) & 3
  uint32_t wl[2];
#pragma unroll
  for (int e = 0; e < 2; ++e)
    wl[e] = (uint32_t)(((lane >> 2) & 1) * W4_SLOT + (wave >> 1) * W4_HIMG + ((wave & 1) * 64 + (lane >> 3)) * 64 +
                       ((((lane & 3) ^ ((2 * e + (lane >> 5)) & 3))) << 4));
  // operands travel global -> registers -> LDS (an LDS-DMA piece costs 60-100 cycles of ISSUE time, and with one wave per SIMD nobody
  // else feeds the matrix pipe meanwhile; global_load_dwordx4 + ds_write_b128 are ~20).  Two register sets of sixteen pieces: tile
  // U (steps 2U, 2U+1) is requested in step 2U-6, written to LDS slots 2U%4 and (2U+1)%4 in step 2U-2 (four steps of latency
  // budget), its fragments are read in steps 2U-1 .. 2U+1.
  u32x4 lA[16], lB[16];
  auto ld = [&](u32x4 (&l)[16], int tile, bool on) __attribute__((always_inline)) {
    const uint32_t m = on ? 0xffffffffu : 0u;
#pragma unroll
    for (int q = 0; q < 16; ++q) l[q] = *reinterpret_cast<const u32x4*>((q < 8 ? gA : gB) + tile * 128 + (go[q] & m));
  };
  auto st = [&](const u32x4 (&l)[16], int pair) __attribute__((always_inline)) {
    char* base = smem + pair * 2 * W4_SLOT;
#pragma unroll
    for (int q = 0; q < 16; ++q) *reinterpret_cast<u32x4*>(base + (q >> 3) * 2 * W4_HIMG + (q & 7) * 512 + wl[q & 1]) = l[q];
  };
  bf16x8 a0[AI], b0[NJ], a1[AI], b1[NJ];  // fragments of k 0-15 / k 16-31 of a step
  const char* fa_ = smem + wr * W4_HIMG;        // + slot: this wave's A half image
  const char* fb_ = smem + (2 + wc) * W4_HIMG;  // ... B half image
// step S = s + C (s a multiple of 4, C a constant: LDS slots and register sets are compile-time).  The order is pinned gap by gap
// (the scheduler's group pipelines came apart on three instruction classes).  MFMA gaps 0-7: read this step's k 16-31 fragments;
// 16-23: read step S+1's k 0-15 fragments.  EVEN steps also write the tile of steps S+2, S+3 from register set L and request the
// tile of steps S+6, S+7 into it: A pieces in gaps 8-15, B pieces in gaps 24-31.  s_waitcnt lgkmcnt(0) closes the step.
#define W4_STEP(C, L, EVEN)                                                                                                    \
  do {                                                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                                         \
    __builtin_amdgcn_s_barrier();                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                                         \
    char* ws_ = smem + (((C) + 2) & 2) * W4_SLOT;                                                                              \
    const uint32_t lm_ = s + (C) + 6 < nsteps ? 0xffffffffu : 0u;                                                              \
    _Pragma("unroll") for (int g_ = 0; g_ < 32; ++g_) {                                                                        \
      if (g_ < 16) acc[g_ >> 2][g_ & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[g_ >> 2], b0[g_ & 3], acc[g_ >> 2][g_ & 3], 0, 0, 0); \
      else acc[(g_ - 16) >> 2][g_ & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[(g_ - 16) >> 2], b1[g_ & 3], acc[(g_ - 16) >> 2][g_ & 3], 0, 0, 0); \
      if (g_ < 4) a1[g_] = read_frag<false, W4_BK, 128>(fa_ + ((C) & 3) * W4_SLOT, g_ * 32, 1, lane);                          \
      else if (g_ < 8) b1[g_ - 4] = read_frag<false, W4_BK, 128>(fb_ + ((C) & 3) * W4_SLOT, (g_ - 4) * 32, 1, lane);           \
      else if (g_ >= 16 && g_ < 20) a0[g_ - 16] = read_frag<false, W4_BK, 128>(fa_ + (((C) + 1) & 3) * W4_SLOT, (g_ - 16) * 32, 0, lane); \
      else if (g_ >= 20 && g_ < 24) b0[g_ - 20] = read_frag<false, W4_BK, 128>(fb_ + (((C) + 1) & 3) * W4_SLOT, (g_ - 20) * 32, 0, lane); \
      else if (EVEN) {                                                                                                         \
        const int q_ = g_ < 16 ? g_ - 8 : g_ - 16;  /* 0-7: A pieces, 8-15: B pieces */                                        \
        *reinterpret_cast<u32x4*>(ws_ + (q_ >> 3) * 2 * W4_HIMG + (q_ & 7) * 512 + wl[q_ & 1]) = L[q_];                        \
        L[q_] = *reinterpret_cast<const u32x4*>((q_ < 8 ? gA : gB) + ((C) + 6) * (W4_BK * 2) + (go[q_] & lm_));                \
      }                                                                                                                        \
      __builtin_amdgcn_sched_barrier(0);                                                                                       \
    }                                                                                                                          \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                                         \
  } while (0)

  if (nsteps >= 8) {
    ld(lA, 0, true);
    st(lA, 0);
    ld(lA, 1, true);
    ld(lB, 2, true);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      a0[x] = read_frag<false, W4_BK, 128>(fa_, x * 32, 0, lane);
      b0[x] = read_frag<false, W4_BK, 128>(fb_, x * 32, 0, lane);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // one uniform loop, no peeled tail (copies of the step for the last iterations made the register allocator spill the in-flight
    // register sets around the loop, and spill traffic counts on vmcnt): past the end of K the requests collapse onto one 128-B
    // line (offset mask 0), the LDS writes land in slots nobody reads again, the fragment reads fetch values nobody uses
    for (int s = 0; s < nsteps; s += 4) {
      W4_STEP(0, lA, true);
      W4_STEP(1, lA, false);
      W4_STEP(2, lB, true);
      W4_STEP(3, lB, false);
      gA += 4 * W4_BK * 2;
      gB += 4 * W4_BK * 2;
    }
  }
  __syncthreads();
  w4_epilogue_lean<(EPI & 1) != 0, (EPI & 4) != 0>(acc, P, smem, m0 + wr * WM, n0 + wc * WN, wave, lane);
}

template <int EPI>
void launch_w4(const LaunchTable& tab, hipStream_t s) {
  constexpr int lds = 4 * W4_SLOT;  // the K loop uses two slots; the shared epilogue restages 4 x 32 KiB, the bare one 4 x 16.5
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
  if (tab.count != 1 || !table_is_plain(tab)) return false;
  const Problem& p = tab.p[0];
  const EpiArgs& e = p.epi;
  return p.nsplit == 1 && p.K >= 256 && p.K % 128 == 0 && !e.R && !e.drop_thr && !e.rowsum2 && !e.c_f32;
}
void launch_gemm_w4(const LaunchTable& tab, hipStream_t s) {
  const EpiArgs& e = tab.p[0].epi;
  const int epi = 2 + (e.rowstat ? 1 : 0) + (e.ln_stats ? 4 : 0);
  switch (epi) {
    case 2: launch_w4<2>(tab, s); break;
    case 3: launch_w4<3>(tab, s); break;
    case 6: launch_w4<6>(tab, s); break;
    default: launch_w4<7>(tab, s); break;
  }
}
