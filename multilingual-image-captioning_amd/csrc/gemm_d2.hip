// gemm_d2.hip — 256 x 128 bf16 tiles on four waves (one per SIMD, wave tile 128 x 64 = 8 accumulator blocks of 32x32 = 128 AGPRs),
// TWO blocks per CU.  gemm_w4.hip gives every wave the whole 512-register file and a 128 x 128 tile: the best FLOPs per staged byte,
// but one instruction stream per SIMD — an LDS-DMA request costs 60-100 cycles of issue during which nobody feeds that SIMD's matrix
// pipe (the K loop runs at ~2/3 of the pipe), and a block's prologue, barrier waits and epilogue (a third of a K = 1024 tile's time)
// are dead time for its CU.  Here a block needs half the registers (<= 256 per lane) and 72 KiB of LDS, so two INDEPENDENT blocks
// share a CU: every SIMD has two instruction streams that take no barrier together — one block's DMA issue, barrier wait, epilogue
// and first-byte latency run under the other's MFMAs.  The price: 1.5x the operand bytes per FLOP (48 B/clk/CU at full pipe rate;
// the XCD-aware tile order delivers 38-42), 32-k steps (64-B row segments per DMA lane group, one barrier per 16 MFMAs).
//
//   * LDS: a ring of three 24-KiB slots, step S (32 k) in slot S % 3 = [A 256 rows | B 128 rows] x 64 B, 16-B chunk position p of row R
//     holds source chunk p ^ ((R >> 2) & 3) (conflict-free for ds_read_b128, gemm_common.h read_frag<false, 32>); the swizzle sits on
//     the DMA's SOURCE address (LDS-DMA writes lane-linear).  1-KiB pieces = 16 rows x 64 B; a wave brings six per step.
//   * Step s: [8 MFMAs of k-half 0 | this step's k-half-1 fragments are read; three pieces of step s+2 go out]
//     {lgkmcnt(0): slot s%3 is read out; vmcnt(6): this wave's pieces of step s+1 have landed} s_barrier
//     [8 MFMAs of k-half 1 | step s+1's k-half-0 fragments are read; three pieces of step s+3 go into slot s%3, now dead].
//     Requests run two steps ahead of their first read; a fragment register is re-read half a step behind its last use (48 VGPRs of
//     fragments in all).  Past the end of K the requests go through a zero-record resource (dropped, still counted by vmcnt).
//   * Epilogue per wave, no block barrier: four passes of 32 rows through a private fp32 LDS image, read back as (row, 8 columns)
//     units.  Bare launches (bf16 C = alpha acc + bias, optional softmax partials) store straight away; everything else — activation,
//     saved pre-activation, dGELU, dropout, residual, fp32 C, accumulate — goes through the shared per-unit helpers with the side
//     operands of a pass requested before its restage.  With a second block computing on the same CU the epilogue is cover, not
//     a stall, so it can afford the general form.
//   * Split-K writes one fp32 slab per split (split-major block order, as gemm_w4.hip).  Single-problem NT launches, K % 64 == 0.
//   * Measured (profiles/NOTES_r5.md): 4096^3 977 TF/s against the four-wave kernel's 1277 — two resident blocks reach 69 % of the
//     matrix pipe where one four-wave block reaches 66-78 %: whatever holds an LDS-fed bf16 K loop at ~2/3 of the pipe (the library's
//     kernels sit at 72 % in cycles) is not the lack of a second instruction stream.  What the second block does buy is the epilogue:
//     LM head with softmax partials 647 us against 636-639 in isolation, decoder step 2.35 against 2.42 ms in situ (+2.7 % captions/s),
//     train step level.  A 192 x 128 variant on whole cache lines in two 40-KiB slots (0.42 L2 requests per clock instead of 0.75)
//     ran its K loop 8 % faster (4096^3 1052 TF/s) and changed neither leg in situ: not kept.
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

constexpr int D2_AIMG = 256 * 64, D2_BIMG = 128 * 64, D2_SLOT = D2_AIMG + D2_BIMG, D2_NS = 3;  // bytes
constexpr int D2_EP = 68;  // floats per restaged row (64 columns + 4)

// byte offset (k = 0 of the block's K range) of this lane's 16-B source chunk for 1-KiB piece q (16 rows x 64 B) of an image whose
// row 0 is x0: lane l lands at LDS byte 16 l of the piece = row q*16 + (l >> 2), chunk position l & 3
__device__ __forceinline__ uint32_t d2_source(int ld, int x0, int lim, int q, int lane) {
  const int R = q * 16 + (lane >> 2), c = (lane & 3) ^ ((R >> 2) & 3);
  int gx = x0 + R;
  gx = gx < lim ? gx : lim - 1;
  return ((uint32_t)gx * (uint32_t)ld + (uint32_t)(c * 8)) * 2u;
}

// one 32-row pass of the epilogue: the wave's accumulator blocks accp[0..1] (rows mp .. mp+31, columns nw .. nw+63) through the
// wave's private LDS image Cw
template <bool STATS>
__device__ __forceinline__ void d2_epilogue_pass(f32x16 (&accp)[2], const EpiArgs& E, float* Cw, int M, int N, int mp, int nw, int lane, bool bare) {
  const int urow = lane >> 3, c8 = (lane & 7) * 8;
  const int n = nw + c8;
  const bool nok = n + 8 <= N;  // (the launcher admits N % 8 == 0 only: a unit is inside or outside as a whole)
  const bool stat_all = STATS && nw + 64 <= E.stat_nvalid;
  u32x4 zq[4], rq[4];
  if (!bare) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int m = mp + it * 8 + urow;
      zq[it] = rq[it] = u32x4{0u, 0u, 0u, 0u};
      if (m < M && nok) epilogue_prefetch8(E, m, n, zq[it], rq[it]);
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) Cw[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * D2_EP + j * 32 + (lane & 31)] = accp[j][r];
  // (one wave: its LDS operations complete in order, the reads below see the writes above)
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = it * 8 + urow;
    const int m = mp + row;
    const float* src = Cw + row * D2_EP + c8;
    const float4 lo = *reinterpret_cast<const float4*>(src);
    const float4 hi = *reinterpret_cast<const float4*>(src + 4);
    float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    const bool ok = m < M && nok;
    if (!bare) {
      if (ok) epilogue_store8_pre(E, m, n, v, zq[it], rq[it]);
    } else {
      uint4 u;  // the values as stored
      u.x = f2bf_pk(v[0], v[1]); u.y = f2bf_pk(v[2], v[3]);
      u.z = f2bf_pk(v[4], v[5]); u.w = f2bf_pk(v[6], v[7]);
      if (ok) *reinterpret_cast<uint4*>((uint16_t*)E.C + (size_t)m * E.ldc + n) = u;
      if constexpr (STATS) {
        // (max, sum exp(x - max)) of the values AS STORED over this row's 64-column granule = the wave tile's width = 8 consecutive lanes
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
        float gm, sm;
        granule_stat8(w, stat_all ? 8 : (ok ? E.stat_nvalid - n : 0), gm, sm);
        if ((lane & 7) == 0 && ok) reinterpret_cast<float2*>(E.rowstat)[(size_t)m * E.stat_ld + n / 64] = make_float2(gm, sm);
      }
    }
  }
}

template <bool STATS>
__device__ __forceinline__ void d2_epilogue(f32x16 (&acc)[4][2], const Problem& P, char* smem, int mw, int nw, int split, int wave, int lane) {
  EpiArgs E = P.epi;
  const int M = P.M, N = P.N;
  {
    const float alpha = E.alpha;
    float bj[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = nw + j * 32 + (lane & 31);
      bj[j] = (E.bias && n < N) ? E.bias[n] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = acc[i][j][r] * alpha + bj[j];
    E.alpha = 1.0f;
    E.bias = nullptr;
  }
  if (E.c_f32) E.C = (float*)E.C + (size_t)split * (size_t)P.split_stride;
  const bool bare = !E.c_f32 && !E.act && !E.Zout && !E.dact && !E.R && !E.drop_thr && !E.accumulate;
  float* Cw = reinterpret_cast<float*>(smem) + wave * (32 * D2_EP);
  // one call per 32-row pass with the accumulator rows named by a CONSTANT (a pass loop the compiler does not unroll turns the
  // index into a run-time value and the accumulators into scratch: tests/test_kernel_resources_cpu.py)
  d2_epilogue_pass<STATS>(acc[0], E, Cw, M, N, mw, nw, lane, bare);
  d2_epilogue_pass<STATS>(acc[1], E, Cw, M, N, mw + 32, nw, lane, bare);
  d2_epilogue_pass<STATS>(acc[2], E, Cw, M, N, mw + 64, nw, lane, bare);
  d2_epilogue_pass<STATS>(acc[3], E, Cw, M, N, mw + 96, nw, lane, bare);
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_d2_kernel(LaunchTable tab) {
  constexpr int WM = 128, WN = 64, BM = 256, BN = 128;
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
  int tile = lid, split = 0;
  if (P.nsplit > 1) {  // split-major (see gemm_w4.hip)
    const int T = P.tiles_m * P.tiles_n;
    split = lid / T;
    tile = lid - split * T;
  }
  int tm, tn;
  tile_coords(tile, P.tiles_m, P.tiles_n, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int wr = wave >> 1, wc = wave & 1;

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int nk_total = P.K / 32;
  const int nk_per = (nk_total + P.nsplit - 1) / P.nsplit;
  const int kt0 = min(nk_total, split * nk_per), kt1 = min(nk_total, kt0 + nk_per);
  const int nk = kt1 - kt0;  // 32-k steps of this block

  // the six 1-KiB pieces this wave brings per step: pieces 4w .. 4w+3 of the A image (16), 2w, 2w+1 of the B image (8)
  uint32_t go[6];
#pragma unroll
  for (int i = 0; i < 4; ++i) go[i] = d2_source(P.lda, m0, P.M, wave * 4 + i, lane);
#pragma unroll
  for (int i = 0; i < 2; ++i) go[4 + i] = d2_source(P.ldb, n0, P.N, wave * 2 + i, lane);
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(P.A) + (size_t)kt0 * 64), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(P.B) + (size_t)kt0 * 64), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rZ = __builtin_amdgcn_make_buffer_rsrc((void*)P.A, 0, 0, 0x00020000);  // zero records: every request dropped
  // this lane's fragment addresses inside a slot, per k-half (the swizzle term depends on the lane and the k-half only: fragment rows
  // start at multiples of 32)
  int lpA[2], lpB[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int lp = (lane & 31) * 64 + (((kk * 2 + (lane >> 5)) ^ ((lane >> 2) & 3)) << 4);
    lpA[kk] = lp + wr * (WM * 64);
    lpB[kk] = lp + D2_AIMG + wc * (WN * 64);
  }
  bf16x8 fa[4][2], fb[2][2];  // [32-row block][k-half]

#define D2_DMA(STEP, SLOTOFF, Q)                                                                                                       \
  __builtin_amdgcn_raw_ptr_buffer_load_lds((STEP) < nk ? ((Q) < 4 ? rA : rB) : rZ,                                                     \
                                           LDS_PTR(void, smem + (SLOTOFF) + ((Q) < 4 ? (wave * 4 + (Q)) * 1024 : D2_AIMG + (wave * 2 + (Q) - 4) * 1024)), \
                                           16, go[Q], (STEP) * 64, 0, 0)
#define D2_BARRIER()                       \
  do {                                     \
    __builtin_amdgcn_sched_barrier(0);     \
    __builtin_amdgcn_s_barrier();          \
    __builtin_amdgcn_sched_barrier(0);     \
  } while (0)
#define D2_READ_A(I, KK, SLOTOFF) fa[I][KK] = *reinterpret_cast<const bf16x8*>(smem + (SLOTOFF) + lpA[KK] + (I) * (32 * 64))
#define D2_READ_B(J, KK, SLOTOFF) fb[J][KK] = *reinterpret_cast<const bf16x8*>(smem + (SLOTOFF) + lpB[KK] + (J) * (32 * 64))

  const bool live = m0 + wr * WM < P.M;  // (wave-uniform)
  if (nk > 0) {
    // prologue: steps 0 and 1 whole, the first half of step 2's pieces (the loop's first half-step sends the other half)
#pragma unroll
    for (int q = 0; q < 6; ++q) D2_DMA(0, 0, q);
#pragma unroll
    for (int q = 0; q < 6; ++q) D2_DMA(1, D2_SLOT, q);
#pragma unroll
    for (int q = 0; q < 3; ++q) D2_DMA(2, 2 * D2_SLOT, q);
    asm volatile("s_waitcnt vmcnt(9)" ::: "memory");  // step 0 has landed
    D2_BARRIER();
    MIC_TRACE(1);
    int c0 = 0, c1 = D2_SLOT, c2 = 2 * D2_SLOT;  // slot offsets of steps s, s+1, s+2
    if (live) {
#pragma unroll
      for (int i = 0; i < 4; ++i) D2_READ_A(i, 0, 0);
#pragma unroll
      for (int j = 0; j < 2; ++j) D2_READ_B(j, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      for (int s = 0; s < nk; ++s) {
        // ---- k-half 0: 8 MFMAs; this step's k-half-1 fragments; pieces 3-5 of step s+2 into its slot (dead since the last barrier)
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          acc[g >> 1][g & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g >> 1][0], fb[g & 1][0], acc[g >> 1][g & 1], 0, 0, 0);
          if (g < 4) D2_READ_A(g, 1, c0);
          else if (g < 6) D2_READ_B(g - 4, 1, c0);
          if (g == 1 || g == 3 || g == 5) D2_DMA(s + 2, c2, 3 + (g >> 1));
          __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");  // slot c0 read out; this wave's pieces of step s+1 landed
        D2_BARRIER();
        // ---- k-half 1: 8 MFMAs; step s+1's k-half-0 fragments; pieces 0-2 of step s+3 into slot c0 (dead now)
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          acc[g >> 1][g & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g >> 1][1], fb[g & 1][1], acc[g >> 1][g & 1], 0, 0, 0);
          if (g < 4) D2_READ_A(g, 0, c1);
          else if (g < 6) D2_READ_B(g - 4, 0, c1);
          if (g == 1 || g == 3 || g == 5) D2_DMA(s + 3, c0, g >> 1);
          __builtin_amdgcn_sched_barrier(0);
        }
        const int t = c0;
        c0 = c1; c1 = c2; c2 = t;
      }
    } else {
      // this wave's 128 rows lie past M: it keeps bringing its pieces and meeting the barriers, without MFMAs and fragment reads
      for (int s = 0; s < nk; ++s) {
#pragma unroll
        for (int q = 3; q < 6; ++q) D2_DMA(s + 2, c2, q);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        D2_BARRIER();
#pragma unroll
        for (int q = 0; q < 3; ++q) D2_DMA(s + 3, c0, q);
        const int t = c0;
        c0 = c1; c1 = c2; c2 = t;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (dropped requests still count)
  }
  __syncthreads();
  MIC_TRACE(2);
  if (live) d2_epilogue<(EPI & 1) != 0>(acc, P, smem, m0 + wr * WM, n0 + wc * WN, split, wave, lane);
#ifdef MIC_TRACE_BLOCKS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the stamp below = this wave's stores acknowledged)
#endif
  MIC_TRACE(3);
#undef D2_DMA
#undef D2_BARRIER
#undef D2_READ_A
#undef D2_READ_B
}

template <int EPI>
void launch_d2(const LaunchTable& tab, hipStream_t s) {
  constexpr int lds = D2_NS * D2_SLOT;  // 72 KiB: two blocks per CU (the epilogue restages 4 x 8.5 KiB through the same memory)
  static bool attr_set_dev[64] = {};  // per instantiation and device
  int dev_ = 0;
  (void)hipGetDevice(&dev_);
  bool& attr_set = attr_set_dev[dev_ & 63];
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_d2_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_d2_kernel<EPI>), dim3(tab.total_blocks), dim3(256), lds, s, tab);
}

}  // namespace

// the launches this kernel takes: one problem, NT, K >= 128, whole 8-column units with 16-B aligned rows everywhere; no folded
// LayerNorm / rowsum2 by-products (they stay with the other kernels); split-K as fp32 slabs
bool gemm_d2_takes(const LaunchTable& tab) {
  if (tab.count != 1) return false;
  const Problem& p = tab.p[0];
  const EpiArgs& e = p.epi;
  const long long lim = 0x7fffffffLL;  // buffer resources with 32-bit lane offsets (gemm_w4.hip)
  if ((long long)p.M * p.lda * 2 >= lim || (long long)p.N * p.ldb * 2 >= lim) return false;
  if (p.K < 128 || p.K % 64 != 0 || p.N % 8 != 0 || e.rowsum2 || e.ln_stats) return false;
  if (((uintptr_t)e.C & 15) != 0 || (e.ldc & 7) != 0) return false;
  if ((e.Zout || e.Zin) && ((e.ldz & 7) != 0 || ((uintptr_t)(e.Zout ? e.Zout : e.Zin) & 15) != 0)) return false;
  if (e.R && ((e.ldr & 7) != 0 || ((uintptr_t)e.R & 15) != 0)) return false;
  if (e.dact && e.accumulate && !e.c_f32) return false;  // (epilogue_pre_ok)
  if (p.nsplit > 1 && !(e.c_f32 && p.split_stride > 0)) return false;
  if (e.rowstat && (e.c_f32 || e.act || e.Zout || e.dact || e.R || e.drop_thr || e.accumulate)) return false;
  return true;
}
void launch_gemm_d2(const LaunchTable& tab, hipStream_t s) {
  if (tab.p[0].epi.rowstat) launch_d2<1>(tab, s);
  else launch_d2<0>(tab, s);
}

