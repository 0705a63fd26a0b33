// gemm_kernel.h — the table-driven bf16 / fp8 MFMA GEMM kernel template and its launcher (included by one translation unit per
// tile configuration: gemm_t256.hip, gemm_t128.hip, gemm_t64.hip — the ~70 instantiations compile in parallel).
#pragma once
#include "gemm_common.h"
#include <type_traits>

// Wave tile WM x WN (WM in {32,64,128}, WN in {32,64}); 2 waves along M, WNW along N.  BM = 2*WM, BN = WN*WNW.
//   <32,32,2>  64x64,   4 waves : launches too small to give every CU a 128x128 tile; 32 KiB LDS, several blocks per CU
//   <64,32,4>  128x128, 8 waves : even one block per CU puts two waves on every SIMD, so one wave's staging / LDS waits
//                                 hide under the other's MFMAs (the 4-wave 64x64 wave tile measured 26 % MFMA-busy at
//                                 one block per CU: a single in-order stream cannot overlap its own waits)
//   <128,64,4> 256x256, 8 waves : launches with >= 200 such tiles (2x the FLOPs per staged byte)
// BKT: k-depth of one pipeline stage.  Two LDS buffers; tile t+1 travels through registers while tile t is multiplied
// (loads issued a full iteration before their ds_write), one barrier per stage.
// KG > 1: KG groups of 2*WNW waves share one output tile and split its K range (each with its own pair of LDS stages);
//   the partial accumulators meet in LDS before the epilogue.  For launches whose tile count cannot fill the chip the
//   K loop is a latency chain (load -> ds_write -> barrier -> ds_read -> MFMA, ~0.5 us per 64-k step at one block per CU);
//   KG groups cut the chain KG-fold without atomics or extra launches.
// PLAIN: every problem of the launch has the bare epilogue (C = dropout(alpha*acc + bias) + residual, bf16 or fp32, no
//   activation / Z / accumulate / split): the store loop is then two ds_read_b128, four v_cvt_pk_bf16_f32 and one 16-B
//   store per group, without the per-group feature tests of the generic path.
// F8: 0 = bf16 operands; 1 = fp8 e4m3 x e4m3 (forward); 2 = A e5m2 x B e4m3 (gradients x weights / saved activations).
//   fp8 launches are NT only (both operands k-contiguous: the quantiser writes a transposed copy where one is needed); all
//   sizes below stay in 2-byte units (K, lda, ldb = bytes / 2), so the staging code is shared.
template <int WM, int WN, int WNW, int BKT, bool AK, bool BKM, int KG = 1, bool PLAIN = false, int F8 = 0>
__global__ __launch_bounds__(128 * WNW * KG, KG > 1 ? 1 : 2) void gemm_bf16_kernel(LaunchTable tab) {
  static_assert(F8 == 0 || (AK == BKM && BKT == 64), "fp8: NT (both operands k-contiguous) or TN (both k-major), 128-k stages");
  constexpr bool F8T = F8 != 0 && AK;  // fp8 weight-gradient launches: [128 k][128 x] byte images, ds_read_b64_tr_b8 fragments
  static_assert(!F8T || (WM % 64 == 0 && WN * WNW % 128 == 0), "fp8 TN: 128-wide images (128x128 / 256x256 tiles)");
  constexpr int BM = 2 * WM, BN = WN * WNW, NWAVES = 2 * WNW, NTHREADS = 64 * NWAVES * KG;
  // rows per staged image: 128, or 64 for the 64-wide tiles and for BM = 192 (three 64-row images side by side: a wave's 96 rows
  // start at row 0 or 96 and run across image boundaries — images of a k-contiguous operand are contiguous 128-B rows whose
  // swizzle depends on (row >> 1) & 7 only, so the fragment reads need no image arithmetic)
  constexpr int UA = (BM % 128 == 0) ? 128 : 64, UB = BN < 128 ? BN : 128;
  static_assert(BM % 128 == 0 || BM == 64 || !AK, "BM = 192: k-contiguous A only (NT / NN launches)");
  constexpr int HALF_A = UA * BKT * 2, HALF_B = UB * BKT * 2;
  constexpr int NHA = BM / UA, NHB = BN / UB, STAGE = NHA * HALF_A + NHB * HALF_B, AI = WM / 32, NJ = WN / 32, KSTEPS = BKT / 16;
  constexpr int STAGE_AT = 0;  // k-step in front of which the next tile's LDS writes / global loads are issued (1..3 measured equal)
  using SA = std::conditional_t<F8T, Half8Stager<NWAVES>, HalfStager<AK, NWAVES, BKT, UA>>;
  using SB = std::conditional_t<F8T, Half8Stager<NWAVES>, HalfStager<BKM, NWAVES, BKT, UB>>;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][A halves | B halves]
  const int tid = threadIdx.x, lane = tid & 63, wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = wave_all / NWAVES, wave = wave_all % NWAVES;  // K-group, wave within the group
  // Persistent launches (PLAIN 256x256 instantiations only — the others have no registers to spare for the loop state; grid <
  // total_blocks, a multiple of 8): a block walks the tiles bid, bid + grid, ... — no block retirement / dispatch gap between
  // two tiles of a CU.  Every other instantiation runs the body once.
  constexpr bool PERSIST = PLAIN && WM == 128 && KG == 1;
  int bid = blockIdx.x;
  do {
  // bijective XCD remap: the blocks that land on XCD x (= bid % 8) get a contiguous run of logical block ids
  int lid;
  {
    const int nwg = tab.total_blocks;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  int pi = 0;
#pragma unroll
  for (int i = 1; i < MAX_PROBLEMS; ++i)
    if (i < tab.count && lid >= tab.p[i].block_begin) pi = i;
  const Problem P = tab.p[pi];  // by value: one burst of scalar loads up front instead of a kernarg load (and wait) at every use
  const int local = lid - P.block_begin;
  int tile = local / P.nsplit, split = local - tile * P.nsplit;
  if (tab.count == 1 && P.nsplit > 1 && (P.nsplit & 7) == 0) {
    // split-K with K-range <-> XCD affinity: XCD x owns the K-chunks [x*S, (x+1)*S) and walks them chunk by chunk over ALL
    // output tiles, so the ~32 blocks resident on an XCD read the same K-range of A and B at the same time (one HBM read
    // per operand byte; tile-major order made every block stream private panels: fabric-bound, no gain over no split)
    const int T = P.tiles_m * P.tiles_n, S = P.nsplit >> 3, j = bid >> 3;
    split = (bid & 7) * S + j / T;
    tile = j % T;
  }
  int tm, tn;
  tile_coords(tile, P.tiles_m, P.tiles_n, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int M = P.M, N = P.N;
  const uint16_t* __restrict__ A = P.A;
  const uint16_t* __restrict__ B = P.B;
  const int lda = P.lda, ldb = P.ldb;
  const int wr = wave / WNW, wc = wave % WNW;

  f32x16 acc[AI][NJ];
#pragma unroll
  for (int i = 0; i < AI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // optional row sums of A (bias gradient when A = dy^T): the K-tiles are dealt round-robin to the tile columns tn and the
  // k-steps to the wave columns wc, so every wave adds 8 fragment values on 1/(tiles_n*WNW) of its k-steps
  // (compiled only into the A-k-major instantiations — the bias gradient rides on dW = dy^T x; the NT / NN kernels run at the
  // 256-VGPR limit and have no registers for code they never execute)
  constexpr bool RS = AK && F8 == 0;
  float bsum[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) bsum[i] = 0.0f;
  const bool do_rowsum = RS && P.a_rowsum != nullptr;
  const int rs_tiles_n = P.tiles_n, rs_k = P.rowsum_k;  // registers: a kernarg load inside the K loop also waits on the LDS reads

  const int nk_total = P.K / BKT;
  const int nk_per = (nk_total + P.nsplit - 1) / P.nsplit;
  const int bt0 = split * nk_per, bt1 = min(nk_total, bt0 + nk_per);   // this block's K-tiles
  const int g_per = (max(bt1 - bt0, 0) + KG - 1) / KG;                   // ... dealt to the K-groups
  const int kt0 = bt0 + kg * g_per, kt1 = min(bt1, kt0 + g_per);
  const int nk = max(kt1 - kt0, 0), nk_loop = KG > 1 ? g_per : nk;      // every group runs nk_loop barriers
  char* const gsm = smem + kg * 2 * STAGE;                               // this group's two stages

  u32x4 ra[NHA][SA::PER], rb[NHB][SB::PER];  // the tile in flight
  auto load_regs = [&](int t) __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < NHA; ++h) SA::load(ra[h], A, lda, m0 + h * UA, (kt0 + t) * BKT, M, wave, lane);
#pragma unroll
    for (int h = 0; h < NHB; ++h) SB::load(rb[h], B, ldb, n0 + h * UB, (kt0 + t) * BKT, N, wave, lane);
  };
  // weight-gradient launches (both operands k-major, see mic_gemm_args.k_valid): reduction rows k >= k_valid are written to LDS
  // as zeros; `t` = the K-tile being written
  constexpr bool KV = AK && BKM;
  constexpr int KROWS = F8T ? 2 * BKT : BKT;  // reduction rows per K-tile (fp8: K counts 2-byte units)
  const int k_valid = KV ? P.k_valid : 0x7fffffff;
  auto write_lds = [&](char* buf, int t) __attribute__((always_inline)) {
    const int kr = KV ? k_valid - (kt0 + t) * KROWS : 0x7fffffff;  // valid rows of this K-tile (>= KROWS: all)
#pragma unroll
    for (int h = 0; h < NHA; ++h) SA::store(ra[h], buf + h * HALF_A, wave, lane, kr);
#pragma unroll
    for (int h = 0; h < NHB; ++h) SB::store(rb[h], buf + NHA * HALF_A + h * HALF_B, wave, lane, kr);
  };
  // this wave's operand sub-images
  const int a_half = (wr * WM) / UA, a_off = (wr * WM) % UA;
  const int b_half = (wc * WN) / UB, b_off = (wc * WN) % UB;

  if (nk > 0) {
    load_regs(0);
    write_lds(gsm, 0);
    if (nk > 1) load_regs(1);
  }
  if (nk_loop > 0) __syncthreads();
  for (int t = 0; t < nk_loop; ++t) {
    if (t < nk) {
      const bool rs_tile = do_rowsum && ((kt0 + t) % rs_tiles_n) == tn;  // this block's share of the A row sums
      const char* cur = gsm + (t & 1) * STAGE;
      const char* At = cur + a_half * HALF_A;
      const char* Bt = cur + NHA * HALF_A + b_half * HALF_B;
      if constexpr (F8 != 0 && WM <= 64) {
        // fp8, small wave tiles: both halves' fragment reads in front of the MFMAs (as for bf16 below)
        if (t + 1 < nk) {
          write_lds(gsm + ((t + 1) & 1) * STAGE, t + 1);
          if (t + 2 < nk) load_regs(t + 2);
        }
        i32x8 a8[2][AI], b8[2][NJ];
#pragma unroll
        for (int mm = 0; mm < 2; ++mm) {
#pragma unroll
          for (int i = 0; i < AI; ++i) a8[mm][i] = F8T ? read_frag8_tr(At, a_off + i * 32, mm, lane) : read_frag8(At, a_off + i * 32, mm, lane);
#pragma unroll
          for (int j = 0; j < NJ; ++j) b8[mm][j] = F8T ? read_frag8_tr(Bt, b_off + j * 32, mm, lane) : read_frag8(Bt, b_off + j * 32, mm, lane);
        }
#pragma unroll
        for (int mm = 0; mm < 2; ++mm)
#pragma unroll
          for (int i = 0; i < AI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[mm][i], b8[mm][j], acc[i][j], F8 == 2 ? 1 : 0, 0, 0, 0x7f7f7f7f, 0,
                                                                          0x7f7f7f7f);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * (F8T ? 4 : 2) * (AI + NJ), 0);  // two 16-B reads (TN: four transposing 8-B reads) per fragment
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * AI * NJ, 0);
      } else if constexpr (F8 != 0) {
#pragma unroll
        for (int mm = 0; mm < 2; ++mm) {  // two K = 64 MFMAs per 128-byte stage
          if (mm == 0 && t + 1 < nk) {
            write_lds(gsm + ((t + 1) & 1) * STAGE, t + 1);
            if (t + 2 < nk) load_regs(t + 2);
          }
          i32x8 a8[AI], b8[NJ];
#pragma unroll
          for (int i = 0; i < AI; ++i) a8[i] = F8T ? read_frag8_tr(At, a_off + i * 32, mm, lane) : read_frag8(At, a_off + i * 32, mm, lane);
#pragma unroll
          for (int j = 0; j < NJ; ++j) b8[j] = F8T ? read_frag8_tr(Bt, b_off + j * 32, mm, lane) : read_frag8(Bt, b_off + j * 32, mm, lane);
#pragma unroll
          for (int i = 0; i < AI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[i], b8[j], acc[i][j], F8 == 2 ? 1 : 0, 0, 0, 0x7f7f7f7f, 0,
                                                                          0x7f7f7f7f);  // cbsz/blgp: 0 = e4m3, 1 = e5m2; scales 2^0
        }
      } else if constexpr (WM == 32 || (WM == 64 && !AK)) {
        // small wave tiles (32 x 32: one MFMA per k-step on one accumulator; 64 x 32: two): the K-tile's fragment reads all go
        // out together, then the MFMAs (the compiler otherwise issues k-step kk+1's reads behind the MFMAs of kk and waits a full
        // LDS latency in every k-step; these launches are latency chains, not throughput)
        // Order inside the iteration: this tile's fragment reads FIRST, then the next tile's LDS writes (its buffer was last read
        // in iteration t-1; the data was requested a whole iteration ago) and the global loads of the tile after it, then the
        // MFMAs.  The LDS port serves the reads first, so the MFMAs start as soon as the first fragments arrive and run beside the
        // remaining reads AND the writes; with the writes in front (as before) every wave's MFMAs waited for the write pass.
        bf16x8 af[KSTEPS][AI], bfr[KSTEPS][NJ];
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
#pragma unroll
          for (int i = 0; i < AI; ++i) af[kk][i] = read_frag<AK, BKT, UA>(At, a_off + i * 32, kk, lane);
#pragma unroll
          for (int j = 0; j < NJ; ++j) bfr[kk][j] = read_frag<BKM, BKT, UB>(Bt, b_off + j * 32, kk, lane);
        }
        if (t + 1 < nk) {
          write_lds(gsm + ((t + 1) & 1) * STAGE, t + 1);
          if (t + 2 < nk) load_regs(t + 2);
        }
        if constexpr (RS) {
          if (rs_tile) {
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
              if ((kk % WNW) != wc) continue;
              const int kb = (kt0 + t) * BKT + kk * 16 + 8 * (lane >> 5);  // this lane's 8 consecutive k
#pragma unroll
              for (int i = 0; i < AI; ++i) {
                const u32x4 u = __builtin_bit_cast(u32x4, af[kk][i]);
                const uint32_t w[4] = {u.x, u.y, u.z, u.w};
                float sacc = 0.0f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  const float lo = __uint_as_float(w[q] << 16), hi = __uint_as_float(w[q] & 0xffff0000u);
                  sacc += (kb + 2 * q < rs_k ? lo : 0.0f) + (kb + 2 * q + 1 < rs_k ? hi : 0.0f);
                }
                bsum[i] += sacc;
              }
            }
          }
        }
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk)
#pragma unroll
          for (int i = 0; i < AI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk][i], bfr[kk][j], acc[i][j], 0, 0, 0);
        // pin the order (the scheduler otherwise sinks every read next to its MFMA): all DS reads, the DS writes, then the MFMAs
        __builtin_amdgcn_sched_group_barrier(0x100, ((AK ? 2 : 1) * AI + (BKM ? 2 : 1) * NJ) * KSTEPS, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, NHA * SA::PER + NHB * SB::PER, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, KSTEPS * AI * NJ, 0);
      } else {
#pragma unroll
      for (int kk = 0; kk < KSTEPS; ++kk) {
        if (kk == STAGE_AT && t + 1 < nk) {
          write_lds(gsm + ((t + 1) & 1) * STAGE, t + 1);  // its buffer was last read in iteration t-1 (barrier below)
          if (t + 2 < nk) load_regs(t + 2);        // a full iteration of MFMAs to land
        }
        bf16x8 af[AI], bfr[NJ];
#pragma unroll
        for (int i = 0; i < AI; ++i) af[i] = read_frag<AK, BKT, UA>(At, a_off + i * 32, kk, lane);
#pragma unroll
        for (int j = 0; j < NJ; ++j) bfr[j] = read_frag<BKM, BKT, UB>(Bt, b_off + j * 32, kk, lane);
        if constexpr (RS) if (rs_tile && (kk % WNW) == wc) {
          const int kb = (kt0 + t) * BKT + kk * 16 + 8 * (lane >> 5);  // this lane's 8 consecutive k
#pragma unroll
          for (int i = 0; i < AI; ++i) {
            const u32x4 u = __builtin_bit_cast(u32x4, af[i]);
            const uint32_t w[4] = {u.x, u.y, u.z, u.w};
            float sacc = 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float lo = __uint_as_float(w[q] << 16), hi = __uint_as_float(w[q] & 0xffff0000u);
              sacc += (kb + 2 * q < rs_k ? lo : 0.0f) + (kb + 2 * q + 1 < rs_k ? hi : 0.0f);
            }
            bsum[i] += sacc;
          }
        }
#pragma unroll
        for (int i = 0; i < AI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
          }
      }
      }
    }
    __syncthreads();
  }

  if constexpr (KG > 1) {  // groups 1.. hand their partial sums to group 0 through LDS (register-image layout, lane-linear)
    float* red = reinterpret_cast<float*>(smem);
    constexpr int PER_WAVE = AI * NJ * 16 * 64;
    if (kg > 0) {
      float* dst = red + ((kg - 1) * NWAVES + wave) * PER_WAVE;
#pragma unroll
      for (int i = 0; i < AI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) dst[((i * NJ + j) * 16 + r) * 64 + lane] = acc[i][j][r];
    }
    __syncthreads();
    if (kg == 0) {
#pragma unroll
      for (int g = 1; g < KG; ++g) {
        const float* src = red + ((g - 1) * NWAVES + wave) * PER_WAVE;
#pragma unroll
        for (int i = 0; i < AI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] += src[((i * NJ + j) * 16 + r) * 64 + lane];
      }
    }
    __syncthreads();
  }

  if constexpr (RS) if (do_rowsum) {
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const float v = bsum[i] + __shfl_xor(bsum[i], 32, 64);  // the two k-halves of the fragment
      const int m = m0 + wr * WM + i * 32 + lane;
      if (lane < 32 && m < M) atomicAdd(P.a_rowsum + m, v);
    }
  }

  gemm_epilogue<WM, WN, WNW, KG, PLAIN, F8>(acc, P, smem, m0, n0, split, kg, wave, lane, tid);
  if constexpr (!PERSIST) break;
  bid += gridDim.x;
  if (bid >= tab.total_blocks) break;
  __syncthreads();  // the next tile's prologue rewrites the LDS the epilogue restaged through
  } while (true);
}

template <int WM, int WN, int WNW, int BKT, int KG, bool PLAIN, int F8 = 0>
static inline void launch_cfg_p(const LaunchTable& tab, int akm, int bkm, hipStream_t s) {
  constexpr int BM = 2 * WM, BN = WN * WNW;
  size_t lds = (size_t)KG * 2 * (BM + BN) * BKT * 2;
  const size_t epi = (size_t)2 * WNW * (WM % 64 == 0 ? 64 : 32) * WN * 4;  // the epilogue restages 64 (or 32) x WN floats per wave
  const size_t red = (size_t)(KG - 1) * 2 * WNW * (WM / 32) * (WN / 32) * 16 * 64 * 4;  // K-group partial sums
  if (epi > lds) lds = epi;
  if (red > lds) lds = red;
  // 256x256 tiles hold a CU alone (128 KiB LDS): launches with more tiles than (free) CUs run as that many persistent blocks
  static const int persist = [] { const char* e = getenv("MIC_GEMM_PERSIST"); return e ? atoi(e) : 1; }();
  int nblk = tab.total_blocks;
  const int cus = mic_cu_budget_now();  // a multiple of 8 (the XCD remap of a persistent grid needs that)
  if (persist && BM == 256 && PLAIN && KG == 1 && nblk > cus) nblk = cus;
  dim3 grid(nblk), block(128 * WNW * KG);
#define LAUNCH(AKM, BKMM)                                                                                                  \
  do {                                                                                                                     \
    static bool attr_set_dev[64] = {}; /* per instantiation AND device (the attribute belongs to the device's code object) */ \
    int dev_ = 0;                                                                                                          \
    (void)hipGetDevice(&dev_);                                                                                             \
    bool& attr_set = attr_set_dev[dev_ & 63];                                                                              \
    if (lds > 65536 && !attr_set) {                                                                                        \
      hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<WM, WN, WNW, BKT, AKM, BKMM, KG, PLAIN, F8>),    \
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                           \
      attr_set = true;                                                                                                     \
    }                                                                                                                      \
    hipLaunchKernelGGL((gemm_bf16_kernel<WM, WN, WNW, BKT, AKM, BKMM, KG, PLAIN, F8>), grid, block, lds, s, tab);                 \
  } while (0)
  if constexpr (F8 != 0) {
    if constexpr ((2 * WM) % 128 == 0 && (WN * WNW) % 128 == 0) {  // 128x128 / 256x256 tiles also carry the TN (weight-gradient) form
      if (akm && bkm) LAUNCH(true, true);
      else LAUNCH(false, false);
    } else {
      LAUNCH(false, false);
    }
  } else if constexpr (WM == 96) {  // BM = 192: k-contiguous A only
    if (!bkm) LAUNCH(false, false);
    else LAUNCH(false, true);
  } else {
    if (!akm && !bkm) LAUNCH(false, false);
    else if (!akm && bkm) LAUNCH(false, true);
    else if (akm && bkm) LAUNCH(true, true);
    else LAUNCH(true, false);
  }
#undef LAUNCH
}

template <int WM, int WN, int WNW, int BKT, int KG = 1>
static inline void launch_cfg(const LaunchTable& tab, int akm, int bkm, hipStream_t s, int f8 = 0) {
  const bool plain = table_is_plain(tab);
  if (f8 == 1) {
    if (plain) launch_cfg_p<WM, WN, WNW, BKT, KG, true, 1>(tab, akm, bkm, s);
    else launch_cfg_p<WM, WN, WNW, BKT, KG, false, 1>(tab, akm, bkm, s);
  } else if (f8 == 2) {
    if (plain) launch_cfg_p<WM, WN, WNW, BKT, KG, true, 2>(tab, akm, bkm, s);
    else launch_cfg_p<WM, WN, WNW, BKT, KG, false, 2>(tab, akm, bkm, s);
  } else if (plain) launch_cfg_p<WM, WN, WNW, BKT, KG, true>(tab, akm, bkm, s);
  else launch_cfg_p<WM, WN, WNW, BKT, KG, false>(tab, akm, bkm, s);
}

