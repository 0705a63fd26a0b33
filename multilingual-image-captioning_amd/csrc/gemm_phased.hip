// gemm_phased.hip — the 256x256x64 bf16 tile as an LDS-DMA kernel with a four-phase K-tile schedule (launches with >= 200
// such tiles: LM head, grouped weight gradients, FFN).  Same launch table, LDS images, fragment reads and epilogue as
// gemm.hip's kernel; what differs is how operands reach LDS and how the waves take turns:
//
//   * 8 waves = 2 (M) x 4 (N), wave tile 128 x 64 = four 64 x 32 quadrants.  A K-tile is consumed in four phases, one
//     quadrant each; the register sub-tiles are read from LDS exactly once per K-tile (A rows 0-63 and B cols 0-31 in phase 1,
//     A rows 64-127 in phase 2, B cols 32-63 in phase 3, nothing in phase 4: 24 ds_read_b128 per wave like the register-
//     staged kernel, but spread so that every phase reads one operand sub-tile at most).
//   * The four 16-KiB LDS half-tiles of a K-tile are cut BY QUADRANT, not by wave row: A0 = the rows every wave reads in
//     phase 1 (tile rows 0-63 and 128-191), A1 = phase 2's rows, B0 / B1 likewise by 32-column group.  A half-tile is
//     therefore dead one phase after it was read and is refilled (for K-tile t+2) right then, one half-tile = two
//     global_load_lds_dwordx4 per wave per phase: B0 in phase 2, A0 in 3, A1 in 4, B1 in phase 1 of the next K-tile.
//     Three half-tiles stay in flight across every barrier; the only wait is one counted s_waitcnt vmcnt(6) per K-tile
//     (phase 4), after which everything the next K-tile reads has landed.  LDS-DMA writes lane-linear, so the XOR swizzle of
//     the images is applied to each lane's SOURCE address (the permutation is an involution inside one 128-B / 256-B row).
//   * The two waves of a SIMD run one barrier apart (waves 4-7 take one extra s_barrier before the loop, waves 0-3 one after
//     it): while one issues its ds_reads and DMA, the other owns the matrix pipe (s_setprio 1 around its 8 MFMAs).
//   * Every phase is  {ds_reads, DMA, s_waitcnt lgkmcnt(0)} s_barrier {8 x v_mfma_f32_32x32x16_bf16} s_barrier.  The reads
//     are retired BEFORE the first barrier, so the refill issued one phase later can never overtake them, also for the wave
//     group that runs a barrier behind.
#include "gemm_common.h"

namespace {

constexpr int HALF = 16384, STAGE = 4 * HALF;  // [A0 | A1 | B0 | B1] per K-tile buffer

// row R (0..127) of LDS half-tile A[q] / B[q]  ->  row / column inside the 256-wide tile
__device__ __forceinline__ int a_tile_row(int q, int R) { return (R >> 6) * 128 + q * 64 + (R & 63); }
__device__ __forceinline__ int b_tile_col(int q, int R) { return (R >> 5) * 64 + q * 32 + (R & 31); }

// per-lane source pointers (K-tile 0) of the two 1-KiB pieces this wave contributes to one half-tile
template <bool KMAJOR, bool IS_A>
__device__ __forceinline__ void half_sources(const uint16_t* (&g)[2], const uint16_t* src, int ld, int x0, int k0, int lim, int q, int wave,
                                             int lane) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int piece = wave * 2 + i;
    if (!KMAJOR) {  // image [128 rows][64 k]: a piece = 8 rows x 128 B; chunk position p of row R holds source chunk p ^ ((R>>1)&7)
      const int R = piece * 8 + (lane >> 3), c = (lane & 7) ^ ((R >> 1) & 7);
      int gx = x0 + (IS_A ? a_tile_row(q, R) : b_tile_col(q, R));
      gx = gx < lim ? gx : lim - 1;
      g[i] = src + (size_t)gx * ld + k0 + c * 8;
    } else {        // image [64 k][128 x]: a piece = 4 k-rows x 256 B; chunk position p of row k holds source chunk p ^ ((k&3)<<2)
      const int k = piece * 4 + (lane >> 4), c = (lane & 15) ^ ((k & 3) << 2);
      int gx = x0 + (IS_A ? a_tile_row(q, c * 8) : b_tile_col(q, c * 8));
      gx = gx < lim ? gx : 0;
      g[i] = src + (size_t)(k0 + k) * ld + gx;
    }
  }
}

__device__ __forceinline__ void dma_half(const uint16_t* const (&g)[2], long koff, char* lds_half, int wave) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
    __builtin_amdgcn_global_load_lds(GLB_PTR(g[i] + koff), LDS_PTR(void, lds_half + (wave * 2 + i) * 1024), 16, 0, 0);
}

#define PH_WAIT_LGKM() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define PH_BARRIER()                       \
  do {                                     \
    __builtin_amdgcn_sched_barrier(0);     \
    __builtin_amdgcn_s_barrier();          \
    __builtin_amdgcn_sched_barrier(0);     \
  } while (0)

template <bool AK, bool BKM, bool PLAIN>
__global__ __launch_bounds__(512, 2) void gemm_phased_kernel(LaunchTable tab) {
  constexpr int WM = 128, WN = 64, WNW = 4, AI = 4, NJ = 2, BM = 256, BN = 256, BKT = 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 K-tiles][A0 | A1 | B0 | B1]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int lid;
  {
    const int bid = blockIdx.x, nwg = tab.total_blocks;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  int pi = 0;
#pragma unroll
  for (int i = 1; i < MAX_PROBLEMS; ++i)
    if (i < tab.count && lid >= tab.p[i].block_begin) pi = i;
  const Problem P = tab.p[pi];
  const int local = lid - P.block_begin;
  int tile = local / P.nsplit, split = local - tile * P.nsplit;
  if (tab.count == 1 && P.nsplit > 1 && (P.nsplit & 7) == 0) {  // split-K with K-range <-> XCD affinity (see gemm.hip)
    const int T = P.tiles_m * P.tiles_n, S = P.nsplit >> 3, j = blockIdx.x >> 3;
    split = (blockIdx.x & 7) * S + j / T;
    tile = j % T;
  }
  int tm, tn;
  tile_coords(tile, P.tiles_m, P.tiles_n, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int wr = wave >> 2, wc = wave & 3;

  f32x16 acc[AI][NJ];
#pragma unroll
  for (int i = 0; i < AI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  float bsum[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) bsum[i] = 0.0f;
  // row sums of A ride on k-major A operands only (the bias gradient of dW = dy^T x): this kernel is built for NT launches, where
  // the code would sit in the K loop as never-taken branches that cost registers (255 -> 234 VGPRs without it)
  constexpr bool RS = AK;
  const bool do_rowsum = RS && P.a_rowsum != nullptr;
  const int rs_tiles_n = P.tiles_n, rs_k = P.rowsum_k;

  const int nk_total = P.K / BKT;
  const int nk_per = (nk_total + P.nsplit - 1) / P.nsplit;
  const int kt0 = split * nk_per, kt1 = min(nk_total, kt0 + nk_per);
  const int nk = max(kt1 - kt0, 0);

  // source pointers of this lane's DMA pieces (K-tile kt0); a K-tile further on is +kstep elements
  const uint16_t *gA0[2], *gA1[2], *gB0[2], *gB1[2];
  half_sources<AK, true>(gA0, P.A, P.lda, m0, kt0 * BKT, P.M, 0, wave, lane);
  half_sources<AK, true>(gA1, P.A, P.lda, m0, kt0 * BKT, P.M, 1, wave, lane);
  half_sources<BKM, false>(gB0, P.B, P.ldb, n0, kt0 * BKT, P.N, 0, wave, lane);
  half_sources<BKM, false>(gB1, P.B, P.ldb, n0, kt0 * BKT, P.N, 1, wave, lane);
  const long ka = AK ? (long)BKT * P.lda : BKT, kb = BKM ? (long)BKT * P.ldb : BKT;

  if (nk > 0) {
    dma_half(gA0, 0, smem, wave);
    dma_half(gB0, 0, smem + 2 * HALF, wave);
    dma_half(gA1, 0, smem + HALF, wave);
    dma_half(gB1, 0, smem + 3 * HALF, wave);
    if (nk > 1) {  // K-tile 1 minus its B1 (phase 1 of K-tile 0 brings that)
      dma_half(gB0, kb, smem + STAGE + 2 * HALF, wave);
      dma_half(gA0, ka, smem + STAGE, wave);
      dma_half(gA1, ka, smem + STAGE + HALF, wave);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  PH_BARRIER();
  if (wr == 1) PH_BARRIER();  // the second wave of every SIMD runs one barrier behind the first

  for (int t = 0; t < nk; ++t) {
    char* cur = smem + (t & 1) * STAGE;
    char* nxt = smem + ((t + 1) & 1) * STAGE;
    const bool rs_tile = do_rowsum && ((kt0 + t) % rs_tiles_n) == tn;
    bf16x8 a0[2][4], a1[2][4], b0[4], b1[4];
    auto rowsum = [&](const bf16x8 (&af)[2][4], int qi) __attribute__((always_inline)) {
      if constexpr (!RS) return;
      if (!rs_tile) return;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        if ((kk % WNW) != wc) continue;
        const int kb0 = (kt0 + t) * BKT + kk * 16 + 8 * (lane >> 5);
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
          const u32x4 u = __builtin_bit_cast(u32x4, af[i2][kk]);
          const uint32_t w[4] = {u.x, u.y, u.z, u.w};
          float sacc = 0.0f;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float lo = __uint_as_float(w[q] << 16), hi = __uint_as_float(w[q] & 0xffff0000u);
            sacc += (kb0 + 2 * q < rs_k ? lo : 0.0f) + (kb0 + 2 * q + 1 < rs_k ? hi : 0.0f);
          }
          bsum[qi * 2 + i2] += sacc;
        }
      }
    };
    // ---- phase 1: quadrant (0,0); refill B1 of K-tile t+1 (last read in phase 3 of K-tile t-1)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) b0[kk] = read_frag<BKM, BKT, 128>(cur + 2 * HALF, wc * 32, kk, lane);
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) a0[i2][kk] = read_frag<AK, BKT, 128>(cur, wr * 64 + i2 * 32, kk, lane);
    if (t + 1 < nk) dma_half(gB1, (long)(t + 1) * kb, nxt + 3 * HALF, wave);
    PH_WAIT_LGKM();
    PH_BARRIER();
    rowsum(a0, 0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2) acc[i2][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[i2][kk], b0[kk], acc[i2][0], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    PH_BARRIER();
    // ---- phase 2: quadrant (1,0); refill B0 for K-tile t+2 (read in phase 1)
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) a1[i2][kk] = read_frag<AK, BKT, 128>(cur + HALF, wr * 64 + i2 * 32, kk, lane);
    if (t + 2 < nk) dma_half(gB0, (long)(t + 2) * kb, cur + 2 * HALF, wave);
    PH_WAIT_LGKM();
    PH_BARRIER();
    rowsum(a1, 1);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2) acc[2 + i2][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[i2][kk], b0[kk], acc[2 + i2][0], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    PH_BARRIER();
    // ---- phase 3: quadrant (1,1); refill A0 for K-tile t+2 (read in phase 1)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) b1[kk] = read_frag<BKM, BKT, 128>(cur + 3 * HALF, wc * 32, kk, lane);
    if (t + 2 < nk) dma_half(gA0, (long)(t + 2) * ka, cur, wave);
    PH_WAIT_LGKM();
    PH_BARRIER();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2) acc[2 + i2][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[i2][kk], b1[kk], acc[2 + i2][1], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    PH_BARRIER();
    // ---- phase 4: quadrant (0,1) from registers; refill A1 for K-tile t+2 (read in phase 2); the one wait of the K-tile:
    //      all but the three newest half-tiles (6 DMA instructions) have landed = everything K-tile t+1 reads
    if (t + 2 < nk) {
      dma_half(gA1, (long)(t + 2) * ka, cur + HALF, wave);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    PH_BARRIER();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2) acc[i2][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[i2][kk], b1[kk], acc[i2][1], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    PH_BARRIER();
  }
  if (wr == 0) PH_BARRIER();
  __syncthreads();

  if (do_rowsum) {
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const float v = bsum[i] + __shfl_xor(bsum[i], 32, 64);
      const int m = m0 + wr * WM + i * 32 + lane;
      if (lane < 32 && m < P.M) atomicAdd(P.a_rowsum + m, v);
    }
  }
  gemm_epilogue<WM, WN, WNW, 1, PLAIN, 0>(acc, P, smem, m0, n0, split, 0, wave, lane, tid);
}

template <bool AK, bool BKM, bool PLAIN>
void launch_one(const LaunchTable& tab, hipStream_t s) {
  constexpr int lds = 2 * STAGE;
  static bool attr_set_dev[64] = {};  // per instantiation and device
  int dev_ = 0;
  (void)hipGetDevice(&dev_);
  bool& attr_set = attr_set_dev[dev_ & 63];
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_phased_kernel<AK, BKM, PLAIN>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_phased_kernel<AK, BKM, PLAIN>), dim3(tab.total_blocks), dim3(512), lds, s, tab);
}

}  // namespace

// NT launches only (both operands k-contiguous): the dispatcher takes this kernel for the single-problem NT 256x256 launches, where
// it measured faster than the register-staged kernel; on the k-major layouts it lost, those instantiations are not built.
void launch_gemm_phased(const LaunchTable& tab, int akm, int bkm, bool plain, hipStream_t s) {
  if (akm || bkm) { mic_set_error("launch_gemm_phased: NT launches only"); return; }
  if (plain) launch_one<false, false, true>(tab, s);
  else launch_one<false, false, false>(tab, s);
}
