// gemm_common.h — device code shared by the MFMA GEMM kernels (gemm.hip: the one-barrier-per-K-tile kernel family;
// gemm_phased.hip: the 256x256 LDS-DMA kernel with the four-phase K-tile schedule): launch table, operand staging through
// registers, MFMA fragment reads from the swizzled LDS images, the XCD-aware tile order and the fused epilogue.
#pragma once
#include "common.h"

#define MAX_PROBLEMS 8
#ifndef MIC_TINY_BELOW
#define MIC_TINY_BELOW 128  // 128x128-tile count under which a launch uses 64x64 tiles (round-1 tile-configuration sweep)
#endif

struct Problem {
  const uint16_t* A; const uint16_t* B;
  int lda, ldb, M, N, K;
  int tiles_m, tiles_n, block_begin, nsplit;
  float* a_rowsum; int rowsum_k;
  int k_valid;  // k-major operands: rows k >= k_valid read as zero (packed batches leave stale rows behind the valid ones)
  long long split_stride;
  const float* sa; const float* sb;  // fp8: device scalars, the operands' dequantisation factors (1 / quantisation scale)
  EpiArgs epi;
};
struct LaunchTable { int count; int total_blocks; Problem p[MAX_PROBLEMS]; };
// CUs the planner sizes one-round launches and persistent grids for (gemm.hip: mic_set_cu_budget / MIC_FREE_CUS; multiple of 8)
int mic_cu_budget_now();
// gemm_phased.hip: 256x256 tiles, LDS-DMA operands, four-phase K-tile schedule (bf16 operands, no K-groups)
void launch_gemm_phased(const LaunchTable& tab, int akm, int bkm, bool plain, hipStream_t s);
// gemm_w4.hip: 256x256 tiles on four waves (128x128 wave tiles, one wave per SIMD), LDS-DMA ring of 32-k steps; NT, single problem
bool gemm_w4_takes(const LaunchTable& tab);
void launch_gemm_w4(const LaunchTable& tab, hipStream_t s);
// gemm_d2.hip: 256x128 tiles on four waves (128x64 wave tiles), two blocks per CU, LDS-DMA ring of 32-k steps; NT, single problem
bool gemm_d2_takes(const LaunchTable& tab);
void launch_gemm_d2(const LaunchTable& tab, hipStream_t s);
// one translation unit per tile configuration of the main kernel (gemm_kernel.h): 256x256 / 128x128 (K-groups 1, 2) / 64x64 (1, 2, 4)
bool table_is_plain(const LaunchTable& t);
void launch_gemm_t256(const LaunchTable& tab, int akm, int bkm, hipStream_t s, int f8);
void launch_gemm_t128(const LaunchTable& tab, int akm, int bkm, hipStream_t s, int f8, int kgroups);
void launch_gemm_t192(const LaunchTable& tab, int bkm, hipStream_t s);  // 192 x 128 tiles, bf16, k-contiguous A (NT / NN)
void launch_gemm_t64(const LaunchTable& tab, int akm, int bkm, hipStream_t s, int f8, int kgroups);

// --- stage one operand image (ROWS x BKT k, or BKT k x ROWS x; ROWS = 128 or 64) HBM/L2 -> registers -> LDS.
//     Measured on gfx950: a global_load_lds (LDS-DMA) instruction costs ~100 cycles of issue time in the issuing wave's
//     in-order stream; 8 of them per K-tile next to 16 MFMAs made every one-block-per-CU shape DMA-issue bound
//     (1.3-1.4k cycles per K-tile against 512 MFMA cycles).  global_load_dwordx4 + ds_write_b128 issue in ~20 cycles a
//     pair, cost 4 VGPRs per piece, and let the swizzle sit on the LDS destination.
//     KMAJOR=false: src is [rows][ld] k-contiguous.  KMAJOR=true: src is [K][ld] x-contiguous.  `lim` = number of valid
//     rows (resp. x) in src; out-of-range rows/chunks are redirected to a valid address (masked at the store).
template <bool KMAJOR, int NWAVES, int BKT, int ROWS>
struct HalfStager {
  static_assert(ROWS == 128 || (ROWS == 64 && BKT == 64), "image = 128 rows (x) of BKT k, or 64 rows of 64 k");
  static constexpr int NINST = ROWS * BKT * 2 / 1024;  // 1 KiB pieces per image
  static constexpr int PER = NINST / NWAVES;
  static_assert(PER >= 1, "too many waves for this image");
  static __device__ __forceinline__ void coords(int q, int lane, int& row, int& c) {
    if (!KMAJOR) {
      if (BKT == 64) { row = q * 8 + (lane >> 3); c = lane & 7; }
      else { row = q * 16 + (lane >> 2); c = lane & 3; }
    } else if (ROWS == 128) { row = q * 4 + (lane >> 4); c = lane & 15; }   // [BKT k][128 x]: 256-B rows
    else { row = q * 8 + (lane >> 3); c = lane & 7; }                       // [64 k][64 x]: 128-B rows
  }
  static __device__ __forceinline__ void load(u32x4 (&r)[PER], const uint16_t* __restrict__ src, int ld, int x0, int k0, int lim,
                                              int wave, int lane) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      int row, c;
      coords(wave * PER + i, lane, row, c);
      const uint16_t* g;
      if (!KMAJOR) {
        int gr = x0 + row;
        gr = gr < lim ? gr : lim - 1;
        g = src + (size_t)gr * ld + k0 + c * 8;
      } else {
        int gx = x0 + c * 8;
        gx = gx < lim ? gx : 0;
        g = src + (size_t)(k0 + row) * ld + gx;
      }
      r[i] = *reinterpret_cast<const u32x4*>(g);
    }
  }
  // k_rows_valid (k-major images only): rows k >= k_rows_valid of this K-tile are written as zeros — the mask sits on the LDS
  // write, where the loaded registers are consumed anyway (masking behind the load would wait for it on the spot)
  static __device__ __forceinline__ void store(const u32x4 (&r)[PER], char* lds_tile, int wave, int lane, int k_rows_valid = 0x7fffffff) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      int row, c;
      coords(wave * PER + i, lane, row, c);
      int off;
      if (!KMAJOR) off = BKT == 64 ? row * 128 + ((c ^ ((row >> 1) & 7)) << 4) : row * 64 + ((c ^ ((row >> 2) & 3)) << 4);
      else if (ROWS == 128) off = row * 256 + ((c ^ ((row & 3) << 2)) << 4);
      else off = row * 128 + ((c ^ (((row >> 1) & 1) << 2)) << 4);
      u32x4 v = r[i];
      if (KMAJOR && row >= k_rows_valid) v = u32x4{0u, 0u, 0u, 0u};
      *reinterpret_cast<u32x4*>(lds_tile + off) = v;
    }
  }
};

// --- read one 32(x) x 16(k) MFMA operand fragment: 8 consecutive k (kk*16 + 8*(lane>>5) ..) of x = xb + (lane&31)
//     k-contiguous images: BKT=64 -> 128-B rows, chunk ^ ((row>>1)&7); BKT=32 -> 64-B rows, chunk ^ ((row>>2)&3); both are
//     conflict-free for ds_read_b128's 16-lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31}.
//     k-major images: [k][128 x] 256-B rows, chunk ^ ((k&3)<<2): the 4 k-rows one ds_read_b64_tr_b16 half-wave touches land
//     on 4 different 64-B bank groups; [k][64 x] 128-B rows, chunk ^ (((k>>1)&1)<<2): rows k and k+2 would share a bank half.
template <bool KMAJOR, int BKT, int ROWS>
__device__ __forceinline__ bf16x8 read_frag(const char* lds_tile, int xb, int kk, int lane) {
  if (!KMAJOR) {
    const int row = xb + (lane & 31);
    const int kc = kk * 2 + (lane >> 5);
    if (BKT == 64) return *reinterpret_cast<const bf16x8*>(lds_tile + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4));
    return *reinterpret_cast<const bf16x8*>(lds_tile + row * 64 + ((kc ^ ((row >> 2) & 3)) << 4));
  } else {
    const int g = lane >> 4, p = lane & 15;
    const int x = xb + 16 * (g & 1) + (p & 3) * 4;
    const int k = kk * 16 + 8 * (g >> 1) + (p >> 2);  // k & 3 == p >> 2 for both halves (k+4 keeps k&3 and (k>>1)&1)
    constexpr int RB = ROWS * 2;
    const int off = ROWS == 128 ? k * 256 + ((((x >> 3) ^ ((k & 3) << 2)) << 4) | ((x & 7) << 1))
                                : k * 128 + ((((x >> 3) ^ (((k >> 1) & 1) << 2)) << 4) | ((x & 7) << 1));
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, lds_tile + off));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, lds_tile + off + 4 * RB));
    s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, r);
  }
}

// fp8 operands (OCP e4m3 / e5m2, one byte per element) ride the SAME k-contiguous LDS image: a 128-B row is 128 k instead of
// 64, and one v_mfma_scale_f32_32x32x64_f8f6f4 takes 32 bytes per lane = two 16-B chunks.  Lane l reads row xb + (l & 31),
// chunks 4*mm + 2*(l >> 5) and the next one.  Which k a (lane half, byte) slot means inside the instruction is irrelevant
// as long as A and B are loaded by the same rule (the contraction is a sum over matching slots).
typedef int i32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ i32x8 read_frag8(const char* lds_tile, int xb, int mm, int lane) {
  const int row = xb + (lane & 31);
  const int kc = mm * 4 + 2 * (lane >> 5);
  const int sw = (row >> 1) & 7;
  const u32x4 lo = *reinterpret_cast<const u32x4*>(lds_tile + row * 128 + (((kc) ^ sw) << 4));
  const u32x4 hi = *reinterpret_cast<const u32x4*>(lds_tile + row * 128 + (((kc + 1) ^ sw) << 4));
  i32x8 r = {(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
  return r;
}

// fp8 K-MAJOR operands (the weight-gradient GEMM dW = dy^T x reads dy and x as the producers wrote them, [rows = k][features = x],
// one byte per element: no transposed copies).  Image = [128 k][128 x] bytes: k-rows of 128 B, 16-B chunk c of row k stored at
// c ^ (((k >> 1) & 3) << 1).  Fragments by ds_read_b64_tr_b8 (tools/probe_tr8.hip: a 16-lane group reads an 8 (k) x 16 (x) byte block,
// lanes 2j / 2j+1 supply the two 8-byte halves of row j, lane l receives column l = 8 consecutive k of ONE x): four of them give a
// lane the 32 consecutive k of x = xb + (lane & 31) that v_mfma_scale_f32_32x32x64_f8f6f4 wants (k half = lane >> 5, as in
// read_frag8).  A half-wave's two groups touch 8 rows x 2 chunks = 16 distinct 16-B slots of the 256-B bank row: conflict-free.
typedef int i32x2 __attribute__((ext_vector_type(2)));
template <int NWAVES>
struct Half8Stager {
  static constexpr int NINST = 16, PER = NINST / NWAVES;  // 1 KiB pieces (8 k-rows) per image
  static_assert(PER >= 1, "too many waves for this image");
  // src: bytes [K rows][ld bytes]; x0 / lim in elements (= bytes), k0 in 2-byte units like every K of the fp8 launches (rows = 2 k0)
  static __device__ __forceinline__ void load(u32x4 (&r)[PER], const uint16_t* __restrict__ src, int ld, int x0, int k0, int lim, int wave, int lane) {
    const uint8_t* s8 = reinterpret_cast<const uint8_t*>(src);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int row = (wave * PER + i) * 8 + (lane >> 3), c = lane & 7;
      int gx = x0 + c * 16;
      gx = gx < lim ? gx : 0;
      r[i] = *reinterpret_cast<const u32x4*>(s8 + (size_t)(2 * k0 + row) * ld + gx);
    }
  }
  static __device__ __forceinline__ void store(const u32x4 (&r)[PER], char* lds_tile, int wave, int lane, int k_rows_valid = 0x7fffffff) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int row = (wave * PER + i) * 8 + (lane >> 3), c = lane & 7;
      u32x4 v = r[i];
      if (row >= k_rows_valid) v = u32x4{0u, 0u, 0u, 0u};
      *reinterpret_cast<u32x4*>(lds_tile + row * 128 + ((c ^ (((row >> 1) & 3) << 1)) << 4)) = v;
    }
  }
};
__device__ __forceinline__ i32x8 read_frag8_tr(const char* lds_tile, int xb, int mm, int lane) {
  const int g = lane >> 4, p = lane & 15;
  const int x = xb + 16 * (g & 1) + 8 * (p & 1);
  const int k = mm * 64 + 32 * (g >> 1) + (p >> 1);  // + 8 r for read r: (k >> 1) & 3 does not depend on r, the four reads are 1 KiB apart
  const char* a = lds_tile + k * 128 + ((((x >> 4) ^ (((k >> 1) & 3) << 1)) << 4) | (x & 15));
  const i32x2 r0 = __builtin_amdgcn_ds_read_tr8_b64_v2i32(LDS_PTR(i32x2, a));
  const i32x2 r1 = __builtin_amdgcn_ds_read_tr8_b64_v2i32(LDS_PTR(i32x2, a + 1024));
  const i32x2 r2 = __builtin_amdgcn_ds_read_tr8_b64_v2i32(LDS_PTR(i32x2, a + 2048));
  const i32x2 r3 = __builtin_amdgcn_ds_read_tr8_b64_v2i32(LDS_PTR(i32x2, a + 3072));
  i32x8 r = {r0[0], r0[1], r1[0], r1[1], r2[0], r2[1], r3[0], r3[1]};
  return r;
}

__device__ __forceinline__ void tile_coords(int lid, int tiles_m, int tiles_n, int& tm, int& tn) {
  const int GROUP_M = 8;  // GROUP_M-tall column panels
  const int per_group = GROUP_M * tiles_n;
  const int gidx = lid / per_group;
  const int first_m = gidx * GROUP_M;
  const int gsz = min(tiles_m - first_m, GROUP_M);
  const int in_g = lid - gidx * per_group;
  tm = first_m + in_g % gsz;
  tn = in_g / gsz;
}


// ---- fused epilogue (shared): accumulators -> LDS restage -> (row, 8 columns) units -> bias / activation / dropout / residual /
// Z / C as 16-B coalesced vectors.  `acc` is the wave's WM x WN block (32x32 MFMA blocks [WM/32][WN/32]); waves are laid out
// 2 x WNW over the tile; only K-group 0 holds the final sums.  Must be entered by every thread of the block after a barrier
// (the LDS stages are reused).
template <int WM, int WN, int WNW, int KG, bool PLAIN, int F8>
__device__ __forceinline__ void gemm_epilogue(f32x16 (&acc)[WM / 32][WN / 32], const Problem& P, char* smem, int m0, int n0, int split,
                                              int kg, int wave, int lane, int tid) {
  constexpr int NWAVES = 2 * WNW, NTHREADS = 64 * NWAVES * KG, AI = WM / 32, NJ = WN / 32;
  const int wr = wave / WNW, wc = wave % WNW;
  const int M = P.M, N = P.N;
  // epilogue.  Everything it needs from the launch table is copied into registers first: P lives in the kernarg segment
  // behind a dynamic index, and inside the store loop every field access was a scalar load + wait (the compiler cannot hoist
  // them across the global stores).  alpha and bias are folded into the accumulators here — a lane owns NJ columns, so the
  // bias is NJ scalar loads per lane instead of a 32-B load (and a vmcnt wait behind the previous stores) per 8-column group.
  EpiArgs E = P.epi;
  const bool is_split = P.nsplit > 1;
  const long long split_stride = P.split_stride;
  {
    float bj[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int n = n0 + wc * WN + j * 32 + (lane & 31);
      bj[j] = (E.bias && !is_split && n < N) ? E.bias[n] : 0.0f;
    }
    float alpha = E.alpha;
    if constexpr (F8 != 0) alpha *= (P.sa ? *P.sa : 1.0f) * (P.sb ? *P.sb : 1.0f);  // dequantise: per-tensor scales of the two operands
#pragma unroll
    for (int i = 0; i < AI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = acc[i][j][r] * alpha + bj[j];
    E.alpha = 1.0f;
    E.bias = nullptr;  // (a folded-LayerNorm launch has no E.bias: its bias' is ln_bias, added per row group in the store loop)
  }
  const bool ln_fold = E.ln_stats != nullptr;
  // each wave restages an RP x WN fp32 block of its accumulators through its own LDS region, then the whole block streams
  // them out: every thread owns 8 consecutive columns of a row (16-B vectors).
  constexpr int RP = WM % 64 == 0 ? 64 : 32;  // rows per pass (WM = 32, 96: 32-row passes)
  static_assert(RP % 32 == 0 && WM % RP == 0, "restage passes are whole 32-row accumulator blocks");
  constexpr int REGION = RP * WN;        // floats per wave region
  constexpr int CPR = WN / 8;            // 8-column chunks per region row
  constexpr int NGRP = NWAVES * RP * CPR;                 // 8-column groups per pass (whole tile)
  constexpr int NIT = (NGRP + NTHREADS - 1) / NTHREADS;   // ... per thread
  float* Cw = reinterpret_cast<float*>(smem) + wave * REGION;
  const bool pre = !is_split && epilogue_pre_ok(E) && epilogue_vec_ok(E, 8);
  Q8Ctx q8c{1.0f, 448.0f, 0.0f};  // fp8 output (fp8 GEMMs, non-PLAIN launches only): scale from the tensor's delayed amax, running max |x|
  constexpr bool Q8EPI = F8 != 0 && !PLAIN && WM <= 64;  // (the 256 x 256 kernels have no registers for it: the planner keeps fp8-C launches off them)
  if constexpr (Q8EPI) {
    if (E.c_q8) q8c = q8_begin(Q8Out{(uint8_t*)E.C, E.ldc, E.q8_state, E.q8_amax, E.c_q8 - 1}, blockIdx.x == 0 && tid == 0);
  }
  // one pass per RP rows of the wave tile (body: gemm_epilogue_pass.inc).  The pass index selects accumulator registers, so it must
  // end up a constant: up to two passes the loop is unrolled by the compiler (the code every kernel had before the 96-row wave
  // tile, same register allocation); with three or four passes `#pragma unroll` gave up, the index stayed a run-time value and the
  // accumulators went to scratch — stored in every K-loop iteration — and a lambda called per pass made the compiler keep the
  // side-load arrays in scratch instead: those configurations get the body textually once per pass with `p` a constant
#define MIC_EPILOGUE_PASS_K(K_) { constexpr int p = K_;
  if constexpr (WM / RP <= 2 && WN <= 64) {
#pragma unroll
    for (int p = 0; p < WM / RP; ++p) {
#include "gemm_epilogue_pass.inc"
    }
  } else {  // (also the 128-column wave tile of gemm_w4.hip: 16 groups per thread and pass, the loop form is not unrolled either)
    static_assert(WM / RP <= 4, "at most four epilogue passes");
    MIC_EPILOGUE_PASS_K(0)
#include "gemm_epilogue_pass.inc"
    }
    if constexpr (WM / RP > 1) MIC_EPILOGUE_PASS_K(1)
#include "gemm_epilogue_pass.inc"
    }
    if constexpr (WM / RP > 2) MIC_EPILOGUE_PASS_K(2)
#include "gemm_epilogue_pass.inc"
    }
    if constexpr (WM / RP > 3) MIC_EPILOGUE_PASS_K(3)
#include "gemm_epilogue_pass.inc"
    }
  }
#undef MIC_EPILOGUE_PASS_K
  if constexpr (Q8EPI) {
    if (E.c_q8) q8_end_wave(Q8Out{(uint8_t*)E.C, E.ldc, E.q8_state, E.q8_amax, E.c_q8 - 1}, q8c, blockIdx.x * (NWAVES * KG) + wave + kg * NWAVES);
  }
}
