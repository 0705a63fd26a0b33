// gemm.hip — MFMA GEMMs with fused epilogues (mic_gemm / mic_gemm_grouped).
//
// bf16 kernel: ONE template, table-driven (a launch carries up to 8 problems), three tile configurations chosen per launch:
//   256x256 tile, 8 waves (2x4), wave tile 128x64 (4x2 v_mfma_f32_32x32x16_bf16 accumulators), 128 KiB LDS, 1 block/CU
//                 — launches with >= 200 such tiles (LM head, grouped dW, FFN): 2x the FLOPs per staged byte;
//   128x128 tile, 8 waves (2x4), wave tile 64x32, 64 KiB LDS, 2 blocks/CU — the default;
//   64x64 tile,   4 waves (2x2), wave tile 32x32, 32 KiB LDS — launches with fewer than 128 tiles of 128x128 (decode-time
//                 GEMMs on ~1k rows, ViT dW);
//   plus K-groups (template KG): 2 (128x128) or 2/4 (64x64) groups of waves share one output tile and split its K range
//   when the grid leaves CUs under-occupied (the K loop is then a latency chain), partial sums meet in LDS.
// BK = 64, two LDS stages, one barrier per K-tile.  Operands travel HBM/L2 -> registers (global_load_dwordx4, issued a full
// iteration ahead) -> swizzled ds_write_b128 -> LDS; LDS-DMA (global_load_lds) was tried three times and lost every time
// (~100 cycles of issue per 1 KiB piece in the issuing wave's in-order stream).  LDS images:
//   k-contiguous operand -> [128|64 rows][64 k] (128-B rows), 16-B chunk c of row r at c ^ ((r >> 1) & 7): conflict-free for
//                           ds_read_b128's lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31}; fragments by ds_read_b128;
//   k-major operand      -> [64 k][128 x] (256-B rows), chunk c of row k at c ^ ((k & 3) << 2), or [64 k][64 x] (128-B rows),
//                           c ^ (((k >> 1) & 1) << 2); fragments by 2 x ds_read_b64_tr_b16 (hardware transpose; lane
//                           layouts verified on gfx950 by tools/probe_layouts.hip).
// So Linear forward (NT), dX (NN) and dW (TN) all run from the tensors as they lie in HBM — no transposed copies.
// Block ids are remapped XCD-aware (block b runs on XCD b % 8): each XCD walks a contiguous run of tiles ordered in
// GROUP_M-tall column panels so neighbouring tiles share A/B panels in that XCD's L2; split-K launches instead give each XCD
// its own K-chunks over all tiles.  Split-K results go to fp32 atomics or, better, to one fp32 slab per split (mic_sum_slabs).
// Epilogue: accumulators are restaged through LDS (free after the main loop) so every thread owns 8 consecutive columns
// of a row: bias / activation / dropout / residual / Z / C all move as 16-B coalesced vectors; side operands (Zin, residual,
// old C) are prefetched for all of a thread's groups before the restage barrier.  Optional by-product: row sums of A (the
// bias gradient when A = dy^T) from the A fragments the waves already hold.
//
// Measured on MI355X: 256^2 tile = ~20 us fixed + 1.7-2.0 us per K-tile (0.95-1.06 PF/s at large K; the loop
// is bound by operand delivery, ~17-20 B/clk/CU through TA/L2, not by MFMA issue), one-round 128^2 launch = 5.2k cycles
// prologue + 2.25k per K-tile pair + 5.4k epilogue (28 B/clk/CU).  MFMA pipe ~45 % busy at the ~1.8 GHz the chip sustains.
//
// f32 kernel (the reference's default dtype; parity mode): 64x64x16 tiles, v_mfma_f32_32x32x2_f32 (bit-exact fp32
// fma chain), generic strides.
#include "gemm_common.h"

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
  static_assert(F8 == 0 || (!AK && !BKM && BKT == 64), "fp8: k-contiguous operands, 128-byte stages");
  constexpr int BM = 2 * WM, BN = WN * WNW, NWAVES = 2 * WNW, NTHREADS = 64 * NWAVES * KG;
  constexpr int UA = BM < 128 ? BM : 128, UB = BN < 128 ? BN : 128;  // rows per staged image (128, or 64 for the 64-wide tiles)
  constexpr int HALF_A = UA * BKT * 2, HALF_B = UB * BKT * 2;
  constexpr int NHA = BM / UA, NHB = BN / UB, STAGE = NHA * HALF_A + NHB * HALF_B, AI = WM / 32, NJ = WN / 32, KSTEPS = BKT / 16;
  constexpr int STAGE_AT = 0;  // k-step in front of which the next tile's LDS writes / global loads are issued (1..3 measured equal)
  using SA = HalfStager<AK, NWAVES, BKT, UA>;
  using SB = HalfStager<BKM, NWAVES, BKT, UB>;
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
  float bsum[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) bsum[i] = 0.0f;
  const bool do_rowsum = P.a_rowsum != nullptr;
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
  auto write_lds = [&](char* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < NHA; ++h) SA::store(ra[h], buf + h * HALF_A, wave, lane);
#pragma unroll
    for (int h = 0; h < NHB; ++h) SB::store(rb[h], buf + NHA * HALF_A + h * HALF_B, wave, lane);
  };
  // this wave's operand sub-images
  const int a_half = (wr * WM) / UA, a_off = (wr * WM) % UA;
  const int b_half = (wc * WN) / UB, b_off = (wc * WN) % UB;

  if (nk > 0) {
    load_regs(0);
    write_lds(gsm);
    if (nk > 1) load_regs(1);
  }
  if (nk_loop > 0) __syncthreads();
  for (int t = 0; t < nk_loop; ++t) {
    if (t < nk) {
      const bool rs_tile = do_rowsum && ((kt0 + t) % rs_tiles_n) == tn;  // this block's share of the A row sums
      const char* cur = gsm + (t & 1) * STAGE;
      const char* At = cur + a_half * HALF_A;
      const char* Bt = cur + NHA * HALF_A + b_half * HALF_B;
      if constexpr (F8 != 0) {
#pragma unroll
        for (int mm = 0; mm < 2; ++mm) {  // two K = 64 MFMAs per 128-byte stage
          if (mm == 0 && t + 1 < nk) {
            write_lds(gsm + ((t + 1) & 1) * STAGE);
            if (t + 2 < nk) load_regs(t + 2);
          }
          i32x8 a8[AI], b8[NJ];
#pragma unroll
          for (int i = 0; i < AI; ++i) a8[i] = read_frag8(At, a_off + i * 32, mm, lane);
#pragma unroll
          for (int j = 0; j < NJ; ++j) b8[j] = read_frag8(Bt, b_off + j * 32, mm, lane);
#pragma unroll
          for (int i = 0; i < AI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[i], b8[j], acc[i][j], F8 == 2 ? 1 : 0, 0, 0, 0x7f7f7f7f, 0,
                                                                          0x7f7f7f7f);  // cbsz/blgp: 0 = e4m3, 1 = e5m2; scales 2^0
        }
      } else {
#pragma unroll
      for (int kk = 0; kk < KSTEPS; ++kk) {
        if (kk == STAGE_AT && t + 1 < nk) {
          write_lds(gsm + ((t + 1) & 1) * STAGE);  // its buffer was last read in iteration t-1 (barrier below)
          if (t + 2 < nk) load_regs(t + 2);        // a full iteration of MFMAs to land
        }
        bf16x8 af[AI], bfr[NJ];
#pragma unroll
        for (int i = 0; i < AI; ++i) af[i] = read_frag<AK, BKT, UA>(At, a_off + i * 32, kk, lane);
#pragma unroll
        for (int j = 0; j < NJ; ++j) bfr[j] = read_frag<BKM, BKT, UB>(Bt, b_off + j * 32, kk, lane);
        if (rs_tile && (kk % WNW) == wc) {
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

  if (do_rowsum) {
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

// ------------------------------------------------------------------------------------------------ f32
// A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn].  64x64 tile, BK = 16, 4 waves each a 32x32 block.
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, long sam, long sak,
                                                       const float* __restrict__ B, long sbk, long sbn, int M, int N,
                                                       int K, EpiArgs epi) {
  __shared__ float As[64][17];
  __shared__ float Bs[16][65];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int wr = wave >> 1, wc = wave & 1;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  for (int k0 = 0; k0 < K; k0 += 16) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * 256;  // 1024 elements per tile
      {
        // A tile: pick the fast-varying index along the contiguous dimension
        int mm, kk;
        if (sak == 1) { mm = e >> 4; kk = e & 15; } else { kk = e >> 6; mm = e & 63; }
        const int gm = m0 + mm, gk = k0 + kk;
        As[mm][kk] = (gm < M && gk < K) ? A[gm * sam + gk * sak] : 0.0f;
      }
      {
        int nn, kk;
        if (sbk == 1) { nn = e >> 4; kk = e & 15; } else { kk = e >> 6; nn = e & 63; }
        const int gn = n0 + nn, gk = k0 + kk;
        Bs[kk][nn] = (gn < N && gk < K) ? B[gk * sbk + gn * sbn] : 0.0f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
      const float a = As[wr * 32 + (lane & 31)][k + (lane >> 5)];
      const float b = Bs[k + (lane >> 5)][wc * 32 + (lane & 31)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  const int n = n0 + wc * 32 + (lane & 31);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (m < M && n < N) epilogue_store<float>(epi, m, n, acc[r]);
  }
}

static int fill_epi(const mic_gemm_args* a, EpiArgs& e) {
  MIC_CHECK(a && a->A && a->B && a->C, "mic_gemm: null pointer");
  MIC_CHECK(a->M > 0 && a->N > 0 && a->K > 0, "mic_gemm: bad shape M=%d N=%d K=%d", a->M, a->N, a->K);
  MIC_CHECK(a->dtype == MIC_BF16 || a->dtype == MIC_F32 || a->dtype == MIC_FP8, "mic_gemm: bad dtype %d", a->dtype);
  if (a->dtype == MIC_FP8) {
    MIC_CHECK(a->c_dtype == MIC_BF16 || a->c_dtype == MIC_F32, "mic_gemm(fp8): C is bf16 or f32");
    MIC_CHECK(!a->a_kmajor && !a->b_kmajor, "mic_gemm(fp8): both operands must be k-contiguous (mic_fp8_quantize writes the transposed copies)");
    MIC_CHECK((a->a_fmt == MIC_E4M3 || a->a_fmt == MIC_E5M2) && a->b_fmt == MIC_E4M3, "mic_gemm(fp8): A is e4m3 or e5m2, B is e4m3");
    MIC_CHECK(a->K % 128 == 0, "mic_gemm(fp8): K=%d must be a multiple of 128 (zero-pad the reduction dim)", a->K);
    MIC_CHECK(a->lda % 16 == 0 && a->ldb % 16 == 0, "mic_gemm(fp8): lda/ldb must be multiples of 16 bytes");
    MIC_CHECK(((uintptr_t)a->A & 15) == 0 && ((uintptr_t)a->B & 15) == 0, "mic_gemm(fp8): A/B must be 16-B aligned");
    MIC_CHECK(a->split_k <= 1 && !a->a_rowsum, "mic_gemm(fp8): no split-K / row sums on the fp8 path");
  } else
  MIC_CHECK(a->c_dtype == a->dtype || a->c_dtype == MIC_F32, "mic_gemm: c_dtype must be dtype or f32");
  MIC_CHECK(!(a->dact && !a->Zin), "mic_gemm: dact needs Zin");
  MIC_CHECK(a->dropout_p >= 0.f && a->dropout_p < 1.f, "mic_gemm: dropout_p out of range");
  e.C = a->C; e.ldc = a->ldc; e.c_f32 = (a->c_dtype == MIC_F32);
  e.bias = a->bias; e.act = a->act; e.Zout = a->Zout; e.ldz = a->ldz; e.Zin = a->Zin; e.dact = a->dact;
  e.R = a->R; e.ldr = a->ldr; e.accumulate = a->accumulate;
  e.drop_thr = a->dropout_p > 0.f ? (uint32_t)fminf(a->dropout_p * 4294967296.0f, 4294967295.0f) : 0u;
  e.drop_seed = a->dropout_seed; e.drop_scale = 1.0f / (1.0f - a->dropout_p);
  e.alpha = a->alpha == 0.f ? 1.0f : a->alpha; e.N = a->N;
  e.rowstat = a->rowstat; e.stat_ld = a->rowstat_ld; e.stat_nvalid = a->rowstat_nvalid > 0 ? a->rowstat_nvalid : a->N;
  if (a->rowstat) {
    MIC_CHECK(a->dtype == MIC_BF16 && a->split_k <= 1 && !a->act && !a->dact && !a->R && !a->accumulate && a->dropout_p == 0.f && !a->Zout,
              "mic_gemm: rowstat goes with the bare bias epilogue of a bf16 GEMM (the LM head)");
    MIC_CHECK(a->N % 64 == 0 && a->ldc % 8 == 0 && ((uintptr_t)a->C & 15) == 0, "mic_gemm: rowstat needs N %% 64 == 0 and a 16-B aligned C");
    MIC_CHECK(a->rowstat_ld >= a->N / 64 && ((uintptr_t)a->rowstat & 7) == 0, "mic_gemm: rowstat needs ld >= N / 64 float2 entries per row");
  }
  e.ln_stats = a->a_ln_stats; e.ln_g = a->a_ln_colsum; e.ln_bias = nullptr; e.ln_inv_d = 0.f; e.ln_eps = a->a_ln_eps;
  e.rowsum2 = a->rowsum2;
  if (a->a_ln_stats) {
    MIC_CHECK(a->dtype == MIC_BF16 && a->a_ln_colsum && a->bias && a->a_ln_width > 0, "mic_gemm: a folded LayerNorm needs bf16 operands, a_ln_colsum, bias' and a_ln_width");
    MIC_CHECK(a->split_k <= 1 && !a->dact && !a->accumulate && (a->alpha == 0.f || a->alpha == 1.f), "mic_gemm: folded LayerNorm: no split-K / dact / accumulate / alpha");
    MIC_CHECK(a->N % 8 == 0 && a->ldc % 8 == 0 && ((uintptr_t)a->C & 15) == 0 && (!a->R || a->ldr % 8 == 0) && (!a->Zout || a->ldz % 8 == 0),
              "mic_gemm: folded LayerNorm needs the vector epilogue (N, ldc, ldr, ldz multiples of 8, 16-B aligned C)");
    e.ln_bias = a->bias; e.bias = nullptr; e.ln_inv_d = 1.0f / (float)a->a_ln_width;
  }
  if (a->rowsum2) {
    MIC_CHECK(a->dtype == MIC_BF16 && a->N % 128 == 0 && a->split_k <= 1 && !a->act && !a->dact && !a->Zout && !a->accumulate,
              "mic_gemm: rowsum2 goes with the bare / residual epilogue of a bf16 GEMM, N %% 128 == 0");
    MIC_CHECK(a->ldc % 8 == 0 && ((uintptr_t)a->C & 15) == 0 && (!a->R || (a->ldr % 8 == 0 && ((uintptr_t)a->R & 15) == 0 && a->c_dtype != MIC_F32)),
              "mic_gemm: rowsum2 needs the PLAIN store path (16-B aligned C / R rows)");
  }
  if (a->dtype == MIC_BF16) {
    MIC_CHECK(a->K % 64 == 0, "mic_gemm(bf16): K=%d must be a multiple of 64 (zero-pad the reduction dim)", a->K);
    MIC_CHECK(a->lda % 8 == 0 && a->ldb % 8 == 0, "mic_gemm(bf16): lda/ldb must be multiples of 8");
    MIC_CHECK(((uintptr_t)a->A & 15) == 0 && ((uintptr_t)a->B & 15) == 0, "mic_gemm(bf16): A/B must be 16-B aligned");
    if (a->a_kmajor) MIC_CHECK(a->M % 8 == 0, "mic_gemm(bf16): k-major A needs M %% 8 == 0");
    if (a->b_kmajor) MIC_CHECK(a->N % 8 == 0, "mic_gemm(bf16): k-major B needs N %% 8 == 0");
    if (a->split_k > 1)
      MIC_CHECK(a->c_dtype == MIC_F32 && !a->bias && !a->act && !a->dact && !a->Zout && !a->R && !a->accumulate && a->dropout_p == 0.f,
                "mic_gemm: split_k accumulates raw fp32 partial sums into a zeroed C (no other epilogue)");
  }
  return MIC_OK;
}

template <int WM, int WN, int WNW, int BKT, int KG, bool PLAIN, int F8 = 0>
static void launch_cfg_p(const LaunchTable& tab, int akm, int bkm, hipStream_t s) {
  constexpr int BM = 2 * WM, BN = WN * WNW;
  size_t lds = (size_t)KG * 2 * (BM + BN) * BKT * 2;
  const size_t epi = (size_t)2 * WNW * (WM < 64 ? WM : 64) * WN * 4;  // the epilogue restages min(WM,64) x WN floats per wave
  const size_t red = (size_t)(KG - 1) * 2 * WNW * (WM / 32) * (WN / 32) * 16 * 64 * 4;  // K-group partial sums
  if (epi > lds) lds = epi;
  if (red > lds) lds = red;
  // 256x256 tiles hold a CU alone (128 KiB LDS): launches with more tiles than CUs run as 256 persistent blocks
  static const int persist = [] { const char* e = getenv("MIC_GEMM_PERSIST"); return e ? atoi(e) : 1; }();
  int nblk = tab.total_blocks;
  if (persist && BM == 256 && PLAIN && KG == 1 && nblk > 256) nblk = 256;
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
    LAUNCH(false, false);
  } else {
    if (!akm && !bkm) LAUNCH(false, false);
    else if (!akm && bkm) LAUNCH(false, true);
    else if (akm && bkm) LAUNCH(true, true);
    else LAUNCH(true, false);
  }
#undef LAUNCH
}

static bool table_is_plain(const LaunchTable& t) {
  for (int i = 0; i < t.count; ++i) {
    const EpiArgs& e = t.p[i].epi;
    if (t.p[i].nsplit > 1 || e.act || e.Zout || e.dact || e.accumulate || (e.ldc & 7)) return false;
    if (((uintptr_t)e.C & 15) != 0) return false;
    if (e.R && ((e.ldr & 7) || ((uintptr_t)e.R & 15) || e.c_f32)) return false;
  }
  return true;
}
template <int WM, int WN, int WNW, int BKT, int KG = 1>
static void launch_cfg(const LaunchTable& tab, int akm, int bkm, hipStream_t s, int f8 = 0) {
  const bool plain = table_is_plain(tab);
  if (f8 == 1) {
    if (plain) launch_cfg_p<WM, WN, WNW, BKT, KG, true, 1>(tab, 0, 0, s);
    else launch_cfg_p<WM, WN, WNW, BKT, KG, false, 1>(tab, 0, 0, s);
  } else if (f8 == 2) {
    if (plain) launch_cfg_p<WM, WN, WNW, BKT, KG, true, 2>(tab, 0, 0, s);
    else launch_cfg_p<WM, WN, WNW, BKT, KG, false, 2>(tab, 0, 0, s);
  } else if (plain) launch_cfg_p<WM, WN, WNW, BKT, KG, true>(tab, akm, bkm, s);
  else launch_cfg_p<WM, WN, WNW, BKT, KG, false>(tab, akm, bkm, s);
}

static int launch_bf16(const mic_gemm_args* args, int count, hipStream_t s) {
  LaunchTable tab;
  tab.count = count;
  const int f8 = args[0].dtype == MIC_FP8 ? (args[0].a_fmt == MIC_E5M2 ? 2 : 1) : 0;
  long tiles_big = 0, tiles_small = 0;
  for (int i = 0; i < count; ++i) {
    const int sp = args[i].split_k > 1 ? args[i].split_k : 1;
    tiles_big += (long)((args[i].M + 255) / 256) * ((args[i].N + 255) / 256) * sp;
    tiles_small += (long)((args[i].M + 127) / 128) * ((args[i].N + 127) / 128) * sp;
  }
  // 256x256 tiles deliver 2x the FLOPs per operand byte but need >= ~0.8 blocks per CU to pay; launches that cannot even
  // give every CU one 128x128 tile (decode-time GEMMs on ~1k rows, the N = 768/1024 projections) run 64x64 tiles, 4 waves,
  // several blocks per CU.  MIC_GEMM_TILE=256|128|64 forces a configuration (benchmarking).
  static const int force = [] { const char* e = getenv("MIC_GEMM_TILE"); return e ? atoi(e) : 0; }();
  static const int tiny_below = [] { const char* e = getenv("MIC_TINY_BELOW"); return e ? atoi(e) : MIC_TINY_BELOW; }();
  int bm = tiles_big >= 200 ? 256 : (tiles_small < tiny_below ? 64 : 128);
  if (force == 256 || force == 128 || force == 64) bm = force;
  for (int i = 0; i < count; ++i)
    if (args[i].rowstat) bm = 256;  // softmax partials per 64-column granule = the wave tile width of this configuration
  int blocks = 0;
  for (int i = 0; i < count; ++i) {
    Problem& p = tab.p[i];
    if (int rc = fill_epi(&args[i], p.epi)) return rc;
    MIC_CHECK(args[i].a_kmajor == args[0].a_kmajor && args[i].b_kmajor == args[0].b_kmajor && args[i].dtype == args[0].dtype &&
                  (args[i].dtype == MIC_BF16 || (args[i].dtype == MIC_FP8 && args[i].a_fmt == args[0].a_fmt)),
              "mic_gemm_grouped: all problems of a group must share dtype (bf16 / fp8 format) and operand layouts");
    p.A = (const uint16_t*)args[i].A; p.B = (const uint16_t*)args[i].B;
    p.lda = args[i].lda; p.ldb = args[i].ldb; p.M = args[i].M; p.N = args[i].N; p.K = args[i].K;
    p.sa = p.sb = nullptr;
    if (f8) {  // the kernel counts K and the leading dimensions in 2-byte units
      p.lda /= 2; p.ldb /= 2; p.K /= 2;
      p.sa = args[i].a_scale_inv; p.sb = args[i].b_scale_inv;
    }
    p.tiles_m = (p.M + bm - 1) / bm; p.tiles_n = (p.N + bm - 1) / bm;
    p.nsplit = args[i].split_k > 1 ? args[i].split_k : 1;
    if (p.nsplit > p.K / 64) p.nsplit = p.K / 64;
    p.split_stride = p.nsplit > 1 ? args[i].split_stride : 0;
    MIC_CHECK(args[i].split_stride >= 0 && (args[i].split_stride == 0 || args[i].split_stride >= (long long)(p.M - 1) * args[i].ldc + p.N),
              "mic_gemm: split_stride must cover one M x N slab");
    MIC_CHECK(p.split_stride == 0 || p.nsplit == args[i].split_k, "mic_gemm: split_k exceeds K/64 with a slab workspace");
    p.a_rowsum = args[i].a_rowsum;
    p.rowsum_k = args[i].rowsum_k > 0 ? args[i].rowsum_k : p.K;
    p.block_begin = blocks;
    blocks += p.tiles_m * p.tiles_n * p.nsplit;
  }
  tab.total_blocks = blocks;
  for (int i = 0; i < count; ++i)
    MIC_CHECK(!args[i].rowsum2 || table_is_plain(tab), "mic_gemm_grouped: rowsum2 needs every problem of the launch on the bare / residual epilogue");
  static const int phased = [] { const char* e = getenv("MIC_GEMM_PHASED"); return e ? atoi(e) : 0; }();  // opt-in: measured slower
  static const int w4 = [] { const char* e = getenv("MIC_GEMM_W4"); return e ? atoi(e) : 0; }();  // opt-in: measured slower
  bool any_rowsum = false;
  for (int i = 0; i < count; ++i) any_rowsum |= args[i].a_rowsum != nullptr;
  bool fits32 = true;  // the v2 kernel addresses its operands through 32-bit buffer offsets
  for (int i = 0; i < count; ++i)
    fits32 &= (size_t)tab.p[i].M * tab.p[i].lda * 2 < 0xFFFFFFFFull && (size_t)tab.p[i].N * tab.p[i].ldb * 2 < 0xFFFFFFFFull;
  if (bm == 256 && f8 == 0 && w4 == 2 && !any_rowsum && !args[0].a_kmajor && !args[0].b_kmajor && fits32)
    launch_gemm_w4v2(tab, table_is_plain(tab), s);
  else if (bm == 256 && f8 == 0 && w4 == 1 && !any_rowsum) launch_gemm_w4(tab, args[0].a_kmajor, args[0].b_kmajor, table_is_plain(tab), s);  // 4 waves, 128x128 wave tiles
  else if (bm == 256 && f8 == 0 && phased) launch_gemm_phased(tab, args[0].a_kmajor, args[0].b_kmajor, table_is_plain(tab), s);  // LDS-DMA, phased
  else if (bm == 256) launch_cfg<128, 64, 4, 64>(tab, args[0].a_kmajor, args[0].b_kmajor, s, f8);     // 256x256x64, 8 waves
  else if (bm == 128) {  // 128x128x64, 8 waves (measured better than the 4-wave 64x64 wave tile at every tile count);
                         // two K-groups (16 waves) when the launch is a single round of at most one block per CU
    static const int kg128 = [] { const char* e = getenv("MIC_GEMM_KG128"); return e ? atoi(e) : -1; }();
    int kmin = 1 << 30;
    for (int i = 0; i < count; ++i) kmin = tab.p[i].K / 64 / tab.p[i].nsplit < kmin ? tab.p[i].K / 64 / tab.p[i].nsplit : kmin;
    const bool two = kg128 >= 0 ? kg128 == 2 : (blocks <= 256 && kmin >= 8);
    if (two) launch_cfg<64, 32, 4, 64, 2>(tab, args[0].a_kmajor, args[0].b_kmajor, s, f8);
    else launch_cfg<64, 32, 4, 64>(tab, args[0].a_kmajor, args[0].b_kmajor, s, f8);
  }
  else {  // 64x64x64 tiles, 4 waves per K-group; K-groups while the grid leaves CUs under-occupied
    static const int kg_force = [] { const char* e = getenv("MIC_GEMM_KG"); return e ? atoi(e) : 0; }();
    int kmin = 1 << 30;
    for (int i = 0; i < count; ++i) kmin = tab.p[i].K / 64 / tab.p[i].nsplit < kmin ? tab.p[i].K / 64 / tab.p[i].nsplit : kmin;
    int kgs = blocks <= 256 && kmin >= 16 ? 4 : (blocks <= 512 && kmin >= 8 ? 2 : 1);
    if (kg_force == 1 || kg_force == 2 || kg_force == 4) kgs = kg_force;
    if (kgs == 4) launch_cfg<32, 32, 2, 64, 4>(tab, args[0].a_kmajor, args[0].b_kmajor, s, f8);
    else if (kgs == 2) launch_cfg<32, 32, 2, 64, 2>(tab, args[0].a_kmajor, args[0].b_kmajor, s, f8);
    else launch_cfg<32, 32, 2, 64>(tab, args[0].a_kmajor, args[0].b_kmajor, s, f8);
  }
  MIC_LAUNCH_CHECK();
  return MIC_OK;
}

extern "C" int mic_gemm(const mic_gemm_args* a, void* stream) {
  MIC_CHECK(a, "mic_gemm: null args");
  hipStream_t s = (hipStream_t)stream;
  if (a->dtype == MIC_BF16 || a->dtype == MIC_FP8) return launch_bf16(a, 1, s);
  EpiArgs e;
  if (int rc = fill_epi(a, e)) return rc;
  MIC_CHECK(a->split_k <= 1, "mic_gemm(f32): split_k is a bf16-path feature");
  MIC_CHECK(!a->a_rowsum, "mic_gemm(f32): a_rowsum is a bf16-path feature (use mic_colsum)");
  dim3 grid((a->N + 63) / 64, (a->M + 63) / 64), block(256);
  const long sam = a->a_kmajor ? 1 : a->lda, sak = a->a_kmajor ? a->lda : 1;
  const long sbk = a->b_kmajor ? a->ldb : 1, sbn = a->b_kmajor ? 1 : a->ldb;
  hipLaunchKernelGGL(gemm_f32_kernel, grid, block, 0, s, (const float*)a->A, sam, sak, (const float*)a->B, sbk, sbn,
                     a->M, a->N, a->K, e);
  MIC_LAUNCH_CHECK();
  return MIC_OK;
}

extern "C" int mic_gemm_grouped(const mic_gemm_args* args, int count, void* stream) {
  MIC_CHECK(args && count >= 1, "mic_gemm_grouped: bad args");
  hipStream_t s = (hipStream_t)stream;
  if (args[0].dtype != MIC_BF16 && args[0].dtype != MIC_FP8) {  // parity mode: plain sequence
    for (int i = 0; i < count; ++i)
      if (int rc = mic_gemm(&args[i], stream)) return rc;
    return MIC_OK;
  }
  for (int i = 0; i < count; i += MAX_PROBLEMS) {
    const int n = count - i < MAX_PROBLEMS ? count - i : MAX_PROBLEMS;
    if (int rc = launch_bf16(args + i, n, s)) return rc;
  }
  return MIC_OK;
}
