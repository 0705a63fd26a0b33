// gemm.hip — MFMA GEMMs with fused epilogues (mic_gemm / mic_gemm_grouped): argument checks, launch-table construction and the
// choice of tile configuration.  The bf16 / fp8 kernel template itself is gemm_kernel.h, instantiated in gemm_t256.hip /
// gemm_t128.hip / gemm_t64.hip (one translation unit per tile configuration, so they compile in parallel); the fp32 kernel
// is below.
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
    MIC_CHECK(a->c_dtype == MIC_BF16 || a->c_dtype == MIC_F32 || a->c_dtype == MIC_FP8, "mic_gemm(fp8): C is bf16, f32 or (fused emission) fp8");
    if (a->c_dtype == MIC_FP8) {
      // the epilogue quantises its result under the output tensor's delayed scale: the vector store path of the non-PLAIN epilogue
      MIC_CHECK(a->c_q8_state && (a->c_q8_fmt == MIC_E4M3 || a->c_q8_fmt == MIC_E5M2), "mic_gemm(fp8 C): c_q8_state / c_q8_fmt");
      MIC_CHECK((a->act || a->dact || a->Zout) && !a->accumulate && a->split_k <= 1 && !a->a_kmajor, "mic_gemm(fp8 C): an activation / dact epilogue of an NT launch, no accumulate / split-K");
      MIC_CHECK(a->N % 8 == 0 && a->ldc % 8 == 0 && ((uintptr_t)a->C & 7) == 0 && (!a->Zout || a->ldz % 8 == 0) && (!a->Zin || a->ldz % 8 == 0) && (!a->R || a->ldr % 8 == 0),
                "mic_gemm(fp8 C): N, ldc, ldz, ldr multiples of 8");
    }
    MIC_CHECK(a->a_kmajor == a->b_kmajor, "mic_gemm(fp8): both operands k-contiguous (NT) or both k-major (TN, the weight-gradient GEMM)");
    if (a->a_kmajor) MIC_CHECK(a->M % 16 == 0 && a->N % 16 == 0, "mic_gemm(fp8, k-major): M and N must be multiples of 16");
    MIC_CHECK((a->a_fmt == MIC_E4M3 || a->a_fmt == MIC_E5M2) && a->b_fmt == MIC_E4M3, "mic_gemm(fp8): A is e4m3 or e5m2, B is e4m3");
    MIC_CHECK(a->K % 128 == 0, "mic_gemm(fp8): K=%d must be a multiple of 128 (zero-pad the reduction dim)", a->K);
    MIC_CHECK(a->lda % 16 == 0 && a->ldb % 16 == 0, "mic_gemm(fp8): lda/ldb must be multiples of 16 bytes");
    MIC_CHECK(((uintptr_t)a->A & 15) == 0 && ((uintptr_t)a->B & 15) == 0, "mic_gemm(fp8): A/B must be 16-B aligned");
    MIC_CHECK(!a->a_rowsum, "mic_gemm(fp8): no row sums on the fp8 path (mic_colsum_q8_grouped)");
    if (a->split_k > 1)
      MIC_CHECK(a->split_stride > 0 && a->c_dtype == MIC_F32 && !a->a_kmajor && !a->bias && !a->act && !a->dact && !a->Zout && !a->R && !a->accumulate && a->dropout_p == 0.f,
                "mic_gemm(fp8): split_k writes raw fp32 partial sums into per-split slabs (split_stride > 0, NT, no other epilogue)");
  } else
  MIC_CHECK(a->c_dtype == a->dtype || a->c_dtype == MIC_F32, "mic_gemm: c_dtype must be dtype or f32");
  MIC_CHECK(!(a->dact && !a->Zin), "mic_gemm: dact needs Zin");
  MIC_CHECK(a->dropout_p >= 0.f && a->dropout_p < 1.f, "mic_gemm: dropout_p out of range");
  MIC_CHECK(a->c_dtype != MIC_FP8 || a->dtype == MIC_FP8, "mic_gemm: an fp8 C belongs to the fp8 GEMMs");
  e.C = a->C; e.ldc = a->ldc; e.c_f32 = (a->c_dtype == MIC_F32);
  e.c_q8 = a->c_dtype == MIC_FP8 ? 1 + a->c_q8_fmt : 0; e.q8_state = a->c_q8_state; e.q8_amax = a->c_q8_amax;
  e.bias = a->bias; e.act = a->act; e.Zout = a->Zout; e.ldz = a->ldz; e.Zin = a->Zin; e.dact = a->dact;
  e.R = a->R; e.ldr = a->ldr; e.accumulate = a->accumulate;
  e.drop_thr = a->dropout_p > 0.f ? (uint32_t)fminf(a->dropout_p * 4294967296.0f, 4294967295.0f) : 0u;
  e.drop_seed = a->dropout_seed; e.drop_scale = 1.0f / (1.0f - a->dropout_p);
  e.alpha = a->alpha == 0.f ? 1.0f : a->alpha; e.N = a->N;
  e.rowstat = a->rowstat; e.stat_ld = a->rowstat_ld; e.stat_nvalid = a->rowstat_nvalid > 0 ? a->rowstat_nvalid : a->N;
  if (a->rowstat) {
    MIC_CHECK((a->dtype == MIC_BF16 || (a->dtype == MIC_FP8 && !a->a_kmajor)) && a->c_dtype == MIC_BF16 && a->split_k <= 1 && !a->act && !a->dact && !a->R &&
                  !a->accumulate && a->dropout_p == 0.f && !a->Zout,
              "mic_gemm: rowstat goes with the bare bias epilogue of a bf16 / fp8 NT GEMM with a bf16 C (the LM head)");
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

bool table_is_plain(const LaunchTable& t) {
  for (int i = 0; i < t.count; ++i) {
    const EpiArgs& e = t.p[i].epi;
    if (t.p[i].nsplit > 1 || e.act || e.Zout || e.dact || e.accumulate || (e.ldc & 7)) return false;
    if (((uintptr_t)e.C & 15) != 0) return false;
    if (e.R && ((e.ldr & 7) || ((uintptr_t)e.R & 15) || e.c_f32)) return false;
  }
  return true;
}
static bool any_rowsum_early(const mic_gemm_args* args, int count) {
  for (int i = 0; i < count; ++i)
    if (args[i].a_rowsum) return true;
  return false;
}
// ---- the CU budget of the tile planner.  Every threshold below that used to say "256" means "the CUs this process's GEMM blocks
// can expect to get": a launch is "one round" when its blocks fit the budget at the configuration's residency.  Default = the
// device (256 on MI355X); a data-parallel job whose collectives occupy CUs (RCCL's channels are persistent blocks, one CU each)
// lowers it — `mic_set_cu_budget`, or MIC_FREE_CUS in the environment — so that a launch sized for 256 free CUs re-plans
// (fewer K-groups, i.e. more blocks per CU) instead of spilling a few blocks into a second round.
static thread_local int g_cu_budget = 0;  // per calling thread: two host threads driving two streams plan independently
int mic_cu_budget_now() {
  static const int env = [] { const char* e = getenv("MIC_FREE_CUS"); return e ? atoi(e) : 0; }();
  int c = g_cu_budget > 0 ? g_cu_budget : (env > 0 ? env : 256);
  c = c < 8 ? 8 : (c > 256 ? 256 : c);
  return c & ~7;  // whole CUs per XCD: persistent grids are multiples of 8
}
extern "C" int mic_set_cu_budget(int cus) {
  MIC_CHECK(cus == 0 || (cus >= 8 && cus <= 1024), "mic_set_cu_budget: %d (0 = default, else 8..1024)", cus);
  g_cu_budget = cus;
  return MIC_OK;
}
extern "C" int mic_get_cu_budget(void) { return mic_cu_budget_now(); }

struct GemmPlan { int bm, bm_m, kgroups, blocks, grid, per_cu, phased; };  // bm_m: tile rows (bm, or 192 with bm = 128)

// tile configuration of one (grouped) bf16 / fp8 launch: pure host arithmetic on the shapes and the CU budget
static GemmPlan plan_bf16(const mic_gemm_args* args, int count) {
  const int cus = mic_cu_budget_now();
  GemmPlan pl{};
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
  int bm = tiles_big >= (200L * cus) / 256 ? 256 : (tiles_small < ((long)tiny_below * cus) / 256 ? 64 : 128);
  // ... and a 256x256 launch that fills the CUs 1.1 times costs two rounds.  Under a reduced CU budget (the defaults for all 256
  // CUs were tuned by measurement and stay as they are; MIC_GEMM_QUANT=1 applies the rule there too) compare the longest per-CU
  // tile queue of both configurations: a 128x128 tile costs ~0.31 of a 256x256 one in CU time (a quarter of the work at ~0.8 of
  // the efficiency).
  static const int quant_env = [] { const char* e = getenv("MIC_GEMM_QUANT"); return e ? atoi(e) : 0; }();
  if (bm == 256 && (cus < 256 || quant_env)) {
    const double c256 = (double)((tiles_big + cus - 1) / cus), c128 = 0.3125 * (double)((tiles_small + cus - 1) / cus);
    if (c128 < c256) bm = 128;
  }
  if (force == 256 || force == 128 || force == 64) bm = force;
  if (args[0].dtype == MIC_FP8 && args[0].a_kmajor && bm < 128) bm = 128;  // fp8 k-major images are 128 wide
  for (int i = 0; i < count; ++i)
    if (args[i].c_dtype == MIC_FP8 && bm == 256) bm = 128;  // the fp8-emitting epilogue lives in the 128 / 64 tile kernels
  for (int i = 0; i < count; ++i)
    if (args[i].rowstat) bm = 256;  // softmax partials per 64-column granule = the wave tile width of this configuration
  // 192 x 128 tiles for the single-problem NT / NN launches whose 128 x 128 tiles would need a second round of the 2-per-CU slots
  // while 192-row tiles fit one (packed decoder rows 2049..3072 and the ViT's 3200 rows against N = 3072 / 4096);
  // MIC_GEMM_T192=0 switches the configuration off (A/B)
  static const int t192 = [] { const char* e = getenv("MIC_GEMM_T192"); return e ? atoi(e) : 1; }();
  int bm_m = bm;
  // (fp8: measured — the 192-row fp8 tile needs 133 registers, i.e. one block per CU; capped at 128 it spills 59 and the step loses 10 %)
  if (bm == 128 && t192 && force == 0 && count == 1 && args[0].dtype == MIC_BF16 && !args[0].a_kmajor && args[0].split_k <= 1) {
    const long t128 = (long)((args[0].M + 127) / 128) * ((args[0].N + 127) / 128), t192n = (long)((args[0].M + 191) / 192) * ((args[0].N + 127) / 128);
    if (t128 > 2L * cus && t192n <= 2L * cus) bm_m = 192;
  }
  int blocks = 0, kmin = 1 << 30;
  for (int i = 0; i < count; ++i) {
    int nsplit = args[i].split_k > 1 ? args[i].split_k : 1;
    const int kt = (args[i].dtype == MIC_FP8 ? args[i].K / 2 : args[i].K) / 64;
    if (nsplit > kt) nsplit = kt;
    if (nsplit < 1) nsplit = 1;
    blocks += ((args[i].M + bm_m - 1) / bm_m) * ((args[i].N + bm - 1) / bm) * nsplit;
    kmin = kt / nsplit < kmin ? kt / nsplit : kmin;
  }
  pl.bm = bm; pl.bm_m = bm_m; pl.blocks = pl.grid = blocks; pl.kgroups = 1; pl.per_cu = 1;
  // LDS-DMA four-phase 256x256 kernel for the single-problem NT launches (both operands k-contiguous: LM-head forward, FFN-in
  // forward), where its deeper operand prefetch wins; MIC_GEMM_PHASED=0 switches it off (A/B).  On every 256x256 launch it
  // measured +1.6 ms per train step: that mode is gone.
  static const int phased_env = [] { const char* e = getenv("MIC_GEMM_PHASED"); return e ? atoi(e) : 2; }();
  const bool f8 = args[0].dtype == MIC_FP8;
  pl.phased = bm == 256 && !f8 && phased_env != 0 && !args[0].a_kmajor && !args[0].b_kmajor && count == 1 && !any_rowsum_early(args, count);
  // ... and of those the four-wave 128x128-wave-tile kernel (gemm_w4.hip) takes, BY DEFAULT, the launches whose shape allows it and
  // whose epilogue — decided at launch, gemm_w4_takes — is a bare one: bf16 C with bias / folded LayerNorm / softmax partials (LM
  // head, all-layer cross k/v projection) or fp32 C, also as split-K slabs (the LM head's backward GEMMs on k-contiguous copies);
  // plan.phased reads 2 for such a shape.  MIC_GEMM_W4=0 keeps them on the four-phase / register-staged kernels (A/B)
  static const int w4_env = [] { const char* e = getenv("MIC_GEMM_W4"); return e ? atoi(e) : 1; }();
  if (pl.phased && w4_env && args[0].K >= 256 && args[0].K % 128 == 0) pl.phased = 2;  // (split-K: fp32 slabs, gemm_w4_takes)
  if (bm == 256) {
    pl.per_cu = 1;  // 128 KiB of LDS: a block holds its CU alone; PLAIN launches with more tiles than CUs run as `cus` persistent blocks
  } else if (bm == 128) {
    // 8 waves (measured better than the 4-wave 64x64 wave tile at every tile count), two blocks per CU; two K-groups (16 waves, one
    // block per CU) when the launch is a single round of at most one block per CU
    static const int kg128 = [] { const char* e = getenv("MIC_GEMM_KG128"); return e ? atoi(e) : -1; }();
    const bool two = bm_m == 128 && (kg128 >= 0 ? kg128 == 2 : (blocks <= cus && kmin >= 8));
    pl.kgroups = two ? 2 : 1;
    pl.per_cu = two ? 1 : 2;
  } else {  // 64x64x64 tiles, 4 waves per K-group (32 KiB of LDS each); K-groups while the grid leaves CUs under-occupied
    static const int kg_force = [] { const char* e = getenv("MIC_GEMM_KG"); return e ? atoi(e) : 0; }();
    int kgs = blocks <= cus && kmin >= 16 ? 4 : (blocks <= 2 * cus && kmin >= 8 ? 2 : 1);
    if (kg_force == 1 || kg_force == 2 || kg_force == 4) kgs = kg_force;
    pl.kgroups = kgs;
    pl.per_cu = 4 / kgs;
  }
  return pl;
}

extern "C" int mic_gemm_plan(const mic_gemm_args* args, int count, mic_gemm_plan_info* out) {
  MIC_CHECK(args && out && count >= 1 && count <= MAX_PROBLEMS, "mic_gemm_plan: bad args (1..%d problems)", MAX_PROBLEMS);
  MIC_CHECK(args[0].dtype == MIC_BF16 || args[0].dtype == MIC_FP8, "mic_gemm_plan: the planner belongs to the bf16 / fp8 kernels");
  const GemmPlan pl = plan_bf16(args, count);
  out->tile = pl.bm; out->tile_m = pl.bm_m; out->kgroups = pl.kgroups; out->blocks = pl.blocks; out->blocks_per_cu = pl.per_cu; out->phased = pl.phased;
  out->cu_budget = mic_cu_budget_now();
  // persistent grid: the PLAIN 256x256 instantiations of the register-staged kernel only (launch_cfg_p) — decided from the
  // epilogue these args describe, and MIC_GEMM_PERSIST
  static const int persist_env = [] { const char* e = getenv("MIC_GEMM_PERSIST"); return e ? atoi(e) : 1; }();
  bool plain = true;
  for (int i = 0; i < count; ++i) {
    const mic_gemm_args& a = args[i];
    if (a.split_k > 1 || a.act || a.Zout || a.dact || a.accumulate || (a.ldc & 7) || ((uintptr_t)a.C & 15)) plain = false;
    if (a.R && ((a.ldr & 7) || ((uintptr_t)a.R & 15) || a.c_dtype == MIC_F32)) plain = false;
  }
  const bool persist = persist_env && plain && pl.bm == 256 && !pl.phased && pl.kgroups == 1 && pl.blocks > out->cu_budget;
  out->grid = persist ? out->cu_budget : pl.blocks;
  // launches with softmax partials (the LM-head forward) run on gemm_d2.hip by default: 256 x 128 tiles, two blocks per CU
  static const int d2_plan_env = [] { const char* e = getenv("MIC_GEMM_D2"); return e ? atoi(e) : 3; }();
  if (d2_plan_env && pl.phased == 2 && count == 1 && args[0].rowstat && args[0].N % 128 == 0) {
    out->tile = 128; out->tile_m = 256; out->blocks_per_cu = 2;
    out->blocks = out->grid = ((args[0].M + 255) / 256) * ((args[0].N + 127) / 128);
  }
  return MIC_OK;
}

static int launch_bf16(const mic_gemm_args* args, int count, hipStream_t s) {
  LaunchTable tab;
  tab.count = count;
  const int f8 = args[0].dtype == MIC_FP8 ? (args[0].a_fmt == MIC_E5M2 ? 2 : 1) : 0;
  const GemmPlan pl = plan_bf16(args, count);
  const int bm = pl.bm, bm_m = pl.bm_m;
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
    if (f8) {  // the kernel counts K and the leading dimensions of k-contiguous operands in 2-byte units (k-major: ld in bytes)
      if (!args[i].a_kmajor) { p.lda /= 2; p.ldb /= 2; }
      p.K /= 2;
      p.sa = args[i].a_scale_inv; p.sb = args[i].b_scale_inv;
    }
    p.tiles_m = (p.M + bm_m - 1) / bm_m; p.tiles_n = (p.N + bm - 1) / bm;
    p.nsplit = args[i].split_k > 1 ? args[i].split_k : 1;
    if (p.nsplit > p.K / 64) p.nsplit = p.K / 64;
    p.split_stride = p.nsplit > 1 ? args[i].split_stride : 0;
    MIC_CHECK(args[i].split_stride >= 0 && (args[i].split_stride == 0 || args[i].split_stride >= (long long)(p.M - 1) * args[i].ldc + p.N),
              "mic_gemm: split_stride must cover one M x N slab");
    MIC_CHECK(p.split_stride == 0 || p.nsplit == args[i].split_k, "mic_gemm: split_k exceeds K/64 with a slab workspace");
    MIC_CHECK(!args[i].a_rowsum || args[i].a_kmajor, "mic_gemm: a_rowsum needs a_kmajor (A = dy^T of the weight-gradient GEMM)");
    p.a_rowsum = args[i].a_rowsum;
    p.rowsum_k = args[i].rowsum_k > 0 ? args[i].rowsum_k : p.K;
    MIC_CHECK(args[i].k_valid >= 0 && (args[i].k_valid == 0 || (args[i].a_kmajor && args[i].b_kmajor)),
              "mic_gemm: k_valid is a feature of the k-major x k-major (weight-gradient) launches");
    p.k_valid = args[i].k_valid > 0 ? args[i].k_valid : 0x7fffffff;
    p.block_begin = blocks;
    blocks += p.tiles_m * p.tiles_n * p.nsplit;
  }
  tab.total_blocks = blocks;
  for (int i = 0; i < count; ++i)
    MIC_CHECK(!args[i].rowsum2 || table_is_plain(tab), "mic_gemm_grouped: rowsum2 needs every problem of the launch on the bare / residual epilogue");
  // The two-blocks-per-CU 256 x 128 kernel (gemm_d2.hip) takes the four-wave kernel's launches that carry softmax partials (the
  // LM-head forward: K = 1024 tiles whose epilogue is a third of their time — under a second block's MFMAs it is cover).  Its K loop
  // is slower than the four-wave kernel's (64-B DMA segments: bound by L2 requests), so the deep-K and bare launches stay there.
  // MIC_GEMM_D2 (A/B): 0 = off; 1 = everything the four-wave kernel would take; 2 = every single-problem NT 256-tile launch its
  // epilogue covers; 3 (default) = the launches with softmax partials
  static const int d2_env = [] { const char* e = getenv("MIC_GEMM_D2"); return e ? atoi(e) : 3; }();
  const bool nt1 = bm == 256 && !f8 && !args[0].a_kmajor && !args[0].b_kmajor && count == 1;
  if (d2_env && nt1 && gemm_d2_takes(tab) && (d2_env == 2 || (pl.phased == 2 && gemm_w4_takes(tab))) && (d2_env != 3 || tab.p[0].epi.rowstat)) {
    Problem& p = tab.p[0];
    p.tiles_n = (p.N + 127) / 128;  // 256 x 128 tiles
    tab.total_blocks = p.tiles_m * p.tiles_n * p.nsplit;
    launch_gemm_d2(tab, s);
  } else
  if (bm == 256 && pl.phased == 2 && gemm_w4_takes(tab)) launch_gemm_w4(tab, s);  // 4 waves x 128x128, bare epilogues (default on)
  else if (bm == 256 && pl.phased && tab.p[0].nsplit == 1) launch_gemm_phased(tab, args[0].a_kmajor, args[0].b_kmajor, table_is_plain(tab), s);  // LDS-DMA, phased
  else if (bm == 256) launch_gemm_t256(tab, args[0].a_kmajor, args[0].b_kmajor, s, f8);     // 256x256x64, 8 waves
  else if (bm == 128 && bm_m == 192) launch_gemm_t192(tab, args[0].b_kmajor, s);                     // 192x128x64, 8 waves, two blocks per CU
  else if (bm == 128) launch_gemm_t128(tab, args[0].a_kmajor, args[0].b_kmajor, s, f8, pl.kgroups);  // 128x128x64, 8 waves per K-group
  else launch_gemm_t64(tab, args[0].a_kmajor, args[0].b_kmajor, s, f8, pl.kgroups);                  // 64x64x64, 4 waves per K-group
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
  MIC_CHECK(a->k_valid == 0, "mic_gemm(f32): k_valid is a bf16-path feature");
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
