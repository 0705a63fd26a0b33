// gemm.hip — MFMA GEMMs with fused epilogues (mic_gemm).
//
// bf16 kernel: 128x128 output tile, BK = 64, 256 threads = 4 waves (2x2), each wave a 64x64 sub-tile as 2x2
// v_mfma_f32_32x32x16_bf16 accumulators (64 accumulator registers).  Operand tiles go HBM -> LDS by direct DMA
// (global_load_lds_dwordx4: destination = wave-uniform base + lane*16), double-buffered (2 x 32 KiB), one barrier per
// K-tile.  LDS images are lane-linear, so the bank-conflict swizzle is applied to the per-lane SOURCE address and the
// matching XOR on the read side:
//   k-contiguous operand  -> image [128 rows][64 k]  (128-B rows), 16-B chunk c of row r stored at c ^ (r & 7),
//                            fragments by ds_read_b128 (8 consecutive k of row lane&31);
//   k-major operand       -> image [64 k][128 x]     (256-B rows), chunk c of row k stored at c ^ ((k & 3) << 2),
//                            fragments by 2 x ds_read_b64_tr_b16 (hardware transpose; lane layouts verified on
//                            gfx950 by tools/probe_layouts.hip).
// So Linear forward (NT), dX (NN) and dW (TN) all run from the tensors as they lie in HBM — no transposed copies.
// Block ids are remapped XCD-aware (block b runs on XCD b % 8): each XCD walks a contiguous run of tiles ordered in
// GROUP_M-tall column panels so neighbouring tiles share A/B panels in that XCD's L2.
//
// f32 kernel (the reference's default dtype; parity mode): 64x64x16 tiles, v_mfma_f32_32x32x2_f32 (bit-exact fp32
// fma chain), generic strides.
#include "common.h"

#define BM 128
#define BN 128
#define BK 64
#define TILE_BYTES (128 * 64 * 2)  // 16 KiB per operand tile

// --- DMA one 16 KiB operand tile into LDS.  KMAJOR=false: src is [rows][ld] k-contiguous, tile = 128 rows x 64 k.
//     KMAJOR=true: src is [K][ld] x-contiguous, tile = 64 k x 128 x.  `lim` = number of valid rows (resp. x) in src;
//     out-of-range rows/chunks are redirected to a valid address (their products are masked at the store).
template <bool KMAJOR>
__device__ __forceinline__ void stage_tile(const uint16_t* __restrict__ src, int ld, int x0, int k0, int lim,
                                           char* lds_tile, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = wave * 4 + i;  // 16 wave-instructions of 1 KiB cover the tile
    const uint16_t* g;
    if (!KMAJOR) {
      const int row = q * 8 + (lane >> 3);
      const int c = (lane & 7) ^ (row & 7);
      int gr = x0 + row;
      gr = gr < lim ? gr : lim - 1;
      g = src + (size_t)gr * ld + k0 + c * 8;
    } else {
      const int k = q * 4 + (lane >> 4);
      const int c = (lane & 15) ^ ((k & 3) << 2);
      int gx = x0 + c * 8;
      gx = gx < lim ? gx : 0;
      g = src + (size_t)(k0 + k) * ld + gx;
    }
    __builtin_amdgcn_global_load_lds(GLB_PTR(g), (__attribute__((address_space(3))) void*)(lds_tile + q * 1024), 16, 0, 0);
  }
}

// --- read one 32(x) x 16(k) MFMA operand fragment: 8 consecutive k (kk*16 + 8*(lane>>5) ..) of x = xb + (lane&31)
template <bool KMAJOR>
__device__ __forceinline__ bf16x8 read_frag(const char* lds_tile, int xb, int kk, int lane) {
  if (!KMAJOR) {
    const int row = xb + (lane & 31);
    const int kc = kk * 2 + (lane >> 5);
    return *reinterpret_cast<const bf16x8*>(lds_tile + row * 128 + ((kc ^ (row & 7)) << 4));
  } else {
    const int g = lane >> 4, p = lane & 15;
    const int x = xb + 16 * (g & 1) + (p & 3) * 4;
    const int k = kk * 16 + 8 * (g >> 1) + (p >> 2);  // k & 3 == p >> 2 for both halves (k+4 keeps k&3)
    const int off = k * 256 + ((((x >> 3) ^ ((k & 3) << 2)) << 4) | ((x & 7) << 1));
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, lds_tile + off));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, lds_tile + off + 4 * 256));
    s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, r);
  }
}

__device__ __forceinline__ void tile_coords(int bid, int tiles_m, int tiles_n, int& tm, int& tn) {
  const int nwg = tiles_m * tiles_n;
  // bijective XCD remap: the blocks that land on XCD x (= bid % 8) get a contiguous run of logical tile ids
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  // GROUP_M-tall column panels
  const int GROUP_M = 8;
  const int per_group = GROUP_M * tiles_n;
  const int gidx = lid / per_group;
  const int first_m = gidx * GROUP_M;
  const int gsz = min(tiles_m - first_m, GROUP_M);
  const int in_g = lid - gidx * per_group;
  tm = first_m + in_g % gsz;
  tn = in_g / gsz;
}

template <bool AK, bool BKM>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const uint16_t* __restrict__ A, int lda,
                                                           const uint16_t* __restrict__ B, int ldb, int M, int N, int K,
                                                           int tiles_m, int tiles_n, EpiArgs epi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][A tile | B tile]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tm, tn;
  tile_coords(blockIdx.x, tiles_m, tiles_n, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int wr = wave >> 1, wc = wave & 1;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int nk = K / BK;
  stage_tile<AK>(A, lda, m0, 0, M, smem, wave, lane);
  stage_tile<BKM>(B, ldb, n0, 0, N, smem + TILE_BYTES, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int t = 0; t < nk; ++t) {
    char* cur = smem + (t & 1) * (2 * TILE_BYTES);
    char* nxt = smem + ((t + 1) & 1) * (2 * TILE_BYTES);
    if (t + 1 < nk) {
      stage_tile<AK>(A, lda, m0, (t + 1) * BK, M, nxt, wave, lane);
      stage_tile<BKM>(B, ldb, n0, (t + 1) * BK, N, nxt + TILE_BYTES, wave, lane);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = read_frag<AK>(cur, wr * 64 + i * 32, kk, lane);
#pragma unroll
      for (int j = 0; j < 2; ++j) bfr[j] = read_frag<BKM>(cur + TILE_BYTES, wc * 64 + j * 32, kk, lane);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // epilogue: restage the 128x128 fp32 tile through LDS (operand buffers are free now: 64 KiB = 128*128*4) so each
  // thread owns 8 consecutive columns of a row -> 16-B coalesced loads/stores for bias / residual / Z / C.
  float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        Cs[(wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 128 + wc * 64 + j * 32 + (lane & 31)] = acc[i][j][r];
  __syncthreads();
  for (int it = 0; it < 8; ++it) {
    const int id = it * 256 + tid;
    const int row = id >> 4, c8 = (id & 15) * 8;
    const int m = m0 + row, n = n0 + c8;
    if (m >= M || n >= N) continue;
    float v[8];
    const float4 lo = *reinterpret_cast<const float4*>(Cs + row * 128 + c8);
    const float4 hi = *reinterpret_cast<const float4*>(Cs + row * 128 + c8 + 4);
    v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
    epilogue_store8<uint16_t>(epi, m, n, v, min(8, N - n));
  }
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

extern "C" int mic_gemm(const mic_gemm_args* a, void* stream) {
  MIC_CHECK(a && a->A && a->B && a->C, "mic_gemm: null pointer");
  MIC_CHECK(a->M > 0 && a->N > 0 && a->K > 0, "mic_gemm: bad shape M=%d N=%d K=%d", a->M, a->N, a->K);
  MIC_CHECK(a->dtype == MIC_BF16 || a->dtype == MIC_F32, "mic_gemm: bad dtype %d", a->dtype);
  MIC_CHECK(a->c_dtype == a->dtype || a->c_dtype == MIC_F32, "mic_gemm: c_dtype must be dtype or f32");
  MIC_CHECK(!(a->dact && !a->Zin), "mic_gemm: dact needs Zin");
  MIC_CHECK(a->dropout_p >= 0.f && a->dropout_p < 1.f, "mic_gemm: dropout_p out of range");
  EpiArgs e;
  e.C = a->C; e.ldc = a->ldc; e.c_f32 = (a->c_dtype == MIC_F32);
  e.bias = a->bias; e.act = a->act; e.Zout = a->Zout; e.ldz = a->ldz; e.Zin = a->Zin; e.dact = a->dact;
  e.R = a->R; e.ldr = a->ldr; e.accumulate = a->accumulate;
  e.drop_thr = a->dropout_p > 0.f ? (uint32_t)fminf(a->dropout_p * 4294967296.0f, 4294967295.0f) : 0u;
  e.drop_seed = a->dropout_seed; e.drop_scale = 1.0f / (1.0f - a->dropout_p);
  e.alpha = a->alpha == 0.f ? 1.0f : a->alpha; e.N = a->N;
  hipStream_t s = (hipStream_t)stream;
  if (a->dtype == MIC_BF16) {
    MIC_CHECK(a->K % BK == 0, "mic_gemm(bf16): K=%d must be a multiple of 64 (zero-pad the reduction dim)", a->K);
    MIC_CHECK(a->lda % 8 == 0 && a->ldb % 8 == 0, "mic_gemm(bf16): lda/ldb must be multiples of 8");
    MIC_CHECK(((uintptr_t)a->A & 15) == 0 && ((uintptr_t)a->B & 15) == 0, "mic_gemm(bf16): A/B must be 16-B aligned");
    if (a->a_kmajor) MIC_CHECK(a->M % 8 == 0, "mic_gemm(bf16): k-major A needs M %% 8 == 0");
    if (a->b_kmajor) MIC_CHECK(a->N % 8 == 0, "mic_gemm(bf16): k-major B needs N %% 8 == 0");
    const int tiles_m = (a->M + BM - 1) / BM, tiles_n = (a->N + BN - 1) / BN;
    dim3 grid(tiles_m * tiles_n), block(256);
    const size_t lds = 4 * TILE_BYTES;
    const uint16_t* A = (const uint16_t*)a->A; const uint16_t* B = (const uint16_t*)a->B;
#define LAUNCH(AKM, BKMM) hipLaunchKernelGGL((gemm_bf16_kernel<AKM, BKMM>), grid, block, lds, s, A, a->lda, B, a->ldb, a->M, a->N, a->K, tiles_m, tiles_n, e)
    if (!a->a_kmajor && !a->b_kmajor) LAUNCH(false, false);
    else if (!a->a_kmajor && a->b_kmajor) LAUNCH(false, true);
    else if (a->a_kmajor && a->b_kmajor) LAUNCH(true, true);
    else LAUNCH(true, false);
#undef LAUNCH
  } else {
    dim3 grid((a->N + 63) / 64, (a->M + 63) / 64), block(256);
    const long sam = a->a_kmajor ? 1 : a->lda, sak = a->a_kmajor ? a->lda : 1;
    const long sbk = a->b_kmajor ? a->ldb : 1, sbn = a->b_kmajor ? 1 : a->ldb;
    hipLaunchKernelGGL(gemm_f32_kernel, grid, block, 0, s, (const float*)a->A, sam, sak, (const float*)a->B, sbk, sbn,
                       a->M, a->N, a->K, e);
  }
  MIC_LAUNCH_CHECK();
  return MIC_OK;
}
