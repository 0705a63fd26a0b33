/* mic_hip.h — C ABI of libmic_hip.so: the MI355X (gfx950) kernels behind the CLIP-Vision + mBART-50
 * captioning hot path (train step and KV-cached greedy / beam generate).
 *
 * The reference (gchhablani/multilingual-image-captioning) is pure Python/Flax: it has NO native boundary
 * for this path — XLA emits every kernel.  The boundary below is therefore build-defined
 * (SURVEY §8b): one entry point per fused op that the reference's module graph implies; each comment cites
 * the reference lines whose arithmetic the op replaces.  `modeling:` =
 * models/flax_clip_vision_mbart/modeling_clip_vision_mbart.py, `gen:` = .../generation_clip_vision_utils.py,
 * 3P = transformers@0085e71 (modeling_flax_clip.py / modeling_flax_mbart.py), un-vendored.
 *
 * Conventions
 *  - All pointers are caller-owned DEVICE pointers; outputs and workspaces are caller-allocated.
 *  - Every call is asynchronous on `stream` (a hipStream_t passed as void*) and re-entrant across streams and
 *    threads.  State the library keeps between calls, all of it host-side planning state (no device memory):
 *      (1) the GEMM tile planner's CU budget — THREAD-LOCAL, set by mic_set_cu_budget (0 = default), read by
 *          the mic_gemm* calls of the same thread and by mic_get_cu_budget / mic_gemm_plan;
 *      (2) A/B switches read ONCE per process from the environment at the first GEMM call and latched:
 *          MIC_FREE_CUS, MIC_GEMM_TILE, MIC_TINY_BELOW, MIC_GEMM_QUANT, MIC_GEMM_T192, MIC_GEMM_PHASED,
 *          MIC_GEMM_W4, MIC_GEMM_D2, MIC_GEMM_KG, MIC_GEMM_KG128, MIC_GEMM_PERSIST, and MIC_LNB_BLOCKS at the
 *          first LayerNorm backward (tools/README.md; the defaults are the measured best, nothing on the
 *          product path sets them);
 *      (3) per-device one-time attributes of the kernels (dynamic-LDS size) and the thread-local
 *          mic_last_error() message.
 *    Nothing else: no caches of caller pointers, no hidden workspaces, no streams of its own.
 *    Return 0 on success, negative MIC_E* on bad arguments (never throws, never exits).
 *  - Activations are row-major [rows][width]; `dtype` selects the storage type of activations
 *    (MIC_BF16 = bf16 storage, fp32 accumulate/statistics;  MIC_F32 = the reference's default dtype).
 *  - Linear weights are stored [out][in] (k-contiguous); LayerNorm / bias vectors and optimizer state fp32.
 */
#ifndef MIC_HIP_H
#define MIC_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MIC_OK 0
#define MIC_EINVAL (-1)   /* bad argument / unsupported shape */
#define MIC_ELAUNCH (-2)  /* HIP launch failure */

#define MIC_BF16 0
#define MIC_F32 1
#define MIC_FP8 2  /* GEMM operands only: OCP fp8, one byte per element (BASELINE configs[4]) */
#define MIC_E4M3 0 /* OCP e4m3fn: max 448 (activations, weights) */
#define MIC_E5M2 1 /* OCP e5m2: max 57344 (gradients) */

/* activation ids for GEMM epilogues */
#define MIC_ACT_NONE 0
#define MIC_ACT_GELU_ERF 1   /* PT-twin mBART "gelu" */
#define MIC_ACT_GELU_TANH 2  /* jax.nn.gelu default at the pinned commit [UNVERIFIED-3P] */
#define MIC_ACT_QUICK_GELU 3 /* CLIP: x*sigmoid(1.702x) */

int mic_version(void);
const char* mic_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * GEMM with fused epilogue:  C[M,N] = epi( op(A)[M,K] * op(B)[K,N] )
 *   a_kmajor = 0: A stored [M][lda>=K] (k contiguous);  1: A stored [K][lda>=M] (m contiguous)
 *   b_kmajor = 0: B stored [N][ldb>=K] (k contiguous);  1: B stored [K][ldb>=N] (n contiguous)
 *   Linear fwd  y = x W^T      : a_kmajor=0, b_kmajor=0  (nn.Dense, 3P; modeling:53-59, 90)
 *   Linear dX   dx = dy W      : a_kmajor=0, b_kmajor=1
 *   Linear dW   dW = dy^T x    : a_kmajor=1, b_kmajor=1
 * epilogue, in this order (each optional):
 *   v = acc (+ bias[n]);  Zout[m,n] = v (pre-activation, saved for backward);  v = act(v);
 *   v *= act'(Zin[m,n]) (activation backward);  v = dropout(v; seed, p, index m*N+n);
 *   v += R[m,n];  v += C_old[m,n] (accumulate);  C[m,n] = v  (c_dtype)
 * Requirements: K % 64 == 0 for MIC_BF16 (callers zero-pad the reduction dimension); lda/ldb % 8 == 0 (bf16).
 * bf16 path: LDS-staged 64x64x64, 128x128x64 or 256x256x64 tiles (operands through registers into XOR-swizzled LDS images),
 * v_mfma_f32_32x32x16_bf16, k-major operands through ds_read_b64_tr_b16.  f32 path: v_mfma_f32_32x32x2_f32 (exact fp32).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  int dtype;         /* dtype of A, B, R, Zin, Zout */
  int c_dtype;       /* dtype of C */
  int M, N, K;
  int a_kmajor, b_kmajor;
  const void* A; int lda;
  const void* B; int ldb;
  void* C; int ldc;
  const float* bias;       /* [N] or NULL */
  int act;                 /* MIC_ACT_* applied to v */
  void* Zout; int ldz;     /* pre-activation out or NULL */
  const void* Zin; int dact; /* multiply by act'(Zin) with activation id `dact` (0 = off) */
  const void* R; int ldr;  /* residual or NULL */
  int accumulate;          /* add existing C */
  float dropout_p; uint32_t dropout_seed; /* p = 0 disables */
  float alpha;             /* scale on acc before bias (0 means 1) */
  int split_k;             /* > 1 (bf16; fp8 NT launches with split_stride > 0): split the reduction over that many workgroups per tile;
                              partial sums are atomically added (fp32) into a caller-zeroed C; no other epilogue allowed */
  long long split_stride;  /* split_k > 1 only.  0: partial sums are atomically added into C (above).  > 0: split s stores its
                              partial tile plainly at C + s * split_stride (elements; C = fp32 workspace of split_k slabs),
                              to be summed by mic_sum_slabs — no atomics, no zero-fill, deterministic */
  float* a_rowsum;         /* optional (bf16, a_kmajor launches only): a_rowsum[m] += sum over k < rowsum_k of A(m,k), fp32 atomics into a
                              caller-zeroed vector.  With A = dy^T (a_kmajor, the weight-gradient GEMM dW = dy^T x) this is
                              the bias gradient colsum(dy) (nn.Dense bias; main.py:696 grads) from operand fragments the
                              kernel holds anyway — no extra pass over dy */
  int rowsum_k;            /* valid reduction rows for a_rowsum (0 = K) */
  /* dtype == MIC_FP8 (BASELINE configs[4]; the reference has no counterpart, its dtypes are main.py:96-101): A and B are OCP
   * fp8 bytes, both k-contiguous (a_kmajor = b_kmajor = 0) or both K-MAJOR (a_kmajor = b_kmajor = 1: the weight-gradient form
   * dW = dy^T x on dy [K][M] and x [K][N] as their producers wrote them, M % 16 == N % 16 == 0, k_valid as for bf16; fragments by
   * ds_read_b64_tr_b8), K % 128 == 0, lda/ldb in bytes and multiples of 16; R / Zin / Zout
   * are bf16, C is c_dtype (bf16 or f32).  v = acc * a_scale_inv[0] * b_scale_inv[0] (device scalars written by
   * mic_fp8_quantize: the per-tensor dequantisation factors) before bias.  v_mfma_scale_f32_32x32x64_f8f6f4 with unit block
   * scales: fp32 accumulate at twice the bf16 MFMA rate, half the operand bytes. */
  int a_fmt, b_fmt;                 /* MIC_E4M3 / MIC_E5M2 (B must be e4m3) */
  const float* a_scale_inv;         /* device scalar or NULL (= 1) */
  const float* b_scale_inv;
  /* optional by-product of the LM-head GEMM (bf16 operands, or fp8 NT operands — the fp8 head of configs[4] —; bf16 C, bias-only
   * epilogue, N % 64 == 0): rowstat[(m * rowstat_ld + g) * 2 + {0, 1}]
   * = max and sum exp(x - max) over the columns [64 g, 64 g + 64) & [0, rowstat_nvalid) of output row m, taken on the values as
   * stored in C.  With them the log-softmax of main.py:672-675 / gen:850 needs no second pass over the [rows][250 054]
   * logits: mic_ce_rows_tiles and mic_row_topk_tiles merge the N / 64 partials of a row. */
  float* rowstat; int rowstat_ld; int rowstat_nvalid;
  /* LayerNorm folded around the GEMM (bf16, decode path; the LayerNorms of modeling_flax_mbart's decoder layer, T2):
   *   LN(x) W^T = rstd (x (gamma o W)^T - mu g) + beta W^T,   g[n] = sum_k gamma[k] W[n][k]
   * so a Linear whose input is LN(x) runs on the RAW x with the pre-scaled weight B = gamma o W (mic_ln_fold_weight) and an
   * epilogue that needs only (sum, sum of squares) of each A row: a_ln_stats int64 [M][2] (2^20 fixed point), a_ln_colsum = g fp32 [N],
   * a_ln_width = the normalised width, a_ln_eps; `bias` must then be bias' = bias + beta W^T (also from mic_ln_fold_weight).
   * Producer side: rowsum2 int64 [M][2] (caller-zeroed) receives (sum, sum of squares) x 2^20 of every output row AS STORED
   * (one pair of integer atomics per row and wave-tile column: integer adds commute, so the result does not depend on the
   * order the column tiles finish in — generate stays run-to-run deterministic and independent of the batch order) — the
   * stats the next folded LayerNorm needs, so the normalised activations are never written and the LayerNorm kernel launch
   * disappears.  N % 128 == 0, bare or residual epilogue. */
  const long long* a_ln_stats; const float* a_ln_colsum; int a_ln_width; float a_ln_eps;
  long long* rowsum2;
  /* bf16, a_kmajor && b_kmajor launches (weight gradients dW = dy^T x, reduction over the ROWS of dy and x): rows k >= k_valid of
   * both operands count as zero (0 = all K rows are valid).  K stays a multiple of 64; the buffers need not keep their rows
   * [k_valid, K) zeroed — with variable-length (packed) batches the number of valid rows changes from step to step. */
  int k_valid;
  /* dtype == MIC_FP8 with c_dtype == MIC_FP8: fused emission (mic_fp8_out below) — the epilogue's result leaves as fp8 bytes
   * (C, ldc in bytes) under the OUTPUT tensor's delayed scale: c_q8_state [2] (amax of the previous pass read, 1 / scale written),
   * c_q8_amax this pass's partial maxima (or NULL), c_q8_fmt MIC_E4M3 / MIC_E5M2.  Activation / dact epilogues of NT launches
   * (GELU(FFN-in) -> the e4m3 operand of FFN-out; dGELU-scaled dX of FFN-out -> the e5m2 dy of FFN-in's backward). */
  float* c_q8_state; float* c_q8_amax; int c_q8_fmt;
} mic_gemm_args;
int mic_gemm(const mic_gemm_args* a, void* stream);
/* dst[r][c] (dst_dtype) = sum over s < n_slabs of src[s * slab_stride + r * ld_src + c] (fp32): the second half of a
 * workspace split-K GEMM. */
int mic_sum_slabs(int dst_dtype, int n_slabs, long long slab_stride, int rows, int cols, const float* src, int ld_src,
                  void* dst, int ld_dst, void* stream);
/* `count` GEMMs that share dtype and operand layouts in as few launches as possible (one launch per 8 problems):
 * the weight-gradient GEMMs of a layer have 36..256 output tiles each — grouped they fill the 256 CUs. */
int mic_gemm_grouped(const mic_gemm_args* args, int count, void* stream);
/* The CU budget of the GEMM tile planner: how many of the device's CUs a launch may count on (0 = default: the device's 256, or
 * MIC_FREE_CUS from the environment).  The planner sizes "one round" launches (one 16-wave block with two / four K-groups per CU)
 * and persistent grids for this number; a data-parallel job lowers it by the CUs its collectives occupy — the reference's
 * `lax.pmean` (main.py:698) is scheduled by XLA inside the step, here RCCL's channel blocks sit on CUs of their own beside backward —
 * so that a launch sized for 256 free CUs re-plans instead of spilling a few blocks into a second round.  Process-wide. */
int mic_set_cu_budget(int cus);
int mic_get_cu_budget(void);
/* What mic_gemm_grouped would launch for these problems under the current CU budget (host arithmetic only, nothing is launched;
 * pointers in `args` are not dereferenced): tile edge (256 / 128 / 64; `tile_m` x `tile` when the rows differ), K-groups per block, logical blocks, launched grid
 * (persistent launches: the budget), blocks of this configuration that fit one CU, and whether the LDS-DMA phased kernel is taken
 * (phased = 1; 2 = the shape fits the four-wave kernel gemm_w4.hip — on by default, MIC_GEMM_W4=0 switches it off — which takes the launch
 * if its epilogue is a bare one: bf16 C with bias / folded LayerNorm / softmax partials, or fp32 C, also as split-K slabs; of those
 * the launches WITH softmax partials run on the two-blocks-per-CU 256 x 128 kernel gemm_d2.hip, MIC_GEMM_D2=0 switches that off — for
 * args that carry `rowstat` the plan reports that tiling: tile 128, tile_m 256, blocks_per_cu 2). */
typedef struct { int tile, kgroups, blocks, grid, blocks_per_cu, phased, cu_budget, tile_m; } mic_gemm_plan_info;  /* tile_m: tile rows (= tile, or 192 with tile 128) */
int mic_gemm_plan(const mic_gemm_args* args, int count, mic_gemm_plan_info* out);
/* Operands of a LayerNorm-folded Linear (see mic_gemm_args.a_ln_stats): for w [N][K] (the compute-dtype weight), gamma / beta
 * fp32 [K], bias fp32 [N] or NULL:  w_fold[n][k] = round(w[n][k] * gamma[k]),  colsum[n] = sum_k w_fold[n][k] (of the ROUNDED
 * values: the epilogue subtracts exactly what the MFMAs added),  bias_fold[n] = bias[n] + sum_k w[n][k] beta[k]. */
int mic_ln_fold_weight(int dtype, int N, int K, const void* w, int ldw, const float* gamma, const float* beta, const float* bias,
                       void* w_fold, int ldwf, float* colsum, float* bias_fold, void* stream);

/* ---------------------------------------------------------------------------------------------
 * fp8 operands for mic_gemm (BASELINE configs[4]; no reference counterpart — its dtypes are fp32/fp16/bf16, main.py:96-101).
 * Per-tensor current scaling: mic_fp8_amax accumulates state[0] = max |x| (fp32 atomic max; the caller zeroes `state`),
 * mic_fp8_quantize then writes q[r][c] = fp8(x[r][c] * FMAX / amax) (row-major) and / or the transposed copy
 * qT[c][r] (rows zero-padded to rows_pad, so it can be the k-contiguous operand of the weight-gradient GEMM, which reduces
 * over rows), and state[1] = amax / FMAX — the factor mic_gemm multiplies back (a_scale_inv / b_scale_inv).
 * `items` is a HOST array; src is bf16 [rows][ld]; e4m3 (FMAX 448) for activations and weights, e5m2 (FMAX 57344) for gradients.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  const void* src; int ld;      /* bf16 [rows][ld], cols % 8 == 0 */
  int rows, cols, rows_pad;     /* rows_pad >= rows: qT gets rows_pad columns (0 = rows) */
  void* q; int ldq;             /* [rows][ldq] bytes or NULL */
  void* qT; int ldqT;           /* [cols][ldqT] bytes or NULL */
  float* state;                 /* [2]: amax, 1 / scale */
  float* amax_next;             /* delayed scaling: table of partial maxima (see below) or NULL */
  int fmt;                      /* MIC_E4M3 / MIC_E5M2 */
} mic_fp8_item;
int mic_fp8_amax(const mic_fp8_item* items, int count, void* stream);
int mic_fp8_quantize(const mic_fp8_item* items, int count, void* stream);
/* Delayed scaling: an item with amax_next != NULL is quantised by mic_fp8_quantize ALONE with the scale taken from the amax
 * already in state[0] (values beyond it saturate at +-FMAX) while max |x| of this pass is accumulated into the tensor's table
 * of partial maxima amax_next[0 .. mic_fp8_amax_partials()) (fp32 atomic max, one per 64x64 tile, spread over the table;
 * caller-zeroed) — the tensor is read once instead of twice.  mic_fp8_roll_amax: for slot i < count, state[i * stride_floats]
 * = max(partials[i * P .. (i+1) * P)) if that is > 0, and the partials are cleared — called at the start of a pass. */
int mic_fp8_amax_partials(void);
int mic_fp8_roll_amax(float* state, int stride_floats, float* partials, int count, void* stream);

/* Fused fp8 emission under delayed scaling: the PRODUCER of an fp8 GEMM operand writes the bytes itself — q[r][c] =
 * fp8(round_bf16(x[r][c]) * FMAX / state[0]) with state[0] the amax the tensor had in the previous pass (mic_fp8_roll_amax), records
 * this pass's max |x| in amax_next (as mic_fp8_quantize does under delayed scaling) and writes state[1] = state[0] / FMAX, the
 * dequantisation factor mic_gemm multiplies back.  Same bytes as producer + mic_fp8_quantize on the same scale; one launch less per
 * tensor (the step had 181 of them).  Producers: mic_layernorm_fwd_q8, mic_layernorm_bwd_partials_q8, mic_attn_bwd_q8 /
 * mic_attn_bwd_packed_q8 and the epilogue of an fp8 mic_gemm (mic_gemm_args.c_q8).  No reference counterpart (configs[4]). */
typedef struct {
  void* q; int ldq;      /* fp8 bytes [rows][ldq], ldq % 8 == 0, 8-B aligned */
  float* state;          /* [2]: amax of the previous pass (read), 1 / scale (written) */
  float* amax_next;      /* [mic_fp8_amax_partials()] partial maxima of this pass (atomic max) or NULL */
  int fmt;               /* MIC_E4M3 / MIC_E5M2 */
} mic_fp8_out;

/* ---------------------------------------------------------------------------------------------
 * LayerNorm (flax nn.LayerNorm: biased variance, fp32 statistics; 3P).
 *   fwd: y = (x-mean)*rstd*gamma + beta, then optional dropout; saves mean/rstd [rows] (may be NULL).
 *   bwd: dx = LNbwd(dy) (+ dres if given);  dgamma/dbeta accumulated with fp32 atomics (caller zeroes);
 *        optional second output dxm = dropout_mask(seed_m) * dx / (1-p_m): the gradient entering the
 *        residual branch that produced x (its epilogue applied that dropout in forward).
 * ------------------------------------------------------------------------------------------- */
int mic_layernorm_fwd(int dtype, int rows, int width, const void* x, const float* gamma, const float* beta,
                      float eps, void* y, float* mean, float* rstd, float dropout_p, uint32_t dropout_seed,
                      void* stream);
int mic_layernorm_bwd(int dtype, int rows, int width, const void* x, const float* gamma, const float* mean,
                      const float* rstd, const void* dy, const void* dres, void* dx, float* dgamma, float* dbeta,
                      void* dxm, float dropout_p, uint32_t dropout_seed, float in_dropout_p, uint32_t in_dropout_seed,
                      void* stream);
/* mic_layernorm_bwd with the gamma / beta gradients as per-block partial column sums (plain stores) instead of fp32 atomics:
 * partials [2][mic_layernorm_bwd_blocks(rows)][width] fp32 (gamma sums first), fully overwritten.  mic_ln_param_grads adds the blocks
 * up in block order — deterministic, and off the critical path: the atomics were a third of the kernel's time at 2.4 k rows. */
int mic_layernorm_bwd_blocks(int rows);
int mic_layernorm_bwd_partials(int dtype, int rows, int width, const void* x, const float* gamma, const float* mean,
                               const float* rstd, const void* dy, const void* dres, void* dx, float* partials, void* dxm,
                               float dropout_p, uint32_t dropout_seed, float in_dropout_p, uint32_t in_dropout_seed, void* stream);
/* bf16 storage, fused fp8 emission (mic_fp8_out above).  fwd: the normalised rows as e4m3 bytes beside y, or — y == NULL — instead
 * of it (the operand of an fp8 q/k/v / FFN-in projection; LayerNorm's backward needs x and the statistics, not y).  bwd: q8_of_dx =
 * 0: the dropout-masked gradient dxm as fp8 (dxm == NULL: only as fp8) — the dy of the fp8 FFN-out projection of the layer below;
 * 1: dx itself also as fp8 (the ViT has no dropout: its residual-stream gradient is that dy). */
int mic_layernorm_fwd_q8(int rows, int width, const void* x, const float* gamma, const float* beta, float eps, void* y, float* mean,
                         float* rstd, float dropout_p, uint32_t dropout_seed, const mic_fp8_out* q8, void* stream);
int mic_layernorm_bwd_partials_q8(int rows, int width, const void* x, const float* gamma, const float* mean, const float* rstd,
                                  const void* dy, const void* dres, void* dx, float* partials, void* dxm, float dropout_p,
                                  uint32_t dropout_seed, float in_dropout_p, uint32_t in_dropout_seed, const mic_fp8_out* q8,
                                  int q8_of_dx, void* stream);
typedef struct { const float* partials; float* dgamma; float* dbeta; int nblk, width, accumulate; } mic_ln_param_item;  /* dgamma / dbeta may be NULL */
int mic_ln_param_grads(const mic_ln_param_item* items, int count, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Attention core (flax dot_product_attention_weights): q scaled by 1/sqrt(D) first,
 * softmax(q k^T + bias) v, heads merged.  q [B*Tq][ldq], k/v [B*Tk][ldk/ldv] with head h at column
 * h*64 (head_dim is 64 for both models); out [B*Tq][ldo].  causal: key j allowed iff j <= i.
 * key_mask int32 [B][Tk] (1 = attend) or NULL.  Disallowed -> -inf bias.  lse [B][H][Tq] saved for bwd.
 * Tq, Tk <= 64.  Covers K4 (ViT, 50x50), K9 (decoder causal+padding, 64x64), K10 (cross, 64x50).
 * ------------------------------------------------------------------------------------------- */
int mic_attn_fwd(int dtype, int B, int H, int Tq, int Tk, const void* q, int ldq, const void* k, int ldk,
                 const void* v, int ldv, void* out, int ldo, const int32_t* key_mask, int causal, float* lse,
                 void* stream);
int mic_attn_bwd(int dtype, int B, int H, int Tq, int Tk, const void* q, int ldq, const void* k, int ldk,
                 const void* v, int ldv, const void* out, int ldo, const void* dout, int lddo, const float* lse,
                 const int32_t* key_mask, int causal, void* dq, int lddq, void* dk, int lddk, void* dv, int lddv,
                 void* stream);

/* The attention weights themselves, for `output_attentions=True` (modeling:499-510 forwards the flag to the Flax modules, which
 * return every layer's softmax weights): out[B][H][Tq][Tk] (fp32) = softmax over the keys of q.k / sqrt(64) with the masks of
 * mic_attn_fwd.  A diagnostic kernel (one wave per query row), never launched by the train / generate paths.  Tk <= 1024. */
int mic_attn_probs(int dtype, int B, int H, int Tq, int Tk, const void* q, int ldq, const void* k, int ldk,
                   const int32_t* key_mask, int causal, float* out, void* stream);

/* The same cores on variable-length ("packed") rows: sequence b owns the q rows [q_off[b], q_off[b] + q_len[b]) of a packed
 * [sum q_len][ld] matrix — padded positions have no rows at all (their loss weight is 0 and no valid position attends to them,
 * main.py:678, 692: every gradient they would contribute is exactly 0).  kv_packed = 1: keys / values are the same packed rows
 * (decoder self-attention: key j allowed iff j <= i, all within q_len[b]); kv_packed = 0: every sequence has its Tk dense rows
 * [b*Tk, (b+1)*Tk) (cross-attention over the encoder states).  q_len[b] <= Tq_max <= 64, Tk <= 64; lse [B][H][Tq_max]. */
int mic_attn_fwd_packed(int dtype, int B, int H, int Tq_max, int Tk, const int32_t* q_off, const int32_t* q_len, int kv_packed,
                        const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* out, int ldo, int causal,
                        float* lse, void* stream);
int mic_attn_bwd_packed(int dtype, int B, int H, int Tq_max, int Tk, const int32_t* q_off, const int32_t* q_len, int kv_packed,
                        const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* out, int ldo,
                        const void* dout, int lddo, const float* lse, int causal, void* dq, int lddq, void* dk, int lddk,
                        void* dv, int lddv, void* stream);
/* mic_attn_bwd / mic_attn_bwd_packed (q_off == NULL: dense rows) of the bf16 single-tile kernel with dQ / dK / dV written as fp8
 * bytes under delayed scaling (mic_fp8_out): dq8 describes dQ's byte matrix and tensor, dk8 dK's; dV (dv8) shares dK's leading
 * dimension, scale and amax table — dK and dV are the two halves of ONE [rows][2d] (or, with dQ, [rows][3d]) gradient tensor. */
int mic_attn_bwd_q8(int B, int H, int Tq, int Tk, const int32_t* q_off, const int32_t* q_len, int kv_packed, const void* q, int ldq,
                    const void* k, int ldk, const void* v, int ldv, const void* out, int ldo, const void* dout, int lddo, const float* lse,
                    const int32_t* key_mask, int causal, const mic_fp8_out* dq8, const mic_fp8_out* dk8, void* dv8, void* stream);

/* Decode-time self-attention over the static max_len-slot cache (3P _concatenate_to_cache; modeling:249-282;
 * SURVEY App. B7): one query per row, validity slot <= cur (cache_index).  The cache is NOT physically
 * reordered by beam (gen:945-953): src_row [R][max_len] int32 says in which row slot s of row r's
 * history lives (beam-parent indirection); NULL = row r reads cache row r / row_div (cross-attention K/V are
 * computed once per image and shared by its beams: row_div = num_beams, cur = S-1).  kc/vc: [rows][max_len] slots of
 * ldc elements each (ldc = H*64 for the self cache; 2*H*64 when k and v are the halves of a fused projection). */
int mic_attn_decode(int dtype, int R, int H, int max_len, int cur, const void* q, int ldq, const void* kc,
                    const void* vc, int ldc, const int32_t* src_row, int row_div, void* out, int ldo, void* stream);
/* writes this step's k,v (columns of the fused qkv projection) into slot `cur` of every row's own cache */
int mic_kv_append(int dtype, int R, int HD, int max_len, int cur, const void* k, int ldk, const void* v, int ldv,
                  void* kc, void* vc, void* stream);

/* ---------------------------------------------------------------------------------------------
 * ViT embeddings (3P FlaxCLIPVisionEmbeddings)
 *   im2col: pixels NHWC fp32 [B][img][img][3] -> patches [B*g*g][ps*ps*3] (dtype), k order (u,v,c) = HWIO.
 *           trunc_int32 = 1 reproduces encode()'s cast of pixel_values to int32 (modeling:330).
 *   assemble: x[b,0] = cls + pos[0]; x[b,1+p] = patch_out[b,p] + pos[1+p]
 *   assemble_bwd: dpatch = dx[:,1:], dpos += sum_b dx, dcls += sum_b dx[:,0]   (fp32 atomics; caller zeroes)
 * ------------------------------------------------------------------------------------------- */
int mic_im2col(int dtype, int B, int img, int ps, const float* pixels, void* patches, int ldp, int trunc_int32,
               void* stream);
int mic_vit_assemble(int dtype, int B, int S, int width, const void* patch_out, int ldp, const float* cls,
                     const float* pos, void* x, void* stream);
int mic_vit_assemble_bwd(int dtype, int B, int S, int width, const void* dx, void* dpatch, int ldp, float* dcls,
                         float* dpos, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Decoder token embedding (3P FlaxMBartDecoder):
 *   h[r] = table[ids[r]] * scale + pos_table[pos_ids[r] + 2]        (table in `dtype`, pos_table fp32)
 *   bwd: dtable[ids[r]] += dh[r]*scale ; dpos_table[pos+2] += dh[r]  (fp32 atomics into the grad buffers); either output
 *        may be NULL (data-parallel runs all-gather the (ids, dh) rows and scatter them after the dense all-reduce)
 * ------------------------------------------------------------------------------------------- */
int mic_embed_fwd(int dtype, int rows, int width, const int32_t* ids, const int32_t* pos_ids, const void* table,
                  const float* pos_table, float scale, void* h, void* stream);
int mic_embed_bwd(int dtype, int rows, int width, const int32_t* ids, const int32_t* pos_ids, const void* dh,
                  float scale, float* dtable, float* dpos_table, void* stream);
/* The token-embedding scatter of the DATA-PARALLEL step (main.py:698 pmean of a gradient whose embedding part is sparse): dtable[ids[i]]
 * += scale * dh[i] for the n all-gathered rows of all ranks, DETERMINISTIC — the occurrences of an id are added by a single writer in a
 * fixed order (a function of the index set only), so every rank computes the same bits and the replicas stay identical (fp32 atomics do
 * not guarantee that).  ids < 0 are skipped (padding); n <= 65536, width % 64 == 0.  ws: mic_embed_rows_add_det_ws(vocab) ints, initialised ONCE by the caller (ints [0, vocab) 0x7fffffff,
 * [vocab, 2 vocab) 0, [2 vocab, 3 vocab) -1, the rest 0); the call leaves its first 3 vocab + 1 ints in that state (the rest is scratch). */
int mic_embed_rows_add_det(int dtype, int n, int width, int vocab, const int32_t* ids, const void* dh, float scale, float* dtable,
                           int32_t* ws, void* stream);
int64_t mic_embed_rows_add_det_ws(int vocab);

/* ---------------------------------------------------------------------------------------------
 * Masked (label-smoothed) softmax cross-entropy over materialised logits (main.py:658-680).
 *   logits [rows][ld] (dtype), columns >= V are padding.  Pass 1 (mic_ce_rows): per-row lse, nll(ls) -> row_loss,
 *   Pass 2 (mic_ce_bwd): logits <- dlogits = mask/denom * (softmax - soft_label) in place, padding columns <- 0,
 *   where denom = sum(mask) is read from device (denom[0]).  loss = sum(row_loss*mask)/denom by mic_ce_reduce.
 * ------------------------------------------------------------------------------------------- */
int mic_ce_rows(int dtype, int rows, int V, const void* logits, int ld, const int32_t* labels,
                const int32_t* mask, float label_smoothing, float* row_lse, float* row_loss, void* stream);
/* mic_ce_rows without the pass over the logits: row_lse from the head GEMM's per-granule partials (mic_gemm_args.rowstat,
 * [rows][stat_ld] float2, 64 columns each), row_loss = lse - logits[label] (label_smoothing 0: plain NLL, main.py:674 with a one-hot target) */
int mic_ce_rows_tiles(int dtype, int rows, int V, const void* logits, int ld, const float* rowstat, int stat_ld,
                      const int32_t* labels, float* row_lse, float* row_loss, void* stream);
int mic_ce_reduce(int rows, const float* row_loss, const int32_t* mask, float* loss_out, float* denom_out,
                  void* stream);
int mic_ce_bwd(int dtype, int rows, int V, int Vpad, void* logits, int ld, const int32_t* labels,
               const int32_t* mask, float label_smoothing, const float* row_lse, const float* denom,
               float loss_scale, void* stream);
/* mic_ce_bwd (bf16 logits) with the gradient leaving as fp8 bytes q8->q [rows][ldq] (e5m2) instead of in place — the logits stay as they
 * are.  The scale of this tensor is known in closed form: every entry of g = mask (softmax - soft_label) lies in [-1, 1], so q = fp8(g *
 * FMAX) and the dequantisation factor q8->state[1] = loss_scale / (denom * FMAX) (state[0] = its reciprocal) are written with no amax
 * history (q8->amax_next is not used) and nothing saturates.  Padding columns V .. Vpad are zero bytes.  With `colsum` != NULL the column
 * sums of the fp32 gradient are ADDED to colsum[0 .. Vpad) (fp32 atomics: the gradient of final_logits_bias, modeling:178).  The fp8 LM
 * head (BASELINE configs[4]): dX = dlogits . E^T-copy and dE = dlogits^T . h (both operands k-major) read this ONE copy — main.py:692-698
 * through the tied head, modeling:170-174.
 * label_coef != NULL: the label entry of every row — w (p_label - conf), at least half of the row's gradient energy — is NOT rounded to
 * two mantissa bits: the byte at [row][labels[row]] is zero and label_coef[row] receives the fp32 gradient; mic_head_label_terms adds
 * its two products exactly. */
int mic_ce_bwd_q8(int rows, int V, int Vpad, const void* logits, int ld, const int32_t* labels, const int32_t* mask,
                  float label_smoothing, const float* row_lse, const float* denom, float loss_scale, const mic_fp8_out* q8,
                  float* colsum, float* label_coef, void* stream);
/* The label entries of dlogits (mic_ce_bwd_q8 label_coef; bf16 E [V][lde] and h [rows][ldh]): dx_slab[m][0 .. width) = coef[m] *
 * E[labels[m]][:] — one more fp32 slab for mic_sum_slabs behind the split-K dX GEMM — and dE[labels[m]][:] += coef[m] * h[m][:] (fp32
 * atomics into the gradient the dE GEMM has written). */
int mic_head_label_terms(int rows, int width, const int32_t* labels, const float* coef, const void* E, int lde, const void* h, int ldh,
                         float* dx_slab, int ldx, float* dE, int ldde, void* stream);
/* mic_ce_bwd (bf16 logits) that ALSO writes the transposed gradient dlogits_t [Vpad][ld_t] (bf16; columns rows .. rows_pad — a
 * multiple of 64, 0 = rows rounded up to 64 — are zeros) and, with `colsum` != NULL, ADDS the column sums of the dlogits as stored to colsum[0 .. Vpad) (fp32 atomics —
 * the gradient of final_logits_bias, modeling:178).  The LM head's backward GEMMs (main.py:692-698 through the tied head,
 * modeling:170-174) then run as NT launches: dE = dlogits_t . h_t^T reduces over the rows, dX = dlogits . E_t^T over the vocabulary. */
int mic_ce_bwd_t(int rows, int V, int Vpad, void* logits, int ld, const int32_t* labels, const int32_t* mask, float label_smoothing,
                 const float* row_lse, const float* denom, float loss_scale, void* dlogits_t, int ld_t, int rows_pad, float* colsum,
                 void* stream);
/* dst[c][r] = src[r][c] (bf16), r < rows, c < cols; dst columns rows .. rows_pad (a multiple of 64; 0 = rows rounded up to 64) are
 * written as zeros (the padding of a GEMM's reduction dimension).  Makes h^T and E^T, the k-contiguous operands of the tied head's
 * backward (modeling:170-174; the reference leaves operand layouts to XLA).  cols, ld_src, ld_dst multiples of 8; ld_dst >= rows_pad; 16-B
 * aligned operands. */
int mic_transpose_bf16(int rows, int rows_pad, int cols, const void* src, int ld_src, void* dst, int ld_dst, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Small reductions / elementwise
 * ------------------------------------------------------------------------------------------- */
/* out[n] (+)= sum_m x[m,n]  (bias gradients).  accumulate=0 overwrites. */
int mic_colsum(int dtype, int rows, int cols, const void* x, int ld, float* out, int accumulate, void* stream);
/* several column sums (always accumulating into caller-zeroed outputs) in one launch per 8 items */
typedef struct { const void* x; float* out; int rows, cols, ld; } mic_colsum_item;
int mic_colsum_grouped(int dtype, const mic_colsum_item* items, int count, void* stream);
/* the same over fp8 tensors (fused emission): out[c] += scale_inv[0] * sum_r fp8(x[r][c]) — the bias gradient of an fp8 projection from
 * the bytes its dy exists as; cols, ld multiples of 8 */
typedef struct { const void* x; float* out; const float* scale_inv; int rows, cols, ld, fmt; } mic_colsum_q8_item;
int mic_colsum_q8_grouped(const mic_colsum_q8_item* items, int count, void* stream);
/* keep-mask (uint8, 1 = keep) that the fused dropout epilogues use for (seed, p) over n elements */
int mic_dropout_mask(uint8_t* out, int64_t n, float p, uint32_t seed, void* stream);
int mic_cast(int src_dtype, int dst_dtype, const void* src, void* dst, int64_t n, void* stream);
/* zero-fill `bytes` bytes on `stream` (the per-step clears: atomically accumulated gradients, row padding of compacted
 * buffers, fp8 amax slots) */
int mic_zero(void* p, int64_t bytes, void* stream);
/* A HIP stream restricted to bits [first_cu, first_cu + n_cus) of the device's CU mask (MI355X: bit i -> XCD i % 8, so a multiple
 * of 8 takes the same number of CUs on every XCD); a proper subset of the device's CUs.  Destroy with mic_stream_destroy.  The reference has no counterpart
 * (XLA schedules the optimizer inside the jitted step, main.py:684-707): this is how the per-bucket AdamW launches get their own
 * CUs beside backward. */
int mic_stream_create_cu_masked(int first_cu, int n_cus, void** stream);
int mic_stream_destroy(void* stream);
/* Stand-in for a collective on ONE GPU (bench.py --emulate-comm; never part of a real data-parallel step): a kernel of `blocks`
 * workgroups that copies `bytes` from src to dst twice (the HBM traffic of a ring all-reduce's local reads and writes) and then
 * holds its CUs until `micros` microseconds have passed since it started (bounded: <= 200 000).  Launched on a CU-masked stream it
 * occupies the CUs and the time an RCCL all-reduce of that bucket is projected to take (main.py:698 `lax.pmean`), so the step's
 * scheduling against a busy collective stream can be measured without a second GPU. */
int mic_comm_emulate(const void* src, void* dst, int64_t bytes, float micros, int blocks, void* stream);
/* row gather/scatter: dst[dst_idx ? dst_idx[i] : i] = src[src_idx ? src_idx[i] : i], i < n.  Used to run the LM head and
 * the cross-entropy only on the label positions whose loss mask is 1 (main.py:678: masked positions contribute exactly 0). */
int mic_copy_rows(int dtype, int n, int width, const void* src, int ld_src, const int32_t* src_idx, void* dst, int ld_dst,
                  const int32_t* dst_idx, void* stream);
/* dst[r][c] (dst_dtype, ld_dst) = src[r][c] (src_dtype, ld_src) */
int mic_cast2d(int src_dtype, int dst_dtype, int rows, int cols, const void* src, int ld_src, void* dst, int ld_dst,
               void* stream);

/* ---------------------------------------------------------------------------------------------
 * AdamW over a flat fp32 parameter buffer (optax.adamw, main.py:629-635):
 *   m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr * ( m/(1-b1^t) / (sqrt(v/(1-b2^t)) + eps) + wd p )
 * `lr` and `t` are read from device scalars (hyper[0] = lr, hyper[1] = t as float) so the step is graph-capturable.
 * b1/b2/eps/wd are doubles so that (1-b1), (1-b2) are formed in double like optax does before rounding to fp32.
 * Also refreshes the bf16 compute copy (p_lp, may be NULL) and scales g by grad_scale (e.g. 1/world).
 * ------------------------------------------------------------------------------------------- */
int mic_adamw(int64_t n, float* p, float* m, float* v, const float* g, void* p_lp, const float* hyper, double b1,
              double b2, double eps, double wd, float grad_scale, void* stream);
/* The same update on a [rows][width] slice of the flat buffers, restricted to the rows with (row_flag[row] != 0) == (want != 0).
 * The tied embedding `shared` takes gradient from two places (modeling:37-44 ties the LM head and the decoder's input embedding): the
 * dense LM-head half is complete right after the head's weight-gradient GEMM, the sparse input-embedding half (the rows of this
 * step's decoder input ids) only at the very end of backward.  AdamW is elementwise, so the rows outside the step's ids are
 * updated early (want = 0, beside backward) and the <= B*T flagged rows after the scatter (want = 1): same arithmetic, bit for bit.
 * mic_row_flags zero-fills flags[n_rows] and sets flags[ids[i]] = 1 (ids outside [0, n_rows) are ignored). */
int mic_adamw_rows(int64_t rows, int width, const uint8_t* row_flag, int want, float* p, float* m, float* v, const float* g,
                   void* p_lp, const float* hyper, double b1, double b2, double eps, double wd, float grad_scale, void* stream);
int mic_row_flags(const int32_t* ids, int n_ids, uint8_t* flags, int n_rows, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Generation epilogues
 *   mic_row_lse_topk: per row of logits [R][ld] (dtype): lse over V and the top-k (k <= 64) of
 *       processed log-probs (forced_token >= 0: everything -inf except forced_token := 0; min-length: eos := -inf
 *       when suppress_eos), plus row_bias[row] (the beam's running score, gen:857) — candidates ordered
 *       (value desc, index asc), i.e. lax.top_k (gen:850-873).  raw_logits = 1: no log-softmax (greedy, gen:497-499).
 *   mic_beam_step: the whole bookkeeping of one beam_search_body_fn iteration (gen:857-966) per batch item.
 *   mic_greedy_step: argmax + EOS->PAD substitution + append (gen:499-512).
 * ------------------------------------------------------------------------------------------- */
int mic_row_lse_topk(int dtype, int R, int V, const void* logits, int ld, int k, int forced_token,
                     int suppress_eos, int eos_token_id, int raw_logits, const float* row_bias, float* top_val,
                     int32_t* top_idx, void* stream);

/* mic_row_lse_topk's results (no forced token) from the head GEMM's per-granule partials: lse by merging ceil(V / 64) pairs, the
 * top-k by scanning only the 64-column granules whose maximum reaches the k-th largest granule maximum (gen:850-873 without
 * streaming the row) */
int mic_row_topk_tiles(int dtype, int R, int V, const void* logits, int ld, const float* rowstat, int stat_ld, int k,
                       int suppress_eos, int eos_token_id, int raw_logits, const float* row_bias, float* top_val,
                       int32_t* top_idx, void* stream);

typedef struct {
  int B, K, max_len, V;
  int cur_len;                /* tokens already in running_sequences */
  int eos_token_id, pad_token_id;
  float length_penalty; int early_stopping;
  const float* cand_val;      /* [B*K][2K] per-row top candidates: processed log-prob + running score (row_bias) */
  const int32_t* cand_idx;    /* [B*K][2K] */
  int32_t* running_seq;       /* [B][K][max_len] in/out */
  float* running_scores;      /* [B][K] in/out */
  int32_t* seq;               /* [B][K][max_len] finished, in/out */
  float* scores;              /* [B][K] in/out */
  int32_t* finished;          /* [B][K] in/out (0/1) */
  int32_t* src_row;           /* [B*K][max_len] in/out: KV-cache slot ownership (beam-parent indirection) */
  int32_t* next_token;        /* [B*K] out: token each running beam feeds next step */
  int32_t* flags;             /* [B][2] out: per item {all finished, improvement impossible} for the loop cond */
  int32_t* gstate;            /* optional [8], zeroed by the caller before the first step: the loop condition of gen:798-820 kept on
                                 the device — [3] = search has ended (later launches are no-ops), [4] = steps taken; lets the host
                                 enqueue decoder steps ahead instead of synchronising after each one */
} mic_beam_step_args;
int mic_beam_step(const mic_beam_step_args* a, void* stream);

int mic_greedy_step(int B, int max_len, int cur_len, int eos_token_id, int pad_token_id, const int32_t* top_idx,
                    int ld_top, int32_t* sequences, int32_t* finished, int32_t* next_token, void* stream);

/* mic_sample_rows: one draw per row of `jax.random.categorical(key, logits[R][V])` (gen:625-627 inside `_sample`,
 *   gen:537-663) = argmax(logits / temperature + Gumbel noise) with the noise bit-compatible with jax 0.2.16's
 *   threefry2x32 stream for key = (key0, key1) over the whole [R, V] array (counter layout in csrc/decode.hip).
 *   forced_token >= 0: every row draws that token (ForcedBOS/ForcedEOS); suppress_eos: logit[eos] := -inf (MinLength);
 *   min_keep / tie_limit (optional, [R], from mic_warp_thresholds): keep v > min_keep[row], or v == min_keep[row] and
 *   index < tie_limit[row]; everything else is masked.
 * mic_warp_thresholds: FlaxTopKLogitsWarper / FlaxTopPLogitsWarper (gen:338-366; temperature and MinLength applied first)
 *   as a per-row (threshold value, tie index limit) pair: top_k <= 0 disables top-k, top_p >= 1 disables top-p. */
int mic_sample_rows(int dtype, int R, int V, const void* logits, int ld, uint32_t key0, uint32_t key1,
                    float temperature, int forced_token, int suppress_eos, int eos_token_id, const float* min_keep,
                    const int32_t* tie_limit, int32_t* out_idx, void* stream);
int mic_warp_thresholds(int dtype, int R, int V, const void* logits, int ld, float temperature, int suppress_eos,
                        int eos_token_id, int top_k, float top_p, float* thr, int32_t* tie_limit, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Input pipeline: the reference's image Transform (main.py:165-179; evaluation.py:35-60) + the
 * NCHW->NHWC permute of collate_fn (main.py:494) for a batch of uint8 images of arbitrary sizes:
 * Resize([S], bicubic, shorter side) -> CenterCrop(S) -> /255 -> (x - mean) / std.  `items` is a HOST array (the
 * per-image sizes are host data); `src` pointers are device pointers to CHW (hwc = 0, what read_image yields) or HWC
 * (hwc = 1) bytes.  dst: float32 [n][S][S][3] (dst_chw = 0, what the model consumes) or [n][3][S][S] (dst_chw = 1,
 * what Transform.forward returns).
 * ------------------------------------------------------------------------------------------- */
typedef struct { const void* src; int H, W, hwc; } mic_image_item;
int mic_image_transform(const mic_image_item* items, int n, int out_size, const float* mean, const float* std,
                        float* dst, int dst_chw, void* stream);

#ifdef __cplusplus
}
#endif
#endif
