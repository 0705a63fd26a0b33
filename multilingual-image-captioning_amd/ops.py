"""Thin torch-tensor wrappers over the C ABI (include/mic_hip.h).  Tensors give device memory and the current
HIP stream; all arithmetic happens in libmic_hip.so."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib as L


_FP8 = {torch.float8_e4m3fn: L.MIC_E4M3, torch.float8_e5m2: L.MIC_E5M2}


def _dt(t_or_dtype) -> int:
    d = t_or_dtype.dtype if isinstance(t_or_dtype, torch.Tensor) else t_or_dtype
    if d == torch.bfloat16:
        return L.MIC_BF16
    if d == torch.float32:
        return L.MIC_F32
    if d in _FP8:
        return L.MIC_FP8
    raise L.MicError(f"unsupported dtype {d}")


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if not t.is_cuda:
        raise L.MicError("libmic_hip ops need device tensors (no CPU fallback)")
    return t.data_ptr()


_STREAM_OVERRIDE = None


def _stream():
    """HIP stream every kernel is launched on: torch's current stream (looked up per call: ~8 us), or the handle pinned by
    `pinned_stream()` for the duration of a step."""
    if _STREAM_OVERRIDE is not None:
        return _STREAM_OVERRIDE
    return torch.cuda.current_stream().cuda_stream


class pinned_stream:
    """Context manager: resolve torch's current stream once and reuse the raw handle for every launch inside."""

    def __enter__(self):
        global _STREAM_OVERRIDE
        self.prev = _STREAM_OVERRIDE
        _STREAM_OVERRIDE = torch.cuda.current_stream().cuda_stream
        return self

    def __exit__(self, *a):
        global _STREAM_OVERRIDE
        _STREAM_OVERRIDE = self.prev


def gemm_args(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor, M: int, N: int, K: int, *, a_kmajor=False, b_kmajor=False,
              bias=None, act=0, zout=None, zin=None, dact=0, residual=None, accumulate=False, dropout_p=0.0, dropout_seed=0,
              alpha=1.0, lda=None, ldb=None, ldc=None, ldz=None, ldr=None, split_k=0, split_stride=0, a_rowsum=None, rowsum_k=0,
              a_scale_inv=None, b_scale_inv=None, rowstat=None, rowstat_nvalid=0, ln_stats=None, ln_colsum=None, ln_width=0,
              ln_eps=0.0, rowsum2=None, k_valid=0, c_q8=None) -> "L.GemmArgs":
    """fp8 operands: `a` / `b` are torch.float8_e4m3fn / float8_e5m2 tensors (both k-contiguous, or both k-major: the weight-gradient
    form), `a_scale_inv` / `b_scale_inv` the device scalars their quantiser / producer wrote.  c_q8 = (state fp32 [2], amax_next or
    None): `out` is an fp8 tensor the epilogue fills under that tensor's delayed scale (fused emission, see `fp8_out`)."""
    g = L.GemmArgs()
    g.dtype, g.c_dtype = _dt(a), _dt(out)
    if g.dtype == L.MIC_FP8:
        g.a_fmt, g.b_fmt = _FP8[a.dtype], _FP8[b.dtype]
        g.a_scale_inv, g.b_scale_inv = _p(a_scale_inv), _p(b_scale_inv)
    g.M, g.N, g.K = M, N, K
    g.a_kmajor, g.b_kmajor = int(a_kmajor), int(b_kmajor)
    g.A, g.lda = _p(a), lda if lda is not None else a.stride(0)
    g.B, g.ldb = _p(b), ldb if ldb is not None else b.stride(0)
    g.C, g.ldc = _p(out), ldc if ldc is not None else out.stride(0)
    g.bias, g.act = _p(bias), act
    zz = zout if zout is not None else zin
    g.Zout, g.Zin, g.dact = _p(zout), _p(zin), dact
    g.ldz = ldz if ldz is not None else (zz.stride(0) if zz is not None else 0)
    g.R, g.ldr = _p(residual), ldr if ldr is not None else (residual.stride(0) if residual is not None else 0)
    g.accumulate, g.dropout_p, g.dropout_seed, g.alpha = int(accumulate), float(dropout_p), int(dropout_seed) & 0xFFFFFFFF, float(alpha)
    g.split_k, g.split_stride = int(split_k), int(split_stride)
    g.a_rowsum, g.rowsum_k = _p(a_rowsum), int(rowsum_k)
    if rowstat is not None:  # fp32 [rows][tiles][2]: softmax partials per 256-column tile (LM head)
        g.rowstat, g.rowstat_ld, g.rowstat_nvalid = _p(rowstat), rowstat.stride(0) // 2, int(rowstat_nvalid)
    if ln_stats is not None:  # LayerNorm folded around the GEMM: b = gamma o W, bias = bias' (ln_fold_weight), stats of the A rows
        g.a_ln_stats, g.a_ln_colsum, g.a_ln_width, g.a_ln_eps = _p(ln_stats), _p(ln_colsum), int(ln_width), float(ln_eps)
    g.rowsum2 = _p(rowsum2)  # int64 [M][2]: (sum, sum of squares) x 2^20 of the stored output rows, accumulated
    g.k_valid = int(k_valid)  # k-major x k-major launches: operand rows k >= k_valid count as zero (0 = all)
    if c_q8 is not None:
        if g.c_dtype != L.MIC_FP8:
            raise L.MicError("c_q8 goes with an fp8 output tensor")
        g.c_q8_state, g.c_q8_amax, g.c_q8_fmt = _p(c_q8[0]), _p(c_q8[1]), _FP8[out.dtype]
    return g


def gemm(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor, M: int, N: int, K: int, **kw):
    """out[M,N] = epi(op(a) @ op(b)); a: [M,K] (or [K,M] if a_kmajor); b: [N,K] (or [K,N] if b_kmajor)."""
    g = gemm_args(a, b, out, M, N, K, **kw)
    L.check(L.lib().mic_gemm(C.byref(g), _stream()), "mic_gemm")
    return out


def gemm_grouped(arg_list):
    """Several GEMMs (same dtype / operand layouts) in as few launches as possible."""
    arr = (L.GemmArgs * len(arg_list))(*arg_list)
    L.check(L.lib().mic_gemm_grouped(arr, len(arg_list), _stream()), "mic_gemm_grouped")


def set_cu_budget(cus: int) -> None:
    """CUs the GEMM tile planner may count on (0 = default: all 256, or MIC_FREE_CUS); see mic_set_cu_budget"""
    L.check(L.lib().mic_set_cu_budget(int(cus)), "mic_set_cu_budget")


def get_cu_budget() -> int:
    return int(L.lib().mic_get_cu_budget())


def gemm_plan(shapes, *, a_kmajor=False, b_kmajor=False, split_k=0, dtype=None) -> dict:
    """What the planner would launch for the problems [(M, N, K), ...] of one grouped bf16 launch under the current CU budget
    (host arithmetic; nothing runs and no device memory is needed)."""
    arr = (L.GemmArgs * len(shapes))()
    for g, (M, N, K) in zip(arr, shapes):
        g.dtype = g.c_dtype = L.MIC_BF16 if dtype is None else dtype
        g.M, g.N, g.K, g.a_kmajor, g.b_kmajor, g.split_k = M, N, K, int(a_kmajor), int(b_kmajor), int(split_k)
    out = L.GemmPlanInfo()
    L.check(L.lib().mic_gemm_plan(arr, len(shapes), C.byref(out)), "mic_gemm_plan")
    return {k: getattr(out, k) for k, _ in L.GemmPlanInfo._fields_}


def comm_emulate(src: torch.Tensor, dst: torch.Tensor, nbytes: int, micros: float, blocks: int):
    """bench.py --emulate-comm: occupy the current (CU-masked) stream like an all-reduce of `nbytes` projected to take `micros`"""
    L.check(L.lib().mic_comm_emulate(_p(src), _p(dst), int(nbytes), float(micros), int(blocks), _stream()), "mic_comm_emulate")


def ln_fold_weight(w, gamma, beta, bias, w_fold, colsum, bias_fold):
    """w [N][K] -> w_fold = round(w * gamma), colsum[n] = sum_k w_fold[n][k], bias_fold = bias + w . beta (see gemm_args ln_*)"""
    N, K = w.shape
    L.check(L.lib().mic_ln_fold_weight(_dt(w), N, K, _p(w), w.stride(0), _p(gamma), _p(beta), _p(bias), _p(w_fold), w_fold.stride(0),
                                       _p(colsum), _p(bias_fold), _stream()), "mic_ln_fold_weight")


def fp8_item(src, rows, cols, state, fmt_dtype, q=None, qT=None, rows_pad=0, amax_next=None) -> "L.Fp8Item":
    """src bf16 [>=rows][ld]; q fp8 [rows][*] and / or qT fp8 [cols][>= rows_pad]; state fp32 [2] (amax, 1/scale);
    amax_next fp32 [fp8_amax_partials()]: delayed scaling (quantise with the scale in `state`, record this pass's partial maxima)."""
    if src.dtype != torch.bfloat16:
        raise L.MicError("fp8 quantisation reads bf16 tensors (the fp8 path needs the bfloat16 storage mode)")
    it = L.Fp8Item()
    it.src, it.ld, it.rows, it.cols, it.rows_pad = _p(src), src.stride(0), rows, cols, rows_pad
    it.q, it.ldq = _p(q), (q.stride(0) if q is not None else 0)
    it.qT, it.ldqT = _p(qT), (qT.stride(0) if qT is not None else 0)
    it.state, it.amax_next, it.fmt = _p(state), _p(amax_next), _FP8[fmt_dtype]
    return it


def fp8_quantize(items, amax_pass: bool = True):
    """amax pass + quantise pass over `items` (list of Fp8Item; their `state` slots must have been zeroed).
    amax_pass=False: delayed scaling — the scales are already in the items' `state`, only the quantiser runs."""
    arr = (L.Fp8Item * len(items))(*items)
    if amax_pass:
        L.check(L.lib().mic_fp8_amax(arr, len(items), _stream()), "mic_fp8_amax")
    L.check(L.lib().mic_fp8_quantize(arr, len(items), _stream()), "mic_fp8_quantize")


def fp8_amax_partials() -> int:
    """entries of a tensor's partial-maximum table (delayed scaling)"""
    return int(L.lib().mic_fp8_amax_partials())


def fp8_roll_amax(state: torch.Tensor, partials: torch.Tensor, count: int):
    """state fp32 [slots][>= 1] (amax first), partials fp32 [slots][fp8_amax_partials()]: start of a pass under delayed scaling."""
    L.check(L.lib().mic_fp8_roll_amax(_p(state), state.stride(0), _p(partials), count, _stream()), "mic_fp8_roll_amax")


def fp8_out(q, state, amax_next=None) -> "L.Fp8Out":
    """descriptor of a fused fp8 emission: q fp8 [rows][ld] (its dtype gives the format), state fp32 [2] (amax of the previous pass
    read, 1 / scale written), amax_next fp32 [fp8_amax_partials()] (this pass's partial maxima) or None"""
    o = L.Fp8Out()
    o.q, o.ldq, o.state, o.amax_next, o.fmt = _p(q), q.stride(0), _p(state), _p(amax_next), _FP8[q.dtype]
    return o


def layernorm_fwd(x, gamma, beta, eps, y, mean=None, rstd=None, rows=None, dropout_p=0.0, dropout_seed=0, q8=None):
    """q8 (an `fp8_out`, bf16 storage): the normalised rows also / only (y None) as fp8 bytes under the tensor's delayed scale"""
    rows = rows if rows is not None else x.shape[0]
    if q8 is not None:
        L.check(L.lib().mic_layernorm_fwd_q8(rows, x.shape[-1], _p(x), _p(gamma), _p(beta), float(eps), _p(y), _p(mean), _p(rstd),
                                             float(dropout_p), int(dropout_seed) & 0xFFFFFFFF, C.byref(q8), _stream()), "mic_layernorm_fwd_q8")
        return y
    L.check(L.lib().mic_layernorm_fwd(_dt(x), rows, x.shape[-1], _p(x), _p(gamma), _p(beta), float(eps), _p(y), _p(mean), _p(rstd),
                                      float(dropout_p), int(dropout_seed) & 0xFFFFFFFF, _stream()), "mic_layernorm_fwd")
    return y


def layernorm_bwd(x, gamma, mean, rstd, dy, dx, dgamma, dbeta, rows=None, dres=None, dxm=None, dropout_p=0.0, dropout_seed=0,
                  in_dropout_p=0.0, in_dropout_seed=0):
    rows = rows if rows is not None else x.shape[0]
    L.check(L.lib().mic_layernorm_bwd(_dt(x), rows, x.shape[-1], _p(x), _p(gamma), _p(mean), _p(rstd), _p(dy), _p(dres), _p(dx),
                                      _p(dgamma), _p(dbeta), _p(dxm), float(dropout_p), int(dropout_seed) & 0xFFFFFFFF,
                                      float(in_dropout_p), int(in_dropout_seed) & 0xFFFFFFFF, _stream()), "mic_layernorm_bwd")
    return dx


def layernorm_bwd_blocks(rows: int) -> int:
    """blocks mic_layernorm_bwd launches for `rows` rows = rows of its partials buffer per (gamma | beta)"""
    return int(L.lib().mic_layernorm_bwd_blocks(int(rows)))


def layernorm_bwd_partials(x, gamma, mean, rstd, dy, dx, partials, rows=None, dres=None, dxm=None, dropout_p=0.0, dropout_seed=0,
                           in_dropout_p=0.0, in_dropout_seed=0, q8=None, q8_of_dx=False):
    """layernorm_bwd whose gamma / beta gradients land as per-block partial sums in `partials` (fp32 [2][blocks][width], overwritten)
    for a later `ln_param_grads` instead of atomics.  q8 (an `fp8_out`): the masked gradient dxm (q8_of_dx: dx itself) also as fp8"""
    rows = rows if rows is not None else x.shape[0]
    if q8 is not None:
        L.check(L.lib().mic_layernorm_bwd_partials_q8(rows, x.shape[-1], _p(x), _p(gamma), _p(mean), _p(rstd), _p(dy), _p(dres), _p(dx), _p(partials),
                                                      _p(dxm), float(dropout_p), int(dropout_seed) & 0xFFFFFFFF, float(in_dropout_p),
                                                      int(in_dropout_seed) & 0xFFFFFFFF, C.byref(q8), int(bool(q8_of_dx)), _stream()),
                "mic_layernorm_bwd_partials_q8")
        return dx
    L.check(L.lib().mic_layernorm_bwd_partials(_dt(x), rows, x.shape[-1], _p(x), _p(gamma), _p(mean), _p(rstd), _p(dy), _p(dres), _p(dx),
                                               _p(partials), _p(dxm), float(dropout_p), int(dropout_seed) & 0xFFFFFFFF,
                                               float(in_dropout_p), int(in_dropout_seed) & 0xFFFFFFFF, _stream()), "mic_layernorm_bwd_partials")
    return dx


def ln_param_grads(items):
    """items: [(partials, nblk, width, dgamma, dbeta, accumulate)] -> dgamma / dbeta (+)= sum over the blocks' partials (one grouped launch
    per 8 items)"""
    arr = (L.LnParamItem * len(items))()
    for a, (part, nblk, width, dg, db, acc) in zip(arr, items):
        a.partials, a.dgamma, a.dbeta, a.nblk, a.width, a.accumulate = _p(part), _p(dg), _p(db), int(nblk), int(width), int(bool(acc))
    L.check(L.lib().mic_ln_param_grads(arr, len(items), _stream()), "mic_ln_param_grads")


def attn_fwd(q, k, v, out, B, H, Tq, Tk, *, ldq, ldk, ldv, ldo, key_mask=None, causal=False, lse=None):
    L.check(L.lib().mic_attn_fwd(_dt(q), B, H, Tq, Tk, _p(q), ldq, _p(k), ldk, _p(v), ldv, _p(out), ldo, _p(key_mask), int(causal),
                                 _p(lse), _stream()), "mic_attn_fwd")
    return out


def attn_probs(q, k, out, B, H, Tq, Tk, *, ldq, ldk, key_mask=None, causal=False):
    """out [B,H,Tq,Tk] fp32 = the attention weights of q / k (views into the saved projection buffers); diagnostic only"""
    L.check(L.lib().mic_attn_probs(_dt(q), B, H, Tq, Tk, _p(q), ldq, _p(k), ldk, _p(key_mask), int(causal), _p(out), _stream()), "mic_attn_probs")
    return out


def attn_bwd(q, k, v, out, dout, lse, dq, dk, dv, B, H, Tq, Tk, *, ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv, key_mask=None,
             causal=False):
    L.check(L.lib().mic_attn_bwd(_dt(q), B, H, Tq, Tk, _p(q), ldq, _p(k), ldk, _p(v), ldv, _p(out), ldo, _p(dout), lddo, _p(lse),
                                 _p(key_mask), int(causal), _p(dq), lddq, _p(dk), lddk, _p(dv), lddv, _stream()), "mic_attn_bwd")


def attn_fwd_packed(q, k, v, out, B, H, Tq_max, Tk, q_off, q_len, *, kv_packed, ldq, ldk, ldv, ldo, causal=False, lse=None):
    """attention cores on packed (variable-length) rows: sequence b = rows [q_off[b], q_off[b] + q_len[b])"""
    L.check(L.lib().mic_attn_fwd_packed(_dt(q), B, H, Tq_max, Tk, _p(q_off), _p(q_len), int(kv_packed), _p(q), ldq, _p(k), ldk, _p(v), ldv,
                                        _p(out), ldo, int(causal), _p(lse), _stream()), "mic_attn_fwd_packed")
    return out


def attn_bwd_packed(q, k, v, out, dout, lse, dq, dk, dv, B, H, Tq_max, Tk, q_off, q_len, *, kv_packed, ldq, ldk, ldv, ldo, lddo, lddq,
                    lddk, lddv, causal=False):
    L.check(L.lib().mic_attn_bwd_packed(_dt(q), B, H, Tq_max, Tk, _p(q_off), _p(q_len), int(kv_packed), _p(q), ldq, _p(k), ldk, _p(v), ldv,
                                        _p(out), ldo, _p(dout), lddo, _p(lse), int(causal), _p(dq), lddq, _p(dk), lddk, _p(dv), lddv,
                                        _stream()), "mic_attn_bwd_packed")


def attn_bwd_q8(q, k, v, out, dout, lse, dq8, dk8, dv8, B, H, Tq, Tk, *, ldq, ldk, ldv, ldo, lddo, q_off=None, q_len=None, kv_packed=False,
                key_mask=None, causal=False):
    """attention backward (bf16, one 64x64 tile per sequence; dense rows, or packed rows with q_off / q_len) whose dQ / dK / dV leave
    as fp8 bytes: dq8 / dk8 are `fp8_out` descriptors of dQ's and dK's byte matrices, dv8 the fp8 tensor (view) of dV (same leading
    dimension, scale and amax table as dK)"""
    L.check(L.lib().mic_attn_bwd_q8(B, H, Tq, Tk, _p(q_off), _p(q_len), int(bool(kv_packed)), _p(q), ldq, _p(k), ldk, _p(v), ldv, _p(out), ldo,
                                    _p(dout), lddo, _p(lse), _p(key_mask), int(causal), C.byref(dq8), C.byref(dk8), _p(dv8), _stream()), "mic_attn_bwd_q8")


def colsum_q8_grouped(items):
    """items: [(x fp8 [rows][ld], out fp32 [cols], scale_inv fp32 [1], rows, cols)]: out += scale_inv * column sums of the fp8 values"""
    arr = (L.ColsumQ8Item * len(items))()
    for a, (x, out, sinv, rows, cols) in zip(arr, items):
        a.x, a.out, a.scale_inv, a.rows, a.cols, a.ld, a.fmt = _p(x), _p(out), _p(sinv), int(rows), int(cols), x.stride(0), _FP8[x.dtype]
    L.check(L.lib().mic_colsum_q8_grouped(arr, len(items), _stream()), "mic_colsum_q8_grouped")


def attn_decode(q, kc, vc, out, R, H, max_len, cur, *, ldq, ldo, ldc=None, src_row=None, row_div=1):
    ldc = ldc if ldc is not None else H * 64
    L.check(L.lib().mic_attn_decode(_dt(q), R, H, max_len, cur, _p(q), ldq, _p(kc), _p(vc), ldc, _p(src_row), row_div, _p(out), ldo,
                                    _stream()), "mic_attn_decode")
    return out


def kv_append(k, v, kc, vc, R, HD, max_len, cur, *, ldk, ldv):
    L.check(L.lib().mic_kv_append(_dt(k), R, HD, max_len, cur, _p(k), ldk, _p(v), ldv, _p(kc), _p(vc), _stream()), "mic_kv_append")


def im2col(pixels, patches, B, img, ps, ldp, trunc_int32=False):
    L.check(L.lib().mic_im2col(_dt(patches), B, img, ps, _p(pixels), _p(patches), ldp, int(trunc_int32), _stream()), "mic_im2col")


def vit_assemble(patch_out, cls, pos, x, B, S, width, ldp):
    L.check(L.lib().mic_vit_assemble(_dt(x), B, S, width, _p(patch_out), ldp, _p(cls), _p(pos), _p(x), _stream()), "mic_vit_assemble")


def vit_assemble_bwd(dx, dpatch, dcls, dpos, B, S, width, ldp):
    L.check(L.lib().mic_vit_assemble_bwd(_dt(dx), B, S, width, _p(dx), _p(dpatch), ldp, _p(dcls), _p(dpos), _stream()), "mic_vit_assemble_bwd")


def embed_fwd(ids, pos_ids, table, pos_table, scale, h, rows, width):
    L.check(L.lib().mic_embed_fwd(_dt(h), rows, width, _p(ids), _p(pos_ids), _p(table), _p(pos_table), float(scale), _p(h), _stream()), "mic_embed_fwd")


def embed_bwd(ids, pos_ids, dh, scale, dtable, dpos_table, rows, width):
    L.check(L.lib().mic_embed_bwd(_dt(dh), rows, width, _p(ids), _p(pos_ids), _p(dh), float(scale), _p(dtable), _p(dpos_table), _stream()), "mic_embed_bwd")


def embed_rows_add_det_workspace(vocab: int, device) -> torch.Tensor:
    """the int32 workspace of `embed_rows_add_det` for a table of `vocab` rows, initialised (the calls keep it in that state)"""
    n = int(L.lib().mic_embed_rows_add_det_ws(int(vocab)))
    ws = torch.zeros(n, dtype=torch.int32, device=device)
    ws[:vocab] = 0x7FFFFFFF
    ws[2 * vocab: 3 * vocab] = -1
    return ws


def embed_rows_add_det(ids, dh, scale, dtable, n, width, ws):
    """dtable[ids[i]] += scale * dh[i] over the n rows, deterministically (ids < 0 skipped): the data-parallel embedding-row exchange"""
    L.check(L.lib().mic_embed_rows_add_det(_dt(dh), int(n), int(width), int(dtable.shape[0]) if dtable.dim() == 2 else int(dtable.numel() // width),
                                           _p(ids), _p(dh), float(scale), _p(dtable), _p(ws), _stream()), "mic_embed_rows_add_det")


def ce_rows(logits, ld, V, labels, mask, ls, row_lse, row_loss, rows):
    L.check(L.lib().mic_ce_rows(_dt(logits), rows, V, _p(logits), ld, _p(labels), _p(mask), float(ls), _p(row_lse), _p(row_loss), _stream()), "mic_ce_rows")


def ce_rows_tiles(logits, ld, V, rowstat, labels, row_lse, row_loss, rows):
    L.check(L.lib().mic_ce_rows_tiles(_dt(logits), rows, V, _p(logits), ld, _p(rowstat), rowstat.stride(0) // 2, _p(labels), _p(row_lse),
                                      _p(row_loss), _stream()), "mic_ce_rows_tiles")


def row_topk_tiles(logits, ld, V, rowstat, k, top_val, top_idx, R, *, suppress_eos=False, eos_token_id=2, raw_logits=False, row_bias=None):
    L.check(L.lib().mic_row_topk_tiles(_dt(logits), R, V, _p(logits), ld, _p(rowstat), rowstat.stride(0) // 2, k, int(suppress_eos),
                                       eos_token_id, int(raw_logits), _p(row_bias), _p(top_val), _p(top_idx), _stream()), "mic_row_topk_tiles")


def ce_reduce(row_loss, mask, loss_out, denom_out, rows):
    L.check(L.lib().mic_ce_reduce(rows, _p(row_loss), _p(mask), _p(loss_out), _p(denom_out), _stream()), "mic_ce_reduce")


def ce_bwd(logits, ld, V, Vpad, labels, mask, ls, row_lse, denom, rows, loss_scale=1.0):
    L.check(L.lib().mic_ce_bwd(_dt(logits), rows, V, Vpad, _p(logits), ld, _p(labels), _p(mask), float(ls), _p(row_lse), _p(denom),
                               float(loss_scale), _stream()), "mic_ce_bwd")


def ce_bwd_q8(logits, ld, V, Vpad, labels, mask, ls, row_lse, denom, rows, q8, colsum=None, label_coef=None, loss_scale=1.0):
    """CE backward of bf16 logits with dlogits leaving as fp8 bytes (q8: an `fp8_out`; its state[1] receives the closed-form
    dequantisation factor loss_scale / (denom * FMAX)) instead of in place; column sums of the gradient added to `colsum`;
    label_coef fp32 [rows]: the label entries leave as fp32 coefficients instead (zero bytes in the matrix) (mic_ce_bwd_q8)"""
    L.check(L.lib().mic_ce_bwd_q8(rows, V, Vpad, _p(logits), ld, _p(labels), _p(mask), float(ls), _p(row_lse), _p(denom), float(loss_scale),
                                  C.byref(q8), _p(colsum), _p(label_coef), _stream()), "mic_ce_bwd_q8")


def head_label_terms(labels, coef, E, h, dx_slab, dE, rows, width):
    """dx_slab[m] = coef[m] * E[labels[m]] (fp32), dE[labels[m]] += coef[m] * h[m] (fp32 atomics): the label entries of dlogits that
    mic_ce_bwd_q8 kept out of the fp8 matrix (mic_head_label_terms)"""
    L.check(L.lib().mic_head_label_terms(rows, width, _p(labels), _p(coef), _p(E), E.stride(0), _p(h), h.stride(0), _p(dx_slab), dx_slab.stride(0),
                                         _p(dE), dE.stride(0), _stream()), "mic_head_label_terms")


def ce_bwd_t(logits, ld, V, Vpad, labels, mask, ls, row_lse, denom, rows, dlogits_t, rows_pad=0, colsum=None, loss_scale=1.0):
    """ce_bwd on bf16 logits that also writes dlogits^T [Vpad][ld_t] (columns rows .. rows_pad zero) and adds the column sums of the
    stored gradient to `colsum` (mic_ce_bwd_t)"""
    L.check(L.lib().mic_ce_bwd_t(rows, V, Vpad, _p(logits), ld, _p(labels), _p(mask), float(ls), _p(row_lse), _p(denom), float(loss_scale),
                                 _p(dlogits_t), dlogits_t.stride(0), int(rows_pad), _p(colsum), _stream()), "mic_ce_bwd_t")


def transpose_bf16(src, dst, rows, cols, rows_pad=0):
    """dst[c][r] = src[r][c] for the leading rows x cols of a bf16 matrix; dst's columns rows .. rows_pad (0: rows rounded up to 64) are zeroed"""
    L.check(L.lib().mic_transpose_bf16(int(rows), int(rows_pad), int(cols), _p(src), src.stride(0), _p(dst), dst.stride(0), _stream()),
            "mic_transpose_bf16")
    return dst


def colsum(x, out, rows, cols, ld, accumulate=False):
    L.check(L.lib().mic_colsum(_dt(x), rows, cols, _p(x), ld, _p(out), int(accumulate), _stream()), "mic_colsum")


def colsum_grouped(items):
    """items: [(x, out, rows, cols, ld)]: all accumulate into pre-zeroed fp32 outputs, one launch per 8 items."""
    arr = (L.ColsumItem * len(items))(*[L.ColsumItem(_p(x), _p(out), rows, cols, ld) for (x, out, rows, cols, ld) in items])
    L.check(L.lib().mic_colsum_grouped(_dt(items[0][0]), arr, len(items), _stream()), "mic_colsum_grouped")


def dropout_mask(n: int, p: float, seed: int, device) -> torch.Tensor:
    out = torch.empty(n, dtype=torch.uint8, device=device)
    L.check(L.lib().mic_dropout_mask(_p(out), n, float(p), int(seed) & 0xFFFFFFFF, _stream()), "mic_dropout_mask")
    return out


def cast(src, dst, n=None):
    n = n if n is not None else src.numel()
    L.check(L.lib().mic_cast(_dt(src), _dt(dst), _p(src), _p(dst), n, _stream()), "mic_cast")
    return dst


def zero(t: torch.Tensor):
    """zero-fill a contiguous device tensor (or contiguous slice) on the launch stream"""
    if t.numel() == 0:
        return t
    if not t.is_contiguous():
        raise L.MicError("ops.zero needs a contiguous tensor")
    L.check(L.lib().mic_zero(_p(t), t.numel() * t.element_size(), _stream()), "mic_zero")
    return t


def cu_masked_stream(first_cu: int, n_cus: int, device) -> "torch.cuda.Stream":
    """A torch stream over a HIP stream that may use only bits [first_cu, first_cu + n_cus) of the CU mask (mic_stream_create_cu_masked).  The HIP
    stream lives as long as the process and is shared by every caller that asks for the same range."""
    import ctypes

    dev = torch.device(device)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), int(first_cu), int(n_cus))
    st = _masked_streams.get(key)
    if st is None:  # one HIP stream per (device, CU range) and process: every masked stream is a hardware queue of its own, and a
        h = ctypes.c_void_p()  # process that keeps creating them (a Trainer per test, say) ends up time-slicing queues
        with torch.cuda.device(dev):
            L.check(L.lib().mic_stream_create_cu_masked(int(first_cu), int(n_cus), ctypes.byref(h)), "mic_stream_create_cu_masked")
        st = _masked_streams[key] = torch.cuda.ExternalStream(h.value, device=dev)
    return st


_masked_streams = {}
_role_streams = {}


def role_stream(device, role: str) -> "torch.cuda.Stream":
    """One side stream per (device, role) and process ("collective", "optimizer", "step", "tail", "dw"), shared by every reducer /
    engine: HIP multiplexes streams onto a handful of hardware queues and streams that share a queue serialise, so a process should
    not keep minting side streams (a Trainer per test, bench.py's legs).  How streams map onto queues depends on the process's whole
    stream history: even with shared role streams a SECOND Trainer in one process measured 2.5-6 ms more per emulated-exchange step
    than the same schedule run by the process's only Trainer (and +10 ms under GPU_MAX_HW_QUEUES=8) — bench.py runs those legs in fresh
    child processes, one Trainer each, like a data-parallel rank."""
    dev = torch.device(device)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), role)
    st = _role_streams.get(key)
    if st is None:
        # MIC_PRIO_<ROLE> (A/B only): HIP stream priority of that role's stream (-1 = high, 0 = default)
        import os

        prio = int(os.environ.get("MIC_PRIO_" + role.upper(), "0"))
        st = _role_streams[key] = torch.cuda.Stream(device=dev, priority=prio) if prio else torch.cuda.Stream(device=dev)
    return st


def sum_slabs(src, n_slabs, slab_stride, dst, rows, cols, ld_src, ld_dst):
    L.check(L.lib().mic_sum_slabs(_dt(dst), n_slabs, int(slab_stride), rows, cols, _p(src), ld_src, _p(dst), ld_dst, _stream()), "mic_sum_slabs")
    return dst


def cast2d(src, dst, rows, cols, ld_src, ld_dst):
    L.check(L.lib().mic_cast2d(_dt(src), _dt(dst), rows, cols, _p(src), ld_src, _p(dst), ld_dst, _stream()), "mic_cast2d")
    return dst


def copy_rows(src, dst, n, width, *, src_idx=None, dst_idx=None):
    L.check(L.lib().mic_copy_rows(_dt(src), n, width, _p(src), src.stride(0), _p(src_idx), _p(dst), dst.stride(0), _p(dst_idx), _stream()),
            "mic_copy_rows")


def adamw(p, m, v, g, p_lp, hyper, b1, b2, eps, wd, grad_scale=1.0, n=None):
    n = n if n is not None else p.numel()
    L.check(L.lib().mic_adamw(n, _p(p), _p(m), _p(v), _p(g), _p(p_lp), _p(hyper), float(b1), float(b2), float(eps), float(wd),
                              float(grad_scale), _stream()), "mic_adamw")


def adamw_rows(rows, width, row_flag, want, p, m, v, g, p_lp, hyper, b1, b2, eps, wd, grad_scale=1.0):
    """AdamW on the rows of a [rows][width] slice whose flag equals `want` (mic_adamw_rows)."""
    L.check(L.lib().mic_adamw_rows(int(rows), int(width), _p(row_flag), int(want), _p(p), _p(m), _p(v), _p(g), _p(p_lp), _p(hyper), float(b1),
                                   float(b2), float(eps), float(wd), float(grad_scale), _stream()), "mic_adamw_rows")


def row_flags(ids, n_ids, flags):
    """flags[:] = 0; flags[ids[i]] = 1 for i < n_ids (uint8 flags, int32 ids)"""
    L.check(L.lib().mic_row_flags(_p(ids), int(n_ids), _p(flags), flags.numel(), _stream()), "mic_row_flags")


def row_lse_topk(logits, ld, V, k, top_val, top_idx, R, *, forced_token=-1, suppress_eos=False, eos_token_id=2, raw_logits=False,
                 row_bias=None):
    L.check(L.lib().mic_row_lse_topk(_dt(logits), R, V, _p(logits), ld, k, int(forced_token), int(suppress_eos), eos_token_id,
                                     int(raw_logits), _p(row_bias), _p(top_val), _p(top_idx), _stream()), "mic_row_lse_topk")


def beam_step(B, K, max_len, V, cur_len, eos, pad, length_penalty, early_stopping, cand_val, cand_idx, running_seq, running_scores,
              seq, scores, finished, src_row, next_token, flags, gstate=None):
    a = L.BeamStepArgs()
    a.B, a.K, a.max_len, a.V, a.cur_len = B, K, max_len, V, cur_len
    a.eos_token_id, a.pad_token_id, a.length_penalty, a.early_stopping = eos, pad, float(length_penalty), int(bool(early_stopping))
    a.cand_val, a.cand_idx = _p(cand_val), _p(cand_idx)
    a.running_seq, a.running_scores, a.seq, a.scores = _p(running_seq), _p(running_scores), _p(seq), _p(scores)
    a.finished, a.src_row, a.next_token, a.flags = _p(finished), _p(src_row), _p(next_token), _p(flags)
    a.gstate = _p(gstate)
    L.check(L.lib().mic_beam_step(C.byref(a), _stream()), "mic_beam_step")


def greedy_step(B, max_len, cur_len, eos, pad, top_idx, ld_top, sequences, finished, next_token):
    L.check(L.lib().mic_greedy_step(B, max_len, cur_len, eos, pad, _p(top_idx), ld_top, _p(sequences), _p(finished), _p(next_token),
                                    _stream()), "mic_greedy_step")


def sample_rows(logits, ld, V, key, out_idx, R, *, temperature=1.0, forced_token=-1, suppress_eos=False, eos_token_id=2, min_keep=None,
                tie_limit=None):
    L.check(L.lib().mic_sample_rows(_dt(logits), R, V, _p(logits), ld, int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF,
                                    float(temperature), int(forced_token), int(suppress_eos), eos_token_id, _p(min_keep), _p(tie_limit),
                                    _p(out_idx), _stream()), "mic_sample_rows")


def warp_thresholds(logits, ld, V, thr, tie_limit, R, *, temperature=1.0, suppress_eos=False, eos_token_id=2, top_k=0, top_p=1.0):
    L.check(L.lib().mic_warp_thresholds(_dt(logits), R, V, _p(logits), ld, float(temperature), int(suppress_eos), eos_token_id,
                                        int(top_k or 0), float(top_p if top_p is not None else 1.0), _p(thr), _p(tie_limit), _stream()),
            "mic_warp_thresholds")


def image_transform(images, out_size, mean, std, dst, *, chw_out=False):
    """images: list of uint8 device tensors, each [3,H,W] (CHW) or [H,W,3] (HWC); dst float32 [n,S,S,3] (or [n,3,S,S])."""
    import ctypes as C

    n = len(images)
    items = (L.ImageItem * n)()
    for i, im in enumerate(images):
        if im.dtype != torch.uint8 or im.dim() != 3 or not im.is_contiguous():
            raise L.MicError("image_transform: images must be contiguous uint8 [3,H,W] or [H,W,3] tensors")
        hwc = im.shape[-1] == 3 and im.shape[0] != 3
        H, W = (im.shape[0], im.shape[1]) if hwc else (im.shape[1], im.shape[2])
        items[i].src, items[i].H, items[i].W, items[i].hwc = _p(im), H, W, int(hwc)
    m, s = (C.c_float * 3)(*[float(x) for x in mean]), (C.c_float * 3)(*[float(x) for x in std])
    L.check(L.lib().mic_image_transform(items, n, out_size, m, s, _p(dst), int(chw_out), _stream()), "mic_image_transform")
    return dst
