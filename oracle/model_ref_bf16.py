"""ORACLE — TEST INFRASTRUCTURE ONLY.  (parity unpinned, like `oracle.model_ref` whose arithmetic this file re-runs)

bfloat16-STORAGE restatement of the hot path: the same graph as `oracle.model_ref` (same citations), evaluated in fp32 on
the CPU, with every tensor that the bf16 mode keeps in memory rounded to bfloat16 (round-to-nearest-even) at exactly the
point where it is stored — and nowhere else.  That is what the reference's `dtype=jnp.bfloat16` mode means (`main.py:96-101,
425`; `modeling_clip_vision_mbart.py:146-192`: parameters stay fp32, every Dense / LayerNorm / Embed computes and hands on
bf16), and it is what the north-star tolerance "logits within 1e-3 (bf16)" can be asserted against: compared with the plain
fp32 oracle a bf16 logit is off by up to half a bf16 ulp (3.9e-3 relative) from storage rounding alone; compared with THIS
oracle the storage format cancels and what remains is kernel arithmetic (accumulation order, exp / rsqrt approximations).

Storage points (what `multilingual-image-captioning_amd/engine.py` keeps in bf16):
  * compute copies of the Dense / conv / shared-embedding kernels (`ParamStore.w`); biases, LayerNorm scale / bias, class and
    position embeddings and `final_logits_bias` stay fp32 (`ParamStore.f32`);
  * im2col patches, every Linear output (bias added in fp32 before the rounding; an activation acts on the ROUNDED
    pre-activation and its result is rounded again; a residual is added in fp32 before the one rounding of the sum),
    every LayerNorm output (statistics in fp32 from the stored input), the token+position embedding sum, attention contexts;
  * inside the teacher-forced attention core the UN-normalised probabilities exp(s - max) are rounded to bf16 for the P.V
    matrix product while the row sum is taken over the unrounded values (csrc/attention.hip attn_fwd_kernel); the cached
    decode attention works in fp32 on the stored q / k / v (attn_decode_kernel);
  * the LayerNorm-folded decoder step (`modeling_clip_vision_mbart.py::_decode_step`, bf16 generate default): a Linear behind a
    LayerNorm runs on the RAW residual rows with gamma folded into its bf16 weight, `LN(x) W^T = rstd (x (gamma o W)^T - mu g) +
    (b + beta W^T)`, statistics (sum x, sum x^2) from the stored rows — the normalised activations are never rounded.
`forward_logits` / `decode_step` return the logits BEFORE their final rounding (fp32) so that a test can state the kernel
error (|stored logit - bf16(oracle)| up to one rounding flip) separately from the format.

Only `tests/` import this module.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch

from . import model_ref as M

Params = Dict[str, torch.Tensor]
V, D_ = M.V, M.D_


def rb(x: torch.Tensor) -> torch.Tensor:
    """value after a bf16 store + load (round to nearest even)"""
    return x.to(torch.bfloat16).to(torch.float32)


def compute_copy(p: Params) -> Params:
    """the bf16 compute copy of the kernels next to the fp32 leaves the kernels read in fp32 (ParamStore.w / .f32)"""
    out = {}
    for k, v in p.items():
        out[k] = rb(v) if (k.endswith("/kernel") or k == "model/shared/embedding") else v
    return out


def _ln(x, p, name, eps):
    return rb(M.layer_norm(x, p[name + "/scale"], p[name + "/bias"], eps))


ACC64 = False  # True: every Linear / LM-head contraction accumulates in float64 — an fp32-ulp-sized perturbation of each
#                pre-rounding value, used by the tests to measure how far two VALID evaluation orders of the same bf16 arithmetic
#                drift apart through the rounding points of a deep stack (see tests/test_fullsize_gpu.py)


def _mm(x, w):
    return (x.double() @ w.double()).float() if ACC64 else x @ w


def _lin(x, p, name):
    """fp32 accumulate + fp32 bias, NOT yet rounded (the caller rounds once, after residual / at the store)"""
    y = _mm(x, p[name + "/kernel"])
    b = p.get(name + "/bias")
    return y + b if b is not None else y


def attn_train_unrounded(q, k, v, bias):
    """attn_fwd_kernel: s = (q.k) / 8 in fp32; p = exp(s - max) rounded to bf16 for P.V, row sum over the unrounded p; the
    context BEFORE its store"""
    d = q.shape[-1]
    s = torch.einsum("bthd,bshd->bhts", q, k) / math.sqrt(d)
    if bias is not None:
        s = s + bias
    e = torch.exp(s - s.max(dim=-1, keepdim=True).values)
    return torch.einsum("bhts,bshd->bthd", rb(e), v) / e.sum(-1).permute(0, 2, 1)[..., None]


def _attn_train(q, k, v, bias):
    return rb(attn_train_unrounded(q, k, v, bias))


def vit_encoder(cfg: M.RefConfig, pc: Params, pixels_nhwc: torch.Tensor) -> torch.Tensor:
    """engine.vit_forward (bf16): last hidden state [B,S,vd] as stored"""
    B = pixels_nhwc.shape[0]
    ps, g = cfg.patch_size, cfg.image_size // cfg.patch_size
    H, Dh = cfg.v_heads, cfg.v_hidden // cfg.v_heads
    x = rb(pixels_nhwc).reshape(B, g, ps, g, ps, 3).permute(0, 1, 3, 2, 4, 5).reshape(B, g * g, ps * ps * 3)
    pe = rb(_mm(x, pc[V + "embeddings/patch_embedding/kernel"].reshape(ps * ps * 3, cfg.v_hidden)))
    cls = pc[V + "embeddings/class_embedding"].reshape(1, 1, -1).expand(B, 1, cfg.v_hidden)
    x = rb(torch.cat([cls, pe], dim=1) + pc[V + "embeddings/position_embedding/embedding"][None, : cfg.v_seq])
    x = _ln(x, pc, V + "pre_layrnorm", cfg.v_ln_eps)
    S = x.shape[1]
    for i in range(cfg.v_layers):
        L = f"{V}encoder/layers/{i}/"
        a = _ln(x, pc, L + "layer_norm1", cfg.v_ln_eps)
        q = rb(_lin(a, pc, L + "self_attn/q_proj")).reshape(B, S, H, Dh)
        k = rb(_lin(a, pc, L + "self_attn/k_proj")).reshape(B, S, H, Dh)
        v = rb(_lin(a, pc, L + "self_attn/v_proj")).reshape(B, S, H, Dh)
        ctx = _attn_train(q, k, v, None).reshape(B, S, H * Dh)
        xm = rb(_lin(ctx, pc, L + "self_attn/out_proj") + x)
        a = _ln(xm, pc, L + "layer_norm2", cfg.v_ln_eps)
        u = rb(M.quick_gelu(rb(_lin(a, pc, L + "mlp/fc1"))))
        x = rb(_lin(u, pc, L + "mlp/fc2") + xm)
    return x


def encode(cfg: M.RefConfig, pc: Params, pixel_values: torch.Tensor, int32_cast: bool = True) -> torch.Tensor:
    px = pixel_values.to(torch.float32)
    if int32_cast:
        px = torch.trunc(px)
    return rb(_lin(vit_encoder(cfg, pc, px), pc, "model/visual_projection"))


def _embed(cfg, pc, ids, position_ids):
    scale = math.sqrt(cfg.d_model) if cfg.scale_embedding else 1.0
    h = rb(pc["model/shared/embedding"][ids] * scale + pc[D_ + "embed_positions/embedding"][position_ids + 2])
    return _ln(h, pc, D_ + "layernorm_embedding", cfg.decoder_ln_eps)


def decoder_forward(cfg: M.RefConfig, pc: Params, ids, attention_mask, position_ids, ehs) -> torch.Tensor:
    """engine.decoder_forward (bf16, eval): final-layer-normed hidden states as stored"""
    B, T = ids.shape
    H = cfg.d_heads
    eps = cfg.decoder_ln_eps
    causal = torch.tril(torch.ones(T, T, dtype=torch.int32))[None, None]
    bias = M.mask_to_bias(causal * attention_mask.to(torch.int32)[:, None, None, :])
    x = _embed(cfg, pc, ids, position_ids)
    sp = lambda t: M._split(t, H)
    for i in range(cfg.d_layers):
        L = f"{D_}layers/{i}/"
        a = _ln(x, pc, L + "self_attn_layer_norm", eps)
        q, k, v = (sp(rb(_lin(a, pc, L + f"self_attn/{n}_proj"))) for n in ("q", "k", "v"))
        ctx = _attn_train(q, k, v, bias).reshape(x.shape)
        x1 = rb(_lin(ctx, pc, L + "self_attn/out_proj") + x)
        a = _ln(x1, pc, L + "encoder_attn_layer_norm", eps)
        q = sp(rb(_lin(a, pc, L + "encoder_attn/q_proj")))
        k, v = (sp(rb(_lin(ehs, pc, L + f"encoder_attn/{n}_proj"))) for n in ("k", "v"))
        ctx = _attn_train(q, k, v, None).reshape(x.shape)
        x2 = rb(_lin(ctx, pc, L + "encoder_attn/out_proj") + x1)
        a = _ln(x2, pc, L + "final_layer_norm", eps)
        u = rb(M.gelu(rb(_lin(a, pc, L + "fc1")), cfg.gelu))
        x = rb(_lin(u, pc, L + "fc2") + x2)
    return _ln(x, pc, D_ + "layer_norm", eps)


def lm_head(cfg, pc, h):
    """fp32 logits BEFORE the final bf16 rounding of the store"""
    return _mm(h, pc["model/shared/embedding"].T) + pc["final_logits_bias"]


def forward_logits(cfg: M.RefConfig, p: Params, pixel_values, decoder_input_ids, decoder_attention_mask=None) -> torch.Tensor:
    """teacher-forced logits of the bf16 mode, unrounded at the very end (store them with rb() to get the stored values)"""
    pc = compute_copy(p)
    ids = decoder_input_ids.to(torch.int64)
    B, T = ids.shape
    am = torch.ones_like(ids) if decoder_attention_mask is None else decoder_attention_mask
    pos = torch.arange(T)[None].expand(B, T)
    ehs = rb(_lin(vit_encoder(cfg, pc, pixel_values.to(torch.float32)), pc, "model/visual_projection"))
    return lm_head(cfg, pc, decoder_forward(cfg, pc, ids, am, pos, ehs))


# --------------------------------------------------------------------------------------------
# cached decoder step (modeling_clip_vision_mbart.py::_decode_step), explicit LayerNorms or LayerNorm-folded GEMMs
# --------------------------------------------------------------------------------------------
class DecodeState(M.DecodeState):
    pass


def _attn_decode(q, k, v, n_valid):
    """attn_decode_kernel: fp32 softmax(q.k / 8) . v over the first n_valid slots of the stored (bf16) cache; output stored"""
    d = q.shape[-1]
    s = torch.einsum("bthd,bshd->bhts", q / math.sqrt(d), k[:, :n_valid])
    w = torch.softmax(s, dim=-1)
    return rb(torch.einsum("bhts,bshd->bthd", w, v[:, :n_valid]))


def _fold(pc: Params, wname: str, lname: str):
    """mic_ln_fold_weight: (gamma o W rounded to bf16 [in,out], g[n] = sum_k of the ROUNDED products, b' = b + beta . W)"""
    w = pc[wname + "/kernel"]  # [in, out], already the bf16 compute copy
    wf = rb(pc[lname + "/scale"][:, None] * w)
    return wf, wf.sum(0), pc[wname + "/bias"] + pc[lname + "/bias"] @ w


def _ln_folded_lin(x, fold, eps):
    """a_ln_stats epilogue: rstd (x (gamma o W)^T - mu g) + b' with (mu, rstd) from (sum x, sum x^2) of the stored row"""
    wf, g, b = fold
    n = x.shape[-1]
    mu = (x.double().sum(-1, keepdim=True) / n).float()
    var = ((x.double() ** 2).sum(-1, keepdim=True) / n).float() - mu * mu
    rstd = torch.rsqrt(var.clamp_min(0.0) + eps)
    return rstd * (_mm(x, wf) - mu * g) + b


def decode_step(cfg: M.RefConfig, pc: Params, state: DecodeState, ids, position_ids, ehs, ln_fold: bool,
                cross_kv=None) -> torch.Tensor:
    """one cached decoder step of the bf16 mode; `pc` = compute_copy(params); returns UNROUNDED logits [R,1,V].
    ln_fold=True restates the LayerNorm-folded launches (layer 0's first LayerNorm, the embedding and the final LayerNorm stay
    explicit, as in `_decode_step`)."""
    H = cfg.d_heads
    eps = cfg.decoder_ln_eps
    t = state.index
    sp = lambda z: M._split(z, H)
    x = _embed(cfg, pc, ids.to(torch.int64), position_ids.to(torch.int64))
    for i in range(cfg.d_layers):
        L = f"{D_}layers/{i}/"
        if ln_fold and i > 0:
            q, k, v = (rb(_ln_folded_lin(x, _fold(pc, L + f"self_attn/{n}_proj", L + "self_attn_layer_norm"), eps)) for n in ("q", "k", "v"))
        else:
            a = _ln(x, pc, L + "self_attn_layer_norm", eps)
            q, k, v = (rb(_lin(a, pc, L + f"self_attn/{n}_proj")) for n in ("q", "k", "v"))
        state.k[i][:, t] = sp(k)[:, 0]
        state.v[i][:, t] = sp(v)[:, 0]
        ctx = _attn_decode(sp(q), state.k[i], state.v[i], t + 1).reshape(x.shape)
        x1 = rb(_lin(ctx, pc, L + "self_attn/out_proj") + x)
        if ln_fold:
            q = rb(_ln_folded_lin(x1, _fold(pc, L + "encoder_attn/q_proj", L + "encoder_attn_layer_norm"), eps))
        else:
            q = rb(_lin(_ln(x1, pc, L + "encoder_attn_layer_norm", eps), pc, L + "encoder_attn/q_proj"))
        if cross_kv is None:
            ck, cv = (sp(rb(_lin(ehs, pc, L + f"encoder_attn/{n}_proj"))) for n in ("k", "v"))
        else:
            ck, cv = cross_kv[i]
        ctx = _attn_decode(sp(q), ck, cv, ck.shape[1]).reshape(x.shape)
        x2 = rb(_lin(ctx, pc, L + "encoder_attn/out_proj") + x1)
        if ln_fold:
            z = rb(_ln_folded_lin(x2, _fold(pc, L + "fc1", L + "final_layer_norm"), eps))
        else:
            z = rb(_lin(_ln(x2, pc, L + "final_layer_norm", eps), pc, L + "fc1"))
        u = rb(M.gelu(z, cfg.gelu))
        x = rb(_lin(u, pc, L + "fc2") + x2)
    state.index = t + 1
    return lm_head(cfg, pc, _ln(x, pc, D_ + "layer_norm", eps))


def cross_kv(cfg: M.RefConfig, pc: Params, ehs: torch.Tensor):
    """cross-attention K/V projected once per image (`_decode_set_encoder`)"""
    H = cfg.d_heads
    return [tuple(M._split(rb(_lin(ehs, pc, f"{D_}layers/{i}/encoder_attn/{n}_proj")), H) for n in ("k", "v")) for i in range(cfg.d_layers)]


def flips(got_bf16: torch.Tensor, ref_unrounded: torch.Tensor) -> float:
    """fraction of stored values that differ from the rounding of the reference's unrounded value"""
    return (got_bf16.to(torch.float32) != rb(ref_unrounded.to(torch.float32))).float().mean().item()


def stored_error(got_bf16: torch.Tensor, ref_unrounded: torch.Tensor) -> Tuple[float, float]:
    """(max, mean) over elements of the distance between a stored bf16 value and the interval of bf16 roundings of
    [ref - e, ref + e] expressed as the smallest such e, in units of max|ref|: the kernel error with the final rounding
    taken out.  e = max(0, |got - ref| - half_ulp(got))."""
    got = got_bf16.to(torch.float32)
    ref = ref_unrounded.to(torch.float32)
    mag = got.abs().clamp_min(torch.finfo(torch.float32).tiny)
    half_ulp = torch.exp2(torch.floor(torch.log2(mag)) - 8)  # bf16: 8 significant bits -> ulp = 2^(e-7), half = 2^(e-8)
    e = ((got - ref).abs() - half_ulp).clamp_min(0.0)
    s = ref.abs().max().clamp_min(1e-30)
    return (e.max() / s).item(), (e.mean() / s).item()
