"""ORACLE — TEST INFRASTRUCTURE ONLY.  (parity unpinned — see below)

CPU fp32 restatement (torch-CPU) of the arithmetic on the reference's hot path:
CLIP ViT encoder -> Dense 768->1024 -> mBART decoder -> tied LM head.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
package; the product (`multilingual-image-captioning_amd/`) never does.

PARITY UNPINNED.  The reference (`/root/reference`) is Flax/JAX and cannot be imported in the build
container (no jax/flax/optax), it ships no tests / golden vectors, and the transformer arithmetic
lives in an un-vendored dependency: `transformers @ 0085e712ddf80fa5cd5f355498fe7f13b839eafa`
(`requirements.txt:39`: `models/clip/modeling_flax_clip.py`, `models/mbart/modeling_flax_mbart.py`),
`flax==0.3.4`, `jax==0.2.16`, `optax==0.0.9`.  This file restates that published algorithm, anchored on
the reference's own call sites (cited per function), and is pinned against the PyTorch twins of the same
two architectures in `transformers 5.15` (fixtures in `tests/golden/`, generator
`tests/golden/make_golden.py`) — a stand-in, not the reference.

Parameters use the reference's Flax tree (`modeling_clip_vision_mbart.py:36-59,
123-135, 768-770`), flattened with "/" separators; Dense kernels are [in,out], the patch conv is HWIO.

Three numerically relevant facts of the pinned dependency cannot be verified offline and are switches
on `RefConfig`: `gelu` ("tanh" | "erf"), `decoder_ln_eps` (1e-6 | 1e-5), and whether cross-attention
K/V are cached (numerically irrelevant).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch

Params = Dict[str, torch.Tensor]


@dataclass
class RefConfig:
    # CLIP ViT-B/32 (hub config openai/clip-vit-base-patch32)
    v_hidden: int = 768
    v_ffn: int = 3072
    v_layers: int = 12
    v_heads: int = 12
    image_size: int = 224
    patch_size: int = 32
    v_ln_eps: float = 1e-5
    # mBART-large-50 decoder (hub config facebook/mbart-large-50)
    d_model: int = 1024
    d_ffn: int = 4096
    d_layers: int = 12
    d_heads: int = 16
    vocab_size: int = 250054
    max_position_embeddings: int = 1024
    scale_embedding: bool = True
    pad_token_id: int = 1
    bos_token_id: int = 0
    eos_token_id: int = 2
    decoder_start_token_id: int = 2
    forced_eos_token_id: Optional[int] = 2
    dropout: float = 0.1
    # [UNVERIFIED-3P] switches
    gelu: str = "tanh"  # jax.nn.gelu default at the pinned commit is the tanh approximation
    decoder_ln_eps: float = 1e-6  # flax nn.LayerNorm default epsilon

    @property
    def n_patches(self) -> int:
        return (self.image_size // self.patch_size) ** 2

    @property
    def v_seq(self) -> int:
        return self.n_patches + 1


# --------------------------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------------------------
def layer_norm(x: torch.Tensor, scale: torch.Tensor, bias: torch.Tensor, eps: float) -> torch.Tensor:
    """flax nn.LayerNorm: biased variance over the last dim, fp32 statistics."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * scale + bias


def dense(x: torch.Tensor, p: Params, name: str) -> torch.Tensor:
    """flax nn.Dense: y = x @ kernel[in,out] + bias."""
    y = x @ p[name + "/kernel"]
    b = p.get(name + "/bias")
    return y + b if b is not None else y


def quick_gelu(x: torch.Tensor) -> torch.Tensor:
    return x * torch.sigmoid(1.702 * x)


def gelu(x: torch.Tensor, kind: str) -> torch.Tensor:
    if kind == "erf":
        return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))
    if kind == "tanh":
        return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))
    raise ValueError(kind)


ATTN_TAP: Optional[list] = None  # set to a list to collect the attention weights [B,H,T,S] of every attention_core call


def attention_core(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """flax `dot_product_attention_weights`: q scaled by 1/sqrt(D) BEFORE q.k^T,
    softmax over keys of (scores + bias), weights @ v.  q,k,v: [B,T,H,D] / [B,S,H,D]; bias [B,1|H,T,S]."""
    d = q.shape[-1]
    q = q / math.sqrt(d)
    w = torch.einsum("bthd,bshd->bhts", q, k)
    if bias is not None:
        w = w + bias
    w = torch.softmax(w, dim=-1)
    if ATTN_TAP is not None:  # tests of `output_attentions`: the weights of every attention, in call order
        ATTN_TAP.append(w)
    return torch.einsum("bhts,bshd->bthd", w, v)


def mask_to_bias(mask: torch.Tensor) -> torch.Tensor:
    """select(mask > 0, 0, -inf) — the Flax mBART attention-bias construction."""
    return torch.where(mask > 0, torch.zeros((), dtype=torch.float32), torch.full((), float("-inf")))


# --------------------------------------------------------------------------------------------
# CLIP vision encoder   (3P FlaxCLIPVisionModule, instantiated at modeling_clip_vision_mbart.py:46-48)
# --------------------------------------------------------------------------------------------
V = "model/encoder/vision_model/"


def vision_embeddings(cfg: RefConfig, p: Params, pixels_nhwc: torch.Tensor) -> torch.Tensor:
    """Conv(768, 32x32, stride 32, VALID, no bias) on NHWC with an HWIO kernel; class token first;
    + position_embedding[arange(50)]  (SURVEY App. B1)."""
    B = pixels_nhwc.shape[0]
    ps, g = cfg.patch_size, cfg.image_size // cfg.patch_size
    w = p[V + "embeddings/patch_embedding/kernel"]  # [ps,ps,3,hidden]
    x = pixels_nhwc.reshape(B, g, ps, g, ps, 3).permute(0, 1, 3, 2, 4, 5).reshape(B, g * g, ps * ps * 3)
    patches = x @ w.reshape(ps * ps * 3, cfg.v_hidden)
    cls = p[V + "embeddings/class_embedding"].reshape(1, 1, -1).expand(B, 1, cfg.v_hidden)
    emb = torch.cat([cls, patches], dim=1)
    return emb + p[V + "embeddings/position_embedding/embedding"][None, : cfg.v_seq]


def vit_layer(cfg: RefConfig, p: Params, x: torch.Tensor, i: int) -> torch.Tensor:
    """Pre-LN block: x += Attn(LN1(x)); x += fc2(quick_gelu(fc1(LN2(x))))  (SURVEY App. B2)."""
    L = f"{V}encoder/layers/{i}/"
    B, S, _ = x.shape
    H, D = cfg.v_heads, cfg.v_hidden // cfg.v_heads
    h = layer_norm(x, p[L + "layer_norm1/scale"], p[L + "layer_norm1/bias"], cfg.v_ln_eps)
    q = dense(h, p, L + "self_attn/q_proj").reshape(B, S, H, D)
    k = dense(h, p, L + "self_attn/k_proj").reshape(B, S, H, D)
    v = dense(h, p, L + "self_attn/v_proj").reshape(B, S, H, D)
    a = attention_core(q, k, v, None).reshape(B, S, H * D)
    x = x + dense(a, p, L + "self_attn/out_proj")
    h = layer_norm(x, p[L + "layer_norm2/scale"], p[L + "layer_norm2/bias"], cfg.v_ln_eps)
    h = dense(quick_gelu(dense(h, p, L + "mlp/fc1")), p, L + "mlp/fc2")
    return x + h


def vit_encoder(cfg: RefConfig, p: Params, pixels_nhwc: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Returns (last_hidden_state [B,50,768] = output of the last block, NOT post-layernormed;
    pooler_output [B,768] = post_layernorm(CLS))."""
    x = vision_embeddings(cfg, p, pixels_nhwc)
    x = layer_norm(x, p[V + "pre_layrnorm/scale"], p[V + "pre_layrnorm/bias"], cfg.v_ln_eps)
    for i in range(cfg.v_layers):
        x = vit_layer(cfg, p, x, i)
    pooled = layer_norm(x[:, 0], p[V + "post_layernorm/scale"], p[V + "post_layernorm/bias"], cfg.v_ln_eps)
    return x, pooled


def encode(cfg: RefConfig, p: Params, pixel_values: torch.Tensor, int32_cast: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """`encode()` (modeling_clip_vision_mbart.py:284-337): encoder + visual_projection on
    last_hidden_state (315-326).  `pixel_values` are cast to int32 (line 330: truncation toward zero)."""
    px = pixel_values.to(torch.float32)
    if int32_cast:
        px = torch.trunc(px)
    last, pooled = vit_encoder(cfg, p, px)
    return dense(last, p, "model/visual_projection"), pooled


# --------------------------------------------------------------------------------------------
# mBART decoder   (3P FlaxMBartDecoder, instantiated at modeling_clip_vision_mbart.py:49-51)
# --------------------------------------------------------------------------------------------
D_ = "model/decoder/"


def decoder_embed(cfg: RefConfig, p: Params, ids: torch.Tensor, position_ids: torch.Tensor,
                  dropout_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """h = shared[ids]*sqrt(d) + embed_positions[position_ids + 2]; layernorm_embedding; dropout (App. B5)."""
    scale = math.sqrt(cfg.d_model) if cfg.scale_embedding else 1.0
    h = p["model/shared/embedding"][ids] * scale + p[D_ + "embed_positions/embedding"][position_ids + 2]
    h = layer_norm(h, p[D_ + "layernorm_embedding/scale"], p[D_ + "layernorm_embedding/bias"], cfg.decoder_ln_eps)
    return _drop(h, dropout_mask, cfg.dropout)


def _drop(x: torch.Tensor, mask: Optional[torch.Tensor], p: float) -> torch.Tensor:
    """Inverted dropout with an injected keep-mask (1 = keep).  None = deterministic."""
    if mask is None:
        return x
    return x * mask.to(x.dtype) / (1.0 - p)


def _split(x: torch.Tensor, H: int) -> torch.Tensor:
    return x.reshape(x.shape[0], x.shape[1], H, x.shape[2] // H)


def decoder_layer(cfg: RefConfig, p: Params, h: torch.Tensor, i: int, self_bias: torch.Tensor,
                  ehs: torch.Tensor, masks: Optional[Dict[str, torch.Tensor]] = None,
                  kv_override: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> torch.Tensor:
    """Pre-LN block (App. B6): h += drop(SelfAttn(LN(h))); h += drop(CrossAttn(LN(h), ehs));
    h += drop(fc2(gelu(fc1(LN(h))))).  `kv_override` = (k, v) [B,S,H,D] replaces the self-attn keys/values
    (decode-time cache)."""
    L = f"{D_}layers/{i}/"
    H = cfg.d_heads
    eps = cfg.decoder_ln_eps
    m = masks or {}
    r = h
    x = layer_norm(h, p[L + "self_attn_layer_norm/scale"], p[L + "self_attn_layer_norm/bias"], eps)
    q = _split(dense(x, p, L + "self_attn/q_proj"), H)
    if kv_override is None:
        k = _split(dense(x, p, L + "self_attn/k_proj"), H)
        v = _split(dense(x, p, L + "self_attn/v_proj"), H)
    else:
        k, v = kv_override
    a = attention_core(q, k, v, self_bias).reshape(h.shape)
    h = r + _drop(dense(a, p, L + "self_attn/out_proj"), m.get(f"{i}/self"), cfg.dropout)
    r = h
    x = layer_norm(h, p[L + "encoder_attn_layer_norm/scale"], p[L + "encoder_attn_layer_norm/bias"], eps)
    q = _split(dense(x, p, L + "encoder_attn/q_proj"), H)
    k = _split(dense(ehs, p, L + "encoder_attn/k_proj"), H)
    v = _split(dense(ehs, p, L + "encoder_attn/v_proj"), H)
    a = attention_core(q, k, v, None).reshape(h.shape)  # encoder mask is all ones (modeling:87-88)
    h = r + _drop(dense(a, p, L + "encoder_attn/out_proj"), m.get(f"{i}/cross"), cfg.dropout)
    r = h
    x = layer_norm(h, p[L + "final_layer_norm/scale"], p[L + "final_layer_norm/bias"], eps)
    x = dense(gelu(dense(x, p, L + "fc1"), cfg.gelu), p, L + "fc2")
    return r + _drop(x, m.get(f"{i}/ffn"), cfg.dropout)


def decoder_forward(cfg: RefConfig, p: Params, ids: torch.Tensor, attention_mask: torch.Tensor,
                    position_ids: torch.Tensor, ehs: torch.Tensor,
                    masks: Optional[Dict[str, torch.Tensor]] = None,
                    return_layers: bool = False):
    """Teacher-forced decoder: causal AND key-padding mask -> 0/-inf bias (App. B6)."""
    B, T = ids.shape
    causal = torch.tril(torch.ones(T, T, dtype=torch.int32))[None, None]
    allowed = causal * attention_mask.to(torch.int32)[:, None, None, :]
    bias = mask_to_bias(allowed)
    m = masks or {}
    h = decoder_embed(cfg, p, ids, position_ids, m.get("embed"))
    layers: List[torch.Tensor] = []
    for i in range(cfg.d_layers):
        h = decoder_layer(cfg, p, h, i, bias, ehs, masks)
        layers.append(h)
    h = layer_norm(h, p[D_ + "layer_norm/scale"], p[D_ + "layer_norm/bias"], cfg.decoder_ln_eps)
    return (h, layers) if return_layers else h


def lm_head(cfg: RefConfig, p: Params, h: torch.Tensor) -> torch.Tensor:
    """Tied head: h @ shared.embedding^T + final_logits_bias  (modeling:170-178, decode path 600-610)."""
    return h @ p["model/shared/embedding"].T + p["final_logits_bias"]


def forward_logits(cfg: RefConfig, p: Params, pixel_values: torch.Tensor, decoder_input_ids: torch.Tensor,
                   decoder_attention_mask: Optional[torch.Tensor] = None,
                   decoder_position_ids: Optional[torch.Tensor] = None,
                   masks: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """Outer `__call__` (modeling:447-510): mask default ones (488-489), positions default arange (490-494),
    pixels cast to float32 (501).  Module graph modeling:67-115, 146-192."""
    ids = decoder_input_ids.to(torch.int64)
    B, T = ids.shape
    am = torch.ones_like(ids) if decoder_attention_mask is None else decoder_attention_mask
    pos = torch.arange(T)[None].expand(B, T) if decoder_position_ids is None else decoder_position_ids.to(torch.int64)
    last, _ = vit_encoder(cfg, p, pixel_values.to(torch.float32))
    ehs = dense(last, p, "model/visual_projection")
    h = decoder_forward(cfg, p, ids, am, pos, ehs, masks)
    return lm_head(cfg, p, h)


# --------------------------------------------------------------------------------------------
# decode-time: static max_length-slot self-attention cache  (init_cache modeling:249-282; App. B7)
# --------------------------------------------------------------------------------------------
class DecodeState:
    """Per-layer `cached_key/value [R,max_length,H,D]` zeros + scalar `cache_index` (modeling:249-282)."""

    def __init__(self, cfg: RefConfig, rows: int, max_length: int):
        H, D = cfg.d_heads, cfg.d_model // cfg.d_heads
        self.k = [torch.zeros(rows, max_length, H, D) for _ in range(cfg.d_layers)]
        self.v = [torch.zeros(rows, max_length, H, D) for _ in range(cfg.d_layers)]
        self.index = 0
        self.max_length = max_length

    def gather_rows(self, idx: torch.Tensor) -> None:
        """Beam reorder: the reference gathers every cache leaf by beam index (generation:945-953)."""
        self.k = [t[idx] for t in self.k]
        self.v = [t[idx] for t in self.v]


def decode_step(cfg: RefConfig, p: Params, state: DecodeState, ids: torch.Tensor, position_ids: torch.Tensor,
                ehs: torch.Tensor) -> torch.Tensor:
    """`decode()` with a cache (modeling:519-651): one token per row; write k,v at slot cache_index, attend all
    max_length slots with validity slot <= cache_index (causal mask sliced by cache_index; the user mask is all
    ones, modeling:669); cache_index += 1.  Returns logits [R,1,V]."""
    R = ids.shape[0]
    H = cfg.d_heads
    t = state.index
    valid = (torch.arange(state.max_length) <= t).to(torch.int32)[None, None, None, :]
    bias = mask_to_bias(valid)
    h = decoder_embed(cfg, p, ids.to(torch.int64), position_ids.to(torch.int64))
    for i in range(cfg.d_layers):
        L = f"{D_}layers/{i}/"
        x = layer_norm(h, p[L + "self_attn_layer_norm/scale"], p[L + "self_attn_layer_norm/bias"], cfg.decoder_ln_eps)
        state.k[i][:, t] = _split(dense(x, p, L + "self_attn/k_proj"), H)[:, 0]
        state.v[i][:, t] = _split(dense(x, p, L + "self_attn/v_proj"), H)[:, 0]
        h = decoder_layer(cfg, p, h, i, bias, ehs, None, kv_override=(state.k[i], state.v[i]))
    state.index = t + 1
    h = layer_norm(h, p[D_ + "layer_norm/scale"], p[D_ + "layer_norm/bias"], cfg.decoder_ln_eps)
    return lm_head(cfg, p, h)


# --------------------------------------------------------------------------------------------
# parameter construction (random init; normal(0.02) Dense/Embed, LN scale 1 / bias 0)
# --------------------------------------------------------------------------------------------
def param_shapes(cfg: RefConfig) -> Dict[str, Tuple[int, ...]]:
    """The reference's parameter tree, flattened with '/'."""
    s: Dict[str, Tuple[int, ...]] = {}
    hv, fv = cfg.v_hidden, cfg.v_ffn
    s["final_logits_bias"] = (1, cfg.vocab_size)
    s["model/shared/embedding"] = (cfg.vocab_size, cfg.d_model)
    s["model/visual_projection/kernel"] = (hv, cfg.d_model)
    s["model/visual_projection/bias"] = (cfg.d_model,)
    s[V + "embeddings/class_embedding"] = (hv,)
    s[V + "embeddings/patch_embedding/kernel"] = (cfg.patch_size, cfg.patch_size, 3, hv)
    s[V + "embeddings/position_embedding/embedding"] = (cfg.v_seq, hv)
    for ln in ("pre_layrnorm", "post_layernorm"):
        s[V + ln + "/scale"] = (hv,)
        s[V + ln + "/bias"] = (hv,)
    for i in range(cfg.v_layers):
        L = f"{V}encoder/layers/{i}/"
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[L + f"self_attn/{n}/kernel"] = (hv, hv)
            s[L + f"self_attn/{n}/bias"] = (hv,)
        for ln in ("layer_norm1", "layer_norm2"):
            s[L + ln + "/scale"] = (hv,)
            s[L + ln + "/bias"] = (hv,)
        s[L + "mlp/fc1/kernel"] = (hv, fv)
        s[L + "mlp/fc1/bias"] = (fv,)
        s[L + "mlp/fc2/kernel"] = (fv, hv)
        s[L + "mlp/fc2/bias"] = (hv,)
    d, f = cfg.d_model, cfg.d_ffn
    s[D_ + "embed_positions/embedding"] = (cfg.max_position_embeddings + 2, d)
    for ln in ("layernorm_embedding", "layer_norm"):
        s[D_ + ln + "/scale"] = (d,)
        s[D_ + ln + "/bias"] = (d,)
    for i in range(cfg.d_layers):
        L = f"{D_}layers/{i}/"
        for blk in ("self_attn", "encoder_attn"):
            for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
                s[L + f"{blk}/{n}/kernel"] = (d, d)
                s[L + f"{blk}/{n}/bias"] = (d,)
            s[L + blk + "_layer_norm/scale"] = (d,)
            s[L + blk + "_layer_norm/bias"] = (d,)
        s[L + "fc1/kernel"] = (d, f)
        s[L + "fc1/bias"] = (f,)
        s[L + "fc2/kernel"] = (f, d)
        s[L + "fc2/bias"] = (d,)
        s[L + "final_layer_norm/scale"] = (d,)
        s[L + "final_layer_norm/bias"] = (d,)
    return s


def init_params(cfg: RefConfig, seed: int = 0, std: float = 0.02, perturb_ln: bool = False) -> Params:
    g = torch.Generator().manual_seed(seed)
    p: Params = {}
    for name, shape in param_shapes(cfg).items():
        if name.endswith("/scale"):
            p[name] = torch.ones(shape) + (0.1 * torch.randn(shape, generator=g) if perturb_ln else 0.0)
        elif name.endswith("/bias") or name == "final_logits_bias":
            p[name] = 0.02 * torch.randn(shape, generator=g) if perturb_ln else torch.zeros(shape)
        else:
            p[name] = std * torch.randn(shape, generator=g)
    return p
