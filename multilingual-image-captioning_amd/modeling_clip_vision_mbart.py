"""`FlaxCLIPVisionMBartForConditionalGeneration` on MI355X — same class name, method names, keyword arguments and
return shapes as the reference's model (`models/flax_clip_vision_mbart/modeling_clip_vision_mbart.py:195-773`,
base class `modeling_clip_vision_utils.py:36-117`), so `main.py` / `evaluation.py` call sites read unchanged:

    model = FlaxCLIPVisionMBartForConditionalGeneration.from_clip_vision_mbart_pretrained(...)   # main.py:421-427
    logits = model(pixel_values, decoder_input_ids, attention_mask, params=..., train=True)[0]   # main.py:692
    ids = model.generate(pixel_values, forced_bos_token_id=lang, num_beams=4, max_length=64).sequences  # evaluation.py:81

Arrays in: numpy / torch (NHWC pixels, int ids).  Arrays out: torch tensors on the model's device (`.cpu().numpy()` for
numpy).  All arithmetic runs in libmic_hip.so; there is no CPU path.
"""
from __future__ import annotations

import hashlib
import json
import os
from typing import Any, Dict, Optional, Tuple

import numpy as np
import torch

from . import ops
from .configuration_clip_vision_mbart import CLIPVisionMBartConfig
from .engine import Engine
from .generation_clip_vision_utils import FlaxCLIPVisionMBartGenerationMixin
from .params import ParamStore, flatten_tree, unflatten_tree


class ModelOutput(dict):
    """dict with attribute access and positional indexing (what the call sites use: `[0]`, `.logits`, `.sequences`)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __getitem__(self, k):
        if isinstance(k, int):
            return [v for v in self.values() if v is not None][k]
        return super().__getitem__(k)

    def to_tuple(self):
        return tuple(v for v in self.values() if v is not None)


def _torch_dtype(dtype) -> torch.dtype:
    if isinstance(dtype, torch.dtype):
        return dtype
    name = getattr(dtype, "__name__", None) or getattr(dtype, "name", None) or str(dtype)
    name = name.replace("jnp.", "").replace("torch.", "")
    if name in ("float32", "f32", "float"):
        return torch.float32
    if name in ("bfloat16", "bf16"):
        return torch.bfloat16
    if name in ("float16", "f16", "half"):
        raise NotImplementedError("float16 compute is not built for gfx950 here: use float32 or bfloat16")
    raise ValueError(f"unknown dtype {dtype}")


def _seed_from(rng) -> Optional[int]:
    if rng is None:
        return None
    if isinstance(rng, (int, np.integer)):
        return int(rng) & 0xFFFFFFFF
    b = np.asarray(rng.cpu() if isinstance(rng, torch.Tensor) else rng).tobytes()
    return int.from_bytes(hashlib.blake2s(b, digest_size=4).digest(), "little")


class FlaxCLIPVisionMBartPreTrainedModel(FlaxCLIPVisionMBartGenerationMixin):
    """Host mirror of `modeling_clip_vision_utils.py:36-117` (params property/validation, save/load)."""

    config_class = CLIPVisionMBartConfig
    base_model_prefix = "model"

    def __init__(self, config: CLIPVisionMBartConfig, input_shape: Tuple = None, seed: int = 0, dtype=torch.float32,
                 device=None, _do_init: bool = True):
        if config is None:
            raise ValueError("config cannot be None")  # utils:60-61
        self._config = config
        self.dtype = _torch_dtype(dtype)
        self.seed = seed
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("FlaxCLIPVisionMBartForConditionalGeneration needs an MI355X (no CPU fallback)")
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        self.store = ParamStore(config, self.dtype, self.device)
        self.engine = Engine(self.store)
        # captured decoder steps point into the engine's buffers.  A weak reference: engine -> bound method -> model -> engine would keep
        # the ~16 GB of a dropped model alive until the cycle collector gets round to it
        import weakref

        me = weakref.ref(self)
        self.engine.on_free.append(lambda: me() is not None and me().release_decode_plans())
        self._required_params = set(tuple(k.split("/")) for k in self.store.flax_shapes())  # utils:78
        self._params_cache = None
        if _do_init:
            self.store.init_random(seed, float(config.mbart_config.init_std))

    # ------------------------------------------------------------------ properties (utils:91-117)
    @property
    def config(self) -> CLIPVisionMBartConfig:
        return self._config

    @property
    def required_params(self):
        return self._required_params

    @property
    def params(self) -> Dict[str, Any]:
        """The reference's nested parameter pytree (numpy leaves, Flax layouts), exported from the device buffers."""
        if self._params_cache is None:
            self._params_cache = unflatten_tree(self.store.export_flat("master"))
        return self._params_cache

    @params.setter
    def params(self, params: Dict[str, Any]):
        if params is self._params_cache and params is not None:
            return
        flat = flatten_tree(params)
        keys = set(tuple(k.split("/")) for k in flat)
        missing = self.required_params - keys
        if len(missing) > 0:
            raise ValueError("Some parameters are missing. Make sure that `params` include the following "
                             f"parameters {missing}")  # utils:112-116
        self.store.load_flat(flat)
        self._params_cache = params
        if self.engine.fp8:
            self.engine.fp8_weights_changed()

    def invalidate_params_cache(self, by_optimizer: bool = False):
        self._params_cache = None
        self.store.version = getattr(self.store, "version", 0) + 1
        if self.engine.fp8:
            self.engine.fp8_weights_changed(by_optimizer=by_optimizer)

    def _use_params(self, params):
        if params is not None and params is not self._params_cache:
            self.params = params

    # ------------------------------------------------------------------ save / load (utils:398-451, 120-396)
    def save_pretrained(self, save_directory: str, params=None, push_to_hub: bool = False, **kwargs):
        if os.path.isfile(save_directory):
            raise EnvironmentError(f"Provided path ({save_directory}) should be a directory, not a file")
        if push_to_hub:
            raise NotImplementedError("hub push is out of scope (no network)")
        os.makedirs(save_directory, exist_ok=True)
        self.config.save_pretrained(save_directory)
        from .checkpoint import save_flax_msgpack

        flat = flatten_tree(params) if params is not None else self.store.export_flat("master")
        save_flax_msgpack(os.path.join(save_directory, "flax_model.msgpack"), unflatten_tree(flat))

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, dtype=torch.float32, *model_args, **kwargs):
        if not os.path.isdir(pretrained_model_name_or_path):
            raise EnvironmentError(f"{pretrained_model_name_or_path}: only local directories are supported (no network)")
        config = kwargs.pop("config", None) or CLIPVisionMBartConfig.from_pretrained(pretrained_model_name_or_path)
        from .checkpoint import load_flax_msgpack

        path = os.path.join(pretrained_model_name_or_path, "flax_model.msgpack")
        if not os.path.isfile(path):
            raise EnvironmentError(f"Error no file named flax_model.msgpack found in directory {pretrained_model_name_or_path}")
        model = cls(config, *model_args, dtype=dtype, _do_init=True, **kwargs)
        state = flatten_tree(load_flax_msgpack(path))
        cur = model.store.export_flat("master")
        for k in list(state):  # unexpected keys dropped, missing keys keep their random init (utils:355-364)
            if k not in cur:
                state.pop(k)
        cur.update(state)
        model.store.load_flat(cur)
        model._params_cache = None
        return model


class FlaxCLIPVisionMBartForConditionalGeneration(FlaxCLIPVisionMBartPreTrainedModel):
    # ------------------------------------------------------------------ input plumbing
    def _dev(self, x, dtype) -> torch.Tensor:
        if isinstance(x, torch.Tensor):
            return x.to(device=self.device, dtype=dtype).contiguous()
        return torch.from_numpy(np.ascontiguousarray(np.asarray(x))).to(device=self.device, dtype=dtype).contiguous()

    def _check_pixels(self, px: torch.Tensor):
        img = self.config.clip_vision_config.image_size
        if px.dim() != 4 or tuple(px.shape[1:]) != (img, img, 3):
            raise ValueError(f"pixel_values must be NHWC [B,{img},{img},3], got {tuple(px.shape)}")

    # ------------------------------------------------------------------ __call__ (modeling:447-510)
    def __call__(self, pixel_values, decoder_input_ids=None, decoder_attention_mask=None, decoder_position_ids=None,
                 output_attentions=None, output_hidden_states=None, return_dict=None, train: bool = False, params=None,
                 dropout_rng=None):
        self._use_params(params)
        px = self._dev(pixel_values, torch.float32)  # modeling:501
        self._check_pixels(px)
        ids = self._dev(decoder_input_ids, torch.int32)  # modeling:502
        B, T = ids.shape
        mask = torch.ones_like(ids) if decoder_attention_mask is None else self._dev(decoder_attention_mask, torch.int32)  # 488-489
        if decoder_position_ids is None:
            pos = torch.arange(T, dtype=torch.int32, device=self.device)[None].expand(B, T).contiguous()  # 490-494
        else:
            pos = self._dev(decoder_position_ids, torch.int32)
        seed = _seed_from(dropout_rng) if train else None  # deterministic = not train (508)
        output_hidden_states = output_hidden_states if output_hidden_states is not None else getattr(self.config, "output_hidden_states", False)
        output_attentions = output_attentions if output_attentions is not None else getattr(self.config, "output_attentions", False)
        if output_attentions and (self.store.d // self.store.H != 64 or self.store.vd // self.store.vH != 64):
            raise NotImplementedError("output_attentions=True: the weights kernel is built for 64-wide heads")
        logits, ehs = self.engine.forward_logits(px, ids.reshape(-1), pos.reshape(-1), mask, B, T,
                                                 save=bool(output_hidden_states or output_attentions), seed=seed)
        V = self.store.V
        out = logits[: B * T, :V].reshape(B, T, V)
        att = self._attention_weights(B, T, mask) if output_attentions else {}
        if not output_hidden_states:
            if return_dict is False:
                return (out,) + tuple(att[k] for k in ("decoder_attentions", "cross_attentions", "encoder_attentions") if k in att)
            return ModelOutput(logits=out, **att)
        # FlaxSeq2SeqLMOutput fields of the reference's module output (modeling:176-192): the per-layer activations the forward
        # pass kept (`save=True`), copied out of the engine's buffers.  Decoder: the embedding output (after layernorm_embedding)
        # and every layer's output — the last one before the final layer_norm, as the Flax decoder collects them; encoder: the
        # pre-layernormed embeddings and every layer's output.
        eng, st = self.engine, self.store
        S, d, vd = st.S, st.d, st.vd
        dec = [eng.buf("d.x0", B * T, d)] + [eng.buf(f"d{l}.x3", B * T, d) for l in range(st.L)]
        enc = [eng.buf("v.x0", B * S, vd)] + [eng.buf(f"v{l}.xo", B * S, vd) for l in range(st.vL)]
        res = ModelOutput(logits=out, decoder_hidden_states=tuple(t[: B * T].reshape(B, T, d).clone() for t in dec))  # field order of
        for k in ("decoder_attentions", "cross_attentions"):                                                       # FlaxSeq2SeqLMOutput
            if k in att:
                res[k] = att[k]
        res["encoder_last_hidden_state"] = ehs[: B * S].reshape(B, S, d).clone()
        res["encoder_hidden_states"] = tuple(t[: B * S].reshape(B, S, vd).clone() for t in enc)
        if "encoder_attentions" in att:
            res["encoder_attentions"] = att["encoder_attentions"]
        return res if return_dict is not False else res.to_tuple()

    def _attention_weights(self, B: int, T: int, mask: torch.Tensor) -> dict:
        """`output_attentions=True` (modeling:499-510): the softmax weights of every attention of the pass that has just run with
        `save=True`, recomputed from the kept q / k projections by the diagnostic kernel `mic_attn_probs` (the fused attention cores
        never write them).  [B, H, Tq, Tk] per layer, in the compute dtype like the Flax modules return them; they are the
        weights BEFORE attention dropout (the reference's attention_dropout is 0: config defaults)."""
        eng, st = self.engine, self.store
        S, d, vd, H, vH = st.S, st.d, st.vd, st.H, st.vH
        dt = eng.dt

        def probs(q, k, Hh, Tq, Tk, ldq, ldk, key_mask=None, causal=False):
            o = torch.empty((B, Hh, Tq, Tk), dtype=torch.float32, device=self.device)
            ops.attn_probs(q, k, o, B, Hh, Tq, Tk, ldq=ldq, ldk=ldk, key_mask=key_mask, causal=causal)
            return o.to(dt)

        enc, dec, cross = [], [], []
        for l in range(st.vL):
            qkv = eng.buf(f"v{l}.qkv", B * S, 3 * vd)
            enc.append(probs(qkv, qkv[:, vd:], vH, S, S, 3 * vd, 3 * vd))
        hoist = eng.ckv_hoisted()
        kvcat = eng.buf("d.ckvcat", B * S, st.L * 2 * d) if hoist else None
        for l in range(st.L):
            qkv = eng.buf(f"d{l}.qkv", B * T, 3 * d)
            dec.append(probs(qkv, qkv[:, d:], H, T, T, 3 * d, 3 * d, key_mask=mask, causal=True))
            cq = eng.buf(f"d{l}.cq", B * T, d)
            ck, ldk = (kvcat[:, l * 2 * d:], kvcat.stride(0)) if hoist else (eng.buf(f"d{l}.ckv", B * S, 2 * d), 2 * d)
            cross.append(probs(cq, ck, H, T, S, d, ldk))
        return {"decoder_attentions": tuple(dec), "cross_attentions": tuple(cross), "encoder_attentions": tuple(enc)}

    # ------------------------------------------------------------------ encode (modeling:284-337)
    def encode(self, pixel_values, output_attentions=None, output_hidden_states=None, return_dict=None, train: bool = False,
               params=None, dropout_rng=None, _int32_cast: bool = True):
        self._use_params(params)
        px = self._dev(pixel_values, torch.float32)
        self._check_pixels(px)
        B = px.shape[0]
        # modeling:330 — `jnp.array(pixel_values, dtype="i4")`: truncation toward zero, reproduced inside im2col
        with self.engine.storage_dtype_gemms():  # inference stays in the storage dtype (fp8 copies belong to train / eval passes)
            last, ehs = self.engine.vit_forward(px, save=False, trunc_int32=_int32_cast)
        S, d = self.store.S, self.store.d
        pooled = self.engine.vit_pooler(last, B)
        out = ModelOutput(last_hidden_state=ehs[: B * S].reshape(B, S, d).clone(), pooler_output=pooled)
        return out if return_dict is not False else out.to_tuple()

    # ------------------------------------------------------------------ decode (modeling:519-651)
    def decode(self, decoder_input_ids, encoder_outputs, encoder_attention_mask=None, decoder_attention_mask=None,
               decoder_position_ids=None, past_key_values: dict = None, output_attentions=None, output_hidden_states=None,
               return_dict=None, deterministic: bool = True, params=None, dropout_rng=None):
        self._use_params(params)
        ehs = encoder_outputs[0]
        ids = self._dev(decoder_input_ids, torch.int32)
        R, T = ids.shape
        if decoder_position_ids is None:
            if past_key_values is not None:
                raise ValueError("Make sure to provide `decoder_position_ids` when passing `past_key_values`.")  # 558-562
            pos = torch.arange(T, dtype=torch.int32, device=self.device)[None].expand(R, T).contiguous()
        else:
            pos = self._dev(decoder_position_ids, torch.int32)
        ehs = self._dev(ehs, self.dtype)
        S, d, V = self.store.S, self.store.d, self.store.V
        if past_key_values:
            if T != 1:
                raise ValueError("cached decode takes one token per row")
            cache = past_key_values
            if cache.get("cross") is None:
                self._decode_set_encoder(cache, ehs.reshape(R * S, d), R, 1)
            logits = self._decode_step(cache, ids.reshape(-1), pos.reshape(-1))
            out = ModelOutput(logits=logits[:R, :V].reshape(R, 1, V), past_key_values=cache)
            return out if return_dict is not False else (out["logits"], cache)
        mask = torch.ones_like(ids) if decoder_attention_mask is None else self._dev(decoder_attention_mask, torch.int32)
        ehs_buf = self.engine.buf("v.ehs", R * S, d)
        ehs_buf[: R * S].copy_(ehs.reshape(R * S, d))
        with self.engine.storage_dtype_gemms():
            hf = self.engine.decoder_forward(ids.reshape(-1), pos.reshape(-1), mask, ehs_buf, R, T, save=False, seed=None)
        logits = self.engine.head_logits(hf, R * T)
        out = ModelOutput(logits=logits[: R * T, :V].reshape(R, T, V))
        return out if return_dict is not False else out.to_tuple()

    # ------------------------------------------------------------------ cache protocol (modeling:249-282, 653-693)
    def init_cache(self, batch_size: int, max_length: int, encoder_outputs=None) -> dict:
        """Zero self-attention cache with static length `max_length` (modeling:249-282) + cache_index."""
        st = self.store
        return {
            "k": [torch.zeros((batch_size, max_length, st.d), dtype=self.dtype, device=self.device) for _ in range(st.L)],
            "v": [torch.zeros((batch_size, max_length, st.d), dtype=self.dtype, device=self.device) for _ in range(st.L)],
            "cache_index": 0, "max_length": max_length, "rows": batch_size, "src_row": None, "cross": None, "row_div": 1,
        }

    def prepare_inputs_for_generation(self, decoder_input_ids, max_length, attention_mask=None, decoder_attention_mask=None,
                                      encoder_outputs=None, **kwargs):
        ids = self._dev(decoder_input_ids, torch.int32)
        batch_size, seq_length = ids.shape
        past = self.init_cache(batch_size, max_length, encoder_outputs)
        ext = torch.ones((batch_size, max_length), dtype=torch.int32, device=self.device)  # modeling:669
        if decoder_attention_mask is not None:
            dam = self._dev(decoder_attention_mask, torch.int32)
            position_ids = dam.cumsum(-1) - 1
            ext[:, : dam.shape[1]] = dam
        else:
            position_ids = torch.arange(seq_length, dtype=torch.int32, device=self.device)[None].expand(batch_size, seq_length)
        return {"past_key_values": past, "encoder_outputs": encoder_outputs, "encoder_attention_mask": attention_mask,
                "decoder_attention_mask": ext, "decoder_position_ids": position_ids}

    def update_inputs_for_generation(self, model_outputs, model_kwargs):
        model_kwargs["past_key_values"] = model_outputs.past_key_values
        model_kwargs["decoder_position_ids"] = model_kwargs["decoder_position_ids"][:, -1:] + 1  # modeling:690-692
        return model_kwargs

    # ------------------------------------------------------------------ one cached decoder step (R rows, one token each)
    def _decode_set_encoder(self, cache: dict, ehs_rows: torch.Tensor, n_img: int, row_div: int):
        """Cross-attention K/V projected ONCE per generate call for the n_img distinct images (the reference re-projects
        them every step; same values).  ehs_rows: [n_img*S, d]."""
        eng, st = self.engine, self.store
        S, d = st.S, st.d
        ns = cache.get("ns", "")  # scratch-buffer namespace of a decode slice (slices of equal size must not share buffers)
        ehs_b = eng.buf(ns + "g.ehs", n_img * S, d)
        ehs_b[: n_img * S].copy_(ehs_rows)
        cross = []
        for l in range(st.L):
            kv = eng.buf(f"{ns}g.ckv{l}", n_img * S, 2 * d)
            eng.linear(ehs_b, f"dec{l}.ckv", kv, n_img * S, fp8=False)  # generation stays in the storage dtype
            cross.append(kv)
        cache["cross"], cache["row_div"] = cross, row_div
        if eng.decode_ln_fold and self.dtype == torch.bfloat16:
            # the LayerNorm-folded weights of the decoder step, rebuilt here (once per generate call at most) if the weights moved
            for l in range(st.L):
                p = f"dec{l}."
                for wn, ln in ((p + "qkv", p + "ln_sa"), (p + "cq", p + "ln_ca"), (p + "fc1", p + "ln_ff")):
                    if l > 0 or not wn.endswith("qkv"):
                        eng.ln_folded(wn, ln)

    def _decode_step(self, cache: dict, tokens: torch.Tensor, pos: torch.Tensor, stats: bool = False):
        """stats=True: returns (logits, per-tile softmax partials of the head GEMM or None in float32 mode)

        bfloat16 mode folds the decoder layer's LayerNorms around the GEMMs (engine.ln_folded / mic_gemm_args.a_ln_stats): the
        GEMM that writes a residual-stream tensor (self-attention out, cross-attention out, fc2) also accumulates (sum, sum of
        squares) of every row it stores, and the Linear that consumes LN(that tensor) runs on the raw rows with gamma folded
        into its weight and finishes the normalisation in its epilogue — 35 of the 38 LayerNorm launches of a decoder step and
        their activation round trips disappear (MIC_DECODE_LNFOLD=0 keeps the explicit kernels; float32 mode always does)."""
        eng, st = self.engine, self.store
        P = st
        R, Lmax, cur = cache["rows"], cache["max_length"], cache["cache_index"]
        d, f, H, S = st.d, st.ffn, st.H, st.S
        fold = eng.decode_ln_fold and self.dtype == torch.bfloat16
        ns = cache.get("ns", "")
        h0 = eng.buf(ns + "g.h0", R, d)
        ops.embed_fwd(tokens, pos, P.w("shared"), P.f32("dec.pos"), eng.embed_scale, h0, R, d)
        x = eng.buf(ns + "g.x", R, d)
        ops.layernorm_fwd(h0, P.f32("dec.ln_emb.g"), P.f32("dec.ln_emb.b"), eng.dec_eps, x, rows=R)
        a, ctx = eng.buf(ns + "g.a", R, d), eng.buf(ns + "g.ctx", R, d)
        x1, x2, q = eng.buf(ns + "g.x1", R, d), eng.buf(ns + "g.x2", R, d), eng.buf(ns + "g.q", R, d)
        u = eng.buf(ns + "g.u", R, f)
        if fold:
            # (sum, sum of squares) per row of x (layer input; layer 0's comes from an explicit LayerNorm), x1, x2 — zeroed once per step
            lnst = eng.buf(ns + "g.lnstats", st.L * 3 * R, 2, torch.int64)[: st.L * 3 * R].view(st.L, 3, R, 2)  # 2^20 fixed point
            ops.zero(lnst)
        eps = eng.dec_eps
        for l in range(st.L):
            p = f"dec{l}."
            kc, vc = cache["k"][l], cache["v"][l]
            # the fused q/k/v projection as ONE grouped launch of three problems that share the A operand: q goes to its buffer,
            # k and v straight into slot `cur` of every row's cache (row stride max_len * d) — no separate append kernel
            if fold and l > 0:
                w, cs, b = eng.ln_folded(p + "qkv", p + "ln_sa")
                lk = dict(ln_stats=lnst[l, 0], ln_width=d, ln_eps=eps)
                ops.gemm_grouped([ops.gemm_args(x, w[:d], q, R, d, d, bias=b[:d], ln_colsum=cs[:d], **lk),
                                  ops.gemm_args(x, w[d:2 * d], kc[:, cur], R, d, d, bias=b[d:2 * d], ln_colsum=cs[d:2 * d], **lk),
                                  ops.gemm_args(x, w[2 * d:], vc[:, cur], R, d, d, bias=b[2 * d:], ln_colsum=cs[2 * d:], **lk)])
            else:
                ops.layernorm_fwd(x, P.f32(p + "ln_sa.g"), P.f32(p + "ln_sa.b"), eps, a, rows=R)
                w, b = P.w(p + "qkv.w"), P.f32(p + "qkv.b")
                ops.gemm_grouped([ops.gemm_args(a, w[:d], q, R, d, d, bias=b[:d]),
                                  ops.gemm_args(a, w[d:2 * d], kc[:, cur], R, d, d, bias=b[d:2 * d]),
                                  ops.gemm_args(a, w[2 * d:], vc[:, cur], R, d, d, bias=b[2 * d:])])
            ops.attn_decode(q, kc, vc, ctx, R, H, Lmax, cur, ldq=d, ldo=d, src_row=cache["src_row"])
            if fold:
                ops.gemm(ctx, P.w(p + "so.w"), x1, R, d, d, bias=P.f32(p + "so.b"), residual=x, rowsum2=lnst[l, 1])
                w, cs, b = eng.ln_folded(p + "cq", p + "ln_ca")
                ops.gemm(x1, w, q, R, d, d, bias=b, ln_stats=lnst[l, 1], ln_colsum=cs, ln_width=d, ln_eps=eps)
            else:
                eng.linear(ctx, p + "so", x1, R, residual=x, fp8=False)
                ops.layernorm_fwd(x1, P.f32(p + "ln_ca.g"), P.f32(p + "ln_ca.b"), eps, a, rows=R)
                eng.linear(a, p + "cq", q, R, fp8=False)
            kv = cache["cross"][l]
            ops.attn_decode(q, kv, kv[:, d:], ctx, R, H, S, S - 1, ldq=d, ldo=d, ldc=2 * d, row_div=cache["row_div"])
            if fold:
                ops.gemm(ctx, P.w(p + "co.w"), x2, R, d, d, bias=P.f32(p + "co.b"), residual=x1, rowsum2=lnst[l, 2])
                w, cs, b = eng.ln_folded(p + "fc1", p + "ln_ff")
                ops.gemm(x2, w, u, R, f, d, bias=b, act=eng.gelu, ln_stats=lnst[l, 2], ln_colsum=cs, ln_width=d, ln_eps=eps)
                ops.gemm(u, P.w(p + "fc2.w"), x, R, d, f, bias=P.f32(p + "fc2.b"), residual=x2,
                         rowsum2=lnst[l + 1, 0] if l + 1 < st.L else None)
            else:
                eng.linear(ctx, p + "co", x2, R, residual=x1, fp8=False)
                ops.layernorm_fwd(x2, P.f32(p + "ln_ff.g"), P.f32(p + "ln_ff.b"), eps, a, rows=R)
                eng.linear(a, p + "fc1", u, R, act=eng.gelu, fp8=False)
                eng.linear(u, p + "fc2", x, R, residual=x2, fp8=False)
        hf = eng.buf(ns + "g.hf", R, d)
        ops.layernorm_fwd(x, P.f32("dec.ln_f.g"), P.f32("dec.ln_f.b"), eng.dec_eps, hf, rows=R)
        cache["cache_index"] = cur + 1
        return eng.head_logits(hf, R, name=ns + "g.logits", stats=stats)

    # ------------------------------------------------------------------ construction (modeling:703-773)
    @classmethod
    def from_clip_vision_mbart_pretrained(cls, clip_vision_model_name_or_path: str = None, mbart_model_name_or_path: str = None,
                                          *model_args, **kwargs):
        """Builds the composite model from a CLIP-vision and an mBART checkpoint *directory* (local msgpack / config.json;
        there is no network here) or from `mbart_model=` / `clip_vision_model=` objects exposing `.config` and `.params`
        (modeling:730-758).  visual_projection and final_logits_bias keep their random/zero init (modeling:766-770)."""
        kwargs_mbart = {k[len("mbart_"):]: v for k, v in kwargs.items() if k.startswith("mbart_")}
        kwargs_clip = {k[len("clip_vision_"):]: v for k, v in kwargs.items() if k.startswith("clip_vision_")}
        for k in kwargs_mbart:
            del kwargs["mbart_" + k]
        for k in kwargs_clip:
            del kwargs["clip_vision_" + k]
        kwargs_mbart.pop("from_pt", None)
        from .checkpoint import load_component

        mbart_model = kwargs_mbart.pop("model", None)
        if mbart_model is None:
            assert mbart_model_name_or_path is not None, \
                "If `model` is not defined as an argument, a `mbart_model_name_or_path` has to be defined"
            mbart_model = load_component(mbart_model_name_or_path, kwargs_mbart.get("config"))
        clip_model = kwargs_clip.pop("model", None)
        if clip_model is None:
            assert clip_vision_model_name_or_path is not None, \
                "If `model` is not defined as an argument, a `clip_vision_model_name_or_path` has to be defined"
            clip_model = load_component(clip_vision_model_name_or_path, kwargs_clip.get("config"))
        dtype = kwargs.pop("dtype", torch.float32)
        seed = kwargs.pop("seed", 0)
        device = kwargs.pop("device", None)
        config = CLIPVisionMBartConfig.from_clip_vision_mbart_configs(clip_model.config, mbart_model.config, **kwargs)
        model = cls(config, *model_args, seed=seed, dtype=dtype, device=device)
        flat = model.store.export_flat("master")
        from .checkpoint import convert_pt_state_dict

        def leaves(comp, prefix, top):
            """component's Flax leaves (relative names); PyTorch checkpoints (`from_pt=True`, main.py:426) are converted
            against the leaf names this model owns under `prefix`.  Like HF `from_pretrained`, a head-model tree whose
            weights sit under the base-model prefix (FlaxMBartForConditionalGeneration: `model/decoder/...`,
            `model/shared/...` next to `final_logits_bias`; FlaxCLIPModel: `vision_model/...` next to `text_model/...`) is
            unwrapped when none of the expected top-level keys `top` is present."""
            if getattr(comp, "pt_state", None) is not None:
                expected = {k[len(prefix):] for k in flat if k.startswith(prefix)}
                return convert_pt_state_dict(comp.pt_state, expected)
            tree = comp.params
            if isinstance(tree, dict) and not any(t in tree for t in top) and isinstance(tree.get(cls.base_model_prefix), dict):
                tree = tree[cls.base_model_prefix]
            return flatten_tree(tree)

        def graft(comp, src_prefixes, dst_prefix, top, what):
            """copy the component's leaves into `flat`; every leaf this model owns under dst_prefix + src_prefix must be
            supplied with the right shape (the reference's params setter raises on missing keys, utils:112-116; a silent
            partial load would leave random-init weights behind)."""
            got = leaves(comp, dst_prefix, top)
            need = {k for k in flat if any(k.startswith(dst_prefix + sp) for sp in src_prefixes)}
            used = set()
            for k, v in got.items():
                if any(k.startswith(sp) for sp in src_prefixes) and dst_prefix + k in flat:
                    v = np.asarray(v)
                    if tuple(v.shape) != tuple(flat[dst_prefix + k].shape):
                        raise ValueError(f"{what} checkpoint: leaf {k} has shape {tuple(v.shape)}, the model expects "
                                         f"{tuple(flat[dst_prefix + k].shape)}")
                    flat[dst_prefix + k] = v
                    used.add(dst_prefix + k)
            missing = sorted(need - used)
            if missing:
                raise ValueError(f"{what} checkpoint supplies {len(used)} of the {len(need)} parameters the model needs; "
                                 f"missing e.g. {missing[:4]} (keys seen: {sorted(got)[:4]} ...)")

        graft(clip_model, ("vision_model/",), "model/encoder/", ("vision_model",), "CLIP vision")  # modeling:768
        # FlaxMBartModel tree: shared / encoder / decoder; the mBART text encoder is not used (modeling:769-770)
        graft(mbart_model, ("decoder/", "shared/"), "model/", ("decoder", "shared"), "mBART")
        model.store.load_flat(flat)
        model._params_cache = None
        return model
