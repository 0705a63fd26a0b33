"""Parameter store: the reference's Flax parameter pytree (`modeling:36-59, 123-135, 768-770`)
<-> flat device buffers laid out for the MI355X path.

Device layout (one flat fp32 master buffer; same offsets for the compute copy, gradients and AdamW moments):
  * Dense kernels are stored [out][in] (k-contiguous: the layout the MFMA GEMM streams without transposes);
    self-attention q/k/v are fused into one [3d][d] weight + [3d] bias, cross-attention k/v into [2d][d];
    the patch conv HWIO [ps,ps,3,hid] is stored [hid][ps*ps*3]; the shared embedding is zero-padded to a multiple of
    128 rows (V = 250054 -> 250112) so the tied LM-head GEMM needs no tail tiles.
  * Segments are ordered as backward produces their gradients (logits bias, decoder top->bottom, embedding,
    projection, ViT top->bottom) so gradient all-reduce buckets are contiguous slices that complete in order; every
    parameter whose gradient is accumulated with atomics (LayerNorm scale/bias, class/position embeddings) lives in one
    trailing region that a single memset clears (bias column sums included: no per-call memsets).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np
import torch

ALIGN = 64  # elements


def _rup(x: int, m: int) -> int:
    return (x + m - 1) // m * m


@dataclass
class Seg:
    name: str
    offset: int
    shape: Tuple[int, ...]  # device shape

    @property
    def numel(self) -> int:
        n = 1
        for s in self.shape:
            n *= s
        return n


def flatten_tree(tree, prefix="") -> Dict[str, np.ndarray]:
    out = {}
    for k, v in tree.items():
        key = f"{prefix}/{k}" if prefix else str(k)
        if isinstance(v, dict):
            out.update(flatten_tree(v, key))
        else:
            out[key] = v
    return out


def unflatten_tree(flat: Dict[str, np.ndarray]):
    tree: Dict = {}
    for k, v in flat.items():
        parts = k.split("/")
        d = tree
        for p in parts[:-1]:
            d = d.setdefault(p, {})
        d[parts[-1]] = v
    return tree


V_ = "model/encoder/vision_model/"
D_ = "model/decoder/"


class ParamStore:
    def __init__(self, config, dtype: torch.dtype, device, allocate: bool = True):
        """allocate=False: the segment layout only (offsets / shapes / order), no buffers — what the bucket planner needs"""
        self.cfg, self.dtype, self.device = config, dtype, device
        mc, vc = config.mbart_config, config.clip_vision_config
        self.d, self.ffn, self.L, self.H = mc.d_model, mc.decoder_ffn_dim, mc.decoder_layers, mc.decoder_attention_heads
        self.vd, self.vffn, self.vL, self.vH = vc.hidden_size, vc.intermediate_size, vc.num_hidden_layers, vc.num_attention_heads
        self.V, self.Vpad = mc.vocab_size, _rup(mc.vocab_size, 128)
        self.ps, self.img = vc.patch_size, vc.image_size
        self.S = (self.img // self.ps) ** 2 + 1
        self.npos = mc.max_position_embeddings + 2
        self.segs: Dict[str, Seg] = {}
        self._order: List[str] = []
        off = 0

        def add(name, shape):
            nonlocal off
            self.segs[name] = Seg(name, off, tuple(shape))
            self._order.append(name)
            off = _rup(off + self.segs[name].numel, ALIGN)

        d, f, vd, vf = self.d, self.ffn, self.vd, self.vffn
        # ---- dense region (GEMM-written weight gradients), in backward-completion order
        add("flb", (self.Vpad,))
        add("shared", (self.Vpad, d))  # LM-head part is final right after the head dW GEMM; the sparse input-embedding rows are
        #                                added after the dense all-reduce (see Trainer / Engine.defer_embed)
        # the cross-attention k/v projections of ALL layers read the same encoder states and nothing inside the decoder feeds
        # them: they sit side by side ([L*2d][d], layer 0 first) so that forward (ehs W^T), dX (sum over layers) and dW are ONE
        # GEMM each instead of L; their gradients complete with the end of decoder backward, hence behind layer 0's segments
        dec_lin = (("fc2", (d, f)), ("fc1", (f, d)), ("co", (d, d)), ("cq", (d, d)), ("so", (d, d)), ("qkv", (3 * d, d)))
        vit_lin = (("fc2", (vd, vf)), ("fc1", (vf, vd)), ("o", (vd, vd)), ("qkv", (3 * vd, vd)))
        for l in reversed(range(self.L)):
            for n, shp in dec_lin:
                add(f"dec{l}.{n}.w", shp)
        for l in range(self.L):
            add(f"dec{l}.ckv.w", (2 * d, d))
        add("vp.w", (d, vd))
        for l in reversed(range(self.vL)):
            for n, shp in vit_lin:
                add(f"vit{l}.{n}.w", shp)
        add("patch.w", (vd, self.ps * self.ps * 3))
        self.atomic_begin = off
        # ---- gradients accumulated with atomics (bias column sums, LayerNorm scale/bias, class/position embeddings):
        #      one trailing region, cleared by one memset per step
        for l in reversed(range(self.L)):
            for n, shp in dec_lin:
                add(f"dec{l}.{n}.b", (shp[0],))
        for l in range(self.L):
            add(f"dec{l}.ckv.b", (2 * d,))
        add("vp.b", (d,))
        for l in reversed(range(self.vL)):
            for n, shp in vit_lin:
                add(f"vit{l}.{n}.b", (shp[0],))
        add("dec.ln_f.g", (d,)); add("dec.ln_f.b", (d,))
        for l in range(self.L):
            for n in ("ln_sa", "ln_ca", "ln_ff"):
                add(f"dec{l}.{n}.g", (d,)); add(f"dec{l}.{n}.b", (d,))
        add("dec.ln_emb.g", (d,)); add("dec.ln_emb.b", (d,))
        add("dec.pos", (self.npos, d))
        for l in range(self.vL):
            for n in ("ln1", "ln2"):
                add(f"vit{l}.{n}.g", (vd,)); add(f"vit{l}.{n}.b", (vd,))
        add("vit.pre_ln.g", (vd,)); add("vit.pre_ln.b", (vd,))
        add("vit.post_ln.g", (vd,)); add("vit.post_ln.b", (vd,))
        add("vit.pos", (self.S, vd))
        add("vit.cls", (vd,))
        self.numel = _rup(off, 256)
        if not allocate:
            self.master = self.lp = None
            self.grad = self.m = self.v = None
            return
        self.master = torch.zeros(self.numel, dtype=torch.float32, device=device)
        self.lp = self.master if dtype == torch.float32 else torch.zeros(self.numel, dtype=dtype, device=device)
        self.grad = None
        self.m = None
        self.v = None

    # ------------------------------------------------------------------ views
    def _view(self, kind: str, buf: torch.Tensor, name: str) -> torch.Tensor:
        """view of segment `name` inside flat buffer `buf`, made once per (buffer, segment): a train step asks for ~1500 of them and
        a slice + view costs ~5 us of host time each (the views alias the flat buffers, so they never go stale; a re-allocated
        buffer — ensure_grads — is a different object and gets new views)"""
        c = self.__dict__.setdefault("_vcache", {})
        hit = c.get((kind, name))
        if hit is not None and hit[0] is buf:
            return hit[1]
        s = self.segs[name]
        v = buf[s.offset: s.offset + s.numel].view(s.shape)
        c[(kind, name)] = (buf, v)
        return v

    def w(self, name: str) -> torch.Tensor:
        """compute-dtype view (what the kernels read)"""
        return self._view("w", self.lp, name)

    def f32(self, name: str) -> torch.Tensor:
        return self._view("f", self.master, name)

    def g(self, name: str) -> torch.Tensor:
        return self._view("g", self.grad, name)

    def ckv_cat(self, which: str = "w"):
        """(weights [L*2d][d], biases [L*2d]) of the cross-attention k/v projections of all layers as ONE matrix: `which` = "w"
        (compute copy + fp32 bias) or "g" (gradients)"""
        a, b = self.segs["dec0.ckv.w"], self.segs[f"dec{self.L - 1}.ckv.w"]
        ba, bb = self.segs["dec0.ckv.b"], self.segs[f"dec{self.L - 1}.ckv.b"]
        n = self.L * 2 * self.d
        assert b.offset + b.numel - a.offset == n * self.d and bb.offset + bb.numel - ba.offset == n, "ckv segments are not contiguous"
        wbuf, bbuf = (self.lp, self.master) if which == "w" else (self.grad, self.grad)
        c = self.__dict__.setdefault("_vcache", {})
        hit = c.get(("ckv", which))
        if hit is not None and hit[0] is wbuf and hit[1] is bbuf:
            return hit[2]
        out = (wbuf[a.offset: a.offset + n * self.d].view(n, self.d), bbuf[ba.offset: ba.offset + n])
        c[("ckv", which)] = (wbuf, bbuf, out)
        return out

    def ensure_grads(self):
        if self.grad is None:
            self.grad = torch.zeros(self.numel, dtype=torch.float32, device=self.device)

    def ensure_opt_state(self):
        if self.m is None:
            self.m = torch.zeros(self.numel, dtype=torch.float32, device=self.device)
            self.v = torch.zeros(self.numel, dtype=torch.float32, device=self.device)

    def refresh_lp(self):
        self.version = getattr(self, "version", 0) + 1  # anything derived from the weights (folded / quantised copies) is stale
        if self.lp is not self.master:
            from . import ops
            ops.cast(self.master, self.lp)

    # ------------------------------------------------------------------ Flax pytree <-> device layout
    def flax_shapes(self) -> Dict[str, Tuple[int, ...]]:
        """Required leaves of the reference's params pytree ('/'-joined)."""
        s: Dict[str, Tuple[int, ...]] = {}
        d, f, vd, vf = self.d, self.ffn, self.vd, self.vffn
        s["final_logits_bias"] = (1, self.V)
        s["model/shared/embedding"] = (self.V, d)
        s["model/visual_projection/kernel"] = (vd, d)
        s["model/visual_projection/bias"] = (d,)
        s[V_ + "embeddings/class_embedding"] = (vd,)
        s[V_ + "embeddings/patch_embedding/kernel"] = (self.ps, self.ps, 3, vd)
        s[V_ + "embeddings/position_embedding/embedding"] = (self.S, vd)
        for ln in ("pre_layrnorm", "post_layernorm"):
            s[V_ + ln + "/scale"] = (vd,); s[V_ + ln + "/bias"] = (vd,)
        for i in range(self.vL):
            Lp = f"{V_}encoder/layers/{i}/"
            for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
                s[Lp + f"self_attn/{n}/kernel"] = (vd, vd); s[Lp + f"self_attn/{n}/bias"] = (vd,)
            for ln in ("layer_norm1", "layer_norm2"):
                s[Lp + ln + "/scale"] = (vd,); s[Lp + ln + "/bias"] = (vd,)
            s[Lp + "mlp/fc1/kernel"] = (vd, vf); s[Lp + "mlp/fc1/bias"] = (vf,)
            s[Lp + "mlp/fc2/kernel"] = (vf, vd); s[Lp + "mlp/fc2/bias"] = (vd,)
        s[D_ + "embed_positions/embedding"] = (self.npos, d)
        for ln in ("layernorm_embedding", "layer_norm"):
            s[D_ + ln + "/scale"] = (d,); s[D_ + ln + "/bias"] = (d,)
        for i in range(self.L):
            Lp = f"{D_}layers/{i}/"
            for blk in ("self_attn", "encoder_attn"):
                for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
                    s[Lp + f"{blk}/{n}/kernel"] = (d, d); s[Lp + f"{blk}/{n}/bias"] = (d,)
                s[Lp + blk + "_layer_norm/scale"] = (d,); s[Lp + blk + "_layer_norm/bias"] = (d,)
            s[Lp + "fc1/kernel"] = (d, f); s[Lp + "fc1/bias"] = (f,)
            s[Lp + "fc2/kernel"] = (f, d); s[Lp + "fc2/bias"] = (d,)
            s[Lp + "final_layer_norm/scale"] = (d,); s[Lp + "final_layer_norm/bias"] = (d,)
        return s

    def _mapping(self):
        """(device seg, [(flax leaf, transform)]) pairs.  transform: 'T' transpose [in,out]->[out,in]; 'id'; row-block
        index for fused weights."""
        m = []
        d = self.d
        m.append(("flb", [("final_logits_bias", "flb")]))
        m.append(("shared", [("model/shared/embedding", "pad_rows")]))
        m.append(("vp.w", [("model/visual_projection/kernel", "T")]))
        m.append(("vp.b", [("model/visual_projection/bias", "id")]))
        m.append(("vit.cls", [(V_ + "embeddings/class_embedding", "id")]))
        m.append(("patch.w", [(V_ + "embeddings/patch_embedding/kernel", "conv")]))
        m.append(("vit.pos", [(V_ + "embeddings/position_embedding/embedding", "id")]))
        for a, b in (("pre_ln", "pre_layrnorm"), ("post_ln", "post_layernorm")):
            m.append((f"vit.{a}.g", [(V_ + b + "/scale", "id")])); m.append((f"vit.{a}.b", [(V_ + b + "/bias", "id")]))
        for i in range(self.vL):
            Lp, p = f"{V_}encoder/layers/{i}/", f"vit{i}."
            m.append((p + "qkv.w", [(Lp + f"self_attn/{n}/kernel", "T") for n in ("q_proj", "k_proj", "v_proj")]))
            m.append((p + "qkv.b", [(Lp + f"self_attn/{n}/bias", "id") for n in ("q_proj", "k_proj", "v_proj")]))
            m.append((p + "o.w", [(Lp + "self_attn/out_proj/kernel", "T")])); m.append((p + "o.b", [(Lp + "self_attn/out_proj/bias", "id")]))
            for a, b in (("ln1", "layer_norm1"), ("ln2", "layer_norm2")):
                m.append((p + a + ".g", [(Lp + b + "/scale", "id")])); m.append((p + a + ".b", [(Lp + b + "/bias", "id")]))
            for n in ("fc1", "fc2"):
                m.append((p + n + ".w", [(Lp + f"mlp/{n}/kernel", "T")])); m.append((p + n + ".b", [(Lp + f"mlp/{n}/bias", "id")]))
        m.append(("dec.pos", [(D_ + "embed_positions/embedding", "id")]))
        for a, b in (("ln_emb", "layernorm_embedding"), ("ln_f", "layer_norm")):
            m.append((f"dec.{a}.g", [(D_ + b + "/scale", "id")])); m.append((f"dec.{a}.b", [(D_ + b + "/bias", "id")]))
        for i in range(self.L):
            Lp, p = f"{D_}layers/{i}/", f"dec{i}."
            m.append((p + "qkv.w", [(Lp + f"self_attn/{n}/kernel", "T") for n in ("q_proj", "k_proj", "v_proj")]))
            m.append((p + "qkv.b", [(Lp + f"self_attn/{n}/bias", "id") for n in ("q_proj", "k_proj", "v_proj")]))
            m.append((p + "so.w", [(Lp + "self_attn/out_proj/kernel", "T")])); m.append((p + "so.b", [(Lp + "self_attn/out_proj/bias", "id")]))
            m.append((p + "cq.w", [(Lp + "encoder_attn/q_proj/kernel", "T")])); m.append((p + "cq.b", [(Lp + "encoder_attn/q_proj/bias", "id")]))
            m.append((p + "ckv.w", [(Lp + f"encoder_attn/{n}/kernel", "T") for n in ("k_proj", "v_proj")]))
            m.append((p + "ckv.b", [(Lp + f"encoder_attn/{n}/bias", "id") for n in ("k_proj", "v_proj")]))
            m.append((p + "co.w", [(Lp + "encoder_attn/out_proj/kernel", "T")])); m.append((p + "co.b", [(Lp + "encoder_attn/out_proj/bias", "id")]))
            for a, b in (("ln_sa", "self_attn_layer_norm"), ("ln_ca", "encoder_attn_layer_norm"), ("ln_ff", "final_layer_norm")):
                m.append((p + a + ".g", [(Lp + b + "/scale", "id")])); m.append((p + a + ".b", [(Lp + b + "/bias", "id")]))
            for n in ("fc1", "fc2"):
                m.append((p + n + ".w", [(Lp + f"{n}/kernel", "T")])); m.append((p + n + ".b", [(Lp + f"{n}/bias", "id")]))
        return m

    def _buffer(self, which: str) -> torch.Tensor:
        buf = {"master": self.master, "grad": self.grad, "m": self.m, "v": self.v}[which]
        if buf is None:
            raise ValueError(f"ParamStore: the '{which}' buffer has not been allocated")
        return buf

    def _seg_view(self, which: str, name: str) -> torch.Tensor:
        s = self.segs[name]
        return self._buffer(which)[s.offset: s.offset + s.numel].view(s.shape)

    def load_flat(self, flat: Dict[str, np.ndarray], target: str = "master") -> None:
        """flat: {'/'-joined flax leaf: array-like} in the reference layout -> device buffers
        (target: master | m | v — the AdamW moments share the parameter layout)."""
        for seg_name, parts in self._mapping():
            dst = self._seg_view(target, seg_name)
            r0 = 0
            for leaf, tr in parts:
                a = flat[leaf]
                t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.asarray(a, dtype=np.float32))
                t = t.to(torch.float32)
                if tr == "T":
                    t = t.T
                elif tr == "conv":
                    t = t.reshape(-1, t.shape[-1]).T  # [ps*ps*3, hid] -> [hid, ps*ps*3]
                elif tr == "flb":
                    t = t.reshape(-1)
                n = t.shape[0]
                dst[r0: r0 + n].copy_(t.to(self.device), non_blocking=False)
                r0 += n
        if target == "master":
            self.refresh_lp()

    def export_flat(self, source: str = "master") -> Dict[str, np.ndarray]:
        """device buffers -> {'/'-joined flax leaf: np.ndarray} in the reference layout (source: master | grad | m | v)."""
        out: Dict[str, np.ndarray] = {}
        shapes = self.flax_shapes()
        for seg_name, parts in self._mapping():
            src = self._seg_view(source, seg_name).detach().cpu()
            r0 = 0
            for leaf, tr in parts:
                shp = shapes[leaf]
                if tr == "T":
                    n = shp[1]
                    t = src[r0: r0 + n].T
                elif tr == "conv":
                    n = shp[-1]
                    t = src[r0: r0 + n].T.reshape(shp)
                elif tr == "flb":
                    n = self.V
                    t = src[: self.V].reshape(1, self.V)
                elif tr == "pad_rows":
                    n = self.V
                    t = src[: self.V]
                else:
                    n = shp[0]
                    t = src[r0: r0 + n]
                out[leaf] = np.ascontiguousarray(t.numpy())
                r0 += n
        return out

    def init_random(self, seed: int = 0, std: float = 0.02) -> None:
        """normal(std) Dense/Embed kernels, LayerNorm scale 1 / bias 0, zero biases — generated on device."""
        g = torch.Generator(device=self.device).manual_seed(seed)
        self.master.zero_()
        for name, s in self.segs.items():
            v = self.f32(name)
            if name.endswith(".w") or name in ("vit.pos", "vit.cls", "dec.pos"):
                v.normal_(0.0, std, generator=g)
            elif name == "shared":
                v[: self.V].normal_(0.0, std, generator=g)
            elif name.endswith(".g"):
                v.fill_(1.0)
        self.refresh_lp()
