"""Kernel sequencing for the captioning hot path: ViT encoder -> projection -> mBART decoder -> tied head, forward
(+loss) and the hand-derived backward.  Pure orchestration: every tensor op is a libmic_hip.so kernel (ops.*);
torch supplies device memory and the current HIP stream only.

Module graph restated from `modeling_clip_vision_mbart.py:67-115, 146-192` (composition) and SURVEY Appendix B
(third-party block structure).  Activations live in row-padded [rows_pad][width] buffers (rows_pad = multiple of 128,
pad rows stay zero) so weight-gradient GEMMs can reduce over the padded row count.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

from . import _lib as L
from . import ops
from .params import ParamStore, _rup

ROWPAD = 128


def _mix(seed: int, site: int) -> int:
    x = (seed * 0x9E3779B1 + site * 0x85EBCA6B + 0x27D4EB2F) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x2C1B3C6D) & 0xFFFFFFFF
    x ^= x >> 12
    return x & 0xFFFFFFFF


class Engine:
    def __init__(self, store: ParamStore):
        self.P = store
        self.dt = store.dtype
        self.dev = store.device
        self._bufs: Dict[str, torch.Tensor] = {}
        mc = store.cfg.mbart_config
        self.gelu = L.ACT_IDS[getattr(mc, "gelu_variant", "tanh")]
        self.dec_eps = float(getattr(mc, "decoder_ln_eps", 1e-6))
        self.vit_eps = float(store.cfg.clip_vision_config.layer_norm_eps)
        self.p_drop = float(mc.dropout)
        self.embed_scale = math.sqrt(store.d) if mc.scale_embedding else 1.0
        self.defer_embed = False  # True: leave the sparse tied-embedding rows (ids, dh0) to the caller (data parallel)
        self.embed_rows = None     # (ids, dh0 buffer, M) of the last backward when deferred
        self._dw_queue = []
        self._cs_queue = []
        self.on_free = []  # callables run by free_buffers() before the buffers go (decode plans drop their captured graphs)
        self.fp8 = False
        import os
        self.use_head_stats = os.environ.get("MIC_HEAD_STATS", "1") != "0"  # softmax partials out of the LM-head GEMM (A/B switch)
        self.grad_progress = None  # callable(offset): every gradient with flat offset < offset is final (DDP overlap)
        self.on_embed_rows = None  # callable(): single process, right behind the input-embedding scatter at the end of decoder backward
        # weight-gradient GEMMs on a second stream (bf16 / f32 GEMM modes): nothing but the optimizer depends on dW, so a layer's
        # grouped dW launch runs beside the NEXT layer's dX chain (LayerNorm / attention backward and the small dX GEMMs leave
        # most CUs idle).  The gradient operands dW reads are double-buffered by layer parity; A/B switch MIC_DW_OVERLAP=0|1.
        self.dw_overlap = self.dev.type == "cuda" and os.environ.get("MIC_DW_OVERLAP", "1") != "0"
        # decoder-step LayerNorms folded around the GEMMs (bfloat16 generate path): A/B switch
        self.decode_ln_fold = self.dev.type == "cuda" and os.environ.get("MIC_DECODE_LNFOLD", "1") != "0"
        self._ckv_hoist = os.environ.get("MIC_CKV_HOIST", "1") != "0"
        self._lnf = {}
        self._lnf_version = -1
        self._dw_stream = None
        # LM-head backward as NT launches of the four-wave kernel (k-contiguous copies: dlogits^T from the CE backward, h^T, E^T);
        # MIC_HEAD_NT=0: the k-major launches of rounds 1-4 (A/B)
        import os as _os
        self.head_nt = _os.environ.get("MIC_HEAD_NT", "1") != "0"
        # LayerNorm backward: gamma / beta gradients as per-block partial sums reduced later with the layer's weight-gradient launches
        # (no atomics on the critical path, a fixed summation order); MIC_LN_PARTIALS=0: fp32 atomics inside the kernel (A/B)
        self.ln_partials = _os.environ.get("MIC_LN_PARTIALS", "1") != "0"
        self._lnp_queue = []
        self._head_nt_state = None   # (dlogits^T, Kp) of the CE backward that has just run
        self.fp8_head = 0            # fp8 GEMM mode: the tied head's GEMMs on fp8 operands too (set_gemm_dtype)
        self._head8_state = None
        self._head_x8 = None
        self._ET_version, self._ET_event = -1, None
        self._dw_events = []
        # weight-gradient launches behind every n-th layer: 2 in fp8 mode (its two GEMM streams alternate rather than overlap — half as
        # many hand-overs: 14.87 -> 14.75 ms), 1 in bf16 (2 measured level); MIC_DW_EVERY=<n> forces a value (A/B)
        self.dw_every = int(os.environ.get("MIC_DW_EVERY", "0"))
        self.dw_late_flush = os.environ.get("MIC_DW_LATE_FLUSH", "1") != "0"  # see dw_fence (A/B: 0 = the weight-gradient launch in front of that LayerNorm backward)
        self._dw_side = False  # True while launching on the dW stream
        self._ckv_block_name = None  # set while the all-layer cross k/v weight gradient (one launch) sits in the dW queue

    # ------------------------------------------------------------------ fp8 GEMM operands (BASELINE configs[4])
    FP8_KINDS = ("qkv", "cq", "ckv", "fc1", "fc2")  # the QKV and FFN projections of both towers; out-projections and the head stay bf16

    def set_gemm_dtype(self, name, scaling: str = "delayed", head: Optional[str] = None):
        """"fp8": the QKV / FFN projections run as OCP fp8 GEMMs — e4m3 activations and weights, e5m2 gradients, one scale per
        tensor, fp32 accumulate (forward, dX and dW).  None / "bf16": the storage dtype.
        scaling="current": every activation / gradient tensor is scaled by its own absolute maximum (two passes over the
        tensor: amax, then quantise).  scaling="delayed" (default; the usual production recipe): the scale comes from the
        amax the same tensor had in the previous pass, the quantiser records the new amax on the way (one pass); values
        that outgrow the old amax saturate at +-FMAX for that one pass.  A tensor seen for the first time is scaled by its
        current amax.  Weights are re-quantised once per optimizer step — with their current amax ("current"), or ("delayed")
        with the amax of the previous step's weights, recorded by the previous quantiser pass: an optimizer step moves a weight
        by ~lr, so the one-step-old maximum is the current one to fp8 precision and the amax pass over 246 M parameters goes;
        weights that arrive any other way (params setter, checkpoint restore) start again from their current amax.
        Under delayed scaling a tensor WITH a scale history is not quantised by a launch of its own: its producer (LayerNorm forward /
        backward, the GELU / dGELU epilogue of the fp8 GEMM, attention backward) writes the fp8 bytes (`_q8_target`, `fp8_fused`);
        the weight copies of the next pass are made on a side stream behind the optimizer (`fp8_refresh_weights`).
        head = "all" | "bwd" | "0" (default: MIC_FP8_HEAD, else "all"): the tied LM head's three GEMMs / its two backward GEMMs / none
        of them on fp8 operands (see below)."""
        if name in (None, "bf16", "bfloat16", "f32", "float32"):
            self.fp8 = False
            return
        if name != "fp8":
            raise ValueError(f"unknown gemm_dtype {name!r}")
        if scaling not in ("delayed", "current"):
            raise ValueError(f"fp8 scaling must be 'delayed' or 'current', got {scaling!r}")
        if self.dt != torch.bfloat16:
            raise ValueError("gemm_dtype='fp8' needs the bfloat16 storage mode (dtype=bfloat16)")
        import os as _os
        P = self.P
        self.fp8 = True
        self.fp8_scaling = scaling
        # the cross-attention k/v projections of all layers as ONE fp8 matrix "ckvcat" (one scale) when they run hoisted (ckv_hoisted)
        kinds = tuple(k for k in self.FP8_KINDS if not (k == "ckv" and self._ckv_hoist))
        names = [f"dec{l}.{k}" for l in range(P.L) for k in kinds] + [f"vit{l}.{k}" for l in range(P.vL) for k in ("qkv", "fc1", "fc2")]
        if self._ckv_hoist:
            names.append("ckvcat")
        # the tied LM head in fp8 too (MIC_FP8_HEAD = all | bwd | 0; default all): its three GEMMs are 3.7 of the step's 8.4 TFLOP.
        # Backward: dE and dX read ONE e5m2 copy of dlogits written by the CE backward under a closed-form scale (`mic_ce_bwd_q8`; no
        # in-place gradient, no transposed copy), the e4m3 copies of E / E^T (re-made with the other weights) and the e4m3 final hidden
        # states; forward ("all"): the logits from the e4m3 operands, softmax partials as in bf16.  "bwd": forward stays bf16; "0": the
        # head as in the bf16 mode.  Needs the reduction dimensions (d, Vpad) in whole 128-byte fragments.
        mode = head if head is not None else _os.environ.get("MIC_FP8_HEAD", "all")
        if mode not in ("0", "bwd", "all"):
            raise ValueError(f"MIC_FP8_HEAD / head={mode!r}: 0 | bwd | all")
        self.fp8_head = {"0": 0, "bwd": 1, "all": 2}[mode] if (P.d % 128 == 0 and P.Vpad % 128 == 0) else 0
        if self.fp8_head:
            names.append("shared")
        self._head8_state = None  # (dlogits as e5m2 [rows][Vpad], its scale state) of the CE backward that has just run
        self._head_x8 = None      # (final hidden states as e4m3, state) of this pass
        self._w8 = {}
        self._w8_state = torch.zeros((len(names), 2), dtype=torch.float32, device=self.dev)
        for i, n in enumerate(names):
            N, K = self._w8_src(n).shape
            self._w8[n] = (torch.empty((N, K), dtype=torch.float8_e4m3fn, device=self.dev),   # [out][in]: forward  x W^T
                           torch.empty((K, N), dtype=torch.float8_e4m3fn, device=self.dev),   # [in][out]: dX = dy W
                           self._w8_state[i])
        self._w8_stale = True
        self._w8_part = torch.zeros((len(names), ops.fp8_amax_partials()), dtype=torch.float32, device=self.dev) if scaling == "delayed" else None
        self._w8_seen = False  # delayed scaling: _w8_state carries the amax recorded by the previous quantiser pass
        # delayed scaling: the optimizer re-quantises a weight right behind its AdamW pass (`fp8_requantize_range`, on the optimizer's
        # stream, under backward) — names done since the last pass; when that is all of them the next pass only rolls the amax
        self._w8_fresh = set()
        # (opt-in, MIC_FP8_REQUANT_OPT=1: measured 0.03 - 0.2 ms per step, inside the noise, for ~8 more launches per step — the default
        # re-quantises all weights in 11 launches at the start of the next pass)
        self._w8_requant_opt = _os.environ.get("MIC_FP8_REQUANT_OPT", "0") == "1"
        self._w8_async = _os.environ.get("MIC_FP8_W8_ASYNC", "1") != "0"  # (A/B: 0 = weights re-quantised at the start of the next pass, on its stream)
        self._w8_event = None
        self._w8_rolled = False  # this step's roll of the weights' amax has been issued (first of the two refresh calls)
        self._w8_index = {n: i for i, n in enumerate(names)}
        segs = P.segs
        def span(n):
            if n == "ckvcat":
                return segs["dec0.ckv.w"].offset, segs[f"dec{P.L - 1}.ckv.w"].offset + segs[f"dec{P.L - 1}.ckv.w"].numel
            sg = segs["shared" if n == "shared" else n + ".w"]
            return sg.offset, sg.offset + sg.numel
        self._w8_span = {n: span(n) for n in names}
        # one (amax, 1/scale) slot per quantised activation / gradient tensor; delayed scaling: + its table of partial maxima
        self._a8_state = torch.zeros((1024, 2), dtype=torch.float32, device=self.dev)
        self._a8_part = torch.zeros((1024, ops.fp8_amax_partials()), dtype=torch.float32, device=self.dev) if scaling == "delayed" else None
        self._a8_slots: Dict[str, int] = {}
        self._a8_ready = set()   # delayed scaling: tags whose slot carries the previous pass's amax
        self._a8_cache: Dict = {}
        self._dw8_queue = []
        self._cs8_queue = []     # bias gradients of the fp8 projections: column sums over the e5m2 bytes of dy
        self._x8: Dict[str, tuple] = {}  # wname -> (q, state) of the Linear's input as quantised in the training forward (dW reads it)
        # fused emission: under delayed scaling the PRODUCER of an operand (LayerNorm, GELU / dGELU epilogue, attention backward)
        # writes the fp8 bytes itself once the tensor has a scale history (from its second pass on); MIC_FP8_FUSED=0: every operand
        # through mic_fp8_quantize again (A/B)
        self.fp8_fused = scaling == "delayed" and self.ln_partials and _os.environ.get("MIC_FP8_FUSED", "1") != "0"

    def _w8_src(self, name: str):
        """the bf16 matrix behind the fp8 weight entry `name`"""
        if name == "shared":
            return self.P.w("shared")
        return self.P.ckv_cat("w")[0] if name == "ckvcat" else self.P.w(name + ".w")

    def storage_dtype_gemms(self):
        """context: every Linear runs in the storage dtype (bf16) even when the trainer switched the projections to fp8 — the
        inference entry points (`encode`, `decode`, `generate`) are not part of a train / eval pass: nothing re-quantises the
        weights for them or rolls the activation scales, so fp8 launches there would read stale (or never written) copies"""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            keep = self.fp8
            self.fp8 = False
            try:
                yield
            finally:
                self.fp8 = keep
        return ctx()

    def fp8_weights_changed(self, by_optimizer: bool = False):
        self._w8_stale = True
        if not by_optimizer:
            self._w8_seen = False
            self._w8_fresh = set()

    def _w8_items_all(self):
        """the 84 weight items of a delayed-scaling quantiser pass, built once (raw pointers into persistent buffers: the flat bf16
        parameter copy, the fp8 copies, the scale slots) — building them was 0.2 ms of host time in the gap between two steps"""
        key = (self.P.lp.data_ptr(), id(self._w8))
        if getattr(self, "_w8_items_key", None) != key:
            self._w8_items = [ops.fp8_item(self._w8_src(n), q.shape[0], q.shape[1], st, torch.float8_e4m3fn, q=q, qT=qT,
                                           amax_next=self._w8_part[i]) for i, (n, (q, qT, st)) in enumerate(self._w8.items())]
            self._w8_items_key = key
        return self._w8_items

    def fp8_refresh_weights(self, side, upto: Optional[int] = None, wait_event=None):
        """Trainer (delayed scaling): roll the weights' amax and re-quantise them for the NEXT pass on `side` instead of in front of that
        pass.  Two calls per step: `upto` = flat offset up to which the optimizer passes have been issued (`wait_event`: recorded on
        the optimizer's stream behind them) — at the end of decoder backward: the decoder's weights go now, under the ViT's backward;
        then the final call (upto None), behind everything the current stream has enqueued, takes the rest beside the host-issued glue
        between two steps.  The first fp8 GEMM of the next pass waits for the final call's event (`_w8_wait`)."""
        if not (self.fp8 and self.fp8_scaling == "delayed" and self._w8_seen and self._w8_async):
            return
        items = self._w8_items_all()
        names = list(self._w8)
        if upto is not None:
            todo = [i for i, n in enumerate(names) if self._w8_span[n][1] <= upto and n not in self._w8_fresh]
            if not todo:
                return
            with torch.cuda.stream(side):
                if wait_event is not None:
                    side.wait_event(wait_event)
                with ops.pinned_stream():
                    if not self._w8_rolled:
                        ops.fp8_roll_amax(self._w8_state, self._w8_part, len(names))
                        self._w8_rolled = True
                    ops.fp8_quantize([items[i] for i in todo], amax_pass=False)
            self._w8_fresh.update(names[i] for i in todo)
            return
        if not self._w8_stale:
            return
        todo = [i for i, n in enumerate(names) if n not in self._w8_fresh]
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            side.wait_event(ev)
            with ops.pinned_stream():
                if not self._w8_rolled:
                    ops.fp8_roll_amax(self._w8_state, self._w8_part, len(names))
                if todo:
                    ops.fp8_quantize([items[i] for i in todo], amax_pass=False)
            done = torch.cuda.Event()
            done.record(side)
        self._w8_event, self._w8_stale, self._w8_fresh, self._w8_rolled = done, False, set(), False

    def _w8_wait(self):
        if self._w8_event is not None:
            torch.cuda.current_stream().wait_event(self._w8_event)
            self._w8_event = None

    def fp8_requantize_range(self, b: int, e: int):
        """Optimizer side (delayed scaling): the weights whose flat segment ENDS in (b, e] have just been updated — the buckets arrive
        in layout order on one stream, so everything before `e` is final — re-quantise them now, on the caller's stream (beside
        backward), under the amax the step's begin rolled in; the next pass then finds its fp8 weight copies ready."""
        if not (self.fp8 and self.fp8_scaling == "delayed" and self._w8_seen and self._w8_requant_opt):
            return
        # (the tied embedding's flagged rows get their optimizer pass at the end of backward: its copies are left to fp8_refresh_weights)
        todo = [n for n, (o, oe) in self._w8_span.items() if b < oe <= e and n not in self._w8_fresh and n != "shared"]
        if not todo:
            return
        items = []
        for n in todo:
            q, qT, st = self._w8[n]
            items.append(ops.fp8_item(self._w8_src(n), q.shape[0], q.shape[1], st, torch.float8_e4m3fn, q=q, qT=qT,
                                      amax_next=self._w8_part[self._w8_index[n]]))
        ops.fp8_quantize(items, amax_pass=False)
        self._w8_fresh.update(todo)

    def _fp8_begin_pass(self):
        """start of a forward(+backward) pass: re-quantise the weights if the optimizer moved them; current scaling: clear the
        activation slots; delayed scaling: last pass's recorded amax becomes this pass's scale source"""
        if not self.fp8:
            return
        if self._w8_stale and self.fp8_scaling == "delayed" and self._w8_seen and len(self._w8_fresh) == len(self._w8):
            # every weight was re-quantised behind its optimizer pass: only the amax recorded there becomes current
            ops.fp8_roll_amax(self._w8_state, self._w8_part, len(self._w8))
            self._w8_stale = False
        self._w8_fresh = set()
        if self._w8_stale:
            P = self.P
            delayed = self.fp8_scaling == "delayed"
            if delayed and self._w8_seen:
                ops.fp8_roll_amax(self._w8_state, self._w8_part, len(self._w8))
            else:
                ops.zero(self._w8_state)
                if delayed:
                    ops.zero(self._w8_part)
            self._w8_wait()  # (a side-stream refresh of older weights may still be writing the copies)
            items = self._w8_items_all() if delayed else [ops.fp8_item(self._w8_src(n), q.shape[0], q.shape[1], st, torch.float8_e4m3fn, q=q, qT=qT)
                                                          for (n, (q, qT, st)) in self._w8.items()]
            ops.fp8_quantize(items, amax_pass=not (delayed and self._w8_seen))
            self._w8_seen = delayed
            self._w8_stale = False
        n = len(self._a8_slots)
        if self.fp8_scaling == "delayed":
            if n:
                ops.fp8_roll_amax(self._a8_state, self._a8_part, n)
            self._a8_ready = set(self._a8_slots)
        else:
            ops.zero(self._a8_state[: max(n, 1)])
        self._a8_cache = {}
        self._x8 = {}
        self._head_x8 = None

    def _fp8_ok(self, wname: str) -> bool:
        return self.fp8 and wname.split(".")[-1] in self.FP8_KINDS and wname in self._w8

    def _a8_slot(self, tag: str) -> int:
        i = self._a8_slots.get(tag)
        if i is None:
            i = self._a8_slots[tag] = len(self._a8_slots)
            if i >= self._a8_state.shape[0]:
                raise RuntimeError("fp8: out of activation scale slots")
        return i

    def _quant(self, x, rows: int, cols: int, tag: str, buf_tag: str, fmt, cache: bool = False):
        """(q [rows][cols] fp8, state) of bf16 x through mic_fp8_quantize (the unfused path: tensors without a scale history, operands
        no kernel of ours produces in fp8).  No transposed copy: the weight-gradient GEMM reads its operands k-major.  cache=True: x
        keeps its contents for the rest of the pass (the encoder states every layer's cross-attention projects): quantised once."""
        key = (x.data_ptr(), rows, cols, fmt)
        hit = self._a8_cache.get(key) if cache else None
        if hit is not None:
            return hit
        # sized by the CAPACITY of x (its row count is the step-independent [B*T] / [B*S] capacity), not by this step's valid rows:
        # with packed decoder rows `rows` changes from step to step and a buffer per distinct count would never be freed
        cap = max(int(x.shape[0]), _rup(rows, ROWPAD))
        q = self.buf(buf_tag + ".q8", cap, cols, fmt)
        slot = self._a8_slot(tag)
        st = self._a8_state[slot]
        delayed = self.fp8_scaling == "delayed"
        item = ops.fp8_item(x, rows, cols, st, fmt, q=q, amax_next=self._a8_part[slot] if delayed else None)
        ops.fp8_quantize([item], amax_pass=not (delayed and tag in self._a8_ready))
        if cache:
            self._a8_cache[key] = (q, st)
        return q, st

    def _q8_target(self, tag: str, buf_tag: str, rows_cap: int, cols: int, fmt):
        """(q, state, fp8_out descriptor, (state, partials)) when the tensor `tag` can be EMITTED by its producer: fp8 GEMMs on, delayed
        scaling, a scale history for this tensor (second pass on); else None: the producer writes bf16 and `_quant` follows."""
        if not (self.fp8 and self.fp8_fused and tag in self._a8_ready):
            return None
        slot = self._a8_slot(tag)
        st, part = self._a8_state[slot], self._a8_part[slot]
        q = self.buf(buf_tag + ".q8", rows_cap, cols, fmt)
        return q, st, ops.fp8_out(q, st, part), (st, part)

    def ln_x8(self, x, ln: str, eps: float, a_name: str, cap: int, mean, rstd, rows: int, wname: str, q_tag: str, **kw):
        """LayerNorm `ln` of x whose output feeds Linear `wname`: (a bf16 or None, x8 = (q, state) or None).  Fused fp8 emission when
        that Linear runs in fp8 and its input has a scale history: the normalised rows leave as e4m3 bytes only."""
        P = self.P
        width = x.shape[-1]
        t = self._q8_target(wname + ".x", q_tag + wname.split(".")[-1] + ".x", cap, width, torch.float8_e4m3fn) if self._fp8_ok(wname) else None
        if t is not None:
            ops.layernorm_fwd(x, P.f32(ln + ".g"), P.f32(ln + ".b"), eps, None, mean, rstd, rows=rows, q8=t[2], **kw)
            return None, (t[0], t[1])
        a = self.buf(a_name, cap, width)
        ops.layernorm_fwd(x, P.f32(ln + ".g"), P.f32(ln + ".b"), eps, a, mean, rstd, rows=rows, **kw)
        return a, None

    def _done(self, seg_name: str):
        """Report that the gradient segment `seg_name` (and, by layout order, everything before it) is final."""
        if self.grad_progress is not None:
            if self._dw_events and not self._dw_side:
                torch.cuda.current_stream().wait_event(self._dw_events[-1])  # "everything before it" includes the dW stream's work
            s = self.P.segs[seg_name]
            self.grad_progress(s.offset + s.numel)

    # ------------------------------------------------------------------ buffers
    def buf(self, name: str, rows: int, cols: int, dtype=None) -> torch.Tensor:
        dtype = dtype or self.dt
        key = (name, rows, cols, dtype)  # (a tuple: this lookup runs ~1500 times per train step on the host)
        t = self._bufs.get(key)
        if t is None:
            t = torch.zeros((_rup(max(rows, 1), ROWPAD), cols), dtype=dtype, device=self.dev)
            self._bufs[key] = t
        return t

    def vec(self, name: str, n: int, dtype=torch.float32) -> torch.Tensor:
        key = (name, n, dtype)
        t = self._bufs.get(key)
        if t is None:
            t = torch.zeros(n, dtype=dtype, device=self.dev)
            self._bufs[key] = t
        return t

    def free_buffers(self):
        """drop every scratch / activation buffer.  Captured decoder-step graphs hold raw pointers into these buffers (and into
        the LayerNorm-folded weights): whoever captured them registered a callback in `on_free` and is told first."""
        for cb in list(self.on_free):
            cb()
        self._bufs.clear()
        self._ET_version, self._ET_event, self._head_nt_state = -1, None, None
        self._lnf = {}
        self._lnf_version = -1

    # ------------------------------------------------------------------ small helpers
    def linear(self, x, wname, out, M, *, act=0, zout=None, residual=None, drop_seed=None, bias=True, save_tag=None, fp8=True,
               stable_input=False, x8=None, out8=None):
        """save_tag (fp8 mode, training forward): the fp8 copy of x is kept under that name for the weight-gradient GEMM of backward
        (dW = dy^T x reads dy and x k-major, as they lie).  x8 = (q, state): x already exists as fp8 bytes (fused emission by its
        producer; `x` may be None).  out8 = `_q8_target(...)`: the result leaves as fp8 bytes (the GELU output of FFN-in, whose only
        reader is the fp8 FFN-out projection); returns (q, state) then."""
        P = self.P
        w = P.w(wname + ".w")
        N, K = w.shape
        p = self.p_drop if drop_seed is not None else 0.0
        if fp8 and self._fp8_ok(wname):
            self._w8_wait()
            wq, _, ws = self._w8[wname]
            if x8 is None:
                x8 = self._quant(x, M, K, wname + ".x", (save_tag or "f8.") + wname.split(".")[-1] + ".x", torch.float8_e4m3fn,
                                 cache=stable_input)
            if save_tag is not None:
                self._x8[wname] = x8
            kw = dict(bias=P.f32(wname + ".b") if bias else None, act=act, zout=zout, residual=residual, dropout_p=p, dropout_seed=drop_seed or 0,
                      a_scale_inv=x8[1][1:], b_scale_inv=ws[1:])
            if out8 is not None:
                ops.gemm(x8[0], wq, out8[0], M, N, K, c_q8=out8[3], **kw)
                return out8[0], out8[1]
            return ops.gemm(x8[0], wq, out, M, N, K, **kw)
        return ops.gemm(x, w, out, M, N, K, bias=P.f32(wname + ".b") if bias else None, act=act, zout=zout,
                        residual=residual, dropout_p=p, dropout_seed=drop_seed or 0)

    def linear_bwd(self, wname, x, dy, M, *, dx=None, zin=None, dact=0, dx_accumulate=False, bias=True, defer=False, x_tag=True,
                   dy8=None, dx8=None, par=0):
        """dW = dy^T x (fp32, overwrite), db = colsum(dy), optionally dx = (dy W) [* act'(zin)].
        defer=True queues the weight-gradient GEMM (nothing but the optimizer depends on it): the caller keeps dy and x
        intact until flush_dw() launches the layer's queue as ONE grouped GEMM (36..256-tile problems fill the chip
        only together).  fp8 projections: dy8 = (q, state) when dy's producer emitted the e5m2 bytes itself (`dy` may be None),
        dx8 = `_q8_target(...)` when dx is to leave as fp8 bytes (the dGELU-scaled dX of FFN-out = the dy of FFN-in); par = layer
        parity of the buffer an unfused dy is quantised into (it is read by the deferred launches of its layer)."""
        P = self.P
        w = P.w(wname + ".w")
        N, K = w.shape
        Mp = _rup(M, ROWPAD)
        if self._fp8_ok(wname) and x_tag is not None:
            # fp8: dy as e5m2 bytes [M][N] — row-major for dX = dy W (NT against W^T, quantised once per step), k-major for
            # dW = dy^T x (TN: both operands as their producers wrote them); x as saved by the forward pass
            kind = wname.split(".")[0].rstrip("0123456789") + "." + wname.split(".")[-1]
            if dy8 is None:
                dy8 = self._quant(dy, M, N, wname + ".dy", f"f8.{kind}.dy.{par}", torch.float8_e5m2)
            x8 = self._x8.get(wname)
            if x8 is None:
                raise RuntimeError(f"fp8 backward of {wname}: the forward pass did not keep the fp8 copy of its input (save_tag missing)")
            _, wqT, ws = self._w8[wname]
            ga = ops.gemm_args(dy8[0], x8[0], P.g(wname + ".w"), N, K, Mp, a_kmajor=True, b_kmajor=True, k_valid=M,
                               a_scale_inv=dy8[1][1:], b_scale_inv=x8[1][1:])
            cs = (dy8[0], P.g(wname + ".b"), dy8[1][1:], M, N) if bias else None  # (into the pre-zeroed atomic region)
            if defer:
                self._dw8_queue.append((ga, wname, bias))
                if cs is not None:
                    self._cs8_queue.append(cs)
            else:
                ops.gemm_grouped([ga])
                if cs is not None:
                    ops.colsum_q8_grouped([cs])
            if dx8 is not None:
                ops.gemm(dy8[0], wqT, dx8[0], M, K, N, zin=zin, dact=dact, a_scale_inv=dy8[1][1:], b_scale_inv=ws[1:], c_q8=dx8[3])
            elif dx is not None:
                ops.gemm(dy8[0], wqT, dx, M, K, N, zin=zin, dact=dact, accumulate=dx_accumulate, a_scale_inv=dy8[1][1:], b_scale_inv=ws[1:])
            if not defer:
                self._done(wname + ".w")
            return dx
        # bf16: the bias gradient colsum(dy) rides on the dW GEMM (row sums of its A operand dy^T, from the fragments the
        # kernel already holds; fp32 atomics into the pre-zeroed atomic region).  fp32 mode keeps the separate column sum.
        fuse = bias and self.dt == torch.bfloat16
        rs = dict(a_rowsum=P.g(wname + ".b"), rowsum_k=M) if fuse else {}
        if self.dt == torch.bfloat16:
            rs["k_valid"] = M  # reduction rows behind the M valid ones count as zero (packed batches: M changes from step to step)
        if defer:
            self._dw_queue.append((ops.gemm_args(dy, x, P.g(wname + ".w"), N, K, Mp, a_kmajor=True, b_kmajor=True, **rs), wname, bias))
        else:
            ops.gemm(dy, x, P.g(wname + ".w"), N, K, Mp, a_kmajor=True, b_kmajor=True, **rs)
        if bias and not fuse:
            if defer:
                self._cs_queue.append((dy, P.g(wname + ".b"), M, N, dy.stride(0)))
            else:
                ops.colsum(dy, P.g(wname + ".b"), M, N, dy.stride(0), accumulate=True)  # region pre-zeroed once per step
        if dx is not None:
            ops.gemm(dy, w, dx, M, K, N, b_kmajor=True, zin=zin, dact=dact, accumulate=dx_accumulate)
        if not defer:
            self._done(wname + ".w")
        return dx

    def dy8_target(self, wname: str, par: int, cap: int):
        """fused-emission target (see `_q8_target`) for the gradient w.r.t. the OUTPUT of the fp8 Linear `wname` — the e5m2 dy its
        backward GEMMs read; `par`: layer parity (the deferred weight-gradient launch of a layer reads it while the next layer's
        backward already writes the other buffer).  None when the Linear is not fp8 / the tensor has no scale history."""
        if not self._fp8_ok(wname):
            return None
        kind = wname.split(".")[0].rstrip("0123456789") + "." + wname.split(".")[-1]
        return self._q8_target(wname + ".dy", f"f8.{kind}.dy.{par}", cap, self.P.w(wname + ".w").shape[0], torch.float8_e5m2)

    def ln_folded(self, wname: str, ln: str):
        """(gamma o W, colsum, bias') of Linear `wname` behind LayerNorm `ln` (mic_ln_fold_weight), rebuilt when the weights
        changed (ParamStore.version): the operands of a LayerNorm-folded GEMM (mic_gemm_args.a_ln_stats)"""
        P = self.P
        ver = getattr(P, "version", 0)
        if ver != self._lnf_version:
            self._lnf_fresh = set()
            self._lnf_version = ver
        hit = self._lnf.get(wname)
        if hit is None:
            w = P.w(wname + ".w")
            hit = self._lnf[wname] = (torch.empty_like(w), torch.empty(w.shape[0], dtype=torch.float32, device=self.dev),
                                      torch.empty(w.shape[0], dtype=torch.float32, device=self.dev))
        if wname not in self._lnf_fresh:
            ops.ln_fold_weight(P.w(wname + ".w"), P.f32(ln + ".g"), P.f32(ln + ".b"), P.f32(wname + ".b"), *hit)
            self._lnf_fresh.add(wname)
        return hit

    def dyb(self, name: str, l: int, rows: int, cols: int) -> torch.Tensor:
        """gradient buffer that a queued weight-gradient GEMM reads: one per layer parity when dW runs on its own stream (the
        next layer's backward rewrites the other one), a single buffer otherwise"""
        return self.buf(f"{name}.{self._par(l)}" if self._dw_on() else name, rows, cols)

    def ln_bwd(self, tag: str, l: int, x, ln: str, mean, rstd, dy, dx, rows: int, **kw):
        """LayerNorm backward of the LayerNorm named `ln` (parameters ln + ".g" / ".b"): dx (and the kwargs' by-products) now, the
        gamma / beta gradients either by atomics inside the kernel or — default — as block partials in a per-(call site, layer
        parity) buffer, summed by the next flush_dw() (the buffer discipline of dyb())"""
        P = self.P
        if not self.ln_partials:
            return ops.layernorm_bwd(x, P.f32(ln + ".g"), mean, rstd, dy, dx, P.g(ln + ".g"), P.g(ln + ".b"), rows=rows, **kw)
        width = x.shape[-1]
        part = self.buf(f"lnp.{tag}.{self._par(l)}", 2 * ops.layernorm_bwd_blocks(1 << 30), width, torch.float32)  # [2][blocks <= cap][width]
        ops.layernorm_bwd_partials(x, P.f32(ln + ".g"), mean, rstd, dy, dx, part, rows=rows, **kw)  # (kw may carry q8 / q8_of_dx: fused fp8 emission)
        self._lnp_queue.append((part, ops.layernorm_bwd_blocks(rows), width, P.g(ln + ".g"), P.g(ln + ".b"), False))
        return dx

    def _dw_on(self) -> bool:
        return self.dw_overlap

    def _par(self, l: int) -> int:
        """index of layer l's copy of a gradient buffer that a queued weight-gradient launch reads: 2 x dw_every copies in turn"""
        return l % (2 * self._dw_every_now())

    def _dw_every_now(self) -> int:
        if not self._dw_on():
            return 1  # no second stream: single buffers, the launches go out in front of every layer's last LayerNorm backward
        return self.dw_every if self.dw_every > 0 else (2 if self.fp8 else 1)

    def _dw_flush_layer(self, l: int) -> bool:
        """weight-gradient launches go out behind every `dw_every`-th layer (and behind layer 0): with 2 x dw_every buffer copies the
        launch that read a copy is the one before the latest when that copy is rewritten, exactly as with one launch per layer"""
        return l % self._dw_every_now() == 0

    def dw_fence(self) -> bool:
        """The wait half of flush_dw(), issued on its own IN FRONT of the layer's last LayerNorm backward: the step's stream waits
        for the most recent weight-gradient launch (the one that read the parity buffers that LayerNorm backward and the next layer
        rewrite); the launch half follows behind that kernel (`flush_dw(fence=False)`), so the critical chain's kernel is in its
        queue before the weight-gradient group is released onto the chip.  False: no second stream — flush_dw() has to come first."""
        if not (self._dw_on() and self.dw_late_flush):
            return False
        if self._dw_events:
            torch.cuda.current_stream().wait_event(self._dw_events[-1])
        return True

    def flush_dw(self, fence: bool = True):
        """Launch the layer's queued weight-gradient GEMMs (and bias column sums) as grouped launches — on the dW stream when
        enabled: it waits for everything enqueued so far (the operands' producers), main goes on with the next layer and
        only waits for the dW launch of TWO layers back (the one that read the buffers the next layer is about to rewrite;
        fence=False: dw_fence() has issued that wait already)."""
        if not (self._cs_queue or self._dw_queue or self._lnp_queue or (self.fp8 and (self._dw8_queue or self._cs8_queue))):
            return
        side = None
        if self._dw_on():
            if self._dw_stream is None:
                import os

                n = int(os.environ.get("MIC_DW_CUS", "0"))  # A/B: the weight-gradient stream on the LAST n CUs of the mask only
                self._dw_stream = ops.cu_masked_stream(256 - n, n, self.dev) if n > 0 else ops.role_stream(self.dev, "dw")
            side = self._dw_stream
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            side.wait_event(ev)
        import contextlib

        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            with (ops.pinned_stream() if side is not None else contextlib.nullcontext()):
                self._dw_side = side is not None
                try:
                    self._flush_dw_launches()
                finally:
                    self._dw_side = False
            if side is not None:
                done = torch.cuda.Event()
                done.record(side)
                self._dw_events.append(done)
        if side is not None and len(self._dw_events) >= 2:
            if fence:
                torch.cuda.current_stream().wait_event(self._dw_events[-2])
            del self._dw_events[:-2]

    def dw_join(self):
        """end of backward: the step's stream waits for the dW stream (next forward rewrites the saved activations dW reads)"""
        if self._lnp_queue:
            self.flush_dw()  # LayerNorm parameter gradients queued behind the last layer's flush
        if self._dw_events:
            torch.cuda.current_stream().wait_event(self._dw_events[-1])
            self._dw_events = []

    def _flush_dw_launches(self):
        if self._lnp_queue:
            ops.ln_param_grads(self._lnp_queue)
            self._lnp_queue = []
        if self._cs_queue:
            ops.colsum_grouped(self._cs_queue)
            self._cs_queue = []
        names = []
        if self.fp8 and self._cs8_queue:
            ops.colsum_q8_grouped(self._cs8_queue)
            self._cs8_queue = []
        if self.fp8 and self._dw8_queue:
            ops.gemm_grouped([q[0] for q in self._dw8_queue])
            names += [q[1] for q in self._dw8_queue]
            self._dw8_queue = []
        if self._dw_queue:
            ops.gemm_grouped([q[0] for q in self._dw_queue])
            names += [q[1] for q in self._dw_queue]
            self._dw_queue = []
        # every queued gradient of the layer is in flight on this stream: report the one that sits last in the flat layout.  The
        # cross-attention k/v weights of ALL layers form one block behind decoder layer 0 (ParamStore.ckv_cat): a per-layer k/v
        # gradient (fp8 GEMMs / MIC_CKV_HOIST=0) must not count here — its offset lies behind every decoder layer, reporting it at
        # layer L-1 would declare the gradients of layers L-2 .. 0 final before their GEMMs have run.  The hoisted launch queues the
        # whole block under the name of its LAST member at the end of decoder backward; the per-layer path reports the block once,
        # after layer 0 (decoder_backward).
        names = [n for n in names if not n.endswith(".ckv") or n == self._ckv_block_name]
        if names:
            self._done(max(names, key=lambda n: self.P.segs[n + ".w"].offset) + ".w")

    # ------------------------------------------------------------------ ViT
    def vit_forward(self, pixels: torch.Tensor, save: bool, trunc_int32: bool = False):
        """pixels [B,img,img,3] fp32 NHWC -> (last hidden [B*S,vd] NOT post-layernormed, ehs [B*S,d])."""
        P = self.P
        B = pixels.shape[0]
        S, vd, vf, H = P.S, P.vd, P.vffn, P.vH
        Mv, Mp = B * S, B * (S - 1)
        pk = P.ps * P.ps * 3
        patches = self.buf("v.patches", Mp, pk)
        ops.im2col(pixels, patches, B, P.img, P.ps, pk, trunc_int32)
        pe = self.buf("v.pe", Mp, vd)
        ops.gemm(patches, P.w("patch.w"), pe, Mp, vd, pk)
        emb = self.buf("v.emb", Mv, vd)
        ops.vit_assemble(pe, P.f32("vit.cls"), P.f32("vit.pos"), emb, B, S, vd, vd)
        x = self.buf("v.x0", Mv, vd)
        st = self.buf("v.pre.stats", 2, _rup(Mv, ROWPAD), torch.float32)
        ops.layernorm_fwd(emb, P.f32("vit.pre_ln.g"), P.f32("vit.pre_ln.b"), self.vit_eps, x, st[0], st[1], rows=Mv)
        for l in range(P.vL):
            tag = f"v{l}." if save else "v_."
            p = f"vit{l}."
            st1 = self.buf(tag + "st1", 2, _rup(Mv, ROWPAD), torch.float32)
            a1, a1_8 = self.ln_x8(x, p + "ln1", self.vit_eps, tag + "a1", Mv, st1[0], st1[1], Mv, p + "qkv", tag)
            qkv = self.buf(tag + "qkv", Mv, 3 * vd)
            self.linear(a1, p + "qkv", qkv, Mv, save_tag=tag if save else None, x8=a1_8)
            ctx = self.buf(tag + "ctx", Mv, vd)
            lse = self.vec(tag + "lse", B * H * S)
            ops.attn_fwd(qkv, qkv[:, vd:], qkv[:, 2 * vd:], ctx, B, H, S, S, ldq=3 * vd, ldk=3 * vd, ldv=3 * vd, ldo=vd, lse=lse)
            xm = self.buf(tag + "xm", Mv, vd)
            self.linear(ctx, p + "o", xm, Mv, residual=x)
            st2 = self.buf(tag + "st2", 2, _rup(Mv, ROWPAD), torch.float32)
            a2, a2_8 = self.ln_x8(xm, p + "ln2", self.vit_eps, tag + "a2", Mv, st2[0], st2[1], Mv, p + "fc1", tag)
            z = self.buf(tag + "z", Mv, vf)
            # the GELU output's only reader is FFN-out: with a scale history it leaves the FFN-in epilogue as e4m3 bytes
            u8t = self._q8_target(p + "fc2.x", tag + "fc2.x", Mv, vf, torch.float8_e4m3fn) if self._fp8_ok(p + "fc2") else None
            u = self.buf(tag + "u", Mv, vf) if u8t is None else None
            u8 = self.linear(a2, p + "fc1", u, Mv, act=L.ACT_QUICK_GELU, zout=z, save_tag=tag if save else None, x8=a2_8, out8=u8t)
            xo = self.buf(f"v{l}.xo" if save else f"v_.xo{l & 1}", Mv, vd)
            self.linear(u, p + "fc2", xo, Mv, residual=xm, save_tag=tag if save else None, x8=u8 if u8t is not None else None)
            x = xo
        ehs = self.buf("v.ehs", Mv, P.d)
        self.linear(x, "vp", ehs, Mv)
        return x, ehs

    def vit_pooler(self, last: torch.Tensor, B: int) -> torch.Tensor:
        """pooler_output = post_layernorm(CLS) (only `encode()` exposes it; unused by captioning)."""
        P = self.P
        cls_rows = last[: B * P.S].view(B, P.S, P.vd)[:, 0].contiguous()
        out = torch.empty_like(cls_rows)
        ops.layernorm_fwd(cls_rows, P.f32("vit.post_ln.g"), P.f32("vit.post_ln.b"), self.vit_eps, out, rows=B)
        return out

    def vit_backward(self, B: int, dehs: torch.Tensor):
        P = self.P
        S, vd, vf, H = P.S, P.vd, P.vffn, P.vH
        Mv, Mp = B * S, B * (S - 1)
        x_last = self.buf(f"v{P.vL - 1}.xo", Mv, vd)
        dx = self.dyb("vb.dx", P.vL - 1, Mv, vd)
        self.linear_bwd("vp", x_last, dehs, Mv, dx=dx)
        f8 = self.fp8  # fp8 projections: their inputs exist as saved fp8 bytes (Engine._x8); the bf16 names below may hold nothing
        dx8 = None     # (q, state) of dx when the LayerNorm backward that produced it emitted the e5m2 bytes too
        for l in reversed(range(P.vL)):
            tag, p = f"v{l}.", f"vit{l}."
            qkv, ctx, xm = (self.buf(tag + n, Mv, c) for n, c in (("qkv", 3 * vd), ("ctx", vd), ("xm", vd)))
            a1, a2, u = ((None, None, None) if f8 else (self.buf(tag + "a1", Mv, vd), self.buf(tag + "a2", Mv, vd), self.buf(tag + "u", Mv, vf)))
            z = self.buf(tag + "z", Mv, vf)
            st1, st2 = self.buf(tag + "st1", 2, _rup(Mv, ROWPAD), torch.float32), self.buf(tag + "st2", 2, _rup(Mv, ROWPAD), torch.float32)
            lse = self.vec(tag + "lse", B * H * S)
            x_in = self.buf(f"v{l - 1}.xo", Mv, vd) if l > 0 else self.buf("v.x0", Mv, vd)
            dz8t = self.dy8_target(p + "fc1", self._par(l), Mv)  # dz = dGELU-scaled dX of fc2 = the dy of fc1: e5m2 straight from the epilogue
            dz = self.dyb("vb.dz", l, Mv, vf) if dz8t is None else None
            self.linear_bwd(p + "fc2", u, dx, Mv, dx=dz, zin=z, dact=L.ACT_QUICK_GELU, defer=True, dy8=dx8, dx8=dz8t, par=self._par(l))
            da = self.buf("vb.da", Mv, vd)
            self.linear_bwd(p + "fc1", a2, dz, Mv, dx=da, defer=True, dy8=None if dz8t is None else (dz8t[0], dz8t[1]), par=self._par(l))
            dxm = self.dyb("vb.dxm", l, Mv, vd)
            self.ln_bwd("v2", l, xm, p + "ln2", st2[0], st2[1], da, dxm, Mv, dres=dx)
            dctx = self.buf("vb.dctx", Mv, vd)
            self.linear_bwd(p + "o", ctx, dxm, Mv, dx=dctx, defer=True)
            dq8t = self.dy8_target(p + "qkv", self._par(l), Mv) if S <= 64 else None
            if dq8t is not None:
                q8 = dq8t[0]
                ops.attn_bwd_q8(qkv, qkv[:, vd:], qkv[:, 2 * vd:], ctx, dctx, lse, dq8t[2], ops.fp8_out(q8[:, vd:], dq8t[1], dq8t[3][1]), q8[:, 2 * vd:],
                                B, H, S, S, ldq=3 * vd, ldk=3 * vd, ldv=3 * vd, ldo=vd, lddo=vd)
                self.linear_bwd(p + "qkv", a1, None, Mv, dx=da, defer=True, dy8=(dq8t[0], dq8t[1]), par=self._par(l))
            else:
                dqkv = self.dyb("vb.dqkv", l, Mv, 3 * vd)
                ops.attn_bwd(qkv, qkv[:, vd:], qkv[:, 2 * vd:], ctx, dctx, lse, dqkv, dqkv[:, vd:], dqkv[:, 2 * vd:], B, H, S, S,
                             ldq=3 * vd, ldk=3 * vd, ldv=3 * vd, ldo=vd, lddo=vd, lddq=3 * vd, lddk=3 * vd, lddv=3 * vd)
                self.linear_bwd(p + "qkv", a1, dqkv, Mv, dx=da, defer=True, par=self._par(l))
            flush_now = self._dw_flush_layer(l)
            late = self.dw_fence() if flush_now else False
            if flush_now and not late:
                self.flush_dw()  # before LN1 backward overwrites dx (the fc2 gradient operand)
            dx = self.dyb("vb.dx", l - 1, Mv, vd)  # the layer below's residual-stream gradient (= its fc2 dW operand)
            # ... which the layer below's fc2 backward reads as e5m2: emitted here, beside the bf16 dx the residual path needs
            dx8t = self.dy8_target(f"vit{l - 1}.fc2", self._par(l - 1), Mv) if l > 0 else None
            if dx8t is not None:
                self.ln_bwd("v1", l, x_in, p + "ln1", st1[0], st1[1], da, dx, Mv, dres=dxm, q8=dx8t[2], q8_of_dx=True)
                dx8 = (dx8t[0], dx8t[1])
            else:
                self.ln_bwd("v1", l, x_in, p + "ln1", st1[0], st1[1], da, dx, Mv, dres=dxm)
                dx8 = None
            if flush_now and late:
                self.flush_dw(fence=False)
        emb = self.buf("v.emb", Mv, vd)
        st = self.buf("v.pre.stats", 2, _rup(Mv, ROWPAD), torch.float32)
        demb = self.buf("vb.demb", Mv, vd)
        self.ln_bwd("vpre", 0, emb, "vit.pre_ln", st[0], st[1], dx, demb, Mv)
        dpe = self.buf("vb.dpe", Mp, vd)
        ops.vit_assemble_bwd(demb, dpe, P.g("vit.cls"), P.g("vit.pos"), B, S, vd, vd)
        pk = P.ps * P.ps * 3
        patches = self.buf("v.patches", Mp, pk)
        ops.gemm(dpe, patches, P.g("patch.w"), vd, pk, _rup(Mp, ROWPAD), a_kmajor=True, b_kmajor=True)
        self._done("patch.w")

    # ------------------------------------------------------------------ decoder (teacher forced)
    def decoder_forward(self, ids, pos_ids, key_mask, ehs, B: int, T: int, save: bool, seed: Optional[int], pack=None):
        """ids/pos_ids/key_mask int32 [B*T]/[B*T]/[B,T]; ehs [B*S,d].  Returns final-layer-normed hidden [B*T,d].

        pack = (q_off int32 [B], q_len int32 [B], rows): PACKED rows — the decoder runs on the `rows` = sum(q_len) valid positions
        only (sequence b = rows [q_off[b], q_off[b] + q_len[b]); ids / pos_ids are then the packed arrays, key_mask is unused:
        every position of a sequence is valid).  Padded positions carry no loss (main.py:678) and no valid position attends to
        them (main.py:692: the labels' attention mask is the decoder's key mask), so everything they would compute or receive as
        gradient is exactly zero: leaving them out changes no result.  Buffers keep their [B*T] capacity."""
        P = self.P
        d, f, H, S = P.d, P.ffn, P.H, P.S
        Mcap, Mv = B * T, B * S
        M = pack[2] if pack is not None else Mcap  # rows the kernels process
        drop = seed is not None and self.p_drop > 0

        def sd(site):
            return _mix(seed, site) if drop else None

        h0 = self.buf("d.h0", Mcap, d)
        ops.embed_fwd(ids, pos_ids, P.w("shared"), P.f32("dec.pos"), self.embed_scale, h0, M, d)
        x = self.buf("d.x0", Mcap, d)
        ste = self.buf("d.emb.stats", 2, _rup(Mcap, ROWPAD), torch.float32)
        ops.layernorm_fwd(h0, P.f32("dec.ln_emb.g"), P.f32("dec.ln_emb.b"), self.dec_eps, x, ste[0], ste[1], rows=M,
                          dropout_p=self.p_drop if drop else 0.0, dropout_seed=sd(1) or 0)
        kvcat = self.cross_kv_all(ehs, Mv, "d." if save else "d_.", save) if self.ckv_hoisted() else None
        for l in range(P.L):
            tag = f"d{l}." if save else "d_."
            p = f"dec{l}."
            stats = self.buf(tag + "stats", 6, _rup(Mcap, ROWPAD), torch.float32)
            a, a8 = self.ln_x8(x, p + "ln_sa", self.dec_eps, tag + "a_sa", Mcap, stats[0], stats[1], M, p + "qkv", tag)
            qkv = self.buf(tag + "qkv", Mcap, 3 * d)
            self.linear(a, p + "qkv", qkv, M, save_tag=tag if save else None, x8=a8)
            ctx = self.buf(tag + "ctx", Mcap, d)
            lse = self.vec(tag + "lse", B * H * T)
            if pack is not None:
                ops.attn_fwd_packed(qkv, qkv[:, d:], qkv[:, 2 * d:], ctx, B, H, T, T, pack[0], pack[1], kv_packed=True, ldq=3 * d, ldk=3 * d,
                                    ldv=3 * d, ldo=d, causal=True, lse=lse)
            else:
                ops.attn_fwd(qkv, qkv[:, d:], qkv[:, 2 * d:], ctx, B, H, T, T, ldq=3 * d, ldk=3 * d, ldv=3 * d, ldo=d, key_mask=key_mask,
                             causal=True, lse=lse)
            x1 = self.buf(tag + "x1", Mcap, d)
            self.linear(ctx, p + "so", x1, M, residual=x, drop_seed=sd(10 + 3 * l))
            a, a8 = self.ln_x8(x1, p + "ln_ca", self.dec_eps, tag + "a_ca", Mcap, stats[2], stats[3], M, p + "cq", tag)
            q = self.buf(tag + "cq", Mcap, d)
            self.linear(a, p + "cq", q, M, save_tag=tag if save else None, x8=a8)
            if kvcat is not None:
                kv, ldkv = kvcat[:, l * 2 * d:], kvcat.stride(0)
            else:
                kv, ldkv = self.buf(tag + "ckv", Mv, 2 * d), 2 * d
                self.linear(ehs, p + "ckv", kv, Mv, save_tag="d.ehs." if save else None, stable_input=True)
            cctx = self.buf(tag + "cctx", Mcap, d)
            clse = self.vec(tag + "clse", B * H * T)
            if pack is not None:
                ops.attn_fwd_packed(q, kv, kv[:, d:], cctx, B, H, T, S, pack[0], pack[1], kv_packed=False, ldq=d, ldk=ldkv, ldv=ldkv, ldo=d,
                                    lse=clse)
            else:
                ops.attn_fwd(q, kv, kv[:, d:], cctx, B, H, T, S, ldq=d, ldk=ldkv, ldv=ldkv, ldo=d, lse=clse)
            x2 = self.buf(tag + "x2", Mcap, d)
            self.linear(cctx, p + "co", x2, M, residual=x1, drop_seed=sd(11 + 3 * l))
            a, a8 = self.ln_x8(x2, p + "ln_ff", self.dec_eps, tag + "a_ff", Mcap, stats[4], stats[5], M, p + "fc1", tag)
            z = self.buf(tag + "z", Mcap, f)
            u8t = self._q8_target(p + "fc2.x", tag + "fc2.x", Mcap, f, torch.float8_e4m3fn) if self._fp8_ok(p + "fc2") else None
            u = self.buf(tag + "u", Mcap, f) if u8t is None else None
            u8 = self.linear(a, p + "fc1", u, M, act=self.gelu, zout=z, save_tag=tag if save else None, x8=a8, out8=u8t)
            x3 = self.buf(f"d{l}.x3" if save else f"d_.x3{l & 1}", Mcap, d)
            self.linear(u, p + "fc2", x3, M, residual=x2, drop_seed=sd(12 + 3 * l), save_tag=tag if save else None,
                        x8=u8 if u8t is not None else None)
            x = x3
        hf = self.buf("d.hf", Mcap, d)
        stf = self.buf("d.f.stats", 2, _rup(Mcap, ROWPAD), torch.float32)
        ops.layernorm_fwd(x, P.f32("dec.ln_f.g"), P.f32("dec.ln_f.b"), self.dec_eps, hf, stf[0], stf[1], rows=M)
        return hf

    # ------------------------------------------------------------------ cross-attention k/v of all layers as one GEMM
    def ckv_hoisted(self) -> bool:
        """the cross-attention k/v projections of all decoder layers run as ONE GEMM each way (forward, dX, dW): they read the
        same encoder states and their weights sit side by side (ParamStore.ckv_cat).  L one-round launches (3200 x 2048 x 1024:
        400 tiles of 128x128) become one with 1248 tiles of 256x256; the dX sum over the layers becomes a K = L*2d contraction
        with one rounding instead of L bf16 accumulations.  fp8 mode: the same three launches on fp8 operands, the concatenated
        weight and the concatenated k/v gradient each under ONE scale.  MIC_CKV_HOIST=0 switches back (A/B)."""
        return self._ckv_hoist

    def cross_kv_all(self, ehs, Mv: int, tag: str, save: bool = False):
        """[Mv][L*2d]: layer l's (k | v) in columns [l*2d, (l+1)*2d)"""
        P = self.P
        w, b = P.ckv_cat("w")
        kvcat = self.buf(tag + "ckvcat", Mv, P.L * 2 * P.d)
        if self.fp8 and "ckvcat" in self._w8:
            self._w8_wait()
            wq, _, ws = self._w8["ckvcat"]
            x8 = self._quant(ehs, Mv, P.d, "ckvcat.x", tag + "ckvcat.x", torch.float8_e4m3fn)
            if save:
                self._x8["ckvcat"] = x8
            ops.gemm(x8[0], wq, kvcat, Mv, P.L * 2 * P.d, P.d, bias=b, a_scale_inv=x8[1][1:], b_scale_inv=ws[1:])
        else:
            ops.gemm(ehs, w, kvcat, Mv, P.L * 2 * P.d, P.d, bias=b)
        return kvcat

    def head_logits(self, hf, M: int, name: str = "d.logits", stats: bool = False, fp8: bool = False):
        """Tied head (modeling:170-178): logits[M, Vpad] = hf @ shared^T + final_logits_bias (compute dtype).
        stats=True (bf16 mode): also returns the GEMM's softmax partials per 64-column granule [M][Vpad/64][2] (fp32) so that the
        log-softmax downstream (cross-entropy, beam scores) needs no second pass over the logits."""
        P = self.P
        logits = self.buf(name, M, P.Vpad)
        stat = self.head_stats(name, M) if stats else None
        self._head_project(hf, logits, M, stat, fp8=fp8)  # (fp8: the train / eval passes only — inference stays in the storage dtype)
        return (logits, stat) if stats else logits

    def _head8(self) -> bool:
        return self.fp8 and self.fp8_head > 0

    def _head_project(self, hf, logits, M: int, stat, fp8: bool = True):
        """logits[:M] = hf[:M] @ shared^T + final_logits_bias (+ the softmax partials): bf16 operands, or — fp8 GEMMs with
        MIC_FP8_HEAD=all — the e4m3 copies of both (the quantised hidden states stay for the head's weight gradient)"""
        P = self.P
        if fp8 and self._head8() and self.fp8_head == 2:
            x8 = self._head_x8 = self._quant(hf, M, P.d, "head.x", "head.x", torch.float8_e4m3fn)
            E8, _, wst = self._w8["shared"]
            self._w8_wait()
            ops.gemm(x8[0], E8, logits, M, P.Vpad, P.d, bias=P.f32("flb"), rowstat=stat, rowstat_nvalid=P.V,
                     a_scale_inv=x8[1][1:], b_scale_inv=wst[1:])
        else:
            ops.gemm(hf, P.w("shared"), logits, M, P.Vpad, P.d, bias=P.f32("flb"), rowstat=stat, rowstat_nvalid=P.V)

    def head_stats(self, name: str, M: int):
        if self.dt != torch.bfloat16 or not self.use_head_stats:
            return None  # the fp32 (parity) GEMM kernel has no by-products: its consumers stream the logits
        return self.buf(name + ".stat", M, 2 * (self.P.Vpad // 64), torch.float32)  # (max, sum exp) per 64-column granule

    def decoder_backward(self, B: int, T: int, ids, pos_ids, key_mask, ehs, dlogits, seed: Optional[int], rows=None, pack=None):
        """Consumes dlogits [M,Vpad] (or [Mc,Vpad] for the compacted head: rows = (idx int32 [Mc], Mc)); writes all
        decoder/embedding/head grads; returns dehs [B*S,d].  pack: see decoder_forward (the compacted head's rows ARE the packed
        rows then: no gather / scatter between the head and the decoder)."""
        P = self.P
        d, f, H, S = P.d, P.ffn, P.H, P.S
        Mcap, Mv = B * T, B * S
        M = pack[2] if pack is not None else Mcap
        drop = seed is not None and self.p_drop > 0
        pd = self.p_drop if drop else 0.0

        def sd(site):
            return _mix(seed, site) if drop else 0

        dhf = self.buf("db.dhf", Mcap, d)
        if rows is None:
            hf = self.buf("d.hf", Mcap, d)
            Mh = M
        else:
            hf = self.buf("d.hf" if pack is not None else "d.hfc", Mcap, d)  # compacted final hidden states (pad rows zero)
            Mh = rows[1]
        Mhp = _rup(Mh, 64)
        nt = self._head_nt_state
        self._head_nt_state = None
        h8s, self._head8_state = self._head8_state, None
        if h8s is not None:
            # fp8 head: dE = dlogits^T . h over K = rows, both operands k-major (the e5m2 copy of dlogits as the CE backward wrote it,
            # the e4m3 hidden states; rows >= Mh of the 128-row reduction padding count as zero), fp32 straight into the gradient buffer
            x8 = self._head_x8 or self._quant(hf, Mh, d, "head.x", "head.x", torch.float8_e4m3fn)
            ops.gemm(h8s[0], x8[0], P.g("shared"), P.Vpad, d, _rup(Mh, 128), a_kmajor=True, b_kmajor=True, k_valid=Mh,
                     a_scale_inv=h8s[1][1:], b_scale_inv=x8[1][1:])
            # the label entries (kept out of the byte matrix): + coef h into dE's label rows, coef E[label] as one more slab of dX
            tiles = ((Mh + 255) // 256) * ((d + 255) // 256)
            nsp = max(1, min(ops.get_cu_budget() // tiles, P.Vpad // 128 // 2))
            slab_rows = _rup(Mh, 256)
            cap_rows = max(_rup(Mcap, 256), (256 // max(1, (d + 255) // 256)) * 256) + _rup(Mcap, 256)
            d32 = self.buf("db.dhf32", cap_rows, d, torch.float32)
            assert (nsp + 1) * slab_rows <= d32.shape[0], (nsp, slab_rows, d32.shape)
            ops.head_label_terms(h8s[3], h8s[2], P.w("shared"), hf, d32[nsp * slab_rows:], P.g("shared"), Mh, d)
        elif nt is not None:
            # NT launches on the four-wave kernel: dE = dlogits^T . (h^T)^T over K = rows (zero-padded to a multiple of 128 by the CE
            # backward; the bias gradient was summed there), fp32 straight into the gradient buffer
            dT, Kp = nt
            hfT = self.buf("d.hfT", d, dT.shape[1])
            ops.transpose_bf16(hf, hfT, Mh, d, rows_pad=Kp)
            ops.gemm(dT, hfT, P.g("shared"), P.Vpad, d, Kp)
        elif self.dt == torch.bfloat16:  # final_logits_bias gradient = row sums of dlogits^T, fused into the dE GEMM
            ops.zero(P.g("flb"))        # (atomics; this segment sits in front of the pre-zeroed atomic region)
            ops.gemm(dlogits, hf, P.g("shared"), P.Vpad, d, Mhp, a_kmajor=True, b_kmajor=True, a_rowsum=P.g("flb"), rowsum_k=Mh)
        else:
            ops.colsum(dlogits, P.g("flb"), Mh, P.Vpad, dlogits.stride(0))
            ops.gemm(dlogits, hf, P.g("shared"), P.Vpad, d, Mhp, a_kmajor=True, b_kmajor=True)
        # dense (LM head) part of the tied embedding gradient: the first thing backward completes — its exchange starts here.  The
        # dX GEMM below still READS the embedding: the Trainer's reducer issues this bucket's optimizer pass at the next progress
        # report, behind it (GradReducer `defer`)
        self._done("shared")
        dhc = dhf if (rows is None or pack is not None) else self.buf("db.dhfc", Mcap, d)
        if h8s is not None:
            # dX = dlogits . (E^T)^T over K = Vpad on the e5m2 / e4m3 copies, split over K as below; the slabs and the label slab are
            # summed and rounded to bf16 once
            _, ET8, wst = self._w8["shared"]
            self._w8_wait()
            slab = slab_rows * d
            ops.gemm(h8s[0], ET8, d32, Mh, d, P.Vpad, split_k=nsp, split_stride=slab if nsp > 1 else 0, a_scale_inv=h8s[1][1:], b_scale_inv=wst[1:])
            ops.sum_slabs(d32, nsp + 1, slab, dhc, Mh, d, d32.stride(0), dhc.stride(0))
        elif nt is not None:
            # dX = dlogits . (E^T)^T over K = Vpad: ceil(Mh / 256) x d / 256 output tiles, split over K so that one round of blocks
            # fills the chip; one fp32 slab per split, summed and rounded to bf16 once
            tiles = ((Mh + 255) // 256) * ((d + 255) // 256)
            nsp = max(1, min(ops.get_cu_budget() // tiles, P.Vpad // 128 // 2))
            # ONE workspace whatever this batch's row count: a slab is as tall as the valid rows (rounded to the tile), and
            # nsp * slab rows <= budget / (d / 256) tile rows — a buffer keyed on nsp would add ~100 MB per distinct split seen
            # during training on ragged batches.  (The split itself follows Mh: the fp32 summation order of dX is a function of
            # the batch's row count, like every tile choice of the planner.)
            slab_rows = _rup(Mh, 256)
            cap_rows = max(_rup(Mcap, 256), (256 // max(1, (d + 255) // 256)) * 256)
            d32 = self.buf("db.dhf32", cap_rows, d, torch.float32)
            assert nsp * slab_rows <= d32.shape[0], (nsp, slab_rows, d32.shape)
            slab = slab_rows * d
            ops.gemm(dlogits, self.shared_T(), d32, Mh, d, P.Vpad, split_k=nsp, split_stride=slab if nsp > 1 else 0)
            if nsp > 1:
                ops.sum_slabs(d32, nsp, slab, dhc, Mh, d, d32.stride(0), dhc.stride(0))
            else:
                ops.cast2d(d32, dhc, Mh, d, d32.stride(0), dhc.stride(0))
        elif self.dt == torch.bfloat16 and P.Vpad >= 16384:
            # [Mh, d] output, reduction over the whole vocabulary: far too few tiles to fill 256 CUs.  Split-K 32 with K-range
            # <-> XCD affinity (gemm.hip) into per-split fp32 slabs (no atomics), summed and rounded to bf16 once.
            nsp = 32  # 16..64 measure the same (+-0.1 ms/step); the gain over the atomic variant is the absence of atomics
            d32 = self.buf("db.dhf32", nsp * _rup(Mcap, ROWPAD), d, torch.float32)  # one fp32 slab per split, summed below
            slab = _rup(Mcap, ROWPAD) * d
            ops.gemm(dlogits, P.w("shared"), d32, Mh, d, P.Vpad, b_kmajor=True, split_k=nsp, split_stride=slab)
            ops.sum_slabs(d32, nsp, slab, dhc, Mh, d, d32.stride(0), dhc.stride(0))
        else:
            ops.gemm(dlogits, P.w("shared"), dhc, Mh, d, P.Vpad, b_kmajor=True)
        if rows is not None and pack is None:
            ops.zero(dhf[:M])  # masked-out positions receive exactly zero gradient from the loss
            ops.copy_rows(dhc, dhf, Mh, d, dst_idx=rows[0])
        dx = self.buf("db.dx", Mcap, d)
        # masked gradients entering the FFN / cross-attention / self-attention branches and the other operands of the deferred
        # weight-gradient GEMMs come from dyb(): one buffer per layer parity while dW runs on its own stream
        stf = self.buf("d.f.stats", 2, _rup(Mcap, ROWPAD), torch.float32)
        x_last = self.buf(f"d{P.L - 1}.x3", Mcap, d)
        # dxm: the dropout-masked residual-stream gradient = the dy of the top layer's FFN-out projection, its only reader: with a
        # scale history it leaves the LayerNorm backward as e5m2 bytes only
        dxm8t = self.dy8_target(f"dec{P.L - 1}.fc2", self._par(P.L - 1), Mcap)
        dxm = self.dyb("db.dxm_a", P.L - 1, Mcap, d) if dxm8t is None else None
        self.ln_bwd("f", 0, x_last, "dec.ln_f", stf[0], stf[1], dhf, dx, M, dxm=dxm, dropout_p=pd, dropout_seed=sd(12 + 3 * (P.L - 1)),
                    **({} if dxm8t is None else dict(q8=dxm8t[2])))
        dxm8 = None if dxm8t is None else (dxm8t[0], dxm8t[1])
        f8 = self.fp8
        dehs = self.buf("db.dehs", Mv, d)
        hoist = self.ckv_hoisted()
        kvcat = self.buf("d.ckvcat", Mv, P.L * 2 * d) if hoist else None
        hoist8 = hoist and self.fp8 and "ckvcat" in self._w8
        # fp8, hoisted: every layer's dK | dV lands in its column slice of ONE e5m2 matrix under one scale (fused emission) ...
        dkvcat8t = self._q8_target("ckvcat.dy", "f8.ckvcat.dy", Mv, P.L * 2 * d, torch.float8_e5m2) if (hoist8 and T <= 64 and S <= 64) else None
        # ... or, without a scale history, in the bf16 matrix that is quantised once behind the loop
        dkvcat = self.buf("db.dkvcat", Mv, P.L * 2 * d) if (hoist and dkvcat8t is None) else None
        for l in reversed(range(P.L)):
            tag, p = f"d{l}.", f"dec{l}."
            stats = self.buf(tag + "stats", 6, _rup(Mcap, ROWPAD), torch.float32)
            qkv, ctx, x1 = (self.buf(tag + n, Mcap, c) for n, c in (("qkv", 3 * d), ("ctx", d), ("x1", d)))
            cq, cctx, x2 = (self.buf(tag + n, Mcap, d) for n in ("cq", "cctx", "x2"))
            ckv, ldkv = (kvcat[:, l * 2 * d:], kvcat.stride(0)) if hoist else (self.buf(tag + "ckv", Mv, 2 * d), 2 * d)
            z = self.buf(tag + "z", Mcap, f)
            # (fp8 projections read their saved fp8 inputs, Engine._x8: the bf16 copies need not exist)
            a_sa, a_ca, a_ff, u = (None,) * 4 if f8 else (self.buf(tag + "a_sa", Mcap, d), self.buf(tag + "a_ca", Mcap, d),
                                                          self.buf(tag + "a_ff", Mcap, d), self.buf(tag + "u", Mcap, f))
            lse, clse = self.vec(tag + "lse", B * H * T), self.vec(tag + "clse", B * H * T)
            x_in = self.buf(f"d{l - 1}.x3", Mcap, d) if l > 0 else self.buf("d.x0", Mcap, d)
            # --- FFN branch: x3 = x2 + drop(fc2(gelu(fc1(LN(x2)))));  dxm = dropout-masked dx3
            dxm_b, dxm_c = self.dyb("db.dxm_b", l, Mcap, d), self.dyb("db.dxm_c", l, Mcap, d)
            dz8t = self.dy8_target(p + "fc1", self._par(l), Mcap)
            dz = self.dyb("db.dz", l, Mcap, f) if dz8t is None else None
            self.linear_bwd(p + "fc2", u, dxm, M, dx=dz, zin=z, dact=self.gelu, defer=True, dy8=dxm8, dx8=dz8t, par=self._par(l))
            da = self.buf("db.da", Mcap, d)
            self.linear_bwd(p + "fc1", a_ff, dz, M, dx=da, defer=True, dy8=None if dz8t is None else (dz8t[0], dz8t[1]), par=self._par(l))
            dx2 = self.buf("db.dx2", Mcap, d)
            self.ln_bwd("ff", l, x2, p + "ln_ff", stats[4], stats[5], da, dx2, M,
                        dres=dx, dxm=dxm_b, dropout_p=pd, dropout_seed=sd(11 + 3 * l))
            # --- cross-attention branch
            dctx = self.buf("db.dctx", Mcap, d)
            self.linear_bwd(p + "co", cctx, dxm_b, M, dx=dctx, defer=True)
            one_tile = T <= 64 and S <= 64
            dq8t = self.dy8_target(p + "cq", self._par(l), Mcap) if one_tile else None
            if hoist:
                kv8 = None if dkvcat8t is None else dkvcat8t[0][:, l * 2 * d:]
                dkv8t = None if kv8 is None else (kv8, dkvcat8t[1], ops.fp8_out(kv8, dkvcat8t[1], dkvcat8t[3][1]))
            else:
                dkv8t = self.dy8_target(p + "ckv", self._par(l), Mv) if one_tile else None
            if dq8t is not None and dkv8t is not None:
                # dQ [rows][d] and dK | dV [encoder rows][2d] as e5m2 bytes under their own scales, straight out of the attention backward
                ops.attn_bwd_q8(cq, ckv, ckv[:, d:], cctx, dctx, clse, dq8t[2], dkv8t[2], dkv8t[0][:, d:], B, H, T, S, ldq=d, ldk=ldkv, ldv=ldkv,
                                ldo=d, lddo=d, q_off=pack[0] if pack is not None else None, q_len=pack[1] if pack is not None else None, kv_packed=False)
                self.linear_bwd(p + "cq", a_ca, None, M, dx=da, defer=True, dy8=(dq8t[0], dq8t[1]), par=self._par(l))
                if not hoist:
                    self.linear_bwd(p + "ckv", ehs, None, Mv, dx=dehs, dx_accumulate=(l != P.L - 1), defer=True, dy8=(dkv8t[0], dkv8t[1]), par=self._par(l))
            elif hoist and dkvcat8t is not None:
                # (dQ has no scale history yet but the concatenated k/v gradient has: cannot happen after the first pass; keep the bytes consistent)
                raise RuntimeError("fp8: cross-attention dQ without a scale history beside a hoisted k/v gradient with one")
            else:
                dq = self.dyb("db.dq", l, Mcap, d)
                dkv = dkvcat[:, l * 2 * d:] if hoist else self.dyb("db.dkv", l, Mv, 2 * d)
                if pack is not None:
                    ops.attn_bwd_packed(cq, ckv, ckv[:, d:], cctx, dctx, clse, dq, dkv, dkv[:, d:], B, H, T, S, pack[0], pack[1], kv_packed=False,
                                        ldq=d, ldk=ldkv, ldv=ldkv, ldo=d, lddo=d, lddq=d, lddk=ldkv, lddv=ldkv)
                else:
                    ops.attn_bwd(cq, ckv, ckv[:, d:], cctx, dctx, clse, dq, dkv, dkv[:, d:], B, H, T, S, ldq=d, ldk=ldkv, ldv=ldkv, ldo=d,
                                 lddo=d, lddq=d, lddk=ldkv, lddv=ldkv)
                self.linear_bwd(p + "cq", a_ca, dq, M, dx=da, defer=True, par=self._par(l))
                if not hoist:
                    self.linear_bwd(p + "ckv", ehs, dkv, Mv, dx=dehs, dx_accumulate=(l != P.L - 1), defer=True, par=self._par(l))
            dx1 = self.buf("db.dx1", Mcap, d)
            self.ln_bwd("ca", l, x1, p + "ln_ca", stats[2], stats[3], da, dx1, M,
                        dres=dx2, dxm=dxm_c, dropout_p=pd, dropout_seed=sd(10 + 3 * l))
            # --- self-attention branch
            self.linear_bwd(p + "so", ctx, dxm_c, M, dx=dctx, defer=True)
            dqkv8t = self.dy8_target(p + "qkv", self._par(l), Mcap) if T <= 64 else None
            if dqkv8t is not None:
                q8 = dqkv8t[0]
                ops.attn_bwd_q8(qkv, qkv[:, d:], qkv[:, 2 * d:], ctx, dctx, lse, dqkv8t[2], ops.fp8_out(q8[:, d:], dqkv8t[1], dqkv8t[3][1]), q8[:, 2 * d:],
                                B, H, T, T, ldq=3 * d, ldk=3 * d, ldv=3 * d, ldo=d, lddo=d, q_off=pack[0] if pack is not None else None,
                                q_len=pack[1] if pack is not None else None, kv_packed=pack is not None,
                                key_mask=key_mask if pack is None else None, causal=True)
                self.linear_bwd(p + "qkv", a_sa, None, M, dx=da, defer=True, dy8=(dqkv8t[0], dqkv8t[1]), par=self._par(l))
            else:
                dqkv = self.dyb("db.dqkv", l, Mcap, 3 * d)
                if pack is not None:
                    ops.attn_bwd_packed(qkv, qkv[:, d:], qkv[:, 2 * d:], ctx, dctx, lse, dqkv, dqkv[:, d:], dqkv[:, 2 * d:], B, H, T, T, pack[0],
                                        pack[1], kv_packed=True, ldq=3 * d, ldk=3 * d, ldv=3 * d, ldo=d, lddo=d, lddq=3 * d, lddk=3 * d,
                                        lddv=3 * d, causal=True)
                else:
                    ops.attn_bwd(qkv, qkv[:, d:], qkv[:, 2 * d:], ctx, dctx, lse, dqkv, dqkv[:, d:], dqkv[:, 2 * d:], B, H, T, T, ldq=3 * d,
                                 ldk=3 * d, ldv=3 * d, ldo=d, lddo=d, lddq=3 * d, lddk=3 * d, lddv=3 * d, key_mask=key_mask, causal=True)
                self.linear_bwd(p + "qkv", a_sa, dqkv, M, dx=da, defer=True, par=self._par(l))
            flush_now = self._dw_flush_layer(l)
            late = self.dw_fence() if flush_now else False
            if flush_now and not late:
                self.flush_dw()  # the layer's 7 weight-gradient GEMMs as one grouped launch (before dxm_a is rewritten)
            if l > 0:
                dxm8t = self.dy8_target(f"dec{l - 1}.fc2", self._par(l - 1), Mcap)
                dxm = self.dyb("db.dxm_a", l - 1, Mcap, d) if dxm8t is None else None
                self.ln_bwd("sa", l, x_in, p + "ln_sa", stats[0], stats[1], da, dx, M, dres=dx1, dxm=dxm, dropout_p=pd,
                            dropout_seed=sd(12 + 3 * (l - 1)), **({} if dxm8t is None else dict(q8=dxm8t[2])))
                dxm8 = None if dxm8t is None else (dxm8t[0], dxm8t[1])
            else:
                self.ln_bwd("sa", l, x_in, p + "ln_sa", stats[0], stats[1], da, dx, M, dres=dx1)
            if flush_now and late:
                self.flush_dw(fence=False)
        if hoist:
            # the cross-attention k/v projections of all layers at once: dW (+ bias row sums) = dkv^T ehs as one [L*2d][d] weight
            # gradient (on the dW stream), dehs = dkv W as one contraction over K = L*2d (split over K into fp32 slabs: 52 output
            # tiles of 256x256 cannot fill the chip)
            wcat, _ = P.ckv_cat("w")
            gw, gb = P.ckv_cat("g")
            N = P.L * 2 * d
            Mvp = _rup(Mv, ROWPAD)
            if hoist8:
                dkv8 = (dkvcat8t[0], dkvcat8t[1]) if dkvcat8t is not None else self._quant(dkvcat, Mv, N, "ckvcat.dy", "f8.ckvcat.dy", torch.float8_e5m2)
                x8 = self._x8.get("ckvcat")
                if x8 is None:
                    raise RuntimeError("fp8 backward of the hoisted cross k/v projection: the forward pass did not keep the fp8 encoder states")
                _, wqT, ws = self._w8["ckvcat"]
                nsp = min(8, N // 128)
                if nsp > 1:
                    slab = Mvp * d
                    d32 = self.buf("db.dehs32", nsp * Mvp, d, torch.float32)
                    ops.gemm(dkv8[0], wqT, d32, Mv, d, N, split_k=nsp, split_stride=slab, a_scale_inv=dkv8[1][1:], b_scale_inv=ws[1:])
                    ops.sum_slabs(d32, nsp, slab, dehs, Mv, d, d32.stride(0), dehs.stride(0))
                else:
                    ops.gemm(dkv8[0], wqT, dehs, Mv, d, N, a_scale_inv=dkv8[1][1:], b_scale_inv=ws[1:])
                self._dw8_queue.append((ops.gemm_args(dkv8[0], x8[0], gw, N, d, Mvp, a_kmajor=True, b_kmajor=True, k_valid=Mv,
                                                      a_scale_inv=dkv8[1][1:], b_scale_inv=x8[1][1:]), f"dec{P.L - 1}.ckv", True))
                self._cs8_queue.append((dkv8[0], gb, dkv8[1][1:], Mv, N))
                self._ckv_block_name = f"dec{P.L - 1}.ckv"
                try:
                    self.flush_dw()
                finally:
                    self._ckv_block_name = None
            else:
                # (bf16 / fp32 operands)
                fuse = self.dt == torch.bfloat16
                rs = dict(a_rowsum=gb, rowsum_k=Mv) if fuse else {}
                # dX first: it READS the weights, and reporting their gradients final (flush_dw -> _done) lets the per-bucket optimizer
                # rewrite them; the dW stream's launch waits for everything enqueued here, so the bucket's event covers this GEMM too
                nsp = min(8, N // 64)  # K-tiles of 64: a reduced model may have fewer than 8 of them
                if self.dt == torch.bfloat16 and nsp > 1:
                    slab = Mvp * d
                    d32 = self.buf("db.dehs32", nsp * Mvp, d, torch.float32)
                    ops.gemm(dkvcat, wcat, d32, Mv, d, N, b_kmajor=True, split_k=nsp, split_stride=slab)
                    ops.sum_slabs(d32, nsp, slab, dehs, Mv, d, d32.stride(0), dehs.stride(0))
                else:
                    ops.gemm(dkvcat, wcat, dehs, Mv, d, N, b_kmajor=True)
                self._dw_queue.append((ops.gemm_args(dkvcat, ehs, gw, N, d, Mvp, a_kmajor=True, b_kmajor=True, **rs), f"dec{P.L - 1}.ckv", True))
                if not fuse:
                    self._cs_queue.append((dkvcat, gb, Mv, N, dkvcat.stride(0)))
                self._ckv_block_name = f"dec{P.L - 1}.ckv"
                try:
                    self.flush_dw()
                finally:
                    self._ckv_block_name = None
        else:
            # per-layer k/v projections: every layer's dW GEMM (and the dX GEMMs that read these weights) has been issued by now
            self._done(f"dec{P.L - 1}.ckv.w")
        # embedding LayerNorm (+ its dropout) and the token/position embedding scatter
        h0 = self.buf("d.h0", Mcap, d)
        ste = self.buf("d.emb.stats", 2, _rup(Mcap, ROWPAD), torch.float32)
        dh0 = self.buf("db.dh0", Mcap, d)
        self.ln_bwd("emb", 0, h0, "dec.ln_emb", ste[0], ste[1], dx, dh0, M,
                    in_dropout_p=pd, in_dropout_seed=sd(1))
        if self.defer_embed:
            ops.embed_bwd(ids, pos_ids, dh0, self.embed_scale, None, P.g("dec.pos"), M, d)
            # the ranks exchange a FIXED number of (id, dh0) rows and add them up deterministically (mic_embed_rows_add_det: the
            # replicas must stay bit-identical): rows that carry no gradient — behind this rank's valid packed rows, or at padded
            # positions (their dh0 is exactly zero) — travel with id -1 and are skipped
            n_x = Mcap if pack is not None else M
            ids_x = self.vec("db.ids_x", Mcap, torch.int32)
            ids_x[:n_x].copy_(ids[:n_x])
            if pack is not None:
                if M < Mcap:
                    ids_x[M:Mcap].fill_(-1)
            elif key_mask is not None:
                ids_x[:n_x].masked_fill_(key_mask.reshape(-1)[:n_x] == 0, -1)
            self.embed_rows = (ids_x, dh0, n_x)
        else:
            # single process: the scatter lands before the optimizer touches the segment (Trainer holds that bucket)
            ops.embed_bwd(ids, pos_ids, dh0, self.embed_scale, P.g("shared"), P.g("dec.pos"), M, d)
            if self.on_embed_rows is not None:
                self.on_embed_rows()  # the tied embedding's gradient is complete (dense head part + these rows): its late optimizer pass may go
        return dehs

    # ------------------------------------------------------------------ loss (main.py:658-680) on materialised logits
    def loss_and_dlogits(self, logits, labels, mask, M: int, label_smoothing: float, backward: bool, stat=None):
        P = self.P
        lse, rl = self.vec("ce.lse", _rup(M, ROWPAD)), self.vec("ce.rowloss", _rup(M, ROWPAD))
        loss, denom = self.vec("ce.loss", 1), self.vec("ce.denom", 1)
        if stat is not None and label_smoothing == 0.0:
            ops.ce_rows_tiles(logits, logits.stride(0), P.V, stat, labels, lse, rl, M)  # lse from the head GEMM's tile partials
        else:
            ops.ce_rows(logits, logits.stride(0), P.V, labels, mask, label_smoothing, lse, rl, M)
        ops.ce_reduce(rl, mask, loss, denom, M)
        if backward and self._head8():
            # fp8 head: dlogits leave as ONE e5m2 copy (the logits stay) under the scale this tensor has in closed form (mic_ce_bwd_q8)
            ops.zero(P.g("flb"))
            q = self.buf("head.dl.q8", logits.shape[0], P.Vpad, torch.float8_e5m2)
            st = self.vec("head.dl.state", 2)
            # ... except the label entry of every row (at least half of the row's gradient energy): an fp32 coefficient whose two
            # products decoder_backward adds exactly (mic_head_label_terms)
            coef = self.vec("head.dl.coef", _rup(logits.shape[0], ROWPAD))
            ops.ce_bwd_q8(logits, logits.stride(0), P.V, P.Vpad, labels, mask, label_smoothing, lse, denom, M, ops.fp8_out(q, st), colsum=P.g("flb"),
                          label_coef=coef)
            self._head8_state = (q, st, coef, labels)
            self._head_nt_state = None
        elif backward:
            kcap = self._head_nt_kcap(logits)
            if kcap:
                # dlogits in place AND transposed [Vpad][Kp] (reduction padding rows M .. Kp zero), final_logits_bias gradient on the way
                Kp = _rup(M, 128)
                dT = self.buf("d.dlogitsT", P.Vpad, kcap)
                ops.zero(P.g("flb"))
                ops.ce_bwd_t(logits, logits.stride(0), P.V, P.Vpad, labels, mask, label_smoothing, lse, denom, M, dT, rows_pad=Kp, colsum=P.g("flb"))
                self._head_nt_state = (dT, Kp)
            else:
                ops.ce_bwd(logits, logits.stride(0), P.V, P.Vpad, labels, mask, label_smoothing, lse, denom, M)
                self._head_nt_state = None
        return loss

    def _head_nt_kcap(self, logits) -> int:
        """columns of the dlogits^T buffer (the capacity of the row dimension, a multiple of 128) when the head's backward runs as NT
        launches of the four-wave kernel, else 0: bf16 storage, a vocabulary worth the copies, operands inside the 2^31-byte
        window that kernel's buffer resources address"""
        P = self.P
        if not self.head_nt or self.dt != torch.bfloat16 or P.Vpad < 16384 or P.d % 128 != 0 or self._head8():
            return 0
        kcap = _rup(logits.shape[0], 128)
        return kcap if P.Vpad * kcap * 2 < 0x7fffffff and logits.shape[0] * logits.stride(0) * 2 < 0x7fffffff else 0

    def shared_T(self):
        """E^T [d][Vpad] (bf16): the k-contiguous copy of the tied embedding that the head's dX GEMM streams (dX = dlogits E reduces
        over the vocabulary).  Rebuilt when the weights changed (ParamStore.version) — by the Trainer on a side stream behind its
        optimizer passes (`refresh_shared_T`), here on the spot otherwise."""
        P = self.P
        ET = self.buf("w.sharedT", P.d, P.Vpad)
        ver = getattr(P, "version", 0)
        if self._ET_version != ver:
            if self._ET_event is not None:  # a side-stream refresh of an older version may still be writing this buffer
                torch.cuda.current_stream().wait_event(self._ET_event)
            ops.transpose_bf16(P.w("shared"), ET, P.Vpad, P.d)
            self._ET_version, self._ET_event = ver, None
        elif self._ET_event is not None:
            torch.cuda.current_stream().wait_event(self._ET_event)
            self._ET_event = None
        return ET

    def refresh_shared_T(self, side, version: int):
        """Trainer, end of a step: rebuild E^T from the updated embedding on `side`, behind everything the current stream has
        enqueued; the next step's head backward waits for it (shared_T).  `version`: the ParamStore.version the copy belongs to."""
        P = self.P
        ET = self._bufs.get(("w.sharedT", P.d, P.Vpad, self.dt))  # (only once a head backward has built it)
        if not self.head_nt or self.dt != torch.bfloat16 or ET is None or self._head8():
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            side.wait_event(ev)
            with ops.pinned_stream():
                ops.transpose_bf16(P.w("shared"), ET, P.Vpad, P.d)
            done = torch.cuda.Event()
            done.record(side)
        self._ET_version, self._ET_event = version, done

    # ------------------------------------------------------------------ full passes
    def forward_logits(self, pixels, ids, pos_ids, key_mask, B, T, *, save=False, seed=None, trunc_int32=False, stats=False):
        self._fp8_begin_pass()
        _, ehs = self.vit_forward(pixels, save, trunc_int32)
        hf = self.decoder_forward(ids, pos_ids, key_mask, ehs, B, T, save, seed)
        if stats:
            logits, stat = self.head_logits(hf, B * T, stats=True, fp8=True)
            return logits, ehs, stat
        return self.head_logits(hf, B * T, fp8=True), ehs

    def compact_head(self, hf, M: int, rows, packed: bool = False):
        """LM head on the loss-relevant rows only: gather hf[idx] -> [Mc, d] (zero pad to a multiple of 64 rows) and
        project.  Exact: loss_fn multiplies every other position by 0 (main.py:678).  packed: hf already holds exactly those
        rows, in that order (the decoder ran on them only): no gather."""
        P = self.P
        idx, Mc = rows
        Mcp = _rup(Mc, 64)
        if packed:
            hfc = hf
        else:
            hfc = self.buf("d.hfc", M, P.d)
            ops.copy_rows(hf, hfc, Mc, P.d, src_idx=idx)
        if Mcp > Mc:
            ops.zero(hfc[Mc:Mcp])
        logits = self.buf("d.logits", M, P.Vpad)
        stat = self.head_stats("d.logits", M)
        self._head_project(hfc, logits, Mc, stat)
        if Mcp > Mc:
            ops.zero(logits[Mc:Mcp])  # reduction padding of the dE GEMM (rows of an earlier, longer batch may linger here)
        return logits, stat

    def loss_only(self, pixels, ids, pos_ids, key_mask, labels, B, T, *, label_smoothing=0.0, rows=None, row_labels=None, pack=None):
        """eval_step's forward + loss (main.py:710-716)."""
        M = B * T
        pack = self._check_pack(pack, rows, T)
        if rows is None:
            logits, _, stat = self.forward_logits(pixels, ids, pos_ids, key_mask, B, T, save=False, seed=None, stats=True)
            return self.loss_and_dlogits(logits, labels, key_mask.reshape(-1), M, label_smoothing, backward=False, stat=stat)
        self._fp8_begin_pass()
        _, ehs = self.vit_forward(pixels, False)
        hf = self.decoder_forward(ids, pos_ids, key_mask, ehs, B, T, False, None, pack=pack)
        logits, stat = self.compact_head(hf, M, rows, packed=pack is not None)
        return self.loss_and_dlogits(logits, row_labels, self.ones_i32(rows[1]), rows[1], label_smoothing, backward=False, stat=stat)

    def _check_pack(self, pack, rows, T: int):
        """packed rows need the bf16-storage kernels (bf16 or fp8 GEMMs), one attention tile per sequence, and the compacted head on
        the same rows"""
        if pack is None:
            return None
        if rows is None or rows[1] != pack[2]:
            raise ValueError("packed decoder rows go with the compacted LM head on the same rows (rows=(idx, n) with n == pack rows)")
        if self.dt != torch.bfloat16 or T > 64 or self.P.S > 64:
            raise ValueError("packed decoder rows: bfloat16 storage mode, seq_len <= 64 and at most 64 encoder positions")
        return pack

    def ones_i32(self, n: int):
        """int32 ones [n] (the loss mask of the compacted rows): one buffer that only ever grows — the row count changes from step to
        step, a buffer per distinct count would accumulate"""
        t = self._bufs.get("ones_i32")
        if t is None or t.numel() < n:
            t = torch.ones(max(_rup(max(n, 1), 4096), 2 * (t.numel() if t is not None else 0)), dtype=torch.int32, device=self.dev)
            self._bufs["ones_i32"] = t
        return t[:n]

    def loss_and_grads(self, pixels, ids, pos_ids, key_mask, labels, B, T, *, label_smoothing=0.0, seed=None, rows=None,
                       row_labels=None, pack=None):
        """value_and_grad(compute_loss) of train_step (main.py:688-697): grads land in ParamStore.grad.
        rows = (idx int32 [Mc] of the positions with loss mask 1, Mc) and row_labels = labels[idx] switch the LM head,
        the cross-entropy and their backward to those rows only (identical loss and gradients, ~(1 - Mc/M) less head work)."""
        P = self.P
        P.ensure_grads()
        ops.zero(P.grad[P.atomic_begin:])
        M = B * T
        pack = self._check_pack(pack, rows, T)
        if rows is None:
            logits, ehs, stat = self.forward_logits(pixels, ids, pos_ids, key_mask, B, T, save=True, seed=seed, stats=True)
            loss = self.loss_and_dlogits(logits, labels, key_mask.reshape(-1), M, label_smoothing, backward=True, stat=stat)
        else:
            self._fp8_begin_pass()
            _, ehs = self.vit_forward(pixels, True)
            hf = self.decoder_forward(ids, pos_ids, key_mask, ehs, B, T, True, seed, pack=pack)
            logits, stat = self.compact_head(hf, M, rows, packed=pack is not None)
            loss = self.loss_and_dlogits(logits, row_labels, self.ones_i32(rows[1]), rows[1], label_smoothing, backward=True, stat=stat)
        dehs = self.decoder_backward(B, T, ids, pos_ids, key_mask, ehs, logits, seed, rows=rows, pack=pack)
        self.vit_backward(B, dehs)
        self.dw_join()
        return loss
