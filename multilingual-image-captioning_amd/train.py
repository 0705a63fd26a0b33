"""Data-parallel training step with the reference's semantics (`main.py:281-292, 629-635, 658-707, 710-721`):

    loss, grads = value_and_grad(compute_loss)(params)      # per-rank masked-mean loss on the local shard (main.py:679)
    grads = pmean(grads)                                     # RCCL all-reduce over xGMI, bucketed, overlapped (main.py:698)
    params = adamw(params, grads)                            # identical local update on every rank (main.py:701)
    metrics = pmean({loss, learning_rate})                   # main.py:703-704

One process per GPU (`torch.distributed`, backend "nccl" = RCCL on ROCm; "gloo" on CPU for the tests of the host logic).
Gradients live in ONE flat fp32 buffer ordered as backward completes them, so buckets are contiguous slices; each
bucket's all-reduce is issued on a side HIP stream as soon as backward has produced it (event-ordered), overlapping the
rest of backward; AdamW waits on the side stream once.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch

from . import ops


def shift_tokens_right(input_ids: np.ndarray, pad_token_id: int) -> np.ndarray:
    """main.py:362-369."""
    shifted = np.zeros(input_ids.shape, dtype=np.int64)
    shifted[:, 1:] = input_ids[:, :-1]
    shifted[:, 0] = pad_token_id
    return shifted


def create_learning_rate_fn(train_ds_size: int, train_batch_size: int, num_train_epochs: int, num_warmup_steps: int,
                            learning_rate: float) -> Callable[[int], float]:
    """main.py:281-292: linear warmup 0->lr over `num_warmup_steps`, then linear decay lr->0 over the remaining steps."""
    steps_per_epoch = train_ds_size // train_batch_size
    num_train_steps = steps_per_epoch * num_train_epochs

    def schedule(step: int) -> float:
        if step < num_warmup_steps:
            frac = min(max(step / num_warmup_steps, 0.0), 1.0) if num_warmup_steps > 0 else 1.0
            return learning_rate * frac
        n = num_train_steps - num_warmup_steps
        if n <= 0:  # optax.linear_schedule with transition_steps <= 0 is the constant init_value
            return learning_rate
        frac = min(max((step - num_warmup_steps) / n, 0.0), 1.0)
        return learning_rate + (0.0 - learning_rate) * frac

    return schedule


def loss_rows(attention_mask, labels):
    """Collate-side helper: flat indices of the label positions whose loss mask is 1 (row-major over [B,T]) and the labels
    at those positions.  `Trainer` runs the LM head / cross-entropy only there (exact: main.py:678 multiplies the rest by 0).
    Returns (idx int32 [Mc], labels int32 [Mc]) as numpy arrays."""
    m = np.asarray(attention_mask).reshape(-1)
    idx = np.nonzero(m)[0].astype(np.int32)
    return idx, np.asarray(labels).reshape(-1)[idx].astype(np.int32)


def packed_rows(attention_mask, decoder_input_ids):
    """Collate-side helper for the packed decoder: when every row of the [B,T] loss / attention mask is a PREFIX of ones (what
    the tokenizer's `padding="max_length"` produces, main.py:513-523), the decoder can run on the sum(len_b) valid positions
    only — sequence b = packed rows [q_off[b], q_off[b] + q_len[b]), in the same order as `loss_rows`' indices.  Returns
    (q_off int32 [B], q_len int32 [B], ids int32 [B*T], pos int32 [B*T]) — decoder input ids and position ids of the packed
    rows, zero-padded to B*T — or None when some row is not a prefix (the padded path is used then)."""
    m = np.asarray(attention_mask)
    B, T = m.shape
    ln = m.astype(bool).sum(1).astype(np.int32)
    if not np.array_equal(m.astype(bool), np.arange(T)[None, :] < ln[:, None]):
        return None
    off = np.zeros(B, dtype=np.int32)
    off[1:] = np.cumsum(ln)[:-1]
    idx = np.nonzero(m.reshape(-1))[0]
    ids = np.zeros(B * T, dtype=np.int32)
    pos = np.zeros(B * T, dtype=np.int32)
    ids[: idx.size] = np.asarray(decoder_input_ids).reshape(-1)[idx]
    pos[: idx.size] = (idx % T).astype(np.int32)  # default position ids arange(T) (modeling:490-494)
    return off, ln, ids, pos


def plan_buckets(numel: int, bucket_elems: int, boundaries: List[int], max_elems: int = 0,
                 splittable: Tuple[Tuple[int, int, int], ...] = ()) -> List[Tuple[int, int]]:
    """Contiguous [begin, end) slices of the flat gradient buffer, cut at segment boundaries, each >= bucket_elems
    (except the last).  `max_elems` > 0: a slice larger than that is cut further, but only INSIDE the segments listed in
    `splittable` ((begin, end, align): cuts at begin + k * step, step = the largest multiple of `align` <= max_elems) — the tied
    embedding is ONE segment of 256 M elements (1 GB in fp32): whole, its optimizer pass could not start before the last byte of
    its all-reduce had arrived.  Pure host logic (tested on CPU)."""
    cuts, start = [], 0
    for b in sorted(set(boundaries)):
        if b - start >= bucket_elems and b < numel:
            cuts.append((start, b))
            start = b
    cuts.append((start, numel))
    if max_elems <= 0:
        return cuts
    out = []
    for (b, e) in cuts:
        pts = set()
        if e - b > max_elems:
            for (sb, se, align) in splittable:
                step = max(align, (max_elems // align) * align)
                x = sb + step
                while x < min(e, se):
                    if x > b:
                        pts.add(x)
                    x += step
        edges = [b] + sorted(pts) + [e]
        out += [(edges[i], edges[i + 1]) for i in range(len(edges) - 1)]
    return out


MAX_BUCKET_MB = 128.0  # largest piece of the tied embedding that goes on the wire as one collective


TAIL_BUCKET_MB = 12.0  # the LAST bucket (what backward completes last = what the next forward reads first) is kept this small


def bucket_plan(store, bucket_mb: float = 64.0, max_mb: float = MAX_BUCKET_MB, tail_mb: float = TAIL_BUCKET_MB) -> List[Tuple[int, int]]:
    """the gradient-exchange buckets of a ParamStore layout (a store built with allocate=False is enough): >= bucket_mb each, cut
    at segment boundaries in the order backward completes them; the tied embedding in pieces of <= max_mb (whole rows).
    tail_mb > 0: the last bucket is cut once more, at the latest segment boundary that leaves a final piece of at most tail_mb —
    its exchange and optimizer pass are the serial tail of the step (nothing is left to hide them under, and the next forward pass
    starts with exactly these weights): 12 MB of patch embedding / position / pre-LayerNorm parameters instead of 64+ MB."""
    bounds = [s.offset for s in store.segs.values()]
    sh = store.segs["shared"]
    split = ((sh.offset, sh.offset + sh.numel, store.d),) if sh.numel % store.d == 0 else ()
    out = plan_buckets(store.numel, int(bucket_mb * 1024 * 1024 / 4), bounds, max_elems=int(max(max_mb, bucket_mb) * 1024 * 1024 / 4),
                       splittable=split)
    if tail_mb > 0:
        # (the very last bucket is the small region of atomically accumulated bias / LayerNorm gradients; the one in front of it ends
        # with the patch embedding) ... and the piece in front of those to ~3 x that (one ViT layer): its pass is released by the
        # last layer's backward and has the patch-embedding kernels left to hide under
        for k, lim_mb in ((1, tail_mb), (2, tail_mb), (3, 3.0 * tail_mb)):
            if len(out) < k:
                break
            b, e = out[-k]
            lim = int(lim_mb * 1024 * 1024 / 4)
            inside = sorted(x for x in set(bounds) if b < x < e and e - x <= lim)
            if e - b > lim and inside:
                out[len(out) - k:len(out) - k + 1] = [(b, inside[0]), (inside[0], e)]
    return out


def describe_buckets(store, buckets: List[Tuple[int, int]], comm_bytes: int = 4) -> List[Dict]:
    """per bucket: first / last segment it covers, elements and bytes on the wire per rank and exchange (logged by rank 0 at
    Trainer construction; the order is the order backward completes them)"""
    segs = sorted(store.segs.values(), key=lambda s: s.offset)
    out = []
    for i, (b, e) in enumerate(buckets):
        inside = [s.name for s in segs if s.offset < e and s.offset + s.numel > b]
        out.append({"bucket": i, "begin": b, "end": e, "elements": e - b, "MB": round((e - b) * comm_bytes / 1e6, 1),
                    "first": inside[0], "last": inside[-1], "segments": len(inside)})
    return out


# ---- what the gradient exchange costs, from byte counts (no N > 1 hardware has been available to measure it)
XGMI_LINK_GBPS = 64.0      # per direction and peer link, what a ring gets out of one xGMI link (7 links x ~153 GB/s bidirectional per GPU)
BACKWARD_WINDOW_MS = 13.0  # backward of the batch-64 step: the time the exchange can hide under (profiles/README.md)
COMM_CUS_DEFAULT = 32      # RCCL channels (= persistent blocks, one CU each) `configure_rccl` caps the collectives at


def configure_rccl(max_channels: int = COMM_CUS_DEFAULT) -> int:
    """Call BEFORE `torch.distributed.init_process_group("nccl")`: caps the CUs RCCL's collectives can occupy by bounding its channel
    count (one persistent block per channel, one CU each): `NCCL_MAX_NCHANNELS` (and `NCCL_MIN_NCHANNELS` no larger than that).  A
    value the user already exported wins.  Returns the cap now in force; `Trainer` reads it back (`rccl_channel_cap`) as its
    `comm_cus` default, so the CU budget the GEMM tile planner works with is the number RCCL was actually held to — not an assumption."""
    import os

    cap = int(os.environ.setdefault("NCCL_MAX_NCHANNELS", str(int(max_channels))))
    lo = os.environ.get("NCCL_MIN_NCHANNELS")
    if lo is not None and int(lo) > cap:
        os.environ["NCCL_MIN_NCHANNELS"] = str(cap)
    return cap


def rccl_channel_cap() -> int:
    """the channel cap exported to RCCL (`configure_rccl` or the user's environment); 0 = none: RCCL picks its own count"""
    import os

    try:
        return max(0, int(os.environ.get("NCCL_MAX_NCHANNELS", "0")))
    except ValueError:
        return 0


XGMI_LINK_PEAK_GBPS = 153.0  # per peer link as the hardware guide and SURVEY §5 quote it (the optimistic end of the projection)


def allreduce_ms(nbytes: float, world: int, link_gbps: float = XGMI_LINK_GBPS, algo: str = "all_links") -> float:
    """PROJECTED all-reduce time of `nbytes` per rank among `world` fully connected GPUs (no N > 1 hardware run has been possible:
    a byte-count model, not a measurement).
    algo="all_links" (default): every rank sends and receives 2 (world-1)/world of its bytes spread over its world-1 peer links —
      what a direct reduce-scatter + all-gather does, and what world-1 concurrent rings do: 2 nbytes / (world * link).
    algo="ring_one_link": a single classic ring, one link per direction: 2 (world-1)/world * nbytes / link.
    `link_gbps`: XGMI_LINK_GBPS (64, the conservative default: what RCCL rings have been seen to get out of one link) or
    XGMI_LINK_PEAK_GBPS (153, SURVEY §5's direct RS + AG figure: 3.6 ms for the 2.19 GB fp32 gradient at N = 8)."""
    if world <= 1:
        return 0.0
    if algo == "ring_one_link":
        return 2.0 * (world - 1) / world * nbytes / (link_gbps * 1e9) * 1e3
    if algo != "all_links":
        raise ValueError(f"unknown all-reduce model {algo!r}")
    return 2.0 * nbytes / (world * link_gbps * 1e9) * 1e3


def allreduce_projections(nbytes: float, world: int) -> Dict[str, float]:
    """the bracket the byte-count model gives: pessimistic (all links at 64 GB/s — what `choose_comm_dtype` and the emulated
    exchange use), optimistic (direct reduce-scatter + all-gather over all links at 153 GB/s) and a single ring"""
    return {"all_links_64GBps_ms": round(allreduce_ms(nbytes, world), 2),
            "direct_rs_ag_153GBps_ms": round(allreduce_ms(nbytes, world, XGMI_LINK_PEAK_GBPS), 2),
            "single_ring_153GBps_ms": round(allreduce_ms(nbytes, world, XGMI_LINK_PEAK_GBPS, "ring_one_link"), 2)}


def choose_comm_dtype(world: int, numel: int, window_ms: float = BACKWARD_WINDOW_MS, link_gbps: float = XGMI_LINK_GBPS) -> Optional[torch.dtype]:
    """`grad_comm_dtype="auto"`: fp32 — the reference's `lax.pmean` of fp32 gradients (main.py:698) — wherever the projected fp32
    exchange fits under backward, bf16 where it would not (the step would be communication-bound; a bf16 sum changes pmean's
    arithmetic by one rounding per rank and element, tests/test_ddp_gpu.py).  By the byte counts of the 547 M-parameter model at
    the conservative 64 GB/s per link: N = 2: 34 ms, N = 4: 17 ms -> bf16 (17 / 8.5 ms); N = 8: 8.5 ms -> fp32.  With
    `link_gbps=XGMI_LINK_PEAK_GBPS` (direct reduce-scatter + all-gather at 153 GB/s: 14.3 / 7.2 / 3.6 ms) only N = 2 is over the
    window; pass the link rate RCCL was MEASURED at once a multi-GPU node has run tools/first_multi_gpu_run.sh."""
    if world <= 1 or allreduce_ms(4.0 * numel, world, link_gbps) <= window_ms:
        return None
    return torch.bfloat16


class GradReducer:
    """Bucketed all-reduce(mean) of the flat gradient buffer on a side stream (C1 of SURVEY §2.3), followed — on a third stream —
    by a per-bucket callback (the fused AdamW of that slice): HBM-bound optimizer traffic then overlaps the MFMA-bound remainder
    of backward instead of trailing it.

    ONE ordering rule per bucket, `ready_when[i]`: when may the bucket's optimizer pass be issued, beyond "its exchange is done"?
      "exchanged"  right behind the exchange (the default);
      "next"       at the next progress report, behind everything backward has enqueued by then — a kernel issued right after
                   the gradient became final still READS the weights (the LM head's dX GEMM reads the tied embedding);
      "end"        in finish(), after `before()` — the gradient receives a late part (the sparse input-embedding rows).
    """

    def __init__(self, flat_grad: torch.Tensor, buckets: List[Tuple[int, int]], group=None, on_ready=None,
                 ready_when: Optional[List[str]] = None, comm_dtype: Optional[torch.dtype] = None, opt_cus: int = 0,
                 emulate: Optional[Dict] = None):
        """opt_cus: CUs of the optimizer stream's mask (0: no mask).  emulate (one GPU only, bench.py --emulate-comm):
        {"world": N, "cus": CUs of the collective stream, "comm_bytes": bytes per element on the wire} — every bucket's exchange
        is replaced by a kernel that occupies `cus` CUs for the projected duration of that bucket's all-reduce."""
        import torch.distributed as dist

        self.dist, self.group = dist, group
        self.grad, self.buckets = flat_grad, buckets
        self.ready_when = list(ready_when) if ready_when is not None else ["exchanged"] * len(buckets)
        assert len(self.ready_when) == len(buckets) and all(w in ("exchanged", "next", "end") for w in self.ready_when)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.cuda = flat_grad.is_cuda
        self.stream = ops.role_stream(flat_grad.device, "collective") if self.cuda else None    # collectives
        self.opt_stream = ops.role_stream(flat_grad.device, "optimizer") if self.cuda else None  # per-bucket optimizer work
        # ... on CUs of its own (`opt_cus` of them; 0 = no mask): the optimizer is HBM-bound and needs few CUs, while the backward
        # GEMM blocks own a CU's whole register file and never share one — with a mask the two run side by side instead of taking
        # turns.  A CU-masked HIP stream is a BLOCKING stream (it synchronises with the null stream, torch's default): the step then
        # runs on `step_stream`, and the passes postponed to the end of backward on the unmasked `tail_stream`.
        self.step_stream = self.tail_stream = None
        self.emulate = dict(emulate) if (emulate and self.cuda and self.world == 1) else None
        if self.cuda and on_ready is not None and opt_cus > 0:
            try:
                self.opt_stream = ops.cu_masked_stream(0, opt_cus, flat_grad.device)
                self.step_stream = ops.role_stream(flat_grad.device, "step")
                self.tail_stream = ops.role_stream(flat_grad.device, "tail")
            except Exception as ex:  # the stack refuses CU masks: plain streams, as before
                import sys

                print(f"[mic_amd.GradReducer] no CU-masked optimizer stream ({ex}); using a plain stream", file=sys.stderr)
        if self.emulate is not None:
            # the stand-in collectives sit on the LAST `cus` bits of the CU mask (the optimizer's mask starts at bit 0)
            cus = int(self.emulate.get("cus", COMM_CUS_DEFAULT))
            self.stream = ops.cu_masked_stream(256 - cus, cus, flat_grad.device)
            if self.step_stream is None:  # a masked stream is a blocking stream: keep the step off the null stream
                self.step_stream = ops.role_stream(flat_grad.device, "step")
            self._emu_scratch = torch.empty(max(e - b for b, e in buckets), dtype=flat_grad.dtype, device=flat_grad.device)
            self.emulated_ms = 0.0
            self.emulated_events = []
        self.on_ready = on_ready if self.cuda else None
        self._finishing = False
        self.host_s = 0.0
        self.next = 0
        self.handles = []
        self._pending: List[Tuple[int, int, object]] = []  # "next": exchange started, optimizer pass due at the next report
        self._at_end: List[Tuple[int, int]] = []           # "end"
        # opt-in reduced-precision exchange (e.g. torch.bfloat16): every rank rounds its bucket, the collective sums in that
        # dtype, the result is widened back into the fp32 buffer.  Halves the xGMI bytes; NOT the reference's fp32 pmean.
        self.comm_dtype = comm_dtype if (comm_dtype is not None and comm_dtype != flat_grad.dtype) else None
        # two staging buffers used alternately: the narrowing copy of bucket k+1 does not wait for the widening copy of bucket k
        self.stage = ([torch.empty(max(e - b for b, e in buckets), dtype=self.comm_dtype, device=flat_grad.device) for _ in range(2)]
                      if self.comm_dtype is not None and self.world > 1 else None)
        self._stage_i = 0

    def _all_reduce(self, b: int, e: int, async_op: bool = False):
        if self.stage is None:
            return self.dist.all_reduce(self.grad[b:e], op=self.dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        st = self.stage[self._stage_i][: e - b]
        self._stage_i ^= 1
        st.copy_(self.grad[b:e])
        self.dist.all_reduce(st, op=self.dist.ReduceOp.SUM, group=self.group)
        self.grad[b:e].copy_(st)
        return None

    def _emulated_exchange(self, b: int, e: int):
        """one GPU: hold the collective stream's CUs for as long as the all-reduce of this bucket is projected to take"""
        em = self.emulate
        nbytes = (e - b) * int(em.get("comm_bytes", 4))
        ms = allreduce_ms(nbytes, int(em["world"]), float(em.get("link_gbps", XGMI_LINK_GBPS)))
        self.emulated_ms += ms
        ev = None
        if em.get("timing"):  # bench.py: how long the stand-in kernels really held the stream
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        with ops.pinned_stream():
            ops.comm_emulate(self.grad[b:e], self._emu_scratch[: e - b], (e - b) * self.grad.element_size(), ms * 1e3,
                             2 * int(em.get("cus", COMM_CUS_DEFAULT)))
        if ev is not None:
            ev[1].record()
            self.emulated_events.append(ev)

    @property
    def active(self) -> bool:
        return self.world > 1 or self.on_ready is not None or self.emulate is not None

    def start_step(self):
        self.host_s = 0.0
        self.next = 0
        self.handles = []
        self._pending = []
        self._at_end = []
        if self.emulate is not None:
            self.emulated_ms = 0.0
            self.emulated_events = []

    def _issue(self, b: int, e: int, ev, here=None):
        """the bucket's optimizer pass on the optimizer stream, behind its exchange (`ev`) and, for "next" buckets, behind
        everything the step's stream had enqueued at the report that released them (`here`).  Buckets released by finish() — backward
        is over, nothing shares the chip with them — go to the UNMASKED tail stream when the optimizer stream is restricted to a few
        CUs: the pass is HBM-bound and the step waits for it."""
        st = self.tail_stream if (self._finishing and self.tail_stream is not None) else self.opt_stream
        with torch.cuda.stream(st):
            st.wait_event(ev)
            if here is not None:
                st.wait_event(here)
            with ops.pinned_stream():  # launch on that stream, not on the step's pinned main stream
                self.on_ready(b, e)

    def _issue_pending(self):
        if not self._pending:
            return
        here = torch.cuda.Event()
        here.record(torch.cuda.current_stream())
        for (b, e, ev) in self._pending:
            self._issue(b, e, ev, here)
        self._pending = []

    def progress(self, offset_done: int):
        """Backward reports that every gradient with flat offset < offset_done is final."""
        if not self.active:
            return
        import time

        t0 = time.perf_counter()
        try:
            self._progress(offset_done)
        finally:
            self.host_s += time.perf_counter() - t0  # host time this step spent issuing exchanges / optimizer passes (bench.py reports it)

    def _progress(self, offset_done: int):
        if self.cuda:
            self._issue_pending()
        while self.next < len(self.buckets) and self.buckets[self.next][1] <= offset_done:
            b, e = self.buckets[self.next]
            when = self.ready_when[self.next]
            if self.cuda:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
                if self.world > 1 or self.emulate is not None:
                    with torch.cuda.stream(self.stream):
                        self.stream.wait_event(ev)
                        if self.world > 1:
                            self._all_reduce(b, e)
                        else:
                            self._emulated_exchange(b, e)
                        ev = torch.cuda.Event()
                        ev.record(self.stream)
                if self.on_ready is None:
                    pass
                elif when == "end":
                    self._at_end.append((b, e))
                elif when == "next":
                    self._pending.append((b, e, ev))
                else:
                    self._issue(b, e, ev)
            else:  # CPU tensors (host-logic tests over gloo)
                h = self._all_reduce(b, e, async_op=True)
                if h is not None:
                    self.handles.append(h)
            self.next += 1

    def finish(self, before=None, after=None):
        """End of backward: the remaining buckets go out; then, behind every collective and everything backward has enqueued,
        `before()` (the sparse embedding-row exchange), the optimizer passes of the "end" buckets and `after()` (the part of the
        optimizer that needed `before`); finally the step's stream waits for the side streams."""
        self._finishing = True
        try:
            self.progress(self.grad.numel())
        finally:
            self._finishing = False
        if not self.active:
            return
        import time

        t0 = time.perf_counter()
        try:
            self._finish(before, after)
        finally:
            self.host_s += time.perf_counter() - t0

    def _finish(self, before, after):
        if not self.cuda:
            for h in self.handles:
                h.wait()
            return
        self._issue_pending()
        if self.on_ready is None:
            after = None  # the late pass belongs to the per-bucket optimizer
            self._at_end = []
        cur = torch.cuda.current_stream()
        if self._at_end or before is not None or after is not None:
            # these run after backward, with nothing beside them: on the unmasked tail stream when the optimizer stream is
            # restricted to a few CUs
            tail = self.tail_stream if self.tail_stream is not None else self.opt_stream
            with torch.cuda.stream(tail):
                if self.world > 1 or self.emulate is not None:
                    tail.wait_stream(self.stream)
                if tail is not self.opt_stream:
                    tail.wait_stream(self.opt_stream)
                tail.wait_stream(cur)  # the end of backward (the sparse embedding rows come from there)
                with ops.pinned_stream():
                    if before is not None:
                        before()
                    for (b, e) in self._at_end:
                        self.on_ready(b, e)
                    if after is not None:
                        after()
            if tail is not self.opt_stream:
                self.opt_stream.wait_stream(tail)
            self._at_end = []
        cur.wait_stream(self.stream)
        cur.wait_stream(self.opt_stream)
        if self.tail_stream is not None and self.tail_stream is not self.opt_stream:
            # the optimizer passes of the buckets finish() released ran on the tail stream (`_issue`): without this join the next forward
            # pass could read weights those passes were still writing whenever nothing else had been queued there (found by making a side
            # stream lag: tests/test_fp8_gpu.py::test_fp8_weight_copies_made_on_a_lagging_side_stream_are_waited_for)
            cur.wait_stream(self.tail_stream)


OPT_CUS_DEFAULT = 96  # CUs of the optimizer stream (12 per XCD of an MI355X): profiles/README.md, round 3 A/B


class Trainer:
    """TrainState + train_step/eval_step of main.py, for one rank."""

    def __init__(self, model, learning_rate_fn: Callable[[int], float], b1=0.9, b2=0.999, eps=1e-8, weight_decay=0.0,
                 label_smoothing_factor=0.0, seed: int = 42, bucket_mb: float = 64.0, group=None, compact_head: bool = True,
                 overlap_optimizer: bool = True, grad_comm_dtype=None, gemm_dtype: Optional[str] = None,
                 fp8_scaling: str = "delayed", pack_rows: bool = True, comm_cus: Optional[int] = None, emulate_comm: int = 0,
                 fp8_head: Optional[str] = None):
        """pack_rows (bfloat16 mode, with compact_head): the decoder runs on the valid caption positions only (`packed_rows`);
        exact — padded positions neither carry loss nor are attended to — and ~1/3 fewer decoder rows on ragged captions.  Needs
        prefix-shaped masks available on the host: a numpy / CPU `attention_mask`, or `batch["packed_rows"]` from the collate
        function; otherwise the step runs padded.  MIC_PACK_ROWS=0 switches it off.
        grad_comm_dtype: None / torch.float32 (default: the reference's fp32 `lax.pmean`, main.py:698), torch.bfloat16 (opt-in: half
        the xGMI bytes, one more rounding per rank and element) or "auto" (opt-in; `choose_comm_dtype`: fp32 where the PROJECTED
        exchange hides under backward, bf16 where it would not — a projection from byte counts, no N > 1 run backs it).  A
        non-fp32 choice is printed by every rank.
        comm_cus (data parallel): CUs the collectives hold while backward runs — the GEMM tile planner sizes its one-round
        launches for the rest (`mic_set_cu_budget`).  Default: the channel cap exported to RCCL (`configure_rccl` before
        `init_process_group`, or NCCL_MAX_NCHANNELS in the environment) when world > 1 — the budget then equals what RCCL was held to;
        0 (planner untouched) when no cap is in force.  A user-set MIC_FREE_CUS wins over both.
        emulate_comm = N (one GPU only; bench.py --emulate-comm): every bucket's exchange is replaced by a kernel holding
        `comm_cus` CUs for the projected duration of its all-reduce among N ranks — a scheduling probe, not a scaling result."""
        import torch.distributed as dist

        self.model, self.lr_fn = model, learning_rate_fn
        self.b1, self.b2, self.eps, self.wd, self.ls = b1, b2, eps, weight_decay, label_smoothing_factor
        self.compact_head = compact_head
        import os as _os
        self.pack_rows = bool(pack_rows) and compact_head and _os.environ.get("MIC_PACK_ROWS", "1") != "0"
        self.step = 0  # state.step
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.group = group
        self.dropout_seed = (seed * 1000003 + self.rank * 7919) & 0xFFFFFFFF  # per-device dropout stream (main.py:251)
        st = model.store
        st.ensure_grads()
        st.ensure_opt_state()
        if gemm_dtype is not None:
            model.engine.set_gemm_dtype(gemm_dtype, scaling=fp8_scaling, head=fp8_head)  # fp8_head: "all" (default) | "bwd" | "0", see Engine
        self.hyper = torch.zeros(2, dtype=torch.float32, device=model.device)
        self._hyper_pin, self._hyper_ev = None, None
        emu_world = int(emulate_comm) if (self.world == 1 and model.device.type == "cuda") else 0
        eff_world = emu_world if emu_world > 1 else self.world
        if isinstance(grad_comm_dtype, str):
            if grad_comm_dtype != "auto":
                raise ValueError(f"grad_comm_dtype must be 'auto', None or a torch dtype, got {grad_comm_dtype!r}")
            grad_comm_dtype = choose_comm_dtype(eff_world, st.numel)
        self.grad_comm_dtype = grad_comm_dtype if grad_comm_dtype not in (None, torch.float32) else None
        if self.grad_comm_dtype is not None and eff_world > 1:
            import sys

            print(f"[mic_amd.Trainer] rank {self.rank}: gradient exchange in {self.grad_comm_dtype} — NOT the reference's fp32 pmean "
                  "(main.py:698); pass grad_comm_dtype=None for fp32", file=sys.stderr, flush=True)
        if comm_cus is not None:
            self.comm_cus = int(comm_cus)
        elif emu_world > 1:
            self.comm_cus = COMM_CUS_DEFAULT   # the stand-in collectives are held to exactly this many CUs by their stream's mask
        else:
            self.comm_cus = min(rccl_channel_cap(), 128) if self.world > 1 else 0
        # one-round GEMM launches are sized for the CUs the collectives leave (process-wide planner state: set for the duration of a
        # train step, see train_step); a MIC_FREE_CUS exported by the user is the user's budget and stays in force
        self._cu_budget = 256 - self.comm_cus if (model.device.type == "cuda" and eff_world > 1 and self.comm_cus > 0
                                                  and not _os.environ.get("MIC_FREE_CUS")) else 0
        self.buckets = bucket_plan(st, bucket_mb)
        cb = 2 if self.grad_comm_dtype in (torch.bfloat16, torch.float16) else 4
        if eff_world > 1 and self.rank == 0:
            import sys

            desc = describe_buckets(st, self.buckets, cb)
            total = sum(d["MB"] for d in desc)
            print(f"[mic_amd.Trainer] data parallel over {eff_world} ranks{' (EMULATED on one GPU)' if emu_world > 1 else ''}: {len(desc)} gradient "
                  f"buckets in backward-completion order, {total:.0f} MB per exchange in {'bf16' if cb == 2 else 'fp32'} (all-reduce, projected "
                  f"{allreduce_ms(total * 1e6, eff_world):.1f} ms over xGMI against ~{BACKWARD_WINDOW_MS:.0f} ms of backward); "
                  + (f"RCCL capped at {self.comm_cus} channels, GEMM tile planner sized for {256 - self.comm_cus} CUs" if self._cu_budget else
                     "no RCCL channel cap in force (mic_amd.train.configure_rccl before init_process_group), GEMM tile planner sized for all CUs")
                  + f"; first: {desc[0]['first']}..{desc[0]['last']} {desc[0]['MB']:.0f} MB, last: {desc[-1]['first']}..{desc[-1]['last']} "
                  f"{desc[-1]['MB']:.0f} MB; sizes MB {[d['MB'] for d in desc]}", file=sys.stderr, flush=True)
        # MIC_OPT_OVERLAP=0: AdamW as one launch after backward (profiling aid: per-bucket AdamW on its own stream shares HBM with
        # the backward kernels it overlaps, so their individual durations read longer than the kernels are)
        import os

        if os.environ.get("MIC_OPT_OVERLAP", "1") == "0":
            overlap_optimizer = False
        self.overlap_optimizer = overlap_optimizer
        sh = st.segs["shared"]
        # The tied embedding's gradient is complete only at the end of backward (its sparse input-embedding rows): AdamW runs
        # early on every row this step's decoder ids do not touch and late on the <= world*B*T rows they do (`_adamw_slice` /
        # `_adamw_shared_late`; MIC_OPT_SPLIT_SHARED=0: the whole segment waits for the end).  MIC_OPT_CUS: CUs of the optimizer
        # stream's mask.
        self._split_shared = (self.overlap_optimizer and model.device.type == "cuda"
                              and _os.environ.get("MIC_OPT_SPLIT_SHARED", "1") != "0" and sh.numel % st.d == 0)
        self._sh = (sh.offset, sh.offset + sh.numel, st.d)
        self._row_flag = torch.zeros(sh.numel // st.d, dtype=torch.uint8, device=model.device) if self._split_shared else None
        opt_cus = int(_os.environ.get("MIC_OPT_CUS", str(OPT_CUS_DEFAULT))) if (model.device.type == "cuda" and self._split_shared) else 0
        import weakref

        me = weakref.ref(self)  # the reducer's callbacks must not own the Trainer (Trainer -> reducer -> bound method -> Trainer is a cycle
        #                         that keeps ~16 GB of device state alive after `del trainer, model` until the cycle collector runs)
        # the pieces of the tied embedding: "next" (the head's dX GEMM, issued right behind the dE GEMM that completes the dense
        # half of the gradient, still reads the weights) when the row split is on, "end" otherwise
        gate = "next" if self._split_shared else "end"
        when = [gate if (b < sh.offset + sh.numel and e > sh.offset) else "exchanged" for (b, e) in self.buckets]
        self.reducer = GradReducer(st.grad, self.buckets, group, on_ready=(lambda b, e: me()._adamw_slice(b, e)) if self.overlap_optimizer else None,
                                   ready_when=when, comm_dtype=self.grad_comm_dtype, opt_cus=opt_cus,
                                   emulate=dict(world=emu_world, cus=self.comm_cus or COMM_CUS_DEFAULT, comm_bytes=cb) if emu_world > 1 else None)
        self.metrics_buf = torch.zeros(2, dtype=torch.float32, device=model.device)
        self._pos = {}
        self._det_ws = None  # workspace of the deterministic embedding-row scatter (data parallel)
        self._late_done = False
        self._late_early = _os.environ.get("MIC_LATE_EARLY", "1") != "0"  # (A/B: 0 = the flagged embedding rows' pass behind the step, as before)

    def _adamw_slice(self, b: int, e: int):
        """AdamW on flat slice [b, e) — runs on the reducer's side stream right after that bucket's all-reduce.  The part of the
        slice inside the tied embedding (whole rows: the bucket cuts inside that segment are row-aligned) skips the rows flagged
        for this step when the row split is on (`_adamw_shared_late` takes those)."""
        st = self.model.store
        sb, se, width = self._sh
        lo, hi = max(b, sb), min(e, se)
        if not self._split_shared or lo >= hi:
            ops.adamw(st.master[b:e], st.m[b:e], st.v[b:e], st.grad[b:e], None if st.lp is st.master else st.lp[b:e], self.hyper,
                      self.b1, self.b2, self.eps, self.wd, grad_scale=1.0 / self.world, n=e - b)
            self.model.engine.fp8_requantize_range(b, e)  # fp8 GEMMs: the bucket's weights as fp8 for the NEXT step, here under backward
            return
        assert (lo - sb) % width == 0 and (hi - sb) % width == 0, "bucket cuts inside the tied embedding are row-aligned"
        for (x, y) in ((b, lo), (hi, e)):
            if y > x:
                ops.adamw(st.master[x:y], st.m[x:y], st.v[x:y], st.grad[x:y], None if st.lp is st.master else st.lp[x:y], self.hyper,
                          self.b1, self.b2, self.eps, self.wd, grad_scale=1.0 / self.world, n=y - x)
        self._adamw_shared(0, (lo - sb) // width, (hi - sb) // width)
        self.model.engine.fp8_requantize_range(b, e)

    def _adamw_shared(self, want: int, r0: int = 0, r1: Optional[int] = None):
        """AdamW on rows [r0, r1) of the tied embedding whose flag equals `want`"""
        st = self.model.store
        sb, se, width = self._sh
        r1 = (se - sb) // width if r1 is None else r1
        x, y = sb + r0 * width, sb + r1 * width
        ops.adamw_rows(r1 - r0, width, self._row_flag[r0:r1], want, st.master[x:y], st.m[x:y], st.v[x:y], st.grad[x:y],
                       None if st.lp is st.master else st.lp[x:y], self.hyper, self.b1, self.b2, self.eps, self.wd,
                       grad_scale=1.0 / self.world)

    def _adamw_shared_late(self):
        """end of backward (after the sparse embedding rows have been added): the flagged rows of the tied embedding"""
        self._adamw_shared(1)

    def _embed_rows_done(self):
        """engine hook (world 1): the late pass of the tied embedding on the optimizer stream, behind everything enqueued so far"""
        r = self.reducer
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        with torch.cuda.stream(r.opt_stream):
            r.opt_stream.wait_event(ev)
            with ops.pinned_stream():
                self._adamw_shared_late()
            ev2 = torch.cuda.Event()
            ev2.record(r.opt_stream)
        self._late_done = True
        # fp8 GEMMs: the weights whose optimizer passes are on the optimizer stream by now (the decoder's) become fp8 for the next step
        # here, under the ViT's backward
        if r.next > 0:
            # ... up to the first bucket whose optimizer pass has been released but not ISSUED yet ("next" buckets wait for the following
            # progress report, "end" buckets for finish()): its weights are not final behind `ev2`
            upto = r.buckets[r.next - 1][1]
            for held in (r._pending, r._at_end):
                if held:
                    upto = min(upto, min(h[0] for h in held))
            if upto > 0:
                self.model.engine.fp8_refresh_weights(ops.role_stream(self.model.device, "aux"), upto=upto, wait_event=ev2)

    def _flag_embedding_rows(self, ids: torch.Tensor):
        """flags the rows of `shared` that this step's decoder input ids (of every rank) will add a sparse gradient to"""
        if self.world > 1:
            import torch.distributed as dist

            all_ids = torch.empty((self.world * ids.numel(),), dtype=ids.dtype, device=ids.device)
            dist.all_gather_into_tensor(all_ids, ids.contiguous(), group=self.group)
            ids = all_ids
        ops.row_flags(ids, ids.numel(), self._row_flag)

    def _scatter_embedding_rows(self):
        """Data parallel: the sparse half of the tied-embedding gradient.  Every rank all-gathers (ids, dh0) — <= B*T rows
        per rank — and adds ALL ranks' rows into its already all-reduced dense half (sum over ranks, like the all-reduce; 1/world
        is applied by AdamW) with the DETERMINISTIC scatter: the same bits on every rank, so the replicas' embeddings stay
        identical (the atomics of the single-process scatter do not guarantee an order).  Runs after the last collective."""
        import torch.distributed as dist

        eng, st = self.model.engine, self.model.store
        ids, dh0, M = eng.embed_rows
        all_ids = torch.empty((self.world * M,), dtype=ids.dtype, device=ids.device)
        all_dh = torch.empty((self.world * M, st.d), dtype=dh0.dtype, device=dh0.device)
        dist.all_gather_into_tensor(all_ids, ids[:M].contiguous(), group=self.group)
        dist.all_gather_into_tensor(all_dh, dh0[:M].contiguous(), group=self.group)
        if self._det_ws is None:
            self._det_ws = ops.embed_rows_add_det_workspace(st.g("shared").numel() // st.d, all_ids.device)
        ops.embed_rows_add_det(all_ids, all_dh, eng.embed_scale, st.g("shared").view(-1, st.d), self.world * M, st.d, self._det_ws)

    def _prep(self, batch):
        m = self.model
        px = m._dev(batch["pixel_values"], torch.float32)
        labels = m._dev(batch["input_ids"], torch.int32)
        mask = m._dev(batch["attention_mask"], torch.int32)
        dec_in = m._dev(batch["decoder_input_ids"], torch.int32)
        B, T = labels.shape
        pos = self._pos.get((B, T))  # position ids 0 .. T-1 per sequence: built once per batch shape (two launches per step otherwise)
        if pos is None:
            pos = self._pos[(B, T)] = torch.arange(T, dtype=torch.int32, device=m.device)[None].expand(B, T).contiguous()
        rows = row_labels = None
        pk = batch.get("packed_rows") if self.pack_rows else None
        if self.compact_head:
            if "loss_rows" in batch:  # precomputed by the collate function (no device->host sync in the step)
                idx, rl = batch["loss_rows"]
                idx, rl = m._dev(idx, torch.int32), m._dev(rl, torch.int32)
            else:
                am = batch["attention_mask"]
                am = am.cpu().numpy() if isinstance(am, torch.Tensor) else np.asarray(am)
                lb = batch["input_ids"]
                lb = lb.cpu().numpy() if isinstance(lb, torch.Tensor) else np.asarray(lb)
                idx, rl = loss_rows(am, lb)
                idx, rl = m._dev(idx, torch.int32), m._dev(rl, torch.int32)
                if self.pack_rows and pk is None:
                    di = batch["decoder_input_ids"]
                    pk = packed_rows(am, di.cpu().numpy() if isinstance(di, torch.Tensor) else np.asarray(di))
            if 0 < idx.numel() < B * T:
                rows, row_labels = (idx, int(idx.numel())), rl
        # packed decoder rows: bf16-storage kernels, one attention tile per sequence, compacted head on the same rows
        self._pack = None
        eng = m.engine
        if pk is not None and rows is not None and eng.dt == torch.bfloat16 and T <= 64 and m.store.S <= 64:
            ql = pk[1]
            if not (isinstance(ql, torch.Tensor) and ql.is_cuda):
                # host-side description: it must describe the same positions as `loss_rows` (a collate function that built the two
                # from different masks would make the attention kernels index rows the LM head does not own).  Device-resident
                # descriptions are trusted: checking them would put a device-to-host sync into every step.
                n_pk = int(np.asarray(ql).sum()) if not isinstance(ql, torch.Tensor) else int(ql.sum())
                if n_pk != rows[1]:
                    raise ValueError(f"batch['packed_rows'] describes {n_pk} decoder rows but batch['loss_rows'] has {rows[1]}: both must come from the same attention_mask")
            q_off, q_len, ids_p, pos_p = (m._dev(t, torch.int32) for t in pk)
            self._pack = ((q_off, q_len, rows[1]), ids_p, pos_p)
        self._rows, self._row_labels = rows, row_labels
        return px, labels, mask, dec_in, pos, B, T

    def _set_hyper(self, lr: float, count: float) -> None:
        """(lr, bias-correction count) of this step into the device pair AdamW reads.  From PINNED host memory: a host-to-device
        copy out of pageable memory is carried out by the runtime synchronously behind everything queued on the stream, i.e.
        it was a full device sync per step (the host could not queue step k+1 under step k)."""
        if self.hyper.device.type != "cuda":
            self.hyper.copy_(torch.tensor([lr, count], dtype=torch.float32))
            return
        if self._hyper_pin is None:
            self._hyper_pin = torch.empty((8, 2), dtype=torch.float32).pin_memory()
            self._hyper_ev = [None] * 8
        slot = self.step % 8
        if self._hyper_ev[slot] is not None:
            self._hyper_ev[slot].synchronize()  # the copy that last read this slot (8 steps ago) has run
        self._hyper_pin[slot, 0] = lr
        self._hyper_pin[slot, 1] = count
        self.hyper.copy_(self._hyper_pin[slot], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._hyper_ev[slot] = ev

    def train_step(self, batch: Dict) -> Dict[str, float]:
        """main.py:684-707."""
        if self._cu_budget:
            ops.set_cu_budget(self._cu_budget)
        try:
            ms = self.reducer.step_stream
            if ms is None:
                return self._train_step(batch)
            # the optimizer's stream carries a CU mask, which makes it a blocking stream: the step keeps off the null stream
            cur = torch.cuda.current_stream()
            ms.wait_stream(cur)
            with torch.cuda.stream(ms):
                out = self._train_step(batch)
            cur.wait_stream(ms)
            return out
        finally:
            if self._cu_budget:
                ops.set_cu_budget(0)  # generation / evaluation have no collective beside them

    def _train_step(self, batch: Dict) -> Dict[str, float]:
        m, st, eng = self.model, self.model.store, self.model.engine
        px, labels, mask, dec_in, pos, B, T = self._prep(batch)
        seed = (self.dropout_seed + self.step * 0x9E3779B1) & 0xFFFFFFFF  # split(dropout_rng) per step (main.py:686)
        lr = float(self.lr_fn(self.step))  # schedule at the pre-increment count, bias correction with count+1
        self._set_hyper(lr, float(self.step + 1))
        self.reducer.start_step()
        eng.grad_progress = self.reducer.progress if self.reducer.active else None
        eng.defer_embed = self.world > 1
        # single process: the flagged rows of the tied embedding get their optimizer pass as soon as decoder backward has scattered the
        # input-embedding rows into them — under the ViT's backward instead of behind the step (0.09 ms + a stream hand-over)
        self._late_done = False
        eng.on_embed_rows = self._embed_rows_done if (self.world == 1 and self._split_shared and self.reducer.on_ready is not None
                                                      and self._late_early) else None
        with ops.pinned_stream():
            if self._split_shared:
                self._flag_embedding_rows(self._pack[1] if self._pack is not None else dec_in.reshape(-1))
            if self._pack is not None:  # the decoder sees the valid positions only
                pack, ids_p, pos_p = self._pack
                loss = eng.loss_and_grads(px, ids_p, pos_p, None, labels.reshape(-1), B, T, label_smoothing=self.ls, seed=seed,
                                          rows=self._rows, row_labels=self._row_labels, pack=pack)
            else:
                loss = eng.loss_and_grads(px, dec_in.reshape(-1), pos.reshape(-1), mask, labels.reshape(-1), B, T,
                                          label_smoothing=self.ls, seed=seed, rows=self._rows, row_labels=self._row_labels)
        # pmean(grad) (main.py:698): SUM here, 1/world folded into AdamW's grad_scale.  The remaining buckets go out; behind them the
        # sparse embedding-row exchange, the optimizer passes that had to wait for it, and the flagged rows of the tied embedding
        eng.on_embed_rows = None
        self.reducer.finish(before=self._scatter_embedding_rows if self.world > 1 else None,
                            after=self._adamw_shared_late if (self._split_shared and not self._late_done) else None)
        if not self.overlap_optimizer:
            ops.adamw(st.master, st.m, st.v, st.grad, None if st.lp is st.master else st.lp, self.hyper, self.b1, self.b2, self.eps,
                      self.wd, grad_scale=1.0 / self.world)
            eng.fp8_requantize_range(0, st.numel)
        m.invalidate_params_cache(by_optimizer=True)
        if m.device.type == "cuda":
            eng.fp8_refresh_weights(ops.role_stream(m.device, "aux"))  # fp8 GEMMs: next step's weight copies, beside the glue between the steps
            # the k-contiguous copy of the tied embedding for the NEXT step's head backward: behind this step's optimizer passes, on a
            # stream of its own — it has the whole next forward pass to finish (engine.shared_T waits for it)
            eng.refresh_shared_T(ops.role_stream(m.device, "aux"), st.version)
        self.step += 1
        if self.world == 1:
            # one launch instead of three (two copies + a clone) in the gap between two steps; fresh storage per step (callers keep these)
            out = torch.cat((loss.reshape(1), self.hyper[0:1]))
            return {"loss": out[0], "learning_rate": out[1]}
        self.metrics_buf[0:1].copy_(loss)
        self.metrics_buf[1:2].copy_(self.hyper[0:1])  # lr, device to device (a Python scalar assigned into a device tensor syncs)
        return self._pmean_metrics()

    def eval_step(self, batch: Dict) -> Dict[str, float]:
        """main.py:710-721 (train=False, no dropout)."""
        m, eng = self.model, self.model.engine
        px, labels, mask, dec_in, pos, B, T = self._prep(batch)
        if self._pack is not None:
            pack, ids_p, pos_p = self._pack
            loss = eng.loss_only(px, ids_p, pos_p, None, labels.reshape(-1), B, T, label_smoothing=self.ls, rows=self._rows,
                                 row_labels=self._row_labels, pack=pack)
        else:
            loss = eng.loss_only(px, dec_in.reshape(-1), pos.reshape(-1), mask, labels.reshape(-1), B, T, label_smoothing=self.ls,
                                 rows=self._rows, row_labels=self._row_labels)
        self.metrics_buf[0:1].copy_(loss)
        self.metrics_buf[1:2].zero_()
        out = self._pmean_metrics()
        out.pop("learning_rate")
        return out

    # ------------------------------------------------------------------ checkpoints (main.py:299-345)
    def save_checkpoint(self, save_dir: str, with_opt: bool = False, overwrite: bool = False) -> str:
        """`save_model_checkpoint`: `<save_dir>/ckpt-<step-1>/` with config.json + flax_model.msgpack, and with
        `with_opt` also opt_state.msgpack + training_state.json.  Rank 0 writes (state is replicated, main.py:300)."""
        import os

        from .checkpoint import save_train_state

        ckpt = os.path.join(save_dir, f"ckpt-{self.step - 1}")
        if self.rank != 0 or (os.path.exists(ckpt) and not overwrite):  # main.py:304-305
            return ckpt
        torch.cuda.synchronize(self.model.device) if self.model.device.type == "cuda" else None
        self.model.save_pretrained(ckpt)
        if with_opt:
            save_train_state(ckpt, self.model.store, self.step)
        return ckpt

    def restore_checkpoint(self, ckpt_dir: str) -> int:
        """`restore_model_checkpoint` (main.py:330-345) — applied to this trainer (the reference's `state.replace` line is
        commented out, 345): parameters, AdamW moments, step counter.  Returns the restored step."""
        from .checkpoint import load_train_state

        self.step = load_train_state(ckpt_dir, self.model.store)
        self.model.invalidate_params_cache()
        return self.step

    def _pmean_metrics(self) -> Dict[str, torch.Tensor]:
        """pmean of the 2 scalars (main.py:703-704, 719).  Returned as device tensors (no host sync in the step)."""
        if self.world > 1:
            import torch.distributed as dist

            dist.all_reduce(self.metrics_buf, op=dist.ReduceOp.SUM, group=self.group)
            self.metrics_buf /= self.world
        out = self.metrics_buf.clone()  # fresh storage per step: callers keep these around (main.py:776 `train_metrics.append`)
        return {"loss": out[0], "learning_rate": out[1]}
