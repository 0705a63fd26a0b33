"""`.generate()` — greedy and beam search with the reference's semantics
(`models/flax_clip_vision_mbart/generation_clip_vision_utils.py:128-336, 368-420, 422-535, 665-990`), driven from the
host as a fixed sequence of HIP launches per step:

    decoder step (KV-cached)  ->  tied head GEMM  ->  mic_row_lse_topk (log-softmax + processors + running score +
    index-stable top-2K)  ->  mic_beam_step (all of gen:872-966 for one step, on device)

Differences in mechanism, not in results: cross-attention K/V are projected once (the reference re-projects every
step), and the self-attention cache is never gathered by beam (gen:945-953) — a slot-ownership table is updated
instead.  `_sample` (gen:537-663): the categorical draw is `mic_sample_rows` (Gumbel-argmax with jax's
threefry2x32 stream); by default it draws from the RAW logits exactly like the reference does (its processed/warped logits
are computed and dropped, gen:620-627); `sample_from_processed_logits=True` applies processors, temperature, top-k and
top-p (`mic_warp_thresholds`).

Decode plans and launch-free decoder steps: a decoder step is ~140 kernel launches whose arguments depend only on (batch,
beams, max_length, processors, step index) — never on the data.  Greedy and beam search keep a `_DecodePlan` per such
configuration: its device state (sequences, scores, KV cache, slot table ...) lives at fixed addresses and is re-initialised in
place per call (no 3 GB cache allocation + zero fill per generate call).  With `MIC_DECODE_GRAPHS=1` the second call of a plan
additionally captures step t of the loop into a hipGraph (one graph per step index: the cache slot, the valid length and the
processor switches are baked-in launch arguments) and from then on a step is ONE graph launch.  The host never waits inside
the loop except for a stop-flag poll every 8 steps (`lax.while_loop` in the reference has no host in it either, gen:976).
MEASURED (MI355X, ROCm 7.2, tools/decode_host_probe.py, ms per decoder step, graphs vs eager): batch 256 x 4 beams 2.71 vs
2.70 (GPU-bound either way), batch 32: 1.61 vs 1.62, batch 4: 1.47 vs 1.57 — a replayed 140-node graph costs what 140
back-to-back launches cost: the ~10 us per dependent kernel are the kernels' own durations (a chain of first-tile miss, K loop,
reduction and epilogue), not launch overhead, so the graphs are correct (tests) but buy nothing for a single chain and stay
opt-in there; fewer, fatter kernels per step are what would move small-batch latency.  A beam search cut into image slices
(MIC_DECODE_SLICES) uses the graphs to run the slices' chains as parallel branches.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops

NEG = -1.0e7
_POLL = 8          # decoder steps between two looks at the device-side stop flag
_MAX_PLANS = 2     # decode plans kept per model (each owns a KV cache: 24 x rows x max_length x d_model elements)
_AUTO_SLICES = 1   # MIC_DECODE_SLICES=auto


# Captured graphs of dropped plans are never destroyed while the process lives.  On this stack (ROCm 7.2) destroying an instantiated
# graph with parallel branches — the sliced beam search — breaks the NEXT multi-branch graph: its replay dies inside
# hip::Graph::UpdateStreams (hipGraphLaunch -> GraphExec::Run; rocgdb backtrace in profiles/README.md).  As long as models lingered in
# reference cycles the graphs happened to outlive every later replay; now that a dropped model frees its device state at once, the
# graph objects (host-side nodes, no device memory of their own: nothing is allocated inside a capture) are parked here instead.
_RETIRED_GRAPHS = []


class _DecodePlan:
    """Fixed-address device state of one generate configuration + its captured decoder steps."""

    def __init__(self):
        self.t = {}          # name -> persistent tensor
        self.graphs = {}     # cur_len -> torch.cuda.CUDAGraph
        self.calls = 0
        self.stream = None   # capture stream
        self.subs = []       # beam search: one sub-plan (state tensors + KV cache) per image slice
        self.streams = []    # ... and the streams their chains run on
        self.graph_events = {}  # cur_len -> the fork / join events recorded while that step was captured (kept as long as the graph)
        self._events_now = []

    def __del__(self):
        # see _RETIRED_GRAPHS: multi-branch graphs (a sliced beam search: `streams` is non-empty) and the events recorded into them
        # outlive the plan; single-chain graphs are destroyed with it (a server cycling through plans must not accumulate them)
        try:
            if self.streams:
                _RETIRED_GRAPHS.extend((g, self.graph_events.get(k)) for k, g in self.graphs.items())
                if _RETIRED_GRAPHS and len(_RETIRED_GRAPHS) % 256 == 0:
                    import sys

                    print(f"[mic_amd.generate] {len(_RETIRED_GRAPHS)} retired multi-branch decode graphs are parked (ROCm 7.2 workaround)", file=sys.stderr)
            self.graphs.clear()
        except Exception:  # interpreter shutdown
            pass

    def new_event(self) -> "torch.cuda.Event":
        """an event for the fork / join of the slices' chains; the ones recorded inside a capture are kept as long as the graph (a
        precaution: the capture turns their records into dependencies of the graph's nodes)"""
        ev = torch.cuda.Event()
        self._events_now.append(ev)
        return ev

    def run_step(self, cur_len: int, fn, use_graphs: bool):
        """fn() issues the launches of decoder step `cur_len` on the current stream.  First call of a plan: eager.  Later calls:
        replay the step's graph, capturing it the first time (capture executes nothing; the replay does)."""
        if not use_graphs or self.calls == 0:
            self._events_now = []
            fn()
            self._events_now = []
            return
        g = self.graphs.get(cur_len)
        if g is None:
            if self.stream is None:
                self.stream = torch.cuda.Stream()
            g = torch.cuda.CUDAGraph()
            self._events_now = []
            with torch.cuda.stream(self.stream):
                g.capture_begin(capture_error_mode="thread_local")
                try:
                    fn()
                finally:
                    g.capture_end()
            self.graphs[cur_len] = g
            self.graph_events[cur_len], self._events_now = self._events_now, []
        g.replay()


def _graphs_enabled(dev, slices: int = 1) -> bool:
    """per-step hipGraphs: opt-in for a single chain (a replayed chain costs what the eager launches cost), ON for a sliced beam
    search — there the graph is what makes the slices' chains parallel branches of one launch (issuing them eagerly would cost
    the host `slices` x 1.4 ms per step); MIC_DECODE_GRAPHS=0|1 forces either"""
    import os

    want = os.environ.get("MIC_DECODE_GRAPHS", "auto")
    if dev.type != "cuda" or want == "0":
        return False
    return want == "1" or slices > 1


class FlaxCLIPVisionMBartGenerationMixin:
    # ------------------------------------------------------------------ gen:128-336
    def generate(self, input_ids, max_length: Optional[int] = None, pad_token_id: Optional[int] = None,
                 bos_token_id: Optional[int] = None, eos_token_id: Optional[int] = None,
                 decoder_start_token_id: Optional[int] = None, do_sample: Optional[bool] = None, prng_key=None,
                 top_k: Optional[int] = None, top_p: Optional[float] = None, temperature: Optional[float] = None,
                 num_beams: Optional[int] = None, no_repeat_ngram_size: Optional[int] = None, min_length: Optional[int] = None,
                 forced_bos_token_id: Optional[int] = None, forced_eos_token_id: Optional[int] = None,
                 length_penalty: Optional[float] = None, early_stopping: Optional[bool] = None, trace: bool = True,
                 params=None, **model_kwargs):
        """`FlaxCLIPVisionGenerationMixin.generate` (gen:128-336): same arguments, defaults from `config.mbart_config`, same dispatch —
        greedy (`num_beams == 1`, no sampling), sampling (`do_sample`, one beam) or beam search; `input_ids` = pixel values.
        Limits of this build (the reference has none of the first two; the third is the reference's own): `num_beams <= 32` — the fused
        per-row top-2K kernel keeps k = 2 * num_beams <= 64 candidates per row, a wider search raises NotImplementedError (the reference's
        evaluation runs num_beams = 4, evaluation.py:80-94); `max_length <= max_position_embeddings` (XLA's gather would clamp silently,
        this raises); beam search with sampling raises NotImplementedError as upstream (gen:336)."""
        mc = self.config.mbart_config
        from_processed = bool(model_kwargs.pop("sample_from_processed_logits", False))  # build-only switch, see _sample
        max_length = max_length if max_length is not None else mc.max_length  # gen:205-209
        bos_token_id = bos_token_id if bos_token_id is not None else mc.bos_token_id
        pad_token_id = pad_token_id if pad_token_id is not None else mc.pad_token_id
        eos_token_id = eos_token_id if eos_token_id is not None else mc.eos_token_id
        decoder_start_token_id = decoder_start_token_id if decoder_start_token_id else mc.decoder_start_token_id  # gen:225-229
        if decoder_start_token_id is None and self.config.is_encoder_decoder:
            raise ValueError("`decoder_start_token_id` has to be defined for encoder-decoder generation.")  # gen:232-235
        do_sample = do_sample if do_sample is not None else mc.do_sample
        num_beams = num_beams if num_beams is not None else mc.num_beams
        if max_length < 2:
            raise ValueError(f"max_length={max_length}: generation needs room for the start token and one more")
        if max_length > mc.max_position_embeddings:
            # the learned position table has max_position_embeddings (+2 offset) rows; XLA's gather would clamp silently
            raise ValueError(f"max_length={max_length} exceeds mbart_config.max_position_embeddings={mc.max_position_embeddings}")
        # gen:109-120: `params=` is NOT forwarded to encode (the encoder always uses self.params); the decoder uses it.
        enc = self.encode(input_ids, return_dict=True, **{k: v for k, v in model_kwargs.items()
                                                          if not (k.startswith("decoder_") or k.startswith("cross_attn"))})
        self._use_params(params)
        ehs = enc["last_hidden_state"]
        B = ehs.shape[0]
        # processors (gen:368-420); `no_repeat_ngram_size` is accepted and ignored like in the reference
        min_length = min_length if min_length is not None else mc.min_length
        forced_bos_token_id = forced_bos_token_id if forced_bos_token_id is not None else mc.forced_bos_token_id
        forced_eos_token_id = forced_eos_token_id if forced_eos_token_id is not None else mc.forced_eos_token_id
        procs = dict(min_length=min_length if (min_length is not None and eos_token_id is not None and min_length > -1) else None,
                     forced_bos=forced_bos_token_id, forced_eos=forced_eos_token_id)
        if not do_sample and num_beams == 1:
            return self._greedy_search(ehs, B, decoder_start_token_id, max_length, pad_token_id, eos_token_id, procs)
        elif do_sample and num_beams == 1:
            mcfg = mc
            top_k = top_k if top_k is not None else getattr(mcfg, "top_k", 50)  # gen:349-356
            top_p = top_p if top_p is not None else getattr(mcfg, "top_p", 1.0)
            temperature = temperature if temperature is not None else getattr(mcfg, "temperature", 1.0)
            return self._sample(ehs, B, decoder_start_token_id, max_length, pad_token_id, eos_token_id, prng_key, procs,
                                dict(top_k=top_k, top_p=top_p, temperature=temperature),
                                from_processed)
        elif not do_sample and num_beams > 1:
            length_penalty = length_penalty if length_penalty is not None else mc.length_penalty  # gen:733-742
            early_stopping = early_stopping if early_stopping is not None else mc.early_stopping
            return self._beam_search(ehs, B, num_beams, decoder_start_token_id, max_length, pad_token_id, eos_token_id,
                                     length_penalty, early_stopping, procs)
        else:
            raise NotImplementedError("`Beam sampling is currently not implemented.")  # gen:336

    @staticmethod
    def _proc_args(procs, cur_len: int, max_length: int, eos_token_id: int):
        """(forced_token, suppress_eos) for this step: MinLength -> ForcedBOS -> ForcedEOS (gen:412-419)."""
        forced = -1
        # FlaxMinLengthLogitsProcessor: apply_penalty = 1 - clip(cur_len - min_length, 0, 1)  =>  EOS is -inf while cur_len <= min_length
        suppress = procs["min_length"] is not None and cur_len <= procs["min_length"]
        if procs["forced_bos"] is not None and cur_len == 1:
            forced = procs["forced_bos"]
        if procs["forced_eos"] is not None and cur_len == max_length - 1:
            forced = procs["forced_eos"]
        return forced, suppress

    # ------------------------------------------------------------------ gen:422-535
    def _decode_plan(self, key, build) -> _DecodePlan:
        """LRU of `_MAX_PLANS` decode plans; `build(plan)` allocates a new plan's tensors."""
        from collections import OrderedDict

        plans = self.__dict__.setdefault("_decode_plans", OrderedDict())
        plan = plans.get(key)
        if plan is None:
            while len(plans) >= _MAX_PLANS:
                plans.popitem(last=False)
            plan = plans[key] = _DecodePlan()
            build(plan)
        plans.move_to_end(key)
        return plan

    def release_decode_plans(self):
        """drop the cached decode plans (their KV caches and captured graphs)"""
        self.__dict__.pop("_decode_plans", None)

    def _plan_cache(self, plan: _DecodePlan, rows: int, max_length: int) -> dict:
        """the plan's self-attention cache (modeling:249-282: static length max_length).  Not re-zeroed between calls: a step
        only reads the slots <= its own index, all written earlier in the same call."""
        if "k0" not in plan.t:
            c = self.init_cache(rows, max_length)
            for l, (k, v) in enumerate(zip(c["k"], c["v"])):
                plan.t[f"k{l}"], plan.t[f"v{l}"] = k, v
        L = self.store.L
        return {"k": [plan.t[f"k{l}"] for l in range(L)], "v": [plan.t[f"v{l}"] for l in range(L)], "cache_index": 0,
                "max_length": max_length, "rows": rows, "src_row": None, "cross": None, "row_div": 1}

    def _greedy_search(self, ehs, B, start_token, max_length, pad_token_id, eos_token_id, procs):
        from .modeling_clip_vision_mbart import ModelOutput

        dev, st = self.device, self.store

        def build(plan):
            t = plan.t
            t["sequences"] = torch.empty((B, max_length), dtype=torch.int32, device=dev)
            t["finished"] = torch.empty(B, dtype=torch.int32, device=dev)
            t["next_token"] = torch.empty((B,), dtype=torch.int32, device=dev)
            t["top_val"] = torch.empty((B, 1), dtype=torch.float32, device=dev)
            t["top_idx"] = torch.empty((B, 1), dtype=torch.int32, device=dev)
            t["pos_all"] = torch.arange(max_length, dtype=torch.int32, device=dev).repeat_interleave(B)

        plan = self._decode_plan(("greedy", B, max_length, pad_token_id, eos_token_id, procs["min_length"], procs["forced_eos"], self.dtype),
                                 build)
        t = plan.t
        sequences, finished, next_token, top_val, top_idx, pos_all = (t[k] for k in ("sequences", "finished", "next_token", "top_val",
                                                                                     "top_idx", "pos_all"))
        sequences.fill_(pad_token_id)  # gen:457
        sequences[:, 0].fill_(start_token)  # (a fill kernel: assigning a Python scalar into a device tensor is a synchronous copy)
        finished.zero_()
        next_token.fill_(start_token)
        cache = self._plan_cache(plan, B, max_length)
        self._decode_set_encoder(cache, ehs.reshape(B * st.S, st.d), B, 1)
        use_graphs = _graphs_enabled(dev)

        def step(cur_len):
            def fn():
                with ops.pinned_stream():
                    cache["cache_index"] = cur_len - 1
                    logits = self._decode_step(cache, next_token, pos_all[(cur_len - 1) * B: cur_len * B])
                    forced, suppress = self._proc_args(procs, cur_len, max_length, eos_token_id)
                    ops.row_lse_topk(logits, logits.stride(0), st.V, 1, top_val, top_idx, B, forced_token=forced, suppress_eos=suppress,
                                     eos_token_id=eos_token_id, raw_logits=True)
                    ops.greedy_step(B, max_length, cur_len, eos_token_id, pad_token_id, top_idx, 1, sequences, finished, next_token)
            return fn

        # gen:480-487 stops when every row has finished.  A finished row only ever receives PAD (gen:501-507) into a buffer that
        # is PAD already, so steps issued after that point change nothing: the flag is looked at every _POLL steps, not every step.
        cur_len = 1
        while cur_len < max_length:
            if (cur_len - 1) % _POLL == 0 and cur_len > 1 and bool(finished.all().item()):
                break
            plan.run_step(cur_len, step(cur_len), use_graphs and cur_len > 1)  # step 1 carries the call's forced-BOS id: never captured
            cur_len += 1
        plan.calls += 1
        return ModelOutput(sequences=sequences.clone())

    # ------------------------------------------------------------------ gen:537-663
    def _sample(self, ehs, B, start_token, max_length, pad_token_id, eos_token_id, prng_key, procs, warp, from_processed: bool):
        from . import prng
        from .modeling_clip_vision_mbart import ModelOutput

        dev, st = self.device, self.store
        use_k = from_processed and warp["top_k"] not in (None, 0)
        use_p = from_processed and warp["top_p"] is not None and warp["top_p"] < 1.0
        thr = torch.empty(B, dtype=torch.float32, device=dev) if (use_k or use_p) else None
        lim = torch.empty(B, dtype=torch.int32, device=dev) if (use_k or use_p) else None
        key = prng.prng_key(0 if prng_key is None else prng_key)  # gen:561
        sequences = torch.full((B, max_length), pad_token_id, dtype=torch.int32, device=dev)
        sequences[:, 0].fill_(start_token)  # (a fill kernel: assigning a Python scalar into a device tensor is a synchronous copy)
        finished = torch.zeros(B, dtype=torch.int32, device=dev)
        next_token = torch.full((B,), start_token, dtype=torch.int32, device=dev)
        cache = self.init_cache(B, max_length)
        self._decode_set_encoder(cache, ehs.reshape(B * st.S, st.d), B, 1)
        drawn = torch.empty((B, 1), dtype=torch.int32, device=dev)
        pos = torch.zeros(B, dtype=torch.int32, device=dev)
        cur_len = 1
        while True:
            if cur_len == max_length or bool(finished.all().item()):  # gen:596-603
                break
            k, key = prng.split(key)  # gen:610
            logits = self._decode_step(cache, next_token, pos)
            if from_processed:
                forced, suppress = self._proc_args(procs, cur_len, max_length, eos_token_id)
                temp = warp["temperature"] or 1.0
                if thr is not None and forced < 0:  # gen:338-366: temperature -> top-k -> top-p, as (threshold, tie limit) per row
                    ops.warp_thresholds(logits, logits.stride(0), st.V, thr, lim, B, temperature=temp, suppress_eos=suppress,
                                        eos_token_id=eos_token_id, top_k=warp["top_k"] if use_k else 0,
                                        top_p=warp["top_p"] if use_p else 1.0)
                ops.sample_rows(logits, logits.stride(0), st.V, k, drawn, B, temperature=temp, forced_token=forced,
                                suppress_eos=suppress, eos_token_id=eos_token_id, min_keep=thr, tie_limit=lim)
            else:  # the reference's behaviour: jax.random.categorical(prng_key, model_outputs.logits[:, -1]) (gen:625-627)
                ops.sample_rows(logits, logits.stride(0), st.V, k, drawn, B)
            ops.greedy_step(B, max_length, cur_len, eos_token_id, pad_token_id, drawn, 1, sequences, finished, next_token)  # gen:629-642
            pos += 1
            cur_len += 1
        return ModelOutput(sequences=sequences)

    # ------------------------------------------------------------------ gen:665-990
    def _decode_slices(self, B: int, K: int) -> int:
        """Number of independent image slices a beam search is cut into (opt-in experiment, default 1 = off).  Images do not interact
        inside a beam-search step, but the reference's STOP test is global over the batch (`beam_search_cond_fn`, gen:798-820:
        `jnp.all(...)` over every image) — one image that can still improve keeps all of them stepping.  Each slice evaluates that
        test over its own images only, so with n > 1 a slice may stop earlier than the reference would have stopped it and its
        sequences / scores can differ from the unsliced search; the default therefore stays at 1.  A decoder step is a chain of ~110
        dependent launches most of which leave the chip idle
        (one 64x64 tile per CU, latency-bound) or use one resource only (decode attention: HBM; GEMMs: MFMA + LDS).  Slices run
        the same chain on their own streams — as parallel branches of the step's hipGraph — so one slice's latency chain and
        HBM-bound kernels overlap another's.  MIC_DECODE_SLICES=n forces n (1 = off)."""
        import os

        want = os.environ.get("MIC_DECODE_SLICES", "auto")
        if self.device.type != "cuda":
            return 1
        n = int(want) if want != "auto" else _AUTO_SLICES
        # auto: a slice keeps at least 256 decoder rows (below that a chain is pure launch latency and slicing only adds launches)
        while n > 1 and (B % n != 0 or (want == "auto" and (B // n) * K < 256)):
            n -= 1
        return max(n, 1)

    def _beam_search(self, ehs, B, K, start_token, max_length, pad_token_id, eos_token_id, length_penalty, early_stopping, procs):
        from .modeling_clip_vision_mbart import ModelOutput

        if 2 * K > 64:
            raise NotImplementedError("num_beams > 32 needs a wider per-row top-k than this build ships (k = 2*num_beams <= 64)")
        dev, st = self.device, self.store
        NS = self._decode_slices(B, K)
        Bs = B // NS          # images per slice
        R = Bs * K            # decoder rows per slice

        def build_slice(sub):
            t = sub.t
            t["running_seq"] = torch.empty((Bs, K, max_length), dtype=torch.int32, device=dev)
            t["seq"] = torch.empty((Bs, K, max_length), dtype=torch.int32, device=dev)
            t["finished"] = torch.empty((Bs, K), dtype=torch.int32, device=dev)
            t["running_scores"] = torch.empty((Bs, K), dtype=torch.float32, device=dev)
            t["scores"] = torch.empty((Bs, K), dtype=torch.float32, device=dev)
            t["next_token"] = torch.empty((R,), dtype=torch.int32, device=dev)
            t["src_row"] = torch.empty((R, max_length), dtype=torch.int32, device=dev)
            t["flags"] = torch.empty((Bs, 2), dtype=torch.int32, device=dev)
            t["cand_val"] = torch.empty((R, 2 * K), dtype=torch.float32, device=dev)
            t["cand_idx"] = torch.empty((R, 2 * K), dtype=torch.int32, device=dev)
            t["pos_all"] = torch.arange(max_length, dtype=torch.int32, device=dev).repeat_interleave(R)  # position ids of every step
            t["gstate"] = torch.empty(8, dtype=torch.int32, device=dev)
            t["score0"] = torch.tensor([0.0] + [NEG] * (K - 1), dtype=torch.float32, device=dev).repeat(Bs, 1).contiguous()
            t["rows"] = torch.arange(R, dtype=torch.int32, device=dev)

        def build(plan):
            plan.subs = [_DecodePlan() for _ in range(NS)]
            for sub in plan.subs:
                build_slice(sub)
            plan.streams = [torch.cuda.Stream(device=dev) for _ in range(NS)] if NS > 1 else []

        plan = self._decode_plan(("beam", B, K, NS, max_length, pad_token_id, eos_token_id, float(length_penalty), bool(early_stopping),
                                  procs["min_length"], procs["forced_eos"], self.dtype), build)
        ehs = ehs.reshape(B, st.S * st.d)
        steps_fn = []
        for i, sub in enumerate(plan.subs):
            t = sub.t
            t["running_seq"].fill_(pad_token_id)  # gen:751-757
            t["running_seq"][:, :, 0].fill_(start_token)
            t["seq"].fill_(pad_token_id)
            t["finished"].zero_()  # gen:760
            t["running_scores"].copy_(t["score0"])  # gen:763-765
            t["scores"].fill_(NEG)  # gen:766
            t["next_token"].fill_(start_token)
            t["src_row"].zero_()
            t["src_row"][:, 0] = t["rows"]
            t["flags"].zero_()
            cache = self._plan_cache(sub, R, max_length)
            cache["src_row"] = t["src_row"]
            cache["ns"] = f"s{i}." if NS > 1 else ""
            # encoder states are shared by an image's K beams (gen:299-307 broadcasts them; here: row r reads image r // K)
            self._decode_set_encoder(cache, ehs[i * Bs:(i + 1) * Bs].reshape(Bs * st.S, st.d), Bs, K)
            # beam_search_cond_fn (gen:798-820) lives on the device: mic_beam_step evaluates it on the new state, and once it says
            # stop every later mic_beam_step launch is a no-op.  The host therefore enqueues decoder steps (graph replays once the
            # plan is warm) without waiting and looks at the flag only every _POLL steps.
            t["gstate"].zero_()
            steps_fn.append(self._beam_step_fn(cache, t, Bs, K, max_length, pad_token_id, eos_token_id, length_penalty, early_stopping, procs))
        use_graphs = _graphs_enabled(dev, NS)

        def step(cur_len):
            if NS == 1:
                return lambda: steps_fn[0](cur_len)

            def fn():
                # fork: every slice's chain on its own stream behind what the current stream has enqueued; join: the current
                # stream waits for all of them (inside a capture these are the graph's parallel branches)
                cur = torch.cuda.current_stream()
                fork = plan.new_event()
                fork.record(cur)
                for i, s_ in enumerate(plan.streams):
                    s_.wait_event(fork)
                    with torch.cuda.stream(s_):
                        steps_fn[i](cur_len)
                        done = plan.new_event()
                        done.record(s_)
                    cur.wait_event(done)
            return fn

        gstates = [sub.t["gstate"] for sub in plan.subs]
        cur_len = 1
        while cur_len < max_length:
            plan.run_step(cur_len, step(cur_len), use_graphs and cur_len > 1)  # step 1 carries the call's forced-BOS id: never captured
            cur_len += 1
            if cur_len < max_length and (cur_len - 1) % _POLL == 0 and all(int(g[3].item()) != 0 for g in gstates):
                break
        plan.calls += 1
        steps = max(int(g[4].item()) for g in gstates)
        outs = []
        for sub in plan.subs:
            t = sub.t
            any_fin = t["finished"].bool().any(dim=1)  # gen:980
            out_seq = torch.where(any_fin[:, None, None], t["seq"], t["running_seq"])  # gen:981-983
            out_scores = torch.where(any_fin[:, None], t["scores"], t["running_scores"])  # gen:984
            outs.append((out_seq[:, -1], out_scores[:, -1]))  # gen:987-990
        out = ModelOutput(sequences=torch.cat([o[0] for o in outs]).contiguous(), scores=torch.cat([o[1] for o in outs]).contiguous())
        out["steps"] = steps
        return out

    def _beam_step_fn(self, cache, t, B, K, max_length, pad_token_id, eos_token_id, length_penalty, early_stopping, procs):
        """the launches of one beam-search step (gen:822-966) of one slice of B images: step(cur_len)"""
        st = self.store
        R = B * K
        running_seq, seq, finished, running_scores, scores = t["running_seq"], t["seq"], t["finished"], t["running_scores"], t["scores"]
        next_token, src_row, flags, cand_val, cand_idx, pos_all, gstate = (t[k] for k in ("next_token", "src_row", "flags", "cand_val",
                                                                                          "cand_idx", "pos_all", "gstate"))

        def step(cur_len):
            with ops.pinned_stream():
                cache["cache_index"] = cur_len - 1
                logits, stat = self._decode_step(cache, next_token, pos_all[(cur_len - 1) * R: cur_len * R], stats=True)  # gen:830-840
                forced, suppress = self._proc_args(procs, cur_len, max_length, eos_token_id)
                if stat is not None and forced < 0 and 2 * K <= 16:
                    # log-softmax + top-2K from the head GEMM's per-granule partials: 3908 pairs and a few 64-column granules
                    # per row instead of two passes over the 250 054 logits (same candidates, same order, same fp32 arithmetic)
                    ops.row_topk_tiles(logits, logits.stride(0), st.V, stat, 2 * K, cand_val, cand_idx, R, suppress_eos=suppress,
                                       eos_token_id=eos_token_id, row_bias=running_scores.reshape(-1))
                else:
                    ops.row_lse_topk(logits, logits.stride(0), st.V, 2 * K, cand_val, cand_idx, R, forced_token=forced,
                                     suppress_eos=suppress, eos_token_id=eos_token_id, row_bias=running_scores.reshape(-1))  # gen:850-873
                ops.beam_step(B, K, max_length, st.V, cur_len, eos_token_id, pad_token_id, length_penalty, early_stopping, cand_val,
                              cand_idx, running_seq, running_scores, seq, scores, finished, src_row, next_token, flags, gstate=gstate)  # gen:872-966
        return step
