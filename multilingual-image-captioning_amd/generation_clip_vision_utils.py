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
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops

NEG = -1.0e7


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
    def _greedy_search(self, ehs, B, start_token, max_length, pad_token_id, eos_token_id, procs):
        from .modeling_clip_vision_mbart import ModelOutput

        dev, st = self.device, self.store
        sequences = torch.full((B, max_length), pad_token_id, dtype=torch.int32, device=dev)  # gen:457
        sequences[:, 0] = start_token
        finished = torch.zeros(B, dtype=torch.int32, device=dev)
        next_token = torch.full((B,), start_token, dtype=torch.int32, device=dev)
        cache = self.init_cache(B, max_length)
        self._decode_set_encoder(cache, ehs.reshape(B * st.S, st.d), B, 1)
        top_val = torch.empty((B, 1), dtype=torch.float32, device=dev)
        top_idx = torch.empty((B, 1), dtype=torch.int32, device=dev)
        pos = torch.zeros(B, dtype=torch.int32, device=dev)
        cur_len = 1
        while True:
            if cur_len == max_length or bool(finished.all().item()):  # gen:480-487
                break
            logits = self._decode_step(cache, next_token, pos)
            forced, suppress = self._proc_args(procs, cur_len, max_length, eos_token_id)
            ops.row_lse_topk(logits, logits.stride(0), st.V, 1, top_val, top_idx, B, forced_token=forced, suppress_eos=suppress,
                             eos_token_id=eos_token_id, raw_logits=True)
            ops.greedy_step(B, max_length, cur_len, eos_token_id, pad_token_id, top_idx, 1, sequences, finished, next_token)
            pos += 1
            cur_len += 1
        return ModelOutput(sequences=sequences)

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
        sequences[:, 0] = start_token
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
    def _beam_search(self, ehs, B, K, start_token, max_length, pad_token_id, eos_token_id, length_penalty, early_stopping, procs):
        from .modeling_clip_vision_mbart import ModelOutput

        if 2 * K > 16:
            raise NotImplementedError("num_beams > 8 needs a wider per-row top-k than this build ships (k = 2*num_beams <= 16)")
        dev, st = self.device, self.store
        R = B * K
        running_seq = torch.full((B, K, max_length), pad_token_id, dtype=torch.int32, device=dev)  # gen:751-757
        running_seq[:, :, 0] = start_token
        seq = torch.full((B, K, max_length), pad_token_id, dtype=torch.int32, device=dev)
        finished = torch.zeros((B, K), dtype=torch.int32, device=dev)  # gen:760
        running_scores = torch.tensor([0.0] + [NEG] * (K - 1), dtype=torch.float32, device=dev).repeat(B, 1).contiguous()  # gen:763-765
        scores = torch.full((B, K), NEG, dtype=torch.float32, device=dev)  # gen:766
        next_token = torch.full((R,), start_token, dtype=torch.int32, device=dev)
        src_row = torch.zeros((R, max_length), dtype=torch.int32, device=dev)
        src_row[:, 0] = torch.arange(R, dtype=torch.int32, device=dev)
        flags = torch.zeros((B, 2), dtype=torch.int32, device=dev)
        cache = self.init_cache(R, max_length)
        cache["src_row"] = src_row
        # encoder states are shared by an image's K beams (gen:299-307 broadcasts them; here: row r reads image r // K)
        self._decode_set_encoder(cache, ehs.reshape(B * st.S, st.d), B, K)
        cand_val = torch.empty((R, 2 * K), dtype=torch.float32, device=dev)
        cand_idx = torch.empty((R, 2 * K), dtype=torch.int32, device=dev)
        pos_all = torch.arange(max_length, dtype=torch.int32, device=dev).repeat_interleave(R)  # position ids of every step
        # beam_search_cond_fn (gen:798-820) lives on the device: mic_beam_step evaluates it on the new state, and once it says
        # stop every later mic_beam_step launch is a no-op.  The host therefore enqueues decoder steps without waiting and
        # looks at the flag only every few steps (the reference's lax.while_loop has no host in the loop either, gen:976).
        gstate = torch.zeros(8, dtype=torch.int32, device=dev)
        poll = 8
        cur_len = 1
        while cur_len < max_length:
            with ops.pinned_stream():
                logits, stat = self._decode_step(cache, next_token, pos_all[(cur_len - 1) * R: cur_len * R], stats=True)  # gen:830-840
                forced, suppress = self._proc_args(procs, cur_len, max_length, eos_token_id)
                if stat is not None and forced < 0:
                    # log-softmax + top-2K from the head GEMM's per-granule partials: 3908 pairs and a few 64-column granules per
                    # row instead of two passes over the 250 054 logits (same candidates, same order, same fp32 arithmetic)
                    ops.row_topk_tiles(logits, logits.stride(0), st.V, stat, 2 * K, cand_val, cand_idx, R, suppress_eos=suppress,
                                       eos_token_id=eos_token_id, row_bias=running_scores.reshape(-1))
                else:
                    ops.row_lse_topk(logits, logits.stride(0), st.V, 2 * K, cand_val, cand_idx, R, forced_token=forced, suppress_eos=suppress,
                                     eos_token_id=eos_token_id, row_bias=running_scores.reshape(-1))  # gen:850-873 (per row)
                ops.beam_step(B, K, max_length, st.V, cur_len, eos_token_id, pad_token_id, length_penalty, early_stopping, cand_val,
                              cand_idx, running_seq, running_scores, seq, scores, finished, src_row, next_token, flags, gstate=gstate)  # gen:872-966
            cur_len += 1
            if cur_len < max_length and (cur_len - 1) % poll == 0 and int(gstate[3].item()) != 0:
                break
        steps = int(gstate[4].item())
        any_fin = finished.bool().any(dim=1)  # gen:980
        out_seq = torch.where(any_fin[:, None, None], seq, running_seq)  # gen:981-983
        out_scores = torch.where(any_fin[:, None], scores, running_scores)  # gen:984
        out = ModelOutput(sequences=out_seq[:, -1].contiguous(), scores=out_scores[:, -1].contiguous())  # gen:987-990
        out["steps"] = steps
        return out
