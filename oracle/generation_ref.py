"""ORACLE — TEST INFRASTRUCTURE ONLY.  numpy restatement of the reference's own generation code
(`models/flax_clip_vision_mbart/generation_clip_vision_utils.py`, cited as `gen:`), line by line.

The decoder is abstracted as a *stepper*:
    stepper.step(tokens: int32[R]) -> float32[R, V]   logits of the next position (one new token per row)
    stepper.reorder(src_rows: int64[R]) -> None       row r continues from old row src_rows[r]
so the same algorithm runs on the CPU model oracle, on scripted fake decoders (KATs), and is what the
HIP path is compared against.

The three logits processors are third-party (`transformers@0085e71 generation_flax_logits_process.py`,
imported at gen:10-18) and restated from their published semantics.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Tuple

import numpy as np

NEG = np.float32(-1.0e7)  # gen:764, 766, 811, 890, 919


# ------------------------------------------------------------------ helpers
def top_k(x: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
    """`lax.top_k` on the last axis: values descending, lowest index first among equals."""
    idx = np.argsort(-x, axis=-1, kind="stable")[..., :k]
    return np.take_along_axis(x, idx, axis=-1), idx


def log_softmax(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.float32)
    m = x.max(axis=-1, keepdims=True)
    s = x - m
    return s - np.log(np.exp(s).sum(axis=-1, keepdims=True, dtype=np.float32))


# ------------------------------------------------------------------ logits processors (T4)
@dataclass
class MinLength:
    min_length: int
    eos_token_id: int

    def __call__(self, input_ids, scores, cur_len):
        if cur_len <= self.min_length:  # apply_penalty = 1 - clip(cur_len - min_length, 0, 1): 1 while cur_len <= min_length
            scores = scores.copy()
            scores[:, self.eos_token_id] = -np.inf
        return scores


@dataclass
class ForcedBOS:
    bos_token_id: int

    def __call__(self, input_ids, scores, cur_len):
        if cur_len == 1:
            new = np.full_like(scores, -np.inf)
            new[:, self.bos_token_id] = 0.0
            return new
        return scores


@dataclass
class ForcedEOS:
    max_length: int
    eos_token_id: int

    def __call__(self, input_ids, scores, cur_len):
        if cur_len == self.max_length - 1:
            new = np.full_like(scores, -np.inf)
            new[:, self.eos_token_id] = 0.0
            return new
        return scores


def get_logits_processor(min_length, max_length, eos_token_id, forced_bos_token_id, forced_eos_token_id) -> List[Callable]:
    """gen:368-420.  Order Min -> ForcedBOS -> ForcedEOS (412-419); `no_repeat_ngram_size` is accepted and ignored."""
    procs: List[Callable] = []
    if min_length is not None and eos_token_id is not None and min_length > -1:
        procs.append(MinLength(min_length, eos_token_id))
    if forced_bos_token_id is not None:
        procs.append(ForcedBOS(forced_bos_token_id))
    if forced_eos_token_id is not None:
        procs.append(ForcedEOS(max_length, forced_eos_token_id))
    return procs


def _apply(procs, input_ids, scores, cur_len):
    for pr in procs:
        scores = pr(input_ids, scores, cur_len)
    return scores


# ------------------------------------------------------------------ greedy (gen:422-535)
def greedy_search(stepper, batch_size: int, start_token: int, max_length: int, pad_token_id: int,
                  eos_token_id: int, procs: List[Callable]) -> np.ndarray:
    sequences = np.full((batch_size, max_length), pad_token_id, dtype=np.int32)  # gen:457
    sequences[:, 0] = start_token  # gen:458
    finished = np.zeros(batch_size, dtype=bool)  # gen:461
    running = sequences[:, 0].copy()
    cur_len = 1
    while not (cur_len == max_length or finished.all()):  # gen:480-487
        logits = stepper.step(running).astype(np.float32)  # gen:491-494
        logits = _apply(procs, sequences, logits, cur_len)  # gen:497
        nxt = np.argmax(logits, axis=-1).astype(np.int32)  # gen:499 (first max on ties)
        finished = finished | (nxt == eos_token_id)  # gen:501-503
        nxt = np.where(finished, pad_token_id, nxt).astype(np.int32)  # gen:504-507: EOS itself -> PAD
        sequences[:, cur_len] = nxt  # gen:510-512
        running = nxt
        cur_len += 1
    return sequences


# ------------------------------------------------------------------ beam (gen:665-990)
@dataclass
class BeamResult:
    sequences: np.ndarray  # [B, max_length]
    scores: np.ndarray  # [B]
    steps: int  # number of decoder steps executed


def beam_search(stepper, batch_size: int, num_beams: int, start_token: int, max_length: int, pad_token_id: int,
                eos_token_id: int, length_penalty: float, early_stopping: bool, procs: List[Callable]) -> BeamResult:
    B, K = batch_size, num_beams
    sequences = np.full((B, K, max_length), pad_token_id, dtype=np.int32)  # gen:751-753
    running_sequences = sequences.copy()  # gen:754-757
    running_sequences[:, :, 0] = start_token
    finished = np.zeros((B, K), dtype=bool)  # gen:760
    running_scores = np.tile(np.array([0.0] + [NEG] * (K - 1), dtype=np.float32), (B, 1))  # gen:763-765
    scores = np.full((B, K), NEG, dtype=np.float32)  # gen:766
    cur_len = 1
    lp = np.float32(length_penalty)
    batch_ar = np.arange(B)[:, None]
    steps = 0

    def cond() -> bool:  # gen:798-820
        not_max = cur_len < max_length
        best_running = running_scores[:, -1:] / np.float32(float(max_length) ** float(length_penalty))  # gen:805-807
        worst_finished = np.where(finished, scores.min(axis=1, keepdims=True), NEG)  # gen:808-812
        improve = bool(np.all(worst_finished < best_running))  # gen:813-815
        still_open = not (bool(finished.all()) and early_stopping)  # gen:818
        return not_max and still_open and improve

    first = True
    while first or cond():  # body once unconditionally (gen:969), then the while_loop (gen:976)
        first = False
        tok = running_sequences[:, :, cur_len - 1].reshape(B * K)  # gen:830-836
        logits = stepper.step(tok).astype(np.float32).reshape(B, K, -1)  # gen:837-840
        V = logits.shape[-1]
        steps += 1
        logp = log_softmax(logits)  # gen:850
        logp = _apply(procs, running_sequences.reshape(B * K, -1), logp.reshape(B * K, V), cur_len).reshape(B, K, V)  # gen:851-856
        logp = logp + running_scores[:, :, None]  # gen:857
        flat = logp.reshape(B, K * V)  # gen:859
        topk_lp, topk_idx = top_k(flat, 2 * K)  # gen:872-873
        topk_beam = topk_idx // V  # gen:874
        topk_seq = running_sequences[batch_ar, topk_beam].copy()  # gen:875-877
        topk_seq[:, :, cur_len] = (topk_idx % V).astype(np.int32)  # gen:878-881
        just_fin = topk_seq[:, :, cur_len] == eos_token_id  # gen:889
        topk_lp = topk_lp + just_fin.astype(np.float32) * NEG  # gen:890
        nxt = top_k(topk_lp, K)[1][:, ::-1]  # gen:895-897 (flip -> ascending, best last)
        next_running_sequences = topk_seq[batch_ar, nxt]  # gen:898-903
        next_running_scores = topk_lp[batch_ar, nxt]
        topk_lp = topk_lp / (np.float32(cur_len) ** lp)  # gen:910
        full = np.broadcast_to(finished.all(axis=-1, keepdims=True), just_fin.shape) & early_stopping  # gen:911-917
        add_penalty = (~just_fin) | full  # gen:918
        topk_lp = topk_lp + add_penalty.astype(np.float32) * NEG  # gen:919
        merged_seq = np.concatenate([sequences, topk_seq], axis=1)  # gen:925-927
        merged_scores = np.concatenate([scores, topk_lp], axis=1)  # gen:928
        merged_fin = np.concatenate([finished, just_fin], axis=1)  # gen:929-931
        mi = top_k(merged_scores, K)[1][:, ::-1]  # gen:932-934
        sequences = merged_seq[batch_ar, mi]  # gen:935-940
        scores = merged_scores[batch_ar, mi]
        finished = merged_fin[batch_ar, mi]
        parent = topk_beam[batch_ar, nxt]  # gen:945-947
        stepper.reorder((np.arange(B)[:, None] * K + parent).reshape(-1))  # gen:948-953
        running_sequences, running_scores = next_running_sequences, next_running_scores.astype(np.float32)
        cur_len += 1  # gen:959
    any_fin = finished.any(axis=1)  # gen:980 (named `none_finished` there)
    out_seq = np.where(any_fin[:, None, None], sequences, running_sequences)  # gen:981-983
    out_scores = np.where(any_fin[:, None], scores, running_scores)  # gen:984
    return BeamResult(out_seq[:, -1], out_scores[:, -1], steps)  # gen:987-990


# ------------------------------------------------------------------ generate dispatcher (gen:128-336)
@dataclass
class GenDefaults:
    """Generation defaults the reference reads off `config.mbart_config` (mbart-large-50 hub values)."""
    max_length: int = 200
    pad_token_id: int = 1
    bos_token_id: int = 0
    eos_token_id: int = 2
    decoder_start_token_id: Optional[int] = 2
    num_beams: int = 5
    do_sample: bool = False
    min_length: int = 0
    forced_bos_token_id: Optional[int] = None
    forced_eos_token_id: Optional[int] = 2
    length_penalty: float = 1.0
    early_stopping: bool = True


def generate(make_stepper: Callable[[int], object], batch_size: int, defaults: GenDefaults, max_length=None,
             pad_token_id=None, eos_token_id=None, decoder_start_token_id=None, do_sample=None, num_beams=None,
             min_length=None, forced_bos_token_id=None, forced_eos_token_id=None, length_penalty=None,
             early_stopping=None):
    """`make_stepper(rows)` builds a decoder stepper for `rows` flat rows (row = batch*num_beams + beam),
    with encoder states expanded the way gen:299-307 + 773-776 do."""
    d = defaults
    max_length = max_length if max_length is not None else d.max_length  # gen:205-209
    pad_token_id = pad_token_id if pad_token_id is not None else d.pad_token_id
    eos_token_id = eos_token_id if eos_token_id is not None else d.eos_token_id
    decoder_start_token_id = decoder_start_token_id if decoder_start_token_id else d.decoder_start_token_id  # gen:225-229 truthiness
    if decoder_start_token_id is None:
        raise ValueError("`decoder_start_token_id` has to be defined for encoder-decoder generation.")  # gen:232-235
    do_sample = do_sample if do_sample is not None else d.do_sample
    num_beams = num_beams if num_beams is not None else d.num_beams
    min_length = min_length if min_length is not None else d.min_length  # gen:389-393
    forced_bos_token_id = forced_bos_token_id if forced_bos_token_id is not None else d.forced_bos_token_id
    forced_eos_token_id = forced_eos_token_id if forced_eos_token_id is not None else d.forced_eos_token_id
    procs = get_logits_processor(min_length, max_length, eos_token_id, forced_bos_token_id, forced_eos_token_id)
    if not do_sample and num_beams == 1:
        return greedy_search(make_stepper(batch_size), batch_size, decoder_start_token_id, max_length,
                             pad_token_id, eos_token_id, procs)
    if not do_sample and num_beams > 1:
        length_penalty = length_penalty if length_penalty is not None else d.length_penalty  # gen:733-742
        early_stopping = early_stopping if early_stopping is not None else d.early_stopping
        return beam_search(make_stepper(batch_size * num_beams), batch_size, num_beams, decoder_start_token_id,
                           max_length, pad_token_id, eos_token_id, length_penalty, early_stopping, procs)
    if do_sample and num_beams == 1:
        raise NotImplementedError("use generation_ref.sample() directly (it needs a PRNG key and the warpers)")
    raise NotImplementedError("`Beam sampling is currently not implemented.")  # gen:336


# ------------------------------------------------------------------ sampling (gen:338-366, 537-663)
# jax 0.2.16 PRNG (threefry2x32) restated from the published algorithm [3P]; pinned in tests/test_oracle_cpu.py on the
# Random123 known-answer vectors, `split(PRNGKey(0))` and `uniform(PRNGKey(0))` as printed in the JAX documentation.
_U = np.uint32
_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))


def threefry2x32(key, x0, x1):
    x0, x1 = np.array(x0, dtype=_U), np.array(x1, dtype=_U)
    k0, k1 = _U(key[0]), _U(key[1])
    ks = (k0, k1, _U(k0 ^ k1 ^ _U(0x1BD11BDA)))
    with np.errstate(over="ignore"):
        x0, x1 = (x0 + ks[0]).astype(_U), (x1 + ks[1]).astype(_U)
        for g in range(5):
            for r in _ROT[g % 2]:
                x0 = (x0 + x1).astype(_U)
                x1 = ((x1 << _U(r)) | (x1 >> _U(32 - r))).astype(_U)
                x1 = (x1 ^ x0).astype(_U)
            x0 = (x0 + ks[(g + 1) % 3]).astype(_U)
            x1 = (x1 + ks[(g + 2) % 3] + _U(g + 1)).astype(_U)
    return x0, x1


def prng_key(seed: int) -> np.ndarray:
    return np.array([(int(seed) >> 32) & 0xFFFFFFFF, int(seed) & 0xFFFFFFFF], dtype=_U)


def random_bits(key, n: int) -> np.ndarray:
    cnt = np.arange(n, dtype=_U)
    if n % 2:
        cnt = np.concatenate([cnt, np.zeros(1, _U)])
    h = len(cnt) // 2
    a, b = threefry2x32(key, cnt[:h], cnt[h:])
    return np.concatenate([a, b])[:n]


def prng_split(key, num: int = 2) -> np.ndarray:
    return random_bits(key, 2 * num).reshape(num, 2)


def uniform(key, shape, minval=0.0, maxval=1.0) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    bits = random_bits(key, n)
    f = ((bits >> _U(9)) | _U(0x3F800000)).view(np.float32) - np.float32(1.0)
    lo, hi = np.float32(minval), np.float32(maxval)
    return np.maximum(lo, (f * (hi - lo) + lo).astype(np.float32)).reshape(shape)


def gumbel(key, shape) -> np.ndarray:
    u = uniform(key, shape, minval=np.finfo(np.float32).tiny, maxval=1.0)
    return (-np.log(-np.log(u))).astype(np.float32)


def categorical(key, logits: np.ndarray) -> np.ndarray:
    return np.argmax(gumbel(key, logits.shape) + logits.astype(np.float32), axis=-1).astype(np.int32)


class TemperatureWarper:
    def __init__(self, temperature: float):
        self.t = np.float32(temperature)

    def __call__(self, input_ids, scores, cur_len):
        return (scores / self.t).astype(np.float32)


class TopKWarper:
    def __init__(self, k: int, min_tokens_to_keep: int = 1):
        self.k = max(int(k), min_tokens_to_keep)

    def __call__(self, input_ids, scores, cur_len):
        k = min(self.k, scores.shape[-1])
        v, i = top_k(scores, k)  # index-stable, like lax.top_k
        out = np.full_like(scores, -np.inf)
        np.put_along_axis(out, i, v, axis=-1)
        return out


class TopPWarper:
    def __init__(self, p: float, min_tokens_to_keep: int = 1):
        self.p, self.keep = np.float32(p), min_tokens_to_keep

    def __call__(self, input_ids, scores, cur_len):
        v, i = top_k(scores, scores.shape[-1])
        e = np.exp(v - v[:, :1])
        cum = np.cumsum((e / e.sum(-1, keepdims=True)).astype(np.float32), axis=-1, dtype=np.float32)
        mask = cum < self.p
        mask[:, 1:] |= mask[:, :-1].copy()  # "include the token that is higher than top_p as well"
        mask[:, 0] = True
        mask[:, : self.keep] = True
        out = np.full_like(scores, -np.inf)
        np.put_along_axis(out, i, np.where(mask, v, -np.inf).astype(scores.dtype), axis=-1)
        return out


def get_logits_warper(top_k_=None, top_p=None, temperature=None) -> List[Callable]:
    """gen:338-366."""
    w: List[Callable] = []
    if temperature is not None and temperature != 1.0:
        w.append(TemperatureWarper(temperature))
    if top_k_ is not None and top_k_ != 0:
        w.append(TopKWarper(top_k_, 1))
    if top_p is not None and top_p < 1.0:
        w.append(TopPWarper(top_p, 1))
    return w


def sample(stepper, batch_size: int, start_token: int, max_length: int, pad_token_id: int, eos_token_id: int, key,
           procs: List[Callable], warpers: List[Callable], sample_from_processed_logits: bool = False) -> np.ndarray:
    """gen:537-663.  The reference computes the processed + warped logits and then draws from the RAW model logits
    (gen:620-627): `sample_from_processed_logits=False` reproduces that; True is the evidently intended behaviour."""
    sequences = np.full((batch_size, max_length), pad_token_id, dtype=np.int32)
    sequences[:, 0] = start_token
    finished = np.zeros(batch_size, dtype=bool)
    running = sequences[:, 0].copy()
    cur_len = 1
    key = np.asarray(key, dtype=_U)
    while not (cur_len == max_length or finished.all()):  # gen:596-603
        k, key = prng_split(key)  # gen:610
        raw = stepper.step(running).astype(np.float32)
        logits = _apply(procs, sequences, raw, cur_len)  # gen:620
        for w in warpers:  # gen:622
            logits = w(sequences, logits, cur_len)
        nxt = categorical(k, logits if sample_from_processed_logits else raw)  # gen:625-627
        finished = finished | (nxt == eos_token_id)
        nxt = np.where(finished, pad_token_id, nxt).astype(np.int32)
        sequences[:, cur_len] = nxt
        running = nxt
        cur_len += 1
    return sequences


# ------------------------------------------------------------------ steppers
class ModelStepper:
    """Drives `oracle.model_ref.decode_step` with the reference's cache protocol
    (prepare_inputs_for_generation modeling:653-686, update_inputs_for_generation 688-693)."""

    def __init__(self, cfg, params, ehs_rows, max_length: int):
        import torch
        from . import model_ref

        self._torch, self._m = torch, model_ref
        self.cfg, self.p, self.ehs = cfg, params, ehs_rows
        self.state = model_ref.DecodeState(cfg, ehs_rows.shape[0], max_length)
        self.pos = 0  # position_ids start at arange(seq_len=1) = 0 and advance by +1 (modeling:676-678, 690-692)

    def step(self, tokens: np.ndarray) -> np.ndarray:
        t = self._torch
        ids = t.from_numpy(np.asarray(tokens, dtype=np.int64))[:, None]
        pos = t.full_like(ids, self.pos)
        with t.no_grad():
            logits = self._m.decode_step(self.cfg, self.p, self.state, ids, pos, self.ehs)
        self.pos += 1
        return logits[:, 0].numpy()

    def reorder(self, src_rows: np.ndarray) -> None:
        self.state.gather_rows(self._torch.from_numpy(np.asarray(src_rows, dtype=np.int64)))


class ScriptedStepper:
    """Fake decoder for known-answer tests: logits depend only on (step, row-history hash)."""

    def __init__(self, rows: int, table: Callable[[int, np.ndarray], np.ndarray]):
        self.rows, self.table, self.t = rows, table, 0
        self.hist = [[] for _ in range(rows)]

    def step(self, tokens: np.ndarray) -> np.ndarray:
        for r in range(self.rows):
            self.hist[r].append(int(tokens[r]))
        out = np.stack([self.table(self.t, np.array(self.hist[r])) for r in range(self.rows)]).astype(np.float32)
        self.t += 1
        return out

    def reorder(self, src_rows: np.ndarray) -> None:
        self.hist = [list(self.hist[int(s)]) for s in src_rows]
