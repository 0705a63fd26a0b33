"""Evaluation harness of the reference: the eval loop of `main.py:791-854` / `evaluation.py:142-196` —
per-language `eval_step` loss, beam generation with `decoder_start_token_id = <language code>` (main.py:820) and
BLEU-1..4 over word-tokenised text (`compute_metrics`, main.py:577-603).

The reference scores with `datasets.load_metric("bleu")` (the tensorflow/nmt `compute_bleu`: clipped n-gram counts pooled
over the corpus, geometric mean of the 1..max_order precisions, brevity penalty exp(1 - ref_len/hyp_len) with the SHORTEST
reference length per segment, no smoothing) [UNVERIFIED-3P: restated from the published algorithm; `datasets` cannot fetch
the metric script here] and tokenises with `nltk.word_tokenize` (absent from this image): the tokeniser is a parameter,
default = a Unicode word/punctuation splitter.  All of this is host code; the GPU work is `Trainer.eval_step` and
`model.generate`."""
from __future__ import annotations

import collections
import math
import re
from typing import Callable, Dict, Iterable, List, Optional, Sequence

import numpy as np

_WORD = re.compile(r"\w+|[^\w\s]", re.UNICODE)


def simple_word_tokenize(text: str, language: Optional[str] = None) -> List[str]:
    """Stand-in for `nltk.word_tokenize(text, language=...)` (main.py:581-584): words and single punctuation marks."""
    return _WORD.findall(text)


def _ngrams(tokens: Sequence[str], max_order: int) -> collections.Counter:
    c = collections.Counter()
    for n in range(1, max_order + 1):
        for i in range(len(tokens) - n + 1):
            c[tuple(tokens[i: i + n])] += 1
    return c


def compute_bleu(predictions: Sequence[Sequence[str]], references: Sequence[Sequence[Sequence[str]]], max_order: int = 4,
                 smooth: bool = False) -> Dict[str, object]:
    """`metric.compute(predictions=..., references=..., max_order=i)` (main.py:596-598): corpus BLEU.
    predictions[i] = token list; references[i] = list of reference token lists."""
    matches, possible = [0] * max_order, [0] * max_order
    ref_len = hyp_len = 0
    for hyp, refs in zip(predictions, references):
        ref_len += min(len(r) for r in refs)
        hyp_len += len(hyp)
        merged = collections.Counter()
        for r in refs:
            merged |= _ngrams(r, max_order)
        overlap = _ngrams(hyp, max_order) & merged
        for ng, cnt in overlap.items():
            matches[len(ng) - 1] += cnt
        for n in range(1, max_order + 1):
            if len(hyp) - n + 1 > 0:
                possible[n - 1] += len(hyp) - n + 1
    precisions = []
    for n in range(max_order):
        if smooth:
            precisions.append((matches[n] + 1.0) / (possible[n] + 1.0))
        else:
            precisions.append(matches[n] / possible[n] if possible[n] > 0 else 0.0)
    geo = math.exp(sum(math.log(p) for p in precisions) / max_order) if min(precisions) > 0 else 0.0
    ratio = hyp_len / ref_len if ref_len > 0 else 0.0
    bp = 1.0 if ratio > 1.0 else (math.exp(1.0 - 1.0 / ratio) if ratio > 0 else 0.0)
    return {"bleu": geo * bp, "precisions": precisions, "brevity_penalty": bp, "length_ratio": ratio,
            "translation_length": hyp_len, "reference_length": ref_len}


def compute_metrics(pred_ids, label_ids, batch_decode: Callable[[Iterable[Sequence[int]]], List[str]],
                    word_tokenize: Callable[[str], List[str]] = simple_word_tokenize) -> Dict[str, float]:
    """main.py:586-603: decode ids (special tokens skipped by `batch_decode`), strip, tokenise, BLEU-1..4."""
    preds = [word_tokenize(p.strip()) for p in batch_decode(pred_ids)]
    labels = [[word_tokenize(l.strip())] for l in batch_decode(label_ids)]
    return {f"BLEU-{i}": compute_bleu(preds, labels, max_order=i)["bleu"] for i in range(1, 5)}


def evaluate(trainer, eval_loaders: Dict[str, Iterable[dict]], lang_code_to_id: Dict[str, int],
             batch_decode: Callable[[Iterable[Sequence[int]]], List[str]], *, predict_with_generate: bool = True,
             max_length: int = 64, num_beams: int = 4, word_tokenize: Callable[[str], List[str]] = simple_word_tokenize) -> Dict[str, object]:
    """The evaluation block of the training loop (main.py:791-846).  `eval_loaders` maps a language code ("en_XX", ...) to an
    iterable of batches (the dicts `collate_fn` builds: pixel_values, input_ids, attention_mask, decoder_input_ids).
    Returns {"loss": mean eval loss over all batches, "<lang>": {"BLEU-1"..}} like `eval_metrics` (main.py:838-841)."""
    losses: List[float] = []
    out: Dict[str, object] = {}
    model = trainer.model
    for lang, loader in eval_loaders.items():
        preds, labels = [], []
        for batch in loader:
            losses.append(float(trainer.eval_step(batch)["loss"]))  # main.py:812
            if predict_with_generate:
                gen = model.generate(batch["pixel_values"], max_length=max_length, num_beams=num_beams,
                                     decoder_start_token_id=lang_code_to_id[lang])  # main.py:723-729, 820
                preds.extend(np.asarray(gen.sequences.cpu()).reshape(-1, max_length))
                lb = batch["input_ids"]
                labels.extend(np.asarray(lb.cpu() if hasattr(lb, "cpu") else lb).reshape(-1, np.asarray(lb).shape[-1]))
        if predict_with_generate:
            out[lang] = compute_metrics(preds, labels, batch_decode, word_tokenize)
    out["loss"] = float(np.mean(losses)) if losses else float("nan")
    return out
