"""ORACLE — TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's training-step pieces
(`main.py`, cited per function).  Gradients come from torch autograd over `oracle.model_ref`.

optax 0.0.9 (`requirements.txt:21`) is un-vendored; `adamw` is restated from its published chain
`scale_by_adam(b1,b2,eps) -> add_decayed_weights(wd) -> scale_by_schedule(-lr)` [UNVERIFIED-3P].
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import model_ref


def shift_tokens_right(input_ids: np.ndarray, pad_token_id: int) -> np.ndarray:
    """main.py:362-369: dec_in[:,0] = pad, dec_in[:,1:] = labels[:,:-1] (int64)."""
    out = np.zeros(input_ids.shape, dtype=np.int64)
    out[:, 1:] = input_ids[:, :-1]
    out[:, 0] = pad_token_id
    return out


def loss_fn(logits: torch.Tensor, labels: torch.Tensor, padding_mask: torch.Tensor,
            label_smoothing_factor: float = 0.0) -> torch.Tensor:
    """main.py:658-680 — label-smoothed CE (Flax WMT recipe), masked mean over the *local* batch."""
    vocab = logits.shape[-1]
    confidence = 1.0 - label_smoothing_factor
    low = (1.0 - confidence) / (vocab - 1)
    if confidence < 1.0:
        norm_const = -(confidence * math.log(confidence) + (vocab - 1) * low * math.log(low + 1e-20))
    else:  # jnp: 1*log(1) + (V-1)*0*log(1e-20) = 0
        norm_const = 0.0
    logp = torch.log_softmax(logits.to(torch.float32), dim=-1)
    # optax.softmax_cross_entropy(logits, soft) = -sum(soft * log_softmax(logits))
    nll_label = -logp.gather(-1, labels.to(torch.int64)[..., None])[..., 0]
    loss = confidence * nll_label + low * (-logp.sum(-1) - nll_label) if low > 0 else nll_label
    loss = loss - norm_const
    m = padding_mask.to(torch.float32)
    return (loss * m).sum() / m.sum()


def linear_warmup_decay(step: int, lr: float, warmup_steps: int, total_steps: int) -> float:
    """main.py:281-292: optax.linear_schedule(0->lr, warmup) joined at `warmup_steps` with linear_schedule(lr->0,
    total-warmup).  optax.linear_schedule clips the fraction to [0,1]."""
    if step < warmup_steps:
        frac = min(max(step / warmup_steps, 0.0), 1.0) if warmup_steps > 0 else 1.0
        return lr * frac
    n = total_steps - warmup_steps
    if n <= 0:  # optax.linear_schedule(transition_steps <= 0) returns the constant init_value
        return lr
    frac = min(max((step - warmup_steps) / n, 0.0), 1.0)
    return lr + (0.0 - lr) * frac


def adamw_update(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, count: int, lr_t: float,
                 b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8, wd: float = 0.0):
    """One optax.adamw update (main.py:629-635).  `count` = number of updates already applied (state.step);
    bias correction uses count+1, the schedule is evaluated at `count` (pre-increment) — SURVEY B10.
    Weight decay hits every leaf (no mask)."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    t = count + 1
    mhat = m / (1 - b1 ** t)
    vhat = v / (1 - b2 ** t)
    upd = mhat / (torch.sqrt(vhat) + eps) + wd * p
    return p - lr_t * upd, m, v


def forward_loss(cfg: model_ref.RefConfig, params: Dict[str, torch.Tensor], pixel_values: torch.Tensor,
                 labels: torch.Tensor, attention_mask: torch.Tensor, decoder_input_ids: torch.Tensor,
                 masks: Optional[Dict[str, torch.Tensor]] = None, label_smoothing_factor: float = 0.0,
                 logits_dtype: Optional[torch.dtype] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """compute_loss of train_step (main.py:688-694): labels = batch['input_ids']; the labels' attention mask is both
    the decoder key-padding mask (692) and the loss mask (693)."""
    logits = model_ref.forward_logits(cfg, params, pixel_values, decoder_input_ids, attention_mask, None, masks)
    if logits_dtype is not None:  # B11: logits come out in the compute dtype before CE
        logits = logits.to(logits_dtype).to(torch.float32)
    return loss_fn(logits, labels, attention_mask, label_smoothing_factor), logits


def loss_and_grads(cfg, params, pixel_values, labels, attention_mask, decoder_input_ids, masks=None,
                   label_smoothing_factor: float = 0.0):
    """jax.value_and_grad(compute_loss)(params) (main.py:696-697)."""
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    loss, _ = forward_loss(cfg, leaves, pixel_values, labels, attention_mask, decoder_input_ids, masks,
                           label_smoothing_factor)
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
    return loss.detach(), grads
