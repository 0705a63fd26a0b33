"""Generates tests/golden/twin_small_decode_grads.npz and twin_small_default_decode_grads.npz — run ONLY in the build
container (needs transformers 5.x).  One file per switch setting of make_golden.VARIANTS ("" = the twin as shipped: erf GELU /
eps 1e-5; "_default" = the product defaults: tanh GELU / eps 1e-6).

Second fixture of the PyTorch twins (same reduced config, seed and weights as make_golden.py):
  * KV-cached decoding: the twin's `MBartDecoder(use_cache=True)` fed one token per step with its own `past_key_values`
    (it derives positions from the cache length) -> per-step logits.  Pins the oracle's static-cache `decode_step`
    (modeling:249-282, 519-651) to something other than itself.
  * Gradients: twin autograd of the masked cross-entropy (main.py:658-680 with label smoothing 0 and 0.1) w.r.t. a set of
    leaves from every part of the graph (tied embedding, both towers, projection).  Pins the oracle's backward (torch
    autograd over oracle.model_ref, which the HIP path's hand-derived backward is tested against).
Only inputs and expected outputs are stored — no upstream source.

    python tests/golden/make_golden_decode_grads.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from make_golden import SEED, VARIANTS, load_into_twin, variant_cfg  # noqa: E402

from oracle import model_ref as M  # noqa: E402
from oracle import train_ref  # noqa: E402

# (flax leaf, twin module attribute path, transform back to the Flax layout)
GRAD_LEAVES = [
    ("model/shared/embedding", ("dec", "embed_tokens.weight"), "id"),
    ("model/decoder/embed_positions/embedding", ("dec", "embed_positions.weight"), "id"),
    ("model/decoder/layers/0/fc1/kernel", ("dec", "layers.0.fc1.weight"), "T"),
    ("model/decoder/layers/1/self_attn/q_proj/kernel", ("dec", "layers.1.self_attn.q_proj.weight"), "T"),
    ("model/decoder/layers/1/self_attn/k_proj/bias", ("dec", "layers.1.self_attn.k_proj.bias"), "id"),
    ("model/decoder/layers/0/encoder_attn/v_proj/kernel", ("dec", "layers.0.encoder_attn.v_proj.weight"), "T"),
    ("model/decoder/layers/1/final_layer_norm/scale", ("dec", "layers.1.final_layer_norm.weight"), "id"),
    ("model/decoder/layernorm_embedding/bias", ("dec", "layernorm_embedding.bias"), "id"),
    ("model/encoder/vision_model/encoder/layers/0/mlp/fc2/kernel", ("vit", "encoder.layers.0.mlp.fc2.weight"), "T"),
    ("model/encoder/vision_model/encoder/layers/1/self_attn/out_proj/bias", ("vit", "encoder.layers.1.self_attn.out_proj.bias"), "id"),
    ("model/encoder/vision_model/embeddings/patch_embedding/kernel", ("vit", "embeddings.patch_embedding.weight"), "conv"),
    ("model/encoder/vision_model/embeddings/class_embedding", ("vit", "embeddings.class_embedding"), "id"),
    ("model/encoder/vision_model/pre_layrnorm/scale", ("vit", "pre_layrnorm.weight"), "id"),
]


def sub(a):
    """large leaves are stored as a strided sample of their flattened entries (<= 6 000 values): the fixture stays small"""
    a = np.asarray(a).reshape(-1)
    return a[:: max(1, -(-a.size // 6000))].copy()


def main(suffix=""):
    cfg = variant_cfg(suffix)
    p = M.init_params(cfg, seed=SEED, perturb_ln=True)
    vit, dec = load_into_twin(cfg, p)
    g = torch.Generator().manual_seed(4321)
    B, T, STEPS = 3, 12, 6
    pixels = torch.randn(B, cfg.image_size, cfg.image_size, 3, generator=g).clamp(-1.8, 2.2)
    out = {}
    # ---- cached decode
    step_ids = torch.randint(4, cfg.vocab_size, (B, STEPS), generator=g)
    with torch.no_grad():
        enc_last = vit(pixel_values=pixels.permute(0, 3, 1, 2)).last_hidden_state
        ehs = enc_last @ p["model/visual_projection/kernel"] + p["model/visual_projection/bias"]
        past = None
        logits_steps = []
        for t in range(STEPS):
            o = dec(input_ids=step_ids[:, t:t + 1], encoder_hidden_states=ehs, past_key_values=past, use_cache=True)
            past = o.past_key_values
            logits_steps.append(o.last_hidden_state[:, 0] @ p["model/shared/embedding"].T + p["final_logits_bias"])
    out.update(dec_pixels=pixels.numpy(), dec_step_ids=step_ids.numpy(), dec_ehs=ehs.numpy(), dec_step_logits=torch.stack(logits_steps, 1).numpy())
    # ---- gradients (tied head through the twin's own embedding parameter)
    labels = torch.full((B, T), cfg.pad_token_id, dtype=torch.int64)
    mask = torch.zeros((B, T), dtype=torch.int64)
    for b, n in enumerate((10, 6, 2)):
        labels[b, 0] = cfg.vocab_size - 10 + b
        labels[b, 1:1 + n] = torch.randint(4, cfg.vocab_size - 20, (n,), generator=g)
        labels[b, 1 + n] = cfg.eos_token_id
        mask[b, :n + 2] = 1
    dec_in = torch.full_like(labels, cfg.pad_token_id)
    dec_in[:, 1:] = labels[:, :-1]
    vp_k = p["model/visual_projection/kernel"].clone().requires_grad_(True)
    vp_b = p["model/visual_projection/bias"].clone().requires_grad_(True)
    flb = p["final_logits_bias"].clone().requires_grad_(True)
    for ls in (0.0, 0.1):
        for m in (vit, dec):
            m.zero_grad(set_to_none=True)
        for t in (vp_k, vp_b, flb):
            t.grad = None
        enc_last = vit(pixel_values=pixels.permute(0, 3, 1, 2)).last_hidden_state
        ehs = enc_last @ vp_k + vp_b
        hid = dec(input_ids=dec_in, attention_mask=mask, encoder_hidden_states=ehs,
                  encoder_attention_mask=torch.ones(B, cfg.v_seq, dtype=torch.int64), use_cache=False).last_hidden_state
        logits = hid @ dec.embed_tokens.weight.T + flb
        loss = train_ref.loss_fn(logits, labels, mask, ls)
        loss.backward()
        tag = f"ls{int(ls * 10)}"
        out[f"loss_{tag}"] = np.float64(loss.item())
        mods = {"vit": dict(vit.named_parameters()), "dec": dict(dec.named_parameters())}
        for leaf, (mod, name), tr in GRAD_LEAVES:
            gr = mods[mod][name].grad
            if leaf == "model/shared/embedding":
                # torch's nn.Embedding(padding_idx=pad) drops the INPUT-embedding gradient of the pad row; flax nn.Embed (the
                # reference) has no padding_idx.  The pad row is therefore excluded from the comparison (zeroed on both sides).
                gr = gr.clone()
                gr[cfg.pad_token_id] = 0
            if tr == "T":
                gr = gr.T
            elif tr == "conv":
                gr = gr.permute(2, 3, 1, 0)
            out[f"grad_{tag}|{leaf}"] = sub(gr.contiguous().numpy())
        out[f"grad_{tag}|model/visual_projection/kernel"] = sub(vp_k.grad.numpy())
        out[f"grad_{tag}|model/visual_projection/bias"] = sub(vp_b.grad.numpy())
        out[f"grad_{tag}|final_logits_bias"] = sub(flb.grad.numpy())
    out.update(g_pixels=pixels.numpy(), g_labels=labels.numpy(), g_mask=mask.numpy(), g_dec_in=dec_in.numpy(), seed=np.int64(SEED),
               gelu=np.array(cfg.gelu), decoder_ln_eps=np.float64(cfg.decoder_ln_eps))
    path = os.path.join(HERE, f"twin_small{suffix}_decode_grads.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
    # immediate self-check of the oracle
    with torch.no_grad():
        st = M.DecodeState(cfg, B, STEPS + 2)
        e = torch.from_numpy(out["dec_ehs"])
        worst = 0.0
        for t in range(STEPS):
            lg = M.decode_step(cfg, p, st, step_ids[:, t:t + 1], torch.full((B, 1), t), e)
            worst = max(worst, (lg[:, 0] - torch.from_numpy(out["dec_step_logits"][:, t])).abs().max().item())
    print("cached decode max|d|", worst)
    for ls in (0.0, 0.1):
        l, gr = train_ref.loss_and_grads(cfg, p, pixels, labels, mask, dec_in, label_smoothing_factor=ls)
        tag = f"ls{int(ls * 10)}"
        w = {}
        for k, v in out.items():
            if k.startswith(f"grad_{tag}|"):
                leaf = k.split("|")[1]
                og = gr[leaf].numpy().copy()
                if leaf == "model/shared/embedding":
                    og[cfg.pad_token_id] = 0
                sc = max(np.abs(og).max(), 1e-30)
                w[leaf] = (np.abs(sub(og) - v).max() / sc, sc)
        for leaf, (e, sc) in sorted(w.items(), key=lambda kv: -kv[1][0])[:4]:
            print("   ", leaf, "rel err", e, "scale", sc)
        w = max(e for e, sc in w.values() if sc > 1e-6)
        print(tag, "loss d", abs(l.item() - out[f"loss_{tag}"]), "worst grad rel", w)


if __name__ == "__main__":
    for sfx in VARIANTS:
        main(sfx)
