"""Shared helpers for the model-level tests: the reduced config of tests/golden/make_golden.py and its oracle params."""
import numpy as np
import torch

SMALL = dict(v_hidden=128, v_ffn=256, v_layers=2, v_heads=2, image_size=48, patch_size=16,
             d_model=128, d_ffn=256, d_layers=2, d_heads=2, vocab_size=1003, max_position_embeddings=64)
SEED = 1234


def ref_config(gelu="erf", decoder_ln_eps=1e-5, **over):
    from oracle import model_ref as M

    kw = dict(SMALL)
    kw.update(over)
    return M.RefConfig(gelu=gelu, decoder_ln_eps=decoder_ln_eps, **kw)


def product_config(rc):
    from mic_amd import CLIPVisionMBartConfig

    mb = dict(vocab_size=rc.vocab_size, d_model=rc.d_model, decoder_layers=rc.d_layers, decoder_attention_heads=rc.d_heads,
              decoder_ffn_dim=rc.d_ffn, max_position_embeddings=rc.max_position_embeddings, gelu_variant=rc.gelu,
              decoder_ln_eps=rc.decoder_ln_eps, dropout=rc.dropout)
    cv = dict(hidden_size=rc.v_hidden, intermediate_size=rc.v_ffn, num_hidden_layers=rc.v_layers, num_attention_heads=rc.v_heads,
              image_size=rc.image_size, patch_size=rc.patch_size)
    return CLIPVisionMBartConfig(mbart_config=mb, clip_vision_config=cv)


def make_pair(dtype, dev, gelu="erf", decoder_ln_eps=1e-5, seed=SEED, **over):
    """(oracle cfg, oracle params, product model holding the same weights)."""
    from mic_amd import FlaxCLIPVisionMBartForConditionalGeneration
    from mic_amd.params import unflatten_tree
    from oracle import model_ref as M

    rc = ref_config(gelu, decoder_ln_eps, **over)
    p = M.init_params(rc, seed=seed, perturb_ln=True)
    model = FlaxCLIPVisionMBartForConditionalGeneration(product_config(rc), seed=0, dtype=dtype, device=dev)
    model.params = unflatten_tree({k: v.numpy() for k, v in p.items()})
    return rc, p, model


def batch(rc, B, T, seed=0, ragged=True):
    g = torch.Generator().manual_seed(seed)
    px = torch.randn(B, rc.image_size, rc.image_size, 3, generator=g).clamp(-1.8, 2.2)
    labels = torch.full((B, T), rc.pad_token_id, dtype=torch.int64)
    mask = torch.zeros((B, T), dtype=torch.int64)
    for b in range(B):
        n = int(torch.randint(2, T - 2, (1,), generator=g)) if ragged and b > 0 else T - 2
        labels[b, 0] = rc.vocab_size - 10 + (b % 4)
        labels[b, 1: 1 + n] = torch.randint(4, rc.vocab_size - 20, (n,), generator=g)
        labels[b, 1 + n] = rc.eos_token_id
        mask[b, : n + 2] = 1
    dec_in = torch.full_like(labels, rc.pad_token_id)
    dec_in[:, 1:] = labels[:, :-1]
    return px, labels, mask, dec_in
