"""Model-level parity on the GPU: the HIP path (through the reference-shaped Python API and the C ABI) against the CPU
oracle on the same seeded inputs, and against the committed golden vectors of the PyTorch twin.

Tolerances (stated, per north_star "logits within 1e-3 (bf16), greedy ids bit-exact"):
  float32 mode (the reference's default dtype): logits / loss / grads within 2e-4 of the output scale; token ids exact.
  bfloat16 mode: activations, weights and logits are *stored* in bf16 (8 mantissa bits: 1 ulp = 3.9e-3 relative), so an
  elementwise 1e-3 is below the format's resolution; asserted: max |dlogit| <= 3e-2 * max|logit| and mean |dlogit| <=
  4e-3 * max|logit| against the fp32 oracle, loss within 2e-2.
"""
import os

import numpy as np
import pytest
import torch

from util_small import SEED, batch, make_pair

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "twin_small.npz")


def scale_err(got, ref):
    ref = ref.float()
    d = (got.float().cpu() - ref).abs()
    s = ref.abs().max().clamp_min(1e-9)
    return (d.max() / s).item(), (d.mean() / s).item()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_logits_match_golden_twin(dev, dtype):
    g = np.load(GOLD)
    rc, p, model = make_pair(dtype, dev)
    assert int(g["seed"]) == SEED
    out = model(g["pixels"], g["ids"], g["mask"])
    logits = out[0]
    assert tuple(logits.shape) == g["logits"].shape
    valid = torch.from_numpy(g["mask"]).bool()
    mx, mean = scale_err(logits[valid.to(dev)], torch.from_numpy(g["logits"])[valid])
    if dtype == torch.float32:
        assert mx < 2e-4, (mx, mean)
    else:
        assert mx < 3e-2 and mean < 4e-3, (mx, mean)
    enc = model.encode(g["pixels"], _int32_cast=False)
    mx, mean = scale_err(enc.last_hidden_state, torch.from_numpy(g["ehs"]))
    assert mx < (2e-4 if dtype == torch.float32 else 3e-2), (mx, mean)
    mx, _ = scale_err(enc.pooler_output, torch.from_numpy(g["pooled"]))
    assert mx < (2e-4 if dtype == torch.float32 else 3e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("gelu,eps", [("tanh", 1e-6), ("erf", 1e-5)])
def test_forward_matches_oracle_switches(dev, dtype, gelu, eps):
    from oracle import model_ref as M

    rc, p, model = make_pair(dtype, dev, gelu=gelu, decoder_ln_eps=eps)
    px, labels, mask, dec_in = batch(rc, 4, 12, seed=3)
    with torch.no_grad():
        ref = M.forward_logits(rc, p, px, dec_in, mask)
    got = model(px.numpy(), dec_in.numpy(), mask.numpy())[0]
    mx, mean = scale_err(got[mask.bool().to(dev)], ref[mask.bool()])
    assert (mx < 2e-4) if dtype == torch.float32 else (mx < 3e-2 and mean < 4e-3), (mx, mean)


def test_encode_int32_cast_and_shapes(dev):
    """encode() truncates pixel values toward zero (modeling:330); __call__ does not (modeling:501)."""
    from oracle import model_ref as M

    rc, p, model = make_pair(torch.float32, dev)
    px, *_ = batch(rc, 2, 12, seed=5)
    with torch.no_grad():
        ref, pooled = M.encode(rc, p, px, int32_cast=True)
    enc = model.encode(px.numpy())
    assert tuple(enc.last_hidden_state.shape) == (2, rc.v_seq, rc.d_model) and tuple(enc[0].shape) == (2, rc.v_seq, rc.d_model)
    assert scale_err(enc.last_hidden_state, ref)[0] < 2e-4
    assert scale_err(enc.pooler_output, pooled)[0] < 2e-4
    with pytest.raises(ValueError):
        model.encode(np.zeros((2, 3, rc.image_size, rc.image_size), dtype=np.float32))  # NCHW is rejected


def _oracle_masks(model, rc, seed, B, T):
    """Materialise the keep-masks the fused dropout epilogues use, in the oracle's dict form."""
    from mic_amd import ops
    from mic_amd.engine import _mix

    n = B * T * rc.d_model
    m = {"embed": ops.dropout_mask(n, rc.dropout, _mix(seed, 1), model.device).cpu().reshape(B, T, rc.d_model)}
    for l in range(rc.d_layers):
        for j, name in enumerate(("self", "cross", "ffn")):
            m[f"{l}/{name}"] = ops.dropout_mask(n, rc.dropout, _mix(seed, 10 + 3 * l + j), model.device).cpu().reshape(B, T, rc.d_model)
    return m


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("train,ls,compact", [(False, 0.0, False), (True, 0.0, False), (True, 0.1, False), (True, 0.1, True), (False, 0.0, True)])
def test_loss_and_grads_match_oracle(dev, dtype, train, ls, compact):
    from mic_amd.params import flatten_tree, unflatten_tree
    from oracle import train_ref

    rc, p, model = make_pair(dtype, dev, gelu="tanh", decoder_ln_eps=1e-6)
    B, T = 4, 12
    px, labels, mask, dec_in = batch(rc, B, T, seed=7)
    seed = 4242 if train else None
    masks = _oracle_masks(model, rc, seed, B, T) if train else None
    ref_loss, ref_g = train_ref.loss_and_grads(rc, p, px, labels, mask, dec_in, masks, ls)
    d = lambda x, t: model._dev(x, t)
    pos = torch.arange(T, dtype=torch.int32, device=dev)[None].expand(B, T).contiguous()
    kw = {}
    if compact:  # LM head + CE on the masked-in label positions only: must give the same loss and gradients
        from mic_amd import loss_rows

        idx, rl = loss_rows(mask.numpy(), labels.numpy())
        assert 0 < len(idx) < B * T
        kw = dict(rows=(d(idx, torch.int32), len(idx)), row_labels=d(rl, torch.int32))
    loss = model.engine.loss_and_grads(d(px, torch.float32), d(dec_in, torch.int32).reshape(-1), pos.reshape(-1), d(mask, torch.int32),
                                       d(labels, torch.int32).reshape(-1), B, T, label_smoothing=ls, seed=seed, **kw)
    torch.cuda.synchronize()
    tol_loss = 2e-5 if dtype == torch.float32 else 2e-2
    assert abs(loss.item() - ref_loss.item()) < tol_loss * max(1.0, abs(ref_loss.item())), (loss.item(), ref_loss.item())
    got = model.store.export_flat("grad")
    worst = {}
    for k, rg in ref_g.items():
        gg = torch.from_numpy(got[k]).reshape(rg.shape)
        s = rg.abs().max().item()
        if s < 1e-6:
            # analytically-zero gradients: post_layernorm (output unused, modeling:90) and every attention k_proj bias
            # (softmax is invariant to a per-query constant) — only fp noise on both sides
            assert gg.abs().max().item() < (1e-6 if dtype == torch.float32 else 2e-3), (k, gg.abs().max().item())
            continue
        worst[k] = ((gg - rg).abs().max() / s).item()
    bad = {k: v for k, v in worst.items() if v > (5e-4 if dtype == torch.float32 else 8e-2)}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:8]
    if dtype == torch.bfloat16:  # aggregate: gradient direction agrees
        a = torch.cat([torch.from_numpy(got[k]).reshape(-1) for k in ref_g])
        b = torch.cat([v.reshape(-1) for v in ref_g.values()])
        assert torch.nn.functional.cosine_similarity(a, b, dim=0).item() > 0.995


def test_train_steps_match_oracle_adamw(dev):
    """3 optimizer steps (no dropout, wd > 0, warmup schedule) on the HIP path vs oracle autograd + AdamW restatement."""
    from mic_amd import Trainer, create_learning_rate_fn
    from oracle import train_ref

    rc, p, model = make_pair(torch.float32, dev, gelu="tanh", decoder_ln_eps=1e-6, dropout=0.0)
    B, T = 4, 12
    lr_fn = create_learning_rate_fn(train_ds_size=40, train_batch_size=4, num_train_epochs=1, num_warmup_steps=2, learning_rate=1e-3)
    tr = Trainer(model, lr_fn, weight_decay=0.01, label_smoothing_factor=0.0, seed=42)
    params = {k: v.clone() for k, v in p.items()}
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v = {k: torch.zeros_like(v) for k, v in p.items()}
    for step in range(3):
        px, labels, mask, dec_in = batch(rc, B, T, seed=100 + step)
        out = tr.train_step({"pixel_values": px.numpy(), "input_ids": labels.numpy(), "attention_mask": mask.numpy(),
                             "decoder_input_ids": dec_in.numpy()})
        ref_loss, g = train_ref.loss_and_grads(rc, params, px, labels, mask, dec_in)
        lr = train_ref.linear_warmup_decay(step, 1e-3, 2, 10)
        assert abs(float(out["learning_rate"]) - lr) < 1e-9
        assert abs(float(out["loss"]) - ref_loss.item()) < 5e-5 * max(1.0, ref_loss.item())
        for k in params:
            params[k], m[k], v[k] = train_ref.adamw_update(params[k], g[k], m[k], v[k], step, lr, wd=0.01)
    got = model.params
    from mic_amd.params import flatten_tree

    gf = flatten_tree(got)
    for k, rv in params.items():
        d = (torch.from_numpy(gf[k]) - rv).abs().max().item()
        # Adam's first steps move every weight by ~lr regardless of gradient scale; compare against that step size
        assert d < 2e-5, (k, d)
