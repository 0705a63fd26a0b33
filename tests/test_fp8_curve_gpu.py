"""BASELINE configs[4] against configs[1] over a TRAINING RUN: 200 optimizer steps of the full-size model (ViT-B/32 + mBART-large-50,
batch 64, seq 64, dropout on, AdamW with warm-up) from the same initial weights, the same batches and the same dropout streams,
once with bf16 GEMMs and once with the QKV / FFN projections and the tied LM head in fp8 (e4m3 / e5m2, delayed scaling, fused
emission).  The reference has no fp8 mode (main.py:96-101 offers fp32 / fp16 / bf16): the bf16 run — itself pinned on the fp32
oracle — is the yardstick.
Two instruments.  (1) The loss CURVE: both runs learn (the loss falls to less than a tenth on the four repeated batches); over the
first 100 steps the fp8 loss stays within 6 % of the bf16 loss, over all 200 within 12 %, the means of the last 20 steps within 10 %.
Two training trajectories are chaotic in the memorisation phase (fp32 atomics order alone moves them): thirteen fp8 runs on MI355X
(`tools/fp8_head_curve.py`, every head mode) ended +1.0 ... +4.4 % above bf16 in the last-20 mean with a largest per-step deviation
of 1.7 ... 5.3 %, the head modes indistinguishable inside that spread (profiles/NOTES_r6.md section 11) — so the curve bounds are
wide, and (2) carries the precision: the GRADIENT of one step on identical weights, batch and dropout masks (`tools/fp8_grad_error.py`),
deterministic up to atomics: after 100 bf16 steps the fp8 gradient's relative L2 distance from the bf16 gradient is bounded, and the
fp8 head adds nothing measurable to it — its backward (one e5m2 copy of dlogits under a closed-form scale, the label entries exact)
within 0.01 of the run whose head is bf16, its forward within 0.03."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(dev, gemm_dtype, steps, batches):
    from mic_amd import CLIPVisionMBartConfig, FlaxCLIPVisionMBartForConditionalGeneration, Trainer, create_learning_rate_fn, loss_rows, packed_rows

    cfg = CLIPVisionMBartConfig(mbart_config={}, clip_vision_config={})
    model = FlaxCLIPVisionMBartForConditionalGeneration(cfg, seed=0, dtype=torch.bfloat16, device=dev)
    lr = create_learning_rate_fn(train_ds_size=64 * steps, train_batch_size=64, num_train_epochs=1, num_warmup_steps=20, learning_rate=1e-4)
    tr = Trainer(model, lr, seed=42, gemm_dtype=gemm_dtype)
    dbs = []
    for b in batches:
        db = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
        idx, rl = loss_rows(b["attention_mask"], b["input_ids"])
        db["loss_rows"] = (torch.from_numpy(idx).to(dev), torch.from_numpy(rl).to(dev))
        pk = packed_rows(b["attention_mask"], b["decoder_input_ids"])
        db["packed_rows"] = tuple(torch.from_numpy(t).to(dev) for t in pk)
        dbs.append(db)
    losses = [tr.train_step(dbs[i % len(dbs)])["loss"] for i in range(steps)]
    out = torch.stack(losses).float().cpu().numpy()
    fused = bool(getattr(model.engine, "fp8_fused", False))
    del tr, model
    torch.cuda.empty_cache()
    return out, fused


def test_fullsize_fp8_loss_curve_follows_bf16_over_200_steps(dev):
    sys.path.insert(0, ROOT)
    from bench import synth_batch

    steps = 200
    batches = [synth_batch(64, 64, 250054, 224, 1234 + i) for i in range(4)]
    l16, _ = _run(dev, None, steps, batches)
    l8, fused = _run(dev, "fp8", steps, batches)
    assert fused and np.isfinite(l8).all() and np.isfinite(l16).all()
    rel = np.abs(l8 - l16) / np.abs(l16)
    tail = abs(l8[-20:].mean() - l16[-20:].mean()) / l16[-20:].mean()
    print(f"[fp8 curve] bf16 {l16[0]:.3f} -> {l16[-1]:.3f}, fp8 {l8[0]:.3f} -> {l8[-1]:.3f}; max |fp8 - bf16| / bf16 = {rel.max():.4f} at step {int(rel.argmax())} "
          f"({rel[:100].max():.4f} over the first 100 steps), "
          f"mean {rel.mean():.4f}; last-20 means differ by {tail:.4f}; every 20th step bf16 {[round(float(x), 3) for x in l16[::20]]} fp8 {[round(float(x), 3) for x in l8[::20]]}")
    assert l16[-1] < 0.1 * l16[0] and l8[-1] < 0.1 * l8[0], (l16[0], l16[-1], l8[0], l8[-1])
    assert rel[:100].max() < 0.06, (rel[:100].max(), int(rel[:100].argmax()))
    assert rel.max() < 0.12, (rel.max(), int(rel.argmax()))
    assert tail < 0.10, tail


def test_fullsize_fp8_gradient_error_against_bf16_and_what_the_fp8_head_adds(dev):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fp8_grad_error as G

    dbs = G.device_batches(dev)
    model, loss = G.trained_model(dev, 100, dbs)
    segs = model.store.segs
    g16, l16 = G.grads(model, dbs[0], None, "0")
    again, _ = G.grads(model, dbs[0], None, "0")
    assert G.rel_err(again, g16, segs, None)[0] < 1e-3  # the instrument: two bf16 passes differ by the order of fp32 atomics only
    err, out = {}, []
    for head in ("0", "bwd", "all"):
        g8, l8 = G.grads(model, dbs[0], "fp8", head)
        err[head] = {k: G.rel_err(g8, g16, segs, names) for k, names in G.groups_of(segs).items()}
        out.append(f"head={head}: loss {l8:.4f} (bf16 {l16:.4f}), all {err[head]['all'][0]:.4f} / cos {err[head]['all'][1]:.4f}, "
                   f"shared {err[head]['shared (dE)'][0]:.4f}, decoder {err[head]['decoder weights'][0]:.4f}, ViT {err[head]['ViT weights'][0]:.4f}")
        assert abs(l8 - l16) < 0.08 * l16, (head, l8, l16)  # (bf16-trained weights seen through fp8 GEMMs: +2 ... +4 % measured)
    print("[fp8 gradient error vs bf16 after 100 bf16 steps (loss %.3f)] " % loss + "; ".join(out))
    assert err["0"]["all"][0] < 0.35 and err["0"]["all"][1] > 0.95, err["0"]["all"]
    for k in ("all", "shared (dE)", "decoder weights", "ViT weights"):
        assert abs(err["bwd"][k][0] - err["0"][k][0]) < 0.01, (k, err["bwd"][k], err["0"][k])
        assert err["all"][k][0] < err["0"][k][0] + 0.03, (k, err["all"][k], err["0"][k])
