"""How far the fp8 step's GRADIENT is from the bf16 step's on the same weights, batch and dropout masks — one number per gradient group
and fp8 mode, free of the chaos of two training trajectories.  The full-size model is trained for `steps` bf16 steps (default 150: the
late phase, confident predictions), then every mode runs three train steps with learning rate 0 on batch 0 (the third: every tensor
has its delayed scale) and the flat gradient is compared with the bf16 one: relative L2 error and cosine.
usage: python tools/fp8_grad_error.py [steps]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mic_amd  # noqa: F401,E402
from bench import synth_batch  # noqa: E402
from mic_amd import CLIPVisionMBartConfig, FlaxCLIPVisionMBartForConditionalGeneration, Trainer, create_learning_rate_fn, loss_rows, packed_rows  # noqa: E402


def device_batches(dev, n=4):
    dbs = []
    for b in [synth_batch(64, 64, 250054, 224, 1234 + i) for i in range(n)]:
        db = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
        idx, rl = loss_rows(b["attention_mask"], b["input_ids"])
        db["loss_rows"] = (torch.from_numpy(idx).to(dev), torch.from_numpy(rl).to(dev))
        pk = packed_rows(b["attention_mask"], b["decoder_input_ids"])
        db["packed_rows"] = tuple(torch.from_numpy(t).to(dev) for t in pk)
        dbs.append(db)
    return dbs


def trained_model(dev, steps, dbs):
    cfg = CLIPVisionMBartConfig(mbart_config={}, clip_vision_config={})
    model = FlaxCLIPVisionMBartForConditionalGeneration(cfg, seed=0, dtype=torch.bfloat16, device=dev)
    lr = create_learning_rate_fn(train_ds_size=64 * 200, train_batch_size=64, num_train_epochs=1, num_warmup_steps=20, learning_rate=1e-4)
    tr = Trainer(model, lr, seed=42)
    loss = None
    for i in range(steps):
        loss = tr.train_step(dbs[i % len(dbs)])["loss"]
    del tr
    return model, (float(loss) if loss is not None else float("nan"))


def groups_of(segs):
    return {"all": None, "shared (dE)": ["shared"], "final_logits_bias": ["flb"],
            "decoder weights": [n for n in segs if n.startswith("dec") and n.endswith(".w")],
            "ViT weights": [n for n in segs if n.startswith("vit") and n.endswith(".w")],
            "LayerNorm / biases": [n for n in segs if not n.endswith(".w") and n not in ("shared", "flb")]}


def grads(model, db, gemm_dtype, head):
    """flat gradient (float64, host) and loss of the third of three learning-rate-0 train steps on `db` in the given GEMM mode"""
    keep = os.environ.get("MIC_FP8_HEAD")
    os.environ["MIC_FP8_HEAD"] = head
    try:
        t = Trainer(model, lambda step: 0.0, seed=42, gemm_dtype=gemm_dtype)
        for _ in range(3):
            out = t.train_step(db)
        torch.cuda.synchronize()
        g = model.store.grad.detach().double().cpu().numpy().copy()
        l = float(out["loss"])
        model.engine.set_gemm_dtype(None)
        del t
    finally:
        if keep is None:
            os.environ.pop("MIC_FP8_HEAD", None)
        else:
            os.environ["MIC_FP8_HEAD"] = keep
    return g, l


def rel_err(g, g_ref, segs, names):
    def pick(x):
        return x if names is None else np.concatenate([x[segs[n].offset: segs[n].offset + segs[n].numel] for n in names])
    a, b = pick(g), pick(g_ref)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b)), float(np.dot(a, b) / np.linalg.norm(a) / np.linalg.norm(b))


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    dev = torch.device("cuda:0")
    dbs = device_batches(dev)
    model, loss = trained_model(dev, steps, dbs)
    print(f"{steps} bf16 steps: loss {loss:.4f}", flush=True)
    segs = model.store.segs
    groups = groups_of(segs)
    g16, l16 = grads(model, dbs[0], None, "0")
    rows = [("bf16 again (atomics)",) + grads(model, dbs[0], None, "0")]
    for head in ("0", "bwd", "all"):
        rows.append((f"fp8 head={head}",) + grads(model, dbs[0], "fp8", head))
    print(f"{'mode':<22} {'loss':>8}  " + "  ".join(f"{k:>24}" for k in groups))
    print(f"{'bf16':<22} {l16:>8.4f}")
    for name, g, l in rows:
        cells = ["%9.4f cos %.5f" % rel_err(g, g16, segs, names) for names in groups.values()]
        print(f"{name:<22} {l:>8.4f}  " + "  ".join(f"{c:>24}" for c in cells), flush=True)


if __name__ == "__main__":
    main()
