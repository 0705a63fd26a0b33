"""200 full-size training steps (tests/test_fp8_curve_gpu.py's run) in bf16 and in fp8 with the tied LM head in each of its modes
(MIC_FP8_HEAD = 0 | bwd | all): per-mode deviation of the fp8 loss curve from the bf16 one.  usage: python tools/fp8_head_curve.py [steps]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mic_amd  # noqa: F401,E402
from bench import synth_batch  # noqa: E402


# (tests/test_fp8_curve_gpu.py's run) + after the last step: the trained weights' loss on the four batches with EVERY GEMM in bf16 —
# what an evaluation of the checkpoint would see
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
    with model.engine.storage_dtype_gemms():
        ev = [float(tr.eval_step(db)["loss"]) for db in dbs]
    del tr, model
    torch.cuda.empty_cache()
    return out, ev

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
batches = [synth_batch(64, 64, 250054, 224, 1234 + i) for i in range(4)]
l16, ev16 = _run(dev, None, steps, batches)
print(f"bf16      {l16[0]:.3f} -> {l16[-1]:.4f}  bf16 evaluation of the result on the 4 batches {[round(x, 4) for x in ev16]}  every 20th {[round(float(x), 3) for x in l16[::20]]}", flush=True)
for mode in (sys.argv[2:] or ["0", "bwd", "all", "all"]):
    os.environ["MIC_FP8_HEAD"] = mode
    l8, ev8 = _run(dev, "fp8", steps, batches)
    rel = np.abs(l8 - l16) / np.abs(l16)
    tail = (l8[-20:].mean() - l16[-20:].mean()) / l16[-20:].mean()
    print(f"fp8 head={mode:<4} {l8[0]:.3f} -> {l8[-1]:.4f}  max dev {rel.max():.4f} at step {int(rel.argmax())}, mean {rel.mean():.4f}, "
          f"last-20 means {tail:+.4f}  bf16 evaluation of the result {[round(x, 4) for x in ev8]} ({np.mean(ev8) / np.mean(ev16) - 1:+.4f})  every 20th {[round(float(x), 3) for x in l8[::20]]}", flush=True)
