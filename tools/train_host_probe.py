"""Where does the host spend a train step?  Wraps the pieces of Trainer.train_step with wall-clock timers (no device syncs added)
and prints, per step, the host time of each piece next to the step's total — a piece that blocks on the GPU shows up as long."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import mic_amd  # noqa: F401
from mic_amd import CLIPVisionMBartConfig, FlaxCLIPVisionMBartForConditionalGeneration, Trainer, create_learning_rate_fn, ops, loss_rows, packed_rows

dev = torch.device("cuda", 0)
cfg = CLIPVisionMBartConfig(mbart_config={}, clip_vision_config={})
model = FlaxCLIPVisionMBartForConditionalGeneration(cfg, seed=0, dtype=torch.bfloat16, device=dev)
B, T = 64, 64
lr_fn = create_learning_rate_fn(train_ds_size=10_000_000, train_batch_size=B, num_train_epochs=7, num_warmup_steps=1000, learning_rate=5e-5)
tr = Trainer(model, lr_fn, seed=42)
V, img = cfg.mbart_config.vocab_size, cfg.clip_vision_config.image_size
batches = [bench.synth_batch(B, T, V, img, 1234 + i) for i in range(2)]
dbs = [{k: torch.from_numpy(v).to(dev) for k, v in b.items()} for b in batches]
for b, db in zip(batches, dbs):
    idx, rl = loss_rows(b["attention_mask"], b["input_ids"])
    db["loss_rows"] = (torch.from_numpy(idx).to(dev), torch.from_numpy(rl).to(dev))
    pk = packed_rows(b["attention_mask"], b["decoder_input_ids"])
    db["packed_rows"] = tuple(torch.from_numpy(t).to(dev) for t in pk)

acc = collections.OrderedDict()
def wrap(obj, name, label=None):
    f = getattr(obj, name)
    label = label or name
    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0
    setattr(obj, name, g)

eng = model.engine
wrap(tr, "_prep"); wrap(tr, "_set_hyper"); wrap(tr, "_pmean_metrics")
for n in ("vit_forward", "decoder_forward", "head_logits", "loss_and_dlogits", "decoder_backward", "vit_backward"):
    if hasattr(eng, n):
        wrap(eng, n)
wrap(eng, "loss_and_grads")
for n in ("start_step", "progress", "finish"):
    wrap(tr.reducer, n, "reducer." + n)
wrap(model, "invalidate_params_cache")
wrap(ops, "adamw", "ops.adamw")
for i in range(4):
    tr.train_step(dbs[i % 2])
torch.cuda.synchronize()
for i in range(6):
    acc.clear()
    t0 = time.perf_counter()
    tr.train_step(dbs[i % 2])
    t = time.perf_counter() - t0
    print(f"step {i}: host {t * 1e3:6.2f} ms | " + "  ".join(f"{k} {v * 1e3:.2f}" for k, v in acc.items()), flush=True)
torch.cuda.synchronize()
# the same with a device sync in front of every step: pure host issue time against an idle GPU
for i in range(3):
    torch.cuda.synchronize()
    acc.clear()
    t0 = time.perf_counter()
    tr.train_step(dbs[i % 2])
    t = time.perf_counter() - t0
    torch.cuda.synchronize()
    t2 = time.perf_counter() - t0
    print(f"idle-GPU step {i}: host issue {t * 1e3:6.2f} ms, done after {t2 * 1e3:6.2f} ms | " + "  ".join(f"{k} {v * 1e3:.2f}" for k, v in acc.items()), flush=True)
