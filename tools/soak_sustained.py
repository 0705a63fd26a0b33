"""One Trainer, many steps: ms per step over consecutive windows (does the step time hold over a sustained run?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import mic_amd  # noqa: F401
from mic_amd import CLIPVisionMBartConfig, FlaxCLIPVisionMBartForConditionalGeneration, Trainer, create_learning_rate_fn, loss_rows, packed_rows

dev = torch.device("cuda", 0)
cfg = CLIPVisionMBartConfig(mbart_config={}, clip_vision_config={})
B, T = 64, 64
V, img = cfg.mbart_config.vocab_size, cfg.clip_vision_config.image_size
batches = [bench.synth_batch(B, T, V, img, 1234 + i) for i in range(2)]
model = FlaxCLIPVisionMBartForConditionalGeneration(cfg, seed=0, dtype=torch.bfloat16, device=dev)
GD = os.environ.get("GEMM_DTYPE") or None  # GEMM_DTYPE=fp8: configs[4]
tr = Trainer(model, create_learning_rate_fn(10_000_000, B, 7, 1000, 5e-5), seed=42, gemm_dtype=GD)
dbs = [{k: torch.from_numpy(v).to(dev) for k, v in b.items()} for b in batches]
for b, db in zip(batches, dbs):
    idx, rl = loss_rows(b["attention_mask"], b["input_ids"])
    db["loss_rows"] = (torch.from_numpy(idx).to(dev), torch.from_numpy(rl).to(dev))
    db["packed_rows"] = tuple(torch.from_numpy(t).to(dev) for t in packed_rows(b["attention_mask"], b["decoder_input_ids"]))
for i in range(4):
    tr.train_step(dbs[i % 2])
torch.cuda.synchronize()
W, NW = int(os.environ.get("WINDOW", 50)), int(os.environ.get("WINDOWS", 12))
out = []
for w in range(NW):
    t0 = time.perf_counter()
    for i in range(W):
        tr.train_step(dbs[i % 2])
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t0) / W * 1e3)
last = float(tr.train_step(dbs[0])["loss"])
assert last == last and abs(last) < 1e4, last  # (finite)
print(f"[{GD or 'bf16'}] ms per step over {NW} consecutive windows of {W} steps: " + " ".join(f"{x:.2f}" for x in out) + f"; last loss {last:.4f}")
