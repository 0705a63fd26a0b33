"""Train step with / without the per-bucket AdamW overlapped on the optimizer stream (single GPU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mic_amd, bench
from mic_amd import CLIPVisionMBartConfig, FlaxCLIPVisionMBartForConditionalGeneration, Trainer, create_learning_rate_fn, loss_rows
dev = torch.device("cuda:0")
cfg = CLIPVisionMBartConfig(mbart_config={}, clip_vision_config={})
model = FlaxCLIPVisionMBartForConditionalGeneration(cfg, dtype=torch.bfloat16, device=dev)
b = bench.synth_batch(64, 64, 250054, 224, 1)
db = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
idx, rl = loss_rows(b["attention_mask"], b["input_ids"]); db["loss_rows"] = (torch.from_numpy(idx).to(dev), torch.from_numpy(rl).to(dev))
for rep in range(2):
    for ov in (True, False):
        tr = Trainer(model, create_learning_rate_fn(10**7, 64, 7, 1000, 5e-5), overlap_optimizer=ov)
        for _ in range(3): tr.train_step(db)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): tr.train_step(db)
        torch.cuda.synchronize()
        print(f"overlap_optimizer={ov}: {(time.perf_counter() - t0) * 100:.3f} ms/step")
