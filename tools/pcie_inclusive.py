"""Train-step rate when every batch arrives as host numpy buffers (what main.py's collate_fn hands over): H2D included."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mic_amd, bench
from mic_amd import CLIPVisionMBartConfig, FlaxCLIPVisionMBartForConditionalGeneration, Trainer, create_learning_rate_fn, loss_rows
dev = torch.device("cuda:0")
cfg = CLIPVisionMBartConfig(mbart_config={}, clip_vision_config={})
model = FlaxCLIPVisionMBartForConditionalGeneration(cfg, dtype=torch.bfloat16, device=dev)
tr = Trainer(model, create_learning_rate_fn(10**7, 64, 7, 1000, 5e-5))
bs = [bench.synth_batch(64, 64, 250054, 224, i) for i in range(2)]
for b in bs: b["loss_rows"] = loss_rows(b["attention_mask"], b["input_ids"])
for i in range(3): tr.train_step(bs[i % 2])
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(10): tr.train_step(bs[i % 2])
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f"host-buffer inputs (pageable numpy, 38.5 MB pixels/step): {dt*1e3:.2f} ms/step, {64/dt:.0f} images/s")
