"""Host enqueue time of one train step (Python + ctypes, no device sync) vs the GPU time of the step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mic_amd, bench
from mic_amd import CLIPVisionMBartConfig, FlaxCLIPVisionMBartForConditionalGeneration, Trainer, create_learning_rate_fn, loss_rows
dev = torch.device("cuda:0")
cfg = CLIPVisionMBartConfig(mbart_config={}, clip_vision_config={})
model = FlaxCLIPVisionMBartForConditionalGeneration(cfg, dtype=torch.bfloat16, device=dev)
tr = Trainer(model, create_learning_rate_fn(10**7, 64, 7, 1000, 5e-5))
b = bench.synth_batch(64, 64, 250054, 224, 1)
db = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
idx, rl = loss_rows(b["attention_mask"], b["input_ids"]); db["loss_rows"] = (torch.from_numpy(idx).to(dev), torch.from_numpy(rl).to(dev))
for _ in range(3): tr.train_step(db)
torch.cuda.synchronize()
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.train_step(db)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"host enqueue {1e3*(t1-t0):.2f} ms, until GPU done {1e3*(t2-t0):.2f} ms")
