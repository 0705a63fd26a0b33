"""How a pinned host -> device copy of one batch (38.5 MB of fp32 pixels) behaves beside a train step: time of the copy alone, of the
step alone, and of both overlapped on two streams (bench.py's `h2d_inclusive` leg in small).  Run under `rocprofv3 --kernel-trace
--memory-copy-trace` to see whether the runtime uses an SDMA engine or a blit kernel for it."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mic_amd  # noqa: F401,E402
from bench import synth_batch  # noqa: E402
from mic_amd import CLIPVisionMBartConfig, FlaxCLIPVisionMBartForConditionalGeneration, Trainer, create_learning_rate_fn, loss_rows, packed_rows  # noqa: E402

dev = torch.device("cuda:0")
cfg = CLIPVisionMBartConfig(mbart_config={}, clip_vision_config={})
model = FlaxCLIPVisionMBartForConditionalGeneration(cfg, seed=0, dtype=torch.bfloat16, device=dev)
tr = Trainer(model, create_learning_rate_fn(10_000_000, 64, 7, 1000, 5e-5), seed=42)
b = synth_batch(64, 64, 250054, 224, 1)
db = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
idx, rl = loss_rows(b["attention_mask"], b["input_ids"])
db["loss_rows"] = (torch.from_numpy(idx).to(dev), torch.from_numpy(rl).to(dev))
db["packed_rows"] = tuple(torch.from_numpy(t).to(dev) for t in packed_rows(b["attention_mask"], b["decoder_input_ids"]))
pin = torch.from_numpy(b["pixel_values"]).pin_memory()
dst = torch.empty_like(pin, device=dev)
cs = torch.cuda.Stream(device=dev)
for _ in range(5):
    tr.train_step(db)
torch.cuda.synchronize()


def timed(fn, n=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def copy_only():
    with torch.cuda.stream(cs):
        dst.copy_(pin, non_blocking=True)


def both():
    with torch.cuda.stream(cs):
        dst.copy_(pin, non_blocking=True)
    tr.train_step(db)


print(f"copy alone {timed(copy_only):.3f} ms ({pin.numel() * 4 / 1e6:.1f} MB)  step alone {timed(lambda: tr.train_step(db)):.3f} ms  "
      f"copy on its own stream beside the step {timed(both):.3f} ms")
