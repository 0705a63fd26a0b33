"""mic_ce_bwd alone on the train step's shape (rows x 250 112 bf16 logits, in place): us per launch and HBM rate."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
importlib.import_module("multilingual-image-captioning_amd")
ops = importlib.import_module("multilingual-image-captioning_amd.ops")
dev = torch.device("cuda:0")
R, V, Vpad = int(os.environ.get("ROWS", 2176)), 250054, 250112
logits = torch.randn(R, Vpad, device=dev).bfloat16()
labels = torch.randint(0, V, (R,), device=dev, dtype=torch.int32)
mask = torch.ones(R, device=dev, dtype=torch.int32)
lse = torch.full((R,), 13.0, device=dev)
denom = torch.tensor([float(R)], device=dev)


def run():
    ops.ce_bwd(logits, Vpad, V, Vpad, labels, mask, 0.0, lse, denom, R)


for _ in range(3):
    run()
torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        run()
    e.record(); torch.cuda.synchronize()
    best = min(best, s.elapsed_time(e) / 10 * 1e3)
print(f"ce_bwd {R} x {Vpad}: {best:.1f} us per launch, {2.0 * R * Vpad * 2 / best * 1e-6:.2f} TB/s (read + write)")
