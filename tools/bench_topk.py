import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mic_amd
from mic_amd import ops
dev = torch.device("cuda:0")
R, V, Vpad = 1024, 250054, 250112
logits = (torch.randn(R, Vpad, device=dev) * 0.7).to(torch.bfloat16)
bias = torch.zeros(R, device=dev)
for k, raw, name in ((8, False, "beam k=8"), (1, True, "greedy raw k=1"), (1, False, "lse k=1")):
    tv = torch.empty((R, k), device=dev); ti = torch.empty((R, k), dtype=torch.int32, device=dev)
    for _ in range(2):
        ops.row_lse_topk(logits, Vpad, V, k, tv, ti, R, raw_logits=raw, row_bias=bias)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(5):
        ops.row_lse_topk(logits, Vpad, V, k, tv, ti, R, raw_logits=raw, row_bias=bias)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 5
    print(f"{name:16s} {us:8.1f} us   {R * Vpad * 2 * (1 if raw else 2) / us / 1e6:.2f} TB/s")
