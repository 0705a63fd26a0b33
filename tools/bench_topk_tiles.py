"""Time mic_row_topk_tiles (beam top-2K from the head GEMM's per-granule softmax partials) on a decode-step-sized problem."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mic_amd
from mic_amd import ops
dev = torch.device("cuda:0")
R, V, Vpad, K = 1024, 250054, 250112, 1024
g = torch.Generator().manual_seed(0)
x = torch.randn(R, K, generator=g).to(torch.bfloat16).to(dev)
w = (torch.randn(Vpad, K, generator=g) * 0.02).to(torch.bfloat16).to(dev)
w[V:] = 0
logits = torch.zeros((R, Vpad), dtype=torch.bfloat16, device=dev)
nt = Vpad // 64
stat = torch.zeros((R, nt, 2), device=dev)
ops.gemm(x, w, logits, R, Vpad, K, rowstat=stat, rowstat_nvalid=V)
bias = torch.zeros(R, device=dev)
for k in (8, 2):
    tv = torch.empty((R, k), device=dev); ti = torch.empty((R, k), dtype=torch.int32, device=dev)
    for _ in range(2):
        ops.row_topk_tiles(logits, Vpad, V, stat, k, tv, ti, R, row_bias=bias)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10):
        ops.row_topk_tiles(logits, Vpad, V, stat, k, tv, ti, R, row_bias=bias)
    e1.record(); torch.cuda.synchronize()
    print(f"k={k} stop={os.environ.get('MIC_TOPK_STOP', '0')}: {e0.elapsed_time(e1) * 1e3 / 10:8.1f} us")
