"""A decode-step d x d GEMM (1024 x 1024 x 1024, 64x64 tiles, 4 K-groups) with its weight hot in cache (the same matrix every
launch) and cold (a ring of matrices larger than L2 + Infinity Cache), with and without the folded-LayerNorm / row-sum epilogues."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
importlib.import_module("multilingual-image-captioning_amd")
ops = importlib.import_module("multilingual-image-captioning_amd.ops")
dev = torch.device("cuda:0")
M = N = K = 1024
NW = 320  # 320 x 2 MB = 640 MB of weights
ws = [(torch.randn(N, K, device=dev) * 0.02).bfloat16() for _ in range(NW)]
xs = [torch.randn(M, K, device=dev).bfloat16() for _ in range(8)]
y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
bias = torch.zeros(N, device=dev)

def run(hot, iters=320):
    for i in range(16):
        ops.gemm(xs[i % 8], ws[0 if hot else i], y, M, N, K, bias=bias)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters):
        ops.gemm(xs[i % 8], ws[0 if hot else i % NW], y, M, N, K, bias=bias)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

for _ in range(2):
    print(f"hot weights {run(True):6.2f} us   cold weights {run(False):6.2f} us  (host-issue bound below ~11 us: read the rocprofv3 kernel durations)")
