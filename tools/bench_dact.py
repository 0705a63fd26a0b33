"""dX GEMM with the fused activation derivative (C = (dy W) * act'(Zin)) beside the same GEMM with the bare epilogue."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
importlib.import_module("multilingual-image-captioning_amd")
ops = importlib.import_module("multilingual-image-captioning_amd.ops")
L = importlib.import_module("multilingual-image-captioning_amd._lib")
dev = torch.device("cuda:0")


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for (M, N, K, act) in [(4096, 4096, 1024, L.ACT_IDS["tanh"]), (3200, 3072, 768, L.ACT_QUICK_GELU)]:
    dy = torch.randn(M, K, device=dev).bfloat16()
    w = torch.randn(K, N, device=dev).bfloat16()   # [K][N]: b_kmajor
    z = torch.randn(M, N, device=dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t0 = timeit(lambda: ops.gemm(dy, w, out, M, N, K, b_kmajor=True))
    t1 = timeit(lambda: ops.gemm(dy, w, out, M, N, K, b_kmajor=True, zin=z, dact=act))
    t2 = timeit(lambda: ops.gemm(dy, w, out, M, N, K, b_kmajor=True, residual=z))
    print(f"{M}x{N}x{K} NN: plain {t0:.1f} us, + dact(Zin) {t1:.1f} us, + residual (plain path, same extra read) {t2:.1f} us", flush=True)
