"""Per-K-tile cost of the big (256^2) and small (128^2) tile kernels by layout: one round of 256 tiles, K sweep."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import mic_amd  # noqa: F401
from mic_amd import ops

dev = torch.device("cuda:0")
for (M, N) in ((4096, 4096), (2048, 2048)):
    for lay in ("NT", "NN", "TN"):
        prev = None
        for K in (1024, 4096, 16384):
            akm, bkm = lay[0] == "T", lay[1] == "N"
            A = (torch.randn((K, M) if akm else (M, K), device=dev) * 0.5).to(torch.bfloat16)
            B = (torch.randn((K, N) if bkm else (N, K), device=dev) * 0.5).to(torch.bfloat16)
            out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
            for _ in range(2):
                ops.gemm(A, B, out, M, N, K, a_kmajor=akm, b_kmajor=bkm)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(5):
                ops.gemm(A, B, out, M, N, K, a_kmajor=akm, b_kmajor=bkm)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 200
            slope = "" if prev is None else f"  slope {(us - prev[1]) / ((K - prev[0]) / 64):.3f} us/K-tile"
            prev = (K, us)
            print(f"M={M} N={N} {lay} K={K:6d} {us:9.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF/s{slope}", flush=True)
