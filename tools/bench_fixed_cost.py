"""Fixed cost of a one-round GEMM launch: 4096x1024xK for small K (256 tiles of 128x128), HIP events over back-to-back launches."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import mic_amd  # noqa: F401
from mic_amd import ops

dev = torch.device("cuda:0")
for (M, N) in ((4096, 1024), (1024, 1024), (4096, 4096)):
    for K in (64, 128, 256, 512, 1024, 2048):
        sets = []
        for _ in range(6):
            A = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
            B = (torch.randn(N, K, device=dev) * 0.5).to(torch.bfloat16)
            sets.append((A, B, torch.empty(M, N, dtype=torch.bfloat16, device=dev)))
        for A, B, o in sets:
            ops.gemm(A, B, o, M, N, K)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            for A, B, o in sets:
                ops.gemm(A, B, o, M, N, K)
        e1.record()
        torch.cuda.synchronize()
        print(f"{M}x{N} K={K:5d} {e0.elapsed_time(e1) * 1e3 / 60:7.2f} us", flush=True)
# an empty-ish kernel for the launch floor
x = torch.zeros(64, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(200):
    ops.cast(x, x.clone() if False else x)
e1.record()
torch.cuda.synchronize()
print(f"tiny cast kernel back-to-back: {e0.elapsed_time(e1) * 1e3 / 200:.2f} us per launch")
