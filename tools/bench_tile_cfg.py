"""128x128 vs 64x64 tile configuration on the small GEMM shapes (run twice: MIC_GEMM_TILE=128 / =64)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import mic_amd  # noqa: F401
from mic_amd import ops

dev = torch.device("cuda:0")
shapes = [  # decode (1024 rows) then train one-round shapes
    ("dec qkv", 1024, 3072, 1024, 0, 0), ("dec o", 1024, 1024, 1024, 0, 0), ("dec fc1", 1024, 4096, 1024, 0, 0), ("dec fc2", 1024, 1024, 4096, 0, 0),
    ("greedy256 o", 256, 1024, 1024, 0, 0), ("greedy256 fc1", 256, 4096, 1024, 0, 0),
    ("tr o fwd", 4096, 1024, 1024, 0, 0), ("tr o dX", 4096, 1024, 1024, 0, 1), ("tr qkv fwd", 4096, 3072, 1024, 0, 0), ("tr fc2 fwd", 4096, 1024, 4096, 0, 0),
    ("tr fc1 dX", 4096, 1024, 4096, 0, 1), ("tr ckv", 3200, 2048, 1024, 0, 0),
    ("vit o fwd", 3200, 768, 768, 0, 0), ("vit o dX", 3200, 768, 768, 0, 1), ("vit qkv fwd", 3200, 2304, 768, 0, 0), ("vit fc2 fwd", 3200, 768, 3072, 0, 0),
    ("vit qkv dX", 3200, 768, 2304, 0, 1), ("vit proj dW", 1024, 768, 3200, 1, 1), ("vit o dW", 768, 768, 3200, 1, 1),
]
NSET = 6
for name, M, N, K, akm, bkm in shapes:
    sets = []
    for _ in range(NSET):
        A = (torch.randn((K, M) if akm else (M, K), device=dev) * 0.5).to(torch.bfloat16)
        B = (torch.randn((K, N) if bkm else (N, K), device=dev) * 0.5).to(torch.bfloat16)
        out = torch.empty((M, N), dtype=torch.float32 if akm else torch.bfloat16, device=dev)
        sets.append((A, B, out))
    for A, B, out in sets:
        ops.gemm(A, B, out, M, N, K, a_kmajor=bool(akm), b_kmajor=bool(bkm))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        for A, B, out in sets:
            ops.gemm(A, B, out, M, N, K, a_kmajor=bool(akm), b_kmajor=bool(bkm))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (5 * NSET)
    t128 = ((M + 127) // 128) * ((N + 127) // 128)
    print(f"{name:14s} {M:5d}x{N:5d}x{K:5d} {'T' if akm else 'N'}{'N' if bkm else 'T'} tiles128={t128:4d} {us:8.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF/s", flush=True)
