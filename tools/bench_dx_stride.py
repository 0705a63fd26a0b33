"""Diagnostic: head-dX shaped NN GEMM (M x 1024 x 250112) with the A operand at different row strides (timing only)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import mic_amd  # noqa: F401
from mic_amd import ops

dev = torch.device("cuda:0")
M, N, K = 2176, 1024, 250112
base = (torch.randn(M * K + 4096, device=dev) * 0.1).to(torch.bfloat16)
B = (torch.randn(K, N, device=dev) * 0.05).to(torch.bfloat16)
Bsmall = B[:4096]
for name, lda, sk, bb in (("real lda=K", K, 0, B), ("real lda=K sk16", K, 16, B), ("alias lda=64", 64, 0, B), ("alias lda=64 sk16", 64, 16, B),
                          ("lda=K, B aliased 4096 rows", K, 0, None), ("lda=K sk16, B aliased", K, 16, None)):
    A = base.as_strided((M, K), (lda, 1))
    out = torch.zeros(M, N, dtype=torch.float32 if sk else torch.bfloat16, device=dev)
    g = ops.gemm_args(A, bb if bb is not None else B, out, M, N, K, b_kmajor=True, split_k=sk)
    if bb is None:
        g.ldb = 0  # every k reads the same 2-KB row of B
    for _ in range(2):
        ops.gemm_grouped([g])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(3):
        ops.gemm_grouped([g])
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 3
    print(f"{name:30s} {us:9.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF/s", flush=True)
