"""Cost of the fused GEMM epilogues at one big-tile round (4096x4096x1024) and one small-tile shape."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import mic_amd  # noqa: F401
from mic_amd import ops

dev = torch.device("cuda:0")


def run(name, M, N, K, bkm, **kw):
    A = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    B = (torch.randn((K, N) if bkm else (N, K), device=dev) * 0.05).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    for _ in range(3):
        ops.gemm(A, B, out, M, N, K, b_kmajor=bkm, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        ops.gemm(A, B, out, M, N, K, b_kmajor=bkm, **kw)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"{name:34s} {M}x{N}x{K} {'NN' if bkm else 'NT'} {us:8.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF/s")


for (M, N, K) in ((4096, 4096, 1024), (3200, 3072, 768), (4096, 1024, 1024)):
    z = (torch.randn(M, N, device=dev)).to(torch.bfloat16)
    z2 = torch.empty_like(z)
    bias = torch.randn(N, device=dev)
    run("plain", M, N, K, False)
    run("bias", M, N, K, False, bias=bias)
    run("bias+zout", M, N, K, False, bias=bias, zout=z2)
    run("bias+act tanh", M, N, K, False, bias=bias, act=2)
    run("bias+act tanh+zout", M, N, K, False, bias=bias, act=2, zout=z2)
    run("bias+act quick+zout", M, N, K, False, bias=bias, act=3, zout=z2)
    run("bias+residual", M, N, K, False, bias=bias, residual=z)
    run("bias+residual+dropout", M, N, K, False, bias=bias, residual=z, dropout_p=0.1, dropout_seed=5)
    run("NN plain", M, N, K, True)
    run("NN zin dact tanh", M, N, K, True, zin=z, dact=2)
    run("NN zin dact quick", M, N, K, True, zin=z, dact=3)
    run("NN zin dact erf", M, N, K, True, zin=z, dact=1)
