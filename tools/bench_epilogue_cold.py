"""Same as bench_epilogue.py but every launch touches a different buffer set (working set >> 256 MB MALL): HBM-cold operands
as inside the train step."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import mic_amd  # noqa: F401
from mic_amd import ops

dev = torch.device("cuda:0")
NSET = 12


def run(name, M, N, K, bkm, mk):
    sets = []
    for i in range(NSET):
        A = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
        B = (torch.randn((K, N) if bkm else (N, K), device=dev) * 0.05).to(torch.bfloat16)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        sets.append((A, B, out, mk(M, N)))
    for s in sets[:3]:
        ops.gemm(s[0], s[1], s[2], M, N, K, b_kmajor=bkm, **s[3])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for r in range(2):
        for s in sets:
            ops.gemm(s[0], s[1], s[2], M, N, K, b_kmajor=bkm, **s[3])
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (2 * NSET)
    print(f"{name:34s} {M}x{N}x{K} {'NN' if bkm else 'NT'} {us:8.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF/s", flush=True)


def Z(M, N):
    return torch.randn(M, N, device=dev).to(torch.bfloat16)


for (M, N, K) in ((4096, 4096, 1024), (3200, 3072, 768), (4096, 1024, 1024), (4096, 1024, 4096)):
    run("plain", M, N, K, False, lambda M, N: {})
    run("bias+act tanh+zout", M, N, K, False, lambda M, N: dict(bias=torch.randn(N, device=dev), act=2, zout=Z(M, N)))
    run("bias+residual+dropout", M, N, K, False, lambda M, N: dict(bias=torch.randn(N, device=dev), residual=Z(M, N), dropout_p=0.1, dropout_seed=5))
    run("NN plain", M, N, K, True, lambda M, N: {})
    run("NN zin dact tanh", M, N, K, True, lambda M, N: dict(zin=Z(M, N), dact=2))
