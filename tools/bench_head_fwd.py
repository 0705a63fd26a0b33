"""The LM-head forward launches (train: packed rows, bias + softmax partials; decode: 1024 rows) and two square shapes, mic_gemm beside
torch.matmul (HIP-event time of 10 back-to-back launches, uniform-random bf16 operands).  Run under MIC_GEMM_D2=0/1 to compare kernels."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mic_amd  # noqa: F401,E402
from mic_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def t(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(f"MIC_GEMM_D2={os.environ.get('MIC_GEMM_D2', '0')}")
for (M, N, K, stats) in ((4096, 4096, 4096, False), (8192, 8192, 8192, False), (4096, 4096, 1024, False), (3200, 24576, 1024, False),
                         (2432, 250112, 1024, True), (2176, 250112, 1024, True), (1024, 250112, 1024, True), (1024, 250112, 1024, False)):
    a = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16)
    b = (torch.rand(N, K, device=dev) * 2 - 1).to(torch.bfloat16)
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    bias = torch.zeros(N, device=dev)
    st = torch.zeros((M, 2 * (N // 64)), dtype=torch.float32, device=dev) if stats else None
    us_mic = t(lambda: ops.gemm(a, b, c, M, N, K, bias=bias, rowstat=st, rowstat_nvalid=N - 58 if stats else 0))
    us_lib = t(lambda: torch.matmul(a, b.t(), out=c))
    fl = 2.0 * M * N * K
    print(f"{M:>5} x {N:>6} x {K:>4} {'bias+partials' if stats else 'bias':<13}: mic {us_mic:8.1f} us {fl / us_mic / 1e6:7.1f} TF/s | lib {us_lib:8.1f} us {fl / us_lib / 1e6:7.1f} TF/s | mic/lib {us_mic / us_lib:.2f}")
    del a, b, c, st
