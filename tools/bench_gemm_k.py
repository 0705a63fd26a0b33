import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mic_amd
from mic_amd import ops
dev = torch.device("cuda:0")
for (M, N) in ((4096, 65536), (4096, 1024), (4096, 4096)):
    for K in (256, 512, 1024, 2048, 4096):
        A = (torch.randn((M, K), device=dev) * 0.5).to(torch.bfloat16)
        B = (torch.randn((N, K), device=dev) * 0.5).to(torch.bfloat16)
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        for _ in range(2): ops.gemm(A, B, out, M, N, K)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10): ops.gemm(A, B, out, M, N, K)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        print(f"M={M} N={N} K={K:5d}  {us:9.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF/s")
