"""One fp8 GEMM shape, a few launches (for rocprofv3 --pmc passes): python tools/one_gemm_fp8.py M N K [reps]
LAYOUT=NT (default: both operands k-contiguous) | TN (both k-major: the weight-gradient form, ds_read_b64_tr_b8 fragments)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mic_amd  # noqa: F401,E402
from mic_amd import ops  # noqa: E402

M, N, K = (int(x) for x in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = torch.device("cuda:0")
tn = os.environ.get("LAYOUT", "NT") == "TN"
a = (torch.rand((K, M) if tn else (M, K), device=dev) * 2 - 1).to(torch.float8_e5m2 if tn else torch.float8_e4m3fn)
b = (torch.rand((K, N) if tn else (N, K), device=dev) * 2 - 1).to(torch.float8_e4m3fn)
c = torch.empty(M, N, device=dev, dtype=torch.float32 if tn else torch.bfloat16)
one = torch.ones(1, device=dev)
for _ in range(reps):
    ops.gemm(a, b, c, M, N, K, a_kmajor=tn, b_kmajor=tn, a_scale_inv=one, b_scale_inv=one)
torch.cuda.synchronize()
