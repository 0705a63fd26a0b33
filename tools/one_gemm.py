"""One bf16 GEMM shape, a few launches (for rocprofv3 --pmc passes): python tools/one_gemm.py M N K [reps] [stats]
LAYOUT=NT|NN|TN in the environment picks the operand layouts (default NT; stats only with NT)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mic_amd  # noqa: F401,E402
from mic_amd import ops  # noqa: E402

M, N, K = (int(x) for x in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
stats = len(sys.argv) > 5
dev = torch.device("cuda:0")
lay = os.environ.get("LAYOUT", "NT")
akm, bkm = lay[0] == "T", lay[1] == "N"
a = (torch.rand((K, M) if akm else (M, K), device=dev) * 2 - 1).to(torch.bfloat16)
b = (torch.rand((K, N) if bkm else (N, K), device=dev) * 2 - 1).to(torch.bfloat16)
c = torch.empty(M, N, device=dev, dtype=torch.float32 if akm else torch.bfloat16)
bias = torch.zeros(N, device=dev)
st = torch.zeros((M, 2 * (N // 64)), dtype=torch.float32, device=dev) if stats else None
for _ in range(reps):
    if lay == "NT":
        ops.gemm(a, b, c, M, N, K, bias=bias, rowstat=st, rowstat_nvalid=N - 58 if stats else 0)
    else:
        ops.gemm(a, b, c, M, N, K, a_kmajor=akm, b_kmajor=bkm)
torch.cuda.synchronize()
