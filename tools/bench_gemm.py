"""Per-shape TFLOP/s of mic_gemm (bf16) for the GEMM shapes of the B=64 train step + a 4096^3 reference point."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mic_amd
from mic_amd import ops

dev = torch.device("cuda:0")
shapes = [  # (name, M, N, K, a_kmajor, b_kmajor)
    ("ref 4096^3 NT", 4096, 4096, 4096, 0, 0),
    ("dec qkv fwd", 4096, 3072, 1024, 0, 0), ("dec so fwd", 4096, 1024, 1024, 0, 0), ("dec fc1 fwd", 4096, 4096, 1024, 0, 0),
    ("dec fc2 fwd", 4096, 1024, 4096, 0, 0), ("dec ckv fwd", 3200, 2048, 1024, 0, 0),
    ("vit qkv fwd", 3200, 2304, 768, 0, 0), ("vit o fwd", 3200, 768, 768, 0, 0), ("vit fc1 fwd", 3200, 3072, 768, 0, 0), ("vit fc2 fwd", 3200, 768, 3072, 0, 0),
    ("head fwd", 4096, 250112, 1024, 0, 0),
    ("dec qkv dX", 4096, 1024, 3072, 0, 1), ("dec so dX", 4096, 1024, 1024, 0, 1), ("dec fc1 dX", 4096, 1024, 4096, 0, 1), ("dec fc2 dX", 4096, 4096, 1024, 0, 1),
    ("vit fc2 dX", 3200, 3072, 768, 0, 1), ("vit o dX", 3200, 768, 768, 0, 1),
    ("head dX", 4096, 1024, 250112, 0, 1),
    ("dec qkv dW", 3072, 1024, 4096, 1, 1), ("dec so dW", 1024, 1024, 4096, 1, 1), ("dec fc1 dW", 4096, 1024, 4096, 1, 1), ("dec fc2 dW", 1024, 4096, 4096, 1, 1),
    ("vit qkv dW", 2304, 768, 3200, 1, 1), ("vit o dW", 768, 768, 3200, 1, 1), ("vit fc1 dW", 3072, 768, 3200, 1, 1),
    ("head dW", 250112, 1024, 4096, 1, 1),
    ("head dX splitK8", 4096, 1024, 250112, 0, 1),
    ("cmp head dX", 2176, 1024, 250112, 0, 1), ("cmp head dX splitK8", 2176, 1024, 250112, 0, 1), ("cmp head dX splitK16", 2176, 1024, 250112, 0, 1),
    ("cmp head dX splitK24", 2176, 1024, 250112, 0, 1), ("cmp head dX splitK32", 2176, 1024, 250112, 0, 1), ("cmp head dX splitK64", 2176, 1024, 250112, 0, 1),
    ("cmp2 head dX", 2640, 1024, 250112, 0, 1), ("cmp2 head dX splitK16", 2640, 1024, 250112, 0, 1), ("cmp2 head dX splitK32", 2640, 1024, 250112, 0, 1),
]
only = sys.argv[1] if len(sys.argv) > 1 else None
tot_f, tot_t = 0.0, 0.0
for name, M, N, K, akm, bkm in shapes:
    if only and only not in name:
        continue
    A = (torch.randn((K, M) if akm else (M, K), device=dev) * 0.5).to(torch.bfloat16)
    B = (torch.randn((K, N) if bkm else (N, K), device=dev) * 0.5).to(torch.bfloat16)
    sk = int(name.split("splitK")[1]) if "splitK" in name else 0
    out = torch.empty((M, N), dtype=torch.float32 if (akm or sk) else torch.bfloat16, device=dev)
    reps = 3 if M * N * K > 5e11 else 20
    for _ in range(2):
        ops.gemm(A, B, out, M, N, K, a_kmajor=bool(akm), b_kmajor=bool(bkm), split_k=sk)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        ops.gemm(A, B, out, M, N, K, a_kmajor=bool(akm), b_kmajor=bool(bkm), split_k=sk)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    fl = 2.0 * M * N * K
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    print(f"{name:16s} M={M:6d} N={N:6d} K={K:6d} {'T' if akm else 'N'}{'N' if bkm else 'T'} tiles={tiles:6d}  {us:9.1f} us  {fl / us / 1e6:7.1f} TF/s")
    mult = 12 if ("dec" in name or "vit" in name) else 1
    if "ref" not in name and "splitK" not in name:
        tot_f += fl * mult; tot_t += us * mult
print(f"weighted (x12 layers): {tot_t / 1e3:.2f} ms for {tot_f / 1e12:.2f} TF -> {tot_f / tot_t / 1e6:.1f} TF/s")
