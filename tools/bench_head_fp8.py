"""The LM head's GEMMs as fp8 launches, in isolation (HIP-event time, 10 back-to-back launches each, uniform-random operands):
forward (e4m3 x e4m3, bf16 C + softmax partials), dE (k-major e5m2 x e4m3, fp32 C), dX (e5m2 x e4m3, split-K fp32 slabs).
usage: python tools/bench_head_fp8.py [rows ...]   (default 2432)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mic_amd  # noqa: F401,E402
from mic_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
V, Vpad, d = 250054, 250112, 1024


def timed(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    rows_list = [int(x) for x in sys.argv[1:]] or [2432]
    E4, E5 = torch.float8_e4m3fn, torch.float8_e5m2
    E8 = (torch.rand(Vpad, d, device=dev) - 0.5).to(E4)
    ET8 = (torch.rand(d, Vpad, device=dev) - 0.5).to(E4)
    gE = torch.empty(Vpad, d, dtype=torch.float32, device=dev)
    one = torch.ones(1, device=dev)
    print(f"{'rows':>6} {'op':<44} {'us':>9} {'TF/s':>8}")
    for M in rows_list:
        Mcap = 4096
        h8 = (torch.rand(Mcap, d, device=dev) - 0.5).to(E4)
        dl8 = (torch.rand(Mcap, Vpad, device=dev) - 0.5).to(E5)
        logits = torch.empty(Mcap, Vpad, dtype=torch.bfloat16, device=dev)
        stat = torch.empty(Mcap, 2 * (Vpad // 64), dtype=torch.float32, device=dev)
        flb = torch.zeros(Vpad, device=dev)
        fl = 2.0 * M * Vpad * d

        def row(name, us, flops=0.0):
            print(f"{M:>6} {name:<44} {us:>9.1f} {flops / us / 1e6 if flops else 0:>8.0f}", flush=True)

        row("fwd  fp8 NT, bf16 C + bias", timed(lambda: ops.gemm(h8, E8, logits, M, Vpad, d, bias=flb, a_scale_inv=one, b_scale_inv=one)), fl)
        try:
            row("fwd  fp8 NT, bf16 C + bias + softmax partials", timed(lambda: ops.gemm(h8, E8, logits, M, Vpad, d, bias=flb, rowstat=stat, rowstat_nvalid=V, a_scale_inv=one, b_scale_inv=one)), fl)
        except Exception as e:  # noqa: BLE001
            print("fwd + partials:", str(e)[:200])
        Mp = (M + 127) // 128 * 128
        row("dE   fp8 TN (k-major both), fp32 C", timed(lambda: ops.gemm(dl8, h8, gE, Vpad, d, Mp, a_kmajor=True, b_kmajor=True, k_valid=M, a_scale_inv=one, b_scale_inv=one)), fl)
        tiles = ((M + 255) // 256) * 4
        for nsp in sorted({256 // tiles, 2 * (256 // tiles), 16, 32}):
            slab = Mcap * d
            d32 = torch.empty(nsp * Mcap, d, dtype=torch.float32, device=dev)
            row(f"dX   fp8 NT, split {nsp}", timed(lambda: ops.gemm(dl8, ET8, d32, M, d, Vpad, split_k=nsp, split_stride=slab, a_scale_inv=one, b_scale_inv=one)), fl)
            del d32


if __name__ == "__main__":
    main()
