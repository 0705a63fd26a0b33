"""The LM head's backward launches in isolation (HIP-event time, 10 back-to-back launches each, uniform-random bf16 operands):
NT on the four-wave kernel (dlogits^T / h^T / E^T copies) beside the k-major launches of rounds 1-4.
usage: python tools/bench_head_bwd.py [rows ...]   (default 2176 2432 2632)"""
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
    rows_list = [int(x) for x in sys.argv[1:]] or [2176, 2432, 2632]
    g = torch.Generator(device="cpu").manual_seed(0)
    E = (torch.rand(Vpad, d, generator=g) - 0.5).to(torch.bfloat16).to(dev)
    ET = torch.empty(d, Vpad, dtype=torch.bfloat16, device=dev)
    print(f"E -> E^T ({Vpad} x {d}): {timed(lambda: ops.transpose_bf16(E, ET, Vpad, d)):.1f} us")
    gE = torch.empty(Vpad, d, dtype=torch.float32, device=dev)
    print(f"{'rows':>6} {'op':<34} {'us':>9} {'TF/s':>8}")
    for M in rows_list:
        Mcap, Kp = 4096, (M + 127) // 128 * 128
        logits = ((torch.rand(Mcap, Vpad, generator=g) - 0.5) * 8).to(torch.bfloat16).to(dev)
        hf = (torch.rand(Mcap, d, generator=g) - 0.5).to(torch.bfloat16).to(dev)
        labels = torch.randint(0, V, (Mcap,), generator=g, dtype=torch.int32).to(dev)
        mask = torch.ones(Mcap, dtype=torch.int32, device=dev)
        lse = torch.full((Mcap,), 14.0, device=dev)
        denom = torch.full((1,), float(M), device=dev)
        dT = torch.empty(Vpad, Mcap, dtype=torch.bfloat16, device=dev)
        hfT = torch.empty(d, Mcap, dtype=torch.bfloat16, device=dev)
        flb = torch.zeros(Vpad, device=dev)
        fl = 2.0 * M * Vpad * d

        def row(name, us, flops=0.0):
            print(f"{M:>6} {name:<34} {us:>9.1f} {flops / us / 1e6 if flops else 0:>8.0f}")

        row("ce_bwd (in place)", timed(lambda: ops.ce_bwd(logits, Vpad, V, Vpad, labels, mask, 0.0, lse, denom, M)))
        row("ce_bwd_t (+ dlogits^T, colsum)", timed(lambda: ops.ce_bwd_t(logits, Vpad, V, Vpad, labels, mask, 0.0, lse, denom, M, dT, rows_pad=Kp, colsum=flb)))
        row("h -> h^T", timed(lambda: ops.transpose_bf16(hf, hfT, M, d, rows_pad=Kp)))
        row("dE  NT four-wave, fp32 C", timed(lambda: ops.gemm(dT, hfT, gE, Vpad, d, Kp)), fl)
        Mp = (M + 63) // 64 * 64
        row("dE  TN (rounds 1-4), + a_rowsum", timed(lambda: ops.gemm(logits, hf, gE, Vpad, d, Mp, a_kmajor=True, b_kmajor=True, a_rowsum=flb, rowsum_k=M)), fl)
        tiles = ((M + 255) // 256) * 4
        for nsp in sorted({256 // tiles, 8, 16}):
            slab = Mcap * d
            d32 = torch.empty(nsp * Mcap, d, dtype=torch.float32, device=dev)
            out = torch.empty(Mcap, d, dtype=torch.bfloat16, device=dev)
            row(f"dX  NT four-wave, split {nsp}", timed(lambda: ops.gemm(logits, ET, d32, M, d, Vpad, split_k=nsp, split_stride=slab)), fl)
            row(f"    sum_slabs {nsp}", timed(lambda: ops.sum_slabs(d32, nsp, slab, out, M, d, d32.stride(0), out.stride(0))))
            del d32
        d32 = torch.empty(32 * Mcap, d, dtype=torch.float32, device=dev)
        row("dX  NN (rounds 1-4), split 32", timed(lambda: ops.gemm(logits, E, d32, M, d, Vpad, b_kmajor=True, split_k=32, split_stride=Mcap * d)), fl)
        del d32, logits, dT


if __name__ == "__main__":
    main()
