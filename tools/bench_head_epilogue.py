"""Where the LM-head forward's time goes: the fp8 (and bf16) NT launch [rows x 250112] at K = 128 ... 1024 with the bare bias epilogue
and with the softmax partials — the K -> 0 intercept is prologue + epilogue + store.  usage: python tools/bench_head_epilogue.py [rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mic_amd  # noqa: F401,E402
from mic_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
V, Vpad = 250054, 250112


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


M = int(sys.argv[1]) if len(sys.argv) > 1 else 2432
one = torch.ones(1, device=dev)
logits = torch.empty(4096, Vpad, dtype=torch.bfloat16, device=dev)
stat = torch.empty(4096, 2 * (Vpad // 64), dtype=torch.float32, device=dev)
flb = torch.zeros(Vpad, device=dev)
print(f"{'dtype':<6} {'K':>6} {'bias only us':>14} {'+ partials us':>14}")
for dt in ("fp8", "bf16"):
    for K in (128, 256, 512, 1024, 2048):
        if dt == "fp8":
            a = (torch.rand(4096, K, device=dev) - 0.5).to(torch.float8_e4m3fn)
            b = (torch.rand(Vpad, K, device=dev) - 0.5).to(torch.float8_e4m3fn)
            kw = dict(a_scale_inv=one, b_scale_inv=one)
        else:
            a = (torch.rand(4096, K, device=dev) - 0.5).to(torch.bfloat16)
            b = (torch.rand(Vpad, K, device=dev) - 0.5).to(torch.bfloat16)
            kw = {}
        t0 = timed(lambda: ops.gemm(a, b, logits, M, Vpad, K, bias=flb, **kw))
        t1 = timed(lambda: ops.gemm(a, b, logits, M, Vpad, K, bias=flb, rowstat=stat, rowstat_nvalid=V, **kw))
        print(f"{dt:<6} {K:>6} {t0:>14.1f} {t1:>14.1f}", flush=True)
        del a, b
