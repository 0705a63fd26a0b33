"""LayerNorm backward at the two train shapes; run under rocprofv3 --kernel-trace for GPU-side durations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mic_amd
from mic_amd import ops
dev = torch.device("cuda:0")
for rows, width in ((2432, 1024), (4096, 1024), (3200, 768)):
    sets = []
    for _ in range(8):
        x = torch.randn(rows, width, device=dev).to(torch.bfloat16)
        sets.append((x, torch.randn_like(x), torch.randn_like(x), torch.empty_like(x), torch.empty_like(x)))
    g = torch.ones(width, device=dev); mean = torch.zeros(rows, device=dev); rstd = torch.ones(rows, device=dev)
    dg = torch.zeros(width, device=dev); db = torch.zeros(width, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(2):
        torch.cuda.synchronize(); e0.record()
        for _ in range(5):
            for x, dy, dres, dx, dxm in sets:
                ops.layernorm_bwd(x, g, mean, rstd, dy, dx, None if os.environ.get("NOATOM") else dg, None if os.environ.get("NOATOM") else db, dres=dres, dxm=dxm, dropout_p=float(os.environ.get("LNP", "0.1")), dropout_seed=3)
        e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 40
    print(f"ln_bwd {rows}x{width}: {us:.1f} us  {rows*width*2*5/us/1e6:.2f} TB/s")
