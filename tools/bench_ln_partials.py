"""LayerNorm forward / backward (partials form, as the train step runs it) at the train shapes: HIP-event time of back-to-back launches
over rotating buffers.  MIC_LNB_BLOCKS=<cap> changes the backward's block cap (default 256)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mic_amd
from mic_amd import ops
dev = torch.device("cuda:0")
for rows, width in ((2432, 1024), (2176, 1024), (3200, 768), (4096, 1024)):
    sets = []
    for _ in range(8):
        x = torch.randn(rows, width, device=dev).to(torch.bfloat16)
        sets.append((x, torch.randn_like(x), torch.randn_like(x), torch.empty_like(x), torch.empty_like(x)))
    g = torch.ones(width, device=dev); b = torch.zeros(width, device=dev); mean = torch.zeros(rows, device=dev); rstd = torch.ones(rows, device=dev)
    nb = ops.layernorm_bwd_blocks(rows)
    part = torch.zeros((2 * nb, width), device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    res = {}
    for name in ("bwd", "fwd"):
        for rep in range(2):
            torch.cuda.synchronize(); e0.record()
            for _ in range(5):
                for x, dy, dres, dx, dxm in sets:
                    if name == "bwd":
                        ops.layernorm_bwd_partials(x, g, mean, rstd, dy, dx, part, rows=rows, dres=dres, dxm=dxm, dropout_p=0.1, dropout_seed=3)
                    else:
                        ops.layernorm_fwd(x, g, b, 1e-5, dx, mean, rstd, rows=rows)
            e1.record(); torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) * 1e3 / 40
    print(f"{rows}x{width}: ln_bwd {res['bwd']:.1f} us ({nb} blocks)  ln_fwd {res['fwd']:.1f} us")
