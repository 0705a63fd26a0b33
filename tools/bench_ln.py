import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mic_amd
from mic_amd import ops
dev = torch.device("cuda:0")
rows, width = 4096, 1024
x = torch.randn(rows, width, device=dev).to(torch.bfloat16); dy = torch.randn_like(x); dres = torch.randn_like(x)
g = torch.ones(width, device=dev); mean = torch.zeros(rows, device=dev); rstd = torch.ones(rows, device=dev)
dx = torch.empty_like(x); dxm = torch.empty_like(x); dg = torch.zeros(width, device=dev); db = torch.zeros(width, device=dev)
for _ in range(3): ops.layernorm_bwd(x, g, mean, rstd, dy, dx, dg, db, dres=dres, dxm=dxm, dropout_p=0.1, dropout_seed=3)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(50): ops.layernorm_bwd(x, g, mean, rstd, dy, dx, dg, db, dres=dres, dxm=dxm, dropout_p=0.1, dropout_seed=3)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 50
print(f"ln_bwd {us:.1f} us  {rows*width*2*5/us/1e6:.2f} TB/s")
