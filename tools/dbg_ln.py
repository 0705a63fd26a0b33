import sys, torch
sys.path.insert(0, '.')
import mic_amd
from mic_amd import ops
dev = torch.device('cuda:0')
rows, width = 50, 768
g = torch.Generator().manual_seed(1)
x = torch.randn(rows, width, generator=g) * 2
gamma, beta = 1 + 0.1 * torch.randn(width, generator=g), 0.1 * torch.randn(width, generator=g)
dy = torch.randn(rows, width, generator=g)
y = torch.empty_like(x, device=dev); mean = torch.empty(rows, device=dev); rstd = torch.empty(rows, device=dev)
ops.layernorm_fwd(x.to(dev), gamma.to(dev), beta.to(dev), 1e-5, y, mean, rstd)
keep = ops.dropout_mask(rows * width, 0.1, 3, dev).cpu().reshape(rows, width).float()
def ref(d):
    xr = x.clone().requires_grad_(True)
    torch.nn.functional.layer_norm(xr, (width,), gamma, beta, 1e-5).backward(d)
    return xr.grad
for name, kw in (("plain", {}), ("in_drop", dict(in_dropout_p=0.1, in_dropout_seed=3))):
    dx = torch.empty_like(x, device=dev)
    ops.layernorm_bwd(x.to(dev), gamma.to(dev), mean, rstd, dy.to(dev), dx, None, None, **kw)
    torch.cuda.synchronize()
    a, b = ref(dy), ref(dy * keep / 0.9)
    print(name, "vs unmasked", (dx.cpu() - a).abs().max().item(), "vs masked", (dx.cpu() - b).abs().max().item(), "scale", b.abs().max().item())
