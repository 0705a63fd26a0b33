import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
importlib.import_module("multilingual-image-captioning_amd")
ops = importlib.import_module("multilingual-image-captioning_amd.ops")
dev = torch.device("cuda:0")
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, K) in ((2432, 250112, 1024), (2200, 250112, 1024), (1024, 250112, 1024)):
    a = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16); b = (torch.rand(N, K, device=dev) * 0.1).to(torch.bfloat16)
    c = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    bias = torch.randn(N, device=dev)
    stat = torch.zeros((M, 2 * (N // 64)), dtype=torch.float32, device=dev)
    print(M, N, K, "bias %.1f" % t(lambda: ops.gemm(a, b, c, M, N, K, bias=bias)),
          "bias+stat %.1f" % t(lambda: ops.gemm(a, b, c, M, N, K, bias=bias, rowstat=stat, rowstat_nvalid=N - 58)), flush=True)
    ref = a[-300:].float() @ b[:4096].float().T + bias[:4096]
    print("   relerr last rows", ((c[-300:, :4096].float() - ref).abs().max() / ref.abs().max()).item())
