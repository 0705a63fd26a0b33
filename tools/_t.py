import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
importlib.import_module("multilingual-image-captioning_amd")
ops = importlib.import_module("multilingual-image-captioning_amd.ops")
dev = torch.device("cuda:0")
for (M, N, K) in ((512, 66048, 256), (1024, 66048, 1024), (4096, 4096, 4096)):
    a = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16); b = (torch.rand(N, K, device=dev) * 2 - 1).to(torch.bfloat16)
    c = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    print(ops.gemm_plan([(M, N, K)]), flush=True)
    ops.gemm(a, b, c, M, N, K); torch.cuda.synchronize()
    ref = a.float() @ b.float().T
    err = (c.float() - ref).abs().max().item() / ref.abs().max().item()
    print(M, N, K, "relerr", err, flush=True)
    if err > 0.02:
        bad = ((c.float() - ref).abs() > 0.05 * ref.abs().max()).nonzero()
        print("bad count", bad.shape[0], bad[:10].tolist(), flush=True)
