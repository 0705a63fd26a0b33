"""Does AdamW on a CU-masked stream run beside GEMMs on the complementary CUs inside this process (torch + ctypes launches)?
Times a chain of one-round K-group GEMMs (2176 x 1024 x 1024) and an AdamW pass over 256 M parameters: each alone, then both at once,
for plain streams and for masked streams (MIC_OPT_CUS CUs for AdamW, the rest for the GEMMs)."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
importlib.import_module("multilingual-image-captioning_amd")
ops = importlib.import_module("multilingual-image-captioning_amd.ops")
dev = torch.device("cuda:0")
n_opt = int(os.environ.get("MIC_OPT_CUS", 32))
M, N, K, L = int(os.environ.get("M", 2176)), 1024, 1024, 400
x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.02).bfloat16(); y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
n = 128 << 20
p = torch.zeros(n, device=dev); m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev); g = torch.zeros(n, device=dev)
lp = torch.zeros(n, device=dev, dtype=torch.bfloat16)
hyper = torch.tensor([1e-3, 1.0], device=dev)

def gemms(st):
    with torch.cuda.stream(st):
        for _ in range(L):
            ops.gemm(x, w, y, M, N, K)

def opt(st):
    with torch.cuda.stream(st):
        ops.adamw(p, m, v, g, lp, hyper, 0.9, 0.999, 1e-8, 0.0)

def timed(fs):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for f in fs: f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3

total = torch.cuda.get_device_properties(dev).multi_processor_count
for name, sg, so in (("plain streams", torch.cuda.Stream(), torch.cuda.Stream()),
                     (f"masked: GEMMs on {total - n_opt} CUs, AdamW on {n_opt}", ops.cu_masked_stream(n_opt, total - n_opt, dev), ops.cu_masked_stream(0, n_opt, dev))):
    timed([lambda: gemms(sg), lambda: opt(so)])
    a = min(timed([lambda: gemms(sg)]) for _ in range(3)); b = min(timed([lambda: opt(so)]) for _ in range(3))
    c = min(timed([lambda: opt(so), lambda: gemms(sg)]) for _ in range(3))
    print(f"{name}: {L} GEMMs {a:.2f} ms, AdamW {b:.2f} ms ({n * 30 / b * 1e-9:.2f} TB/s), both {c:.2f} ms")
