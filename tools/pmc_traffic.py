"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (counter_collection.csv) per kernel.
usage: python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> [n_steps] [out.json]
Units and corrections (MI355X_MICROARCH.md, HBM section): counters are KiB-like units of 1024 B as printed by rocprofv3;
on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads -> x2; WRITE_SIZE is calibrated here on adamw_kernel,
whose traffic is known exactly (16 B read, 14 B written per element)."""
import collections
import csv
import sys


def load(path, name):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != name:
            continue
        k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        k = k[:k.index("(")] if "(" in k else k
        if len(k) > 60:
            k = k[:60]
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    return agg


f = load(sys.argv[1], "FETCH_SIZE")
w = load(sys.argv[2], "WRITE_SIZE")
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
N = 547_223_040
KB = 1024.0
ad_f = sum(v[1] for k, v in f.items() if k.startswith("adamw_kernel")) / steps * KB  # adamw_kernel<false> (+ <true>: the row passes)
ad_w = sum(v[1] for k, v in w.items() if k.startswith("adamw_kernel")) / steps * KB
print(f"# adamw_kernel per step: FETCH_SIZE {ad_f / 1e9:.3f} GB raw (expected 16 B x {N} = {16 * N / 1e9:.3f} GB -> factor {16 * N / ad_f:.3f}); "
      f"WRITE_SIZE {ad_w / 1e9:.3f} GB raw (expected 14 B x n = {14 * N / 1e9:.3f} GB -> factor {14 * N / ad_w:.3f})")
cf, cw = 16 * N / ad_f, 14 * N / ad_w
print(f"# corrected = raw x {cf:.3f} (fetch), x {cw:.3f} (write); per step = total / {steps}")
print(f"{'kernel':62s} {'calls/step':>10s} {'fetch GB/step':>14s} {'write GB/step':>14s} {'MB/launch':>10s}")
tf = tw = 0.0
for k in sorted(set(f) | set(w), key=lambda k: -(f[k][1] * cf + w[k][1] * cw)):
    n = max(f[k][0], w[k][0]) / steps
    fb, wb = f[k][1] * KB * cf / steps, w[k][1] * KB * cw / steps
    tf += fb
    tw += wb
    if fb + wb > 1e6:
        print(f"{k:62s} {n:10.1f} {fb / 1e9:14.3f} {wb / 1e9:14.3f} {(fb + wb) / max(n, 1) / 1e6:10.2f}")
print(f"{'TOTAL':62s} {'':10s} {tf / 1e9:14.3f} {tw / 1e9:14.3f}")
if len(sys.argv) > 4:
    import json

    g = [k for k in set(f) | set(w) if k.startswith("gemm_")]
    n = sum(max(f[k][0], w[k][0]) for k in g) / steps
    fb = sum(f[k][1] for k in g) * KB * cf / steps
    wb = sum(w[k][1] for k in g) * KB * cw / steps
    json.dump({"kernel": "gemm_bf16_kernel + gemm_w4_kernel + gemm_d2_kernel + gemm_phased_kernel", "launches_per_step": n, "fetch_GB_per_step": round(fb / 1e9, 3),
               "write_GB_per_step": round(wb / 1e9, 3), "bytes_per_launch": int((fb + wb) / max(n, 1)),
               "calibration": {"fetch_factor": round(cf, 3), "write_factor": round(cw, 3), "on": "adamw_kernel (16 B read + 14 B written per element)"},
               "total_fetch_GB_per_step": round(tf / 1e9, 3), "total_write_GB_per_step": round(tw / 1e9, 3)}, open(sys.argv[4], "w"), indent=1)
