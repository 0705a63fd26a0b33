"""Per-kernel HBM-side bytes of the beam-4 generate leg from two rocprofv3 passes (--pmc FETCH_SIZE / --pmc WRITE_SIZE with
--kernel-trace only) over `bench.py --generate-only`.
usage: python tools/pmc_traffic_gen.py <fetch_counter_collection.csv> <write_counter_collection.csv> <decoder steps in the run> [out.json]
Units and corrections as in pmc_traffic.py (MI355X_MICROARCH.md, HBM section): counter values x 1024 B; FETCH_SIZE x 2 on gfx950;
the factors were calibrated on adamw_kernel in the train-step passes (2.000 / 1.000) — that kernel does not run here."""
import collections
import csv
import json
import sys


def load(path, name):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != name:
            continue
        k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        k = k[:k.index("(")] if "(" in k else k
        agg[k[:70]][0] += 1
        agg[k[:70]][1] += float(r["Counter_Value"])
    return agg


f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
steps = float(sys.argv[3])
print(f"# corrected = FETCH_SIZE x 1024 B x 2 (gfx950), WRITE_SIZE x 1024 B; per decoder step = total / {steps:g}")
print(f"{'kernel':72s} {'calls/step':>10s} {'fetch MB/step':>14s} {'write MB/step':>14s} {'MB/launch':>10s}")
tf = tw = 0.0
rows = {}
for k in sorted(set(f) | set(w), key=lambda k: -(f[k][1] * 2 + w[k][1])):
    n = max(f[k][0], w[k][0])
    fb, wb = f[k][1] * 1024 * 2, w[k][1] * 1024
    tf += fb
    tw += wb
    rows[k] = (n, fb, wb)
    if (fb + wb) / steps > 1e5:
        print(f"{k:72s} {n / steps:10.2f} {fb / steps / 1e6:14.2f} {wb / steps / 1e6:14.2f} {(fb + wb) / n / 1e6:10.2f}")
print(f"{'TOTAL':72s} {'':10s} {tf / steps / 1e6:14.2f} {tw / steps / 1e6:14.2f}")
if len(sys.argv) > 4:
    def group(pred):
        sel = [v for k, v in rows.items() if pred(k)]
        n = sum(v[0] for v in sel)
        return {"launches_per_step": round(n / steps, 2), "bytes_per_launch": int(sum(v[1] + v[2] for v in sel) / max(n, 1)),
                "fetch_MB_per_step": round(sum(v[1] for v in sel) / steps / 1e6, 2), "write_MB_per_step": round(sum(v[2] for v in sel) / steps / 1e6, 2)}
    json.dump({"attention": dict(kernel="attn_decode_kernel + attn_decode_group_kernel", **group(lambda k: k.startswith("attn_decode"))),
               "gemm": dict(kernel="gemm_bf16_kernel + gemm_w4_kernel + gemm_d2_kernel + gemm_phased_kernel", **group(lambda k: k.startswith("gemm_"))),
               "total_GB_per_decoder_step": round((tf + tw) / steps / 1e9, 3)}, open(sys.argv[4], "w"), indent=1)
