"""Per-kernel MFMA utilisation / LDS bank conflicts / L2 hit rate from two rocprofv3 --pmc passes of one train step.
usage: python tools/pmc_mfma.py <mfma_counter_collection.csv> <tcc_counter_collection.csv>
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs), cycles = GRBM_GUI_ACTIVE / 8 (the counter is summed over
the 8 XCDs; calibrated on the LM-head dE GEMM: 41.2 M v_mfma_f32_32x32x16_bf16 x 32 cycles = 1.32e9 vs 1.34e9 counted).
Kernels run ~20-30 % slower under counter collection (serialised dispatches), so utilisation here is a lower bound."""
import collections
import csv
import sys


def load(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt, dur, seen = collections.Counter(), collections.defaultdict(float), set()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        k = (k[:k.index("(")] if "(" in k else k).replace("void ", "")[:58]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            cnt[k] += 1
            dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return agg, cnt, dur


a, c, d = load(sys.argv[1])
t, _, _ = load(sys.argv[2])
print(f"{'kernel':60s} {'calls':>6s} {'ms (profiled)':>13s} {'clock GHz':>9s} {'MFMA util %':>11s} {'LDS conflict %':>14s} {'L2 hit %':>8s}")
for k in sorted(a, key=lambda k: -d[k]):
    v = a[k]
    if not any(s in k for s in ("gemm", "attn")):
        continue
    cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8
    util = 100 * v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024) if cyc else 0
    lds = 100 * v.get("SQ_LDS_BANK_CONFLICT", 0) / v["SQ_LDS_IDX_ACTIVE"] if v.get("SQ_LDS_IDX_ACTIVE") else 0
    tt = t.get(k, {})
    hit = 100 * tt.get("TCC_HIT_sum", 0) / max(1.0, tt.get("TCC_HIT_sum", 0) + tt.get("TCC_MISS_sum", 0))
    print(f"{k:60s} {c[k]:6d} {d[k] / 1e6:13.3f} {cyc / max(d[k], 1):9.2f} {util:11.1f} {lds:14.2f} {hit:8.1f}")
