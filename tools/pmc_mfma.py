"""Per-kernel MFMA utilisation / LDS bank conflicts from a rocprofv3 --pmc pass of one train step.
usage: python tools/pmc_mfma.py <mfma_counter_collection.csv>
(round 2 printed two more columns: "L2 hit %" — its TCC pass never delivered counts on this stack, the column read 0.0 — and a
"clock GHz" derived from GRBM_GUI_ACTIVE over the dispatch duration, which is not physical for sub-30-us dispatches (the counter
includes ramp and drain around the timestamps); both are dropped)
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
        k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        k = (k[:k.index("(")] if "(" in k else k)[:58]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            cnt[k] += 1
            dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return agg, cnt, dur


a, c, d = load(sys.argv[1])
print(f"{'kernel':60s} {'calls':>6s} {'ms (profiled)':>13s} {'MFMA util %':>11s} {'LDS conflict %':>14s}")
for k in sorted(a, key=lambda k: -d[k]):
    v = a[k]
    if not any(s in k for s in ("gemm", "attn")):
        continue
    cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8
    util = 100 * v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024) if cyc else 0
    lds = 100 * v.get("SQ_LDS_BANK_CONFLICT", 0) / v["SQ_LDS_IDX_ACTIVE"] if v.get("SQ_LDS_IDX_ACTIVE") else 0
    print(f"{k:60s} {c[k]:6d} {d[k] / 1e6:13.3f} {util:11.1f} {lds:14.2f}")
