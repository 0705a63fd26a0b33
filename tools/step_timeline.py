"""Timeline of ONE train step from a `rocprofv3 --kernel-trace` CSV: which streams were busy when, and what the tail of the step
consists of.  A step starts at an `im2col_kernel` dispatch (the first kernel of the forward pass).

    python tools/step_timeline.py <kernel_trace.csv> [step_index_from_the_end=1] [bin_ms=1.0]
"""
import csv
import re
import sys
from collections import defaultdict


def cls(name):
    n = re.sub(r"^void ", "", name)
    for k, v in (("comm_emulate", "COMM"), ("adamw", "adamw"), ("gemm_", "gemm"), ("attn_", "attn"), ("ln_", "ln"), ("ce_", "ce"),
                 ("embed", "embed"), ("sum_slabs", "gemm"), ("rccl", "COMM"), ("nccl", "COMM")):
        if k in n:
            return v
    return "other"


def main():
    path = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    binms = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if "im2col" in r[2]]
    if len(starts) < back + 1:
        sys.exit(f"need at least {back + 1} steps in the trace, found {len(starts)}")
    i0, i1 = starts[-back - 1], starts[-back]
    t0, t1 = rows[i0][0], rows[i1][0]
    step = [r for r in rows if t0 <= r[0] < t1]
    print(f"# {path}: step of {(t1 - t0) / 1e6:.3f} ms (start to next start), {len(step)} dispatches")
    per = defaultdict(lambda: [None, 0, 0, defaultdict(float)])
    for s, e, n, q, st in step:
        p = per[(q, st)]
        p[0] = s if p[0] is None else min(p[0], s)
        p[1] = max(p[1], e)
        p[2] += e - s
        p[3][cls(n)] += (e - s) / 1e6
    print(f"{'queue/stream':>14s} {'first_ms':>9s} {'last_ms':>9s} {'busy_ms':>9s}  classes (ms)")
    for k, (a, b, busy, c) in sorted(per.items(), key=lambda kv: kv[1][0]):
        print(f"{str(k[0]) + '/' + str(k[1]):>14s} {(a - t0) / 1e6:9.3f} {(b - t0) / 1e6:9.3f} {busy / 1e6:9.3f}  " +
              ", ".join(f"{x} {y:.2f}" for x, y in sorted(c.items(), key=lambda kv: -kv[1])))
    nb = int((t1 - t0) / 1e6 / binms) + 1
    bins = [defaultdict(float) for _ in range(nb)]
    for s, e, n, q, st in step:
        c = cls(n)
        b0, b1 = int((s - t0) / 1e6 / binms), int((e - t0) / 1e6 / binms)
        for b in range(b0, min(b1, nb - 1) + 1):
            lo, hi = max(s, t0 + b * binms * 1e6), min(e, t0 + (b + 1) * binms * 1e6)
            if hi > lo:
                bins[b][c] += (hi - lo) / 1e6
    print(f"\n# busy ms per {binms:g}-ms bin and kernel class (sum over streams; > bin width = overlap)")
    for b, d in enumerate(bins):
        print(f"{b * binms:7.1f}  " + "  ".join(f"{x}:{y:.2f}" for x, y in sorted(d.items())))
    # the main stream's idle time: gaps between consecutive kernels of the busiest queue, by size and by the kernel that ends the gap
    mainq = max(per.items(), key=lambda kv: kv[1][2])[0]
    ms_ = sorted((s, e, n) for s, e, n, q, st in step if (q, st) == mainq)
    gaps = [(ms_[i + 1][0] - max(x[1] for x in ms_[: i + 1][-8:]), ms_[i][2], ms_[i + 1][2], ms_[i + 1][0]) for i in range(len(ms_) - 1)]
    gaps = [g for g in gaps if g[0] > 0]
    tot = sum(g[0] for g in gaps)
    print(f"\n# main stream {mainq[0]}/{mainq[1]}: {len(ms_)} kernels, {tot / 1e6:.3f} ms idle between them: "
          f"{sum(1 for g in gaps if g[0] > 20000)} gaps > 20 us = {sum(g[0] for g in gaps if g[0] > 20000) / 1e6:.3f} ms, "
          f"{sum(1 for g in gaps if 5000 < g[0] <= 20000)} of 5-20 us = {sum(g[0] for g in gaps if 5000 < g[0] <= 20000) / 1e6:.3f} ms, "
          f"{sum(1 for g in gaps if g[0] <= 5000)} below 5 us = {sum(g[0] for g in gaps if g[0] <= 5000) / 1e6:.3f} ms")
    by_next = defaultdict(lambda: [0, 0])
    for g in gaps:
        k = re.sub(r"^void ", "", g[2])[:60]
        by_next[k][0] += 1
        by_next[k][1] += g[0]
    print("# idle in front of (kernel that ends the gap): count, total us, mean us")
    for k, (c, t) in sorted(by_next.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"  {c:4d} {t / 1e3:9.1f} {t / 1e3 / c:7.1f}  {k}")
    print("# the 12 largest gaps: us, at ms, after -> before")
    for g in sorted(gaps, key=lambda g: -g[0])[:12]:
        print(f"  {g[0] / 1e3:7.1f}  +{(g[3] - t0) / 1e6:7.3f}  {re.sub(r'^void ', '', g[1])[:44]} -> {re.sub(r'^void ', '', g[2])[:44]}")
    # the tail: everything that starts after the last GEMM of the step's main chain has ended
    last_gemm = max((e for s, e, n, q, st in step if cls(n) == "gemm"), default=t0)
    tail = [(s, e, n) for s, e, n, q, st in step if e > last_gemm]
    print(f"\n# after the last GEMM ended (+{(last_gemm - t0) / 1e6:.3f} ms): {len(tail)} kernels still running / starting")
    for s, e, n in tail[:40]:
        print(f"  +{(s - t0) / 1e6:8.3f} .. +{(e - t0) / 1e6:8.3f}  {re.sub(r'^void ', '', n)[:90]}")
    # the head: everything up to the first GEMM of the step (the glue the host issues in front of the forward pass)
    first_gemm = min((s for s, e, n, q, st in step if cls(n) == "gemm"), default=t1)
    head = [(s, e, n, q) for s, e, n, q, st in step if s <= first_gemm]
    print(f"\n# up to the first GEMM (+{(first_gemm - t0) / 1e6:.3f} ms): {len(head)} kernels")
    for s, e, n, q in head[:40]:
        print(f"  +{(s - t0) / 1e6:8.3f} .. +{(e - t0) / 1e6:8.3f}  q{q}  {re.sub(r'^void ', '', n)[:90]}")


if __name__ == "__main__":
    main()
