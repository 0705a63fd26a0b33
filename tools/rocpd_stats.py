"""Per-kernel summary (count, total, average, share) from a rocprofv3 rocpd SQLite database or kernel-trace CSV."""
import csv
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*", "", name)
    return name[:100]


def from_db(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type='table' or type='view'")]
    disp = [t for t in tables if "kernel_dispatch" in t][0]
    cols = [r[1] for r in cur.execute(f"pragma table_info('{disp}')")]
    sym = [t for t in tables if "kernel_symbol" in t][0]
    scols = [r[1] for r in cur.execute(f"pragma table_info('{sym}')")]
    name_col = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else "name")
    q = f"select s.{name_col}, d.start, d.end from '{disp}' d join '{sym}' s on d.kernel_id = s.id"
    return [(n, e - s) for n, s, e in cur.execute(q)]


def from_csv(path):
    out = []
    with open(path) as f:
        for r in csv.DictReader(f):
            out.append((r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return out


def main():
    path = sys.argv[1]
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    rows = from_db(path) if path.endswith(".db") else from_csv(path)
    agg = defaultdict(lambda: [0, 0])
    for n, d in rows:
        a = agg[short(n)]
        a[0] += 1
        a[1] += d
    tot = sum(a[1] for a in agg.values())
    print(f"# {path}: {len(rows)} dispatches, total kernel time {tot / 1e6:.3f} ms over {steps:g} steps ({tot / 1e6 / steps:.3f} ms/step)")
    print(f"{'kernel':100s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'share':>7s} {'ms/step':>9s}")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:100s} {c:7d} {t / 1e6:10.3f} {t / c / 1e3:10.2f} {100 * t / tot:6.1f}% {t / 1e6 / steps:9.3f}")


if __name__ == "__main__":
    main()
