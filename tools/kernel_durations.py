"""Per-call durations of the kernels whose name contains <substring>, from a rocprofv3 --kernel-trace CSV."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for sub in sys.argv[2:]:
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if sub in r["Kernel_Name"]]
    print(sub, len(d))
    print("  ", " ".join(f"{x:.0f}" for x in d))
