#!/usr/bin/env python3
"""List the kernels a rocprofv3 --kernel-trace CSV holds between the first and the last launch whose name contains
<marker> (default: sumsq -- tests/parity_report.py `kernels` brackets the module under study with it).
usage: between_markers.py <kernel_trace.csv> [marker]"""
import collections
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2] if len(sys.argv) > 2 else "sumsq"
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
if len(idx) < 2:
    sys.exit(f"fewer than two '{marker}' launches in the trace")
agg = collections.OrderedDict()
for r in rows[idx[0] + 1:idx[-1]]:
    a = agg.setdefault(r["Kernel_Name"], [0, 0])
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print(f"{'calls':>6s} {'avg_us':>9s}  kernel")
for k, (n, t) in agg.items():
    print(f"{n:6d} {t / n / 1e3:9.1f}  {k[:200]}")
