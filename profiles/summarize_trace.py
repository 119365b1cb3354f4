#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel totals inside bench.py's TIMED region
(between the end of the first xai_ig_accum launch, i.e. the warm-up step's, and the last one).
usage: summarize_trace.py <kernel_trace.csv> [marker-substring [first-marker-index [last-marker-index]]]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "ig_accum"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
acc = [r for r in rows if marker in r["Kernel_Name"]]
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
last = int(sys.argv[4]) if len(sys.argv) > 4 else len(acc) - 1
t0, t1 = int(acc[first]["End_Timestamp"]), int(acc[last]["End_Timestamp"])
sel = [r for r in rows if t0 < int(r["Start_Timestamp"]) <= t1]


def short(n):
    if "elementwise_kernel" in n or "vectorized_elementwise" in n:
        m = re.search(r"(batch_norm\w+|direct_copy\w+|CUDAFunctor_\w+|launch_clamp\w+|threshold\w+|FillFunctor|BinaryFunctor|BUnaryFunctor)", n)
        return "torch-elementwise:" + (m.group(1) if m else n[40:90])
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return n[:70]


agg = collections.defaultdict(lambda: [0, 0])
for r in sel:
    a = agg[short(r["Kernel_Name"])]
    a[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a[1] += 1
tot = sum(v[0] for v in agg.values())
print(f"timed region: wall {(t1 - t0) / 1e6:.1f} ms, kernel-sum {tot / 1e6:.1f} ms, {len(sel)} launches, {last - first} step(s)")
print(f"{'kernel':72s} {'calls':>6s} {'total_ms':>9s} {'avg_us':>9s} {'%':>6s}")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
    print(f"{k:72s} {v[1]:6d} {v[0] / 1e6:9.2f} {v[0] / v[1] / 1e3:9.1f} {100 * v[0] / tot:6.2f}")
