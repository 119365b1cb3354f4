"""Print the launches around each occurrence of a kernel in a rocprofv3 kernel-trace CSV.
usage: around_kernel.py <kernel_trace.csv> <substring> [max_occurrences]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
pat, lim = sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 6
hits = [i for i, r in enumerate(rows) if pat in r["Kernel_Name"]]
print(len(hits), "occurrences")
for i in hits[:lim]:
    for j in range(max(0, i - 4), min(len(rows), i + 3)):
        r = rows[j]
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        print(("  >> " if j == i else "     ") + f"{d:9.1f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>9} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?')):>5}  {r['Kernel_Name'][:110]}")
    print()
