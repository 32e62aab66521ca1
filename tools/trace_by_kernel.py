"""rocprofv3 --kernel-trace CSV -> average duration per (kernel, grid size).  usage: python tools/trace_by_kernel.py trace.csv [n]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 16
d = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"].replace("void sph3d::", "")[:48]
    g = r.get("Grid_Size") or r.get("Grid_Size_X")
    d[(n, g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:top]:
    print("%-50s grid %-9s n=%3d avg %8.1f us" % (k[0], k[1], len(v), sum(v) / len(v)))
