"""Per-dispatch durations from a rocprofv3 --kernel-trace CSV, grouped by kernel name (microseconds).
usage: python tools/kernel_times.py <trace.csv> [substring ...]"""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
want = sys.argv[2:]
d = defaultdict(list)
for r in rows:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    if want and not any(w in n for w in want):
        continue
    d[n[:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in sorted(d.items(), key=lambda x: -sum(x[1])):
    print("%-70s n=%-3d" % (n, len(v)), " ".join("%.1f" % x for x in v[:10]))
