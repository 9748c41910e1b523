"""Per-dispatch durations from a rocprofv3 --kernel-trace output (the CSV or the .db), grouped by kernel name,
in dispatch order of first appearance (microseconds).
usage: python tools/kernel_times.py <trace.csv | results.db> [substring ...]"""
import csv, sqlite3, sys
from collections import defaultdict
path, want = sys.argv[1], sys.argv[2:]
if path.endswith(".db"):
    con = sqlite3.connect(path)
    rows = [(n, s, e, gx // max(wx, 1)) for n, s, e, gx, wx in
            con.execute("select name, start, end, grid_x, workgroup_x from kernels order by start")]
else:
    rows = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
             int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)) for r in csv.DictReader(open(path))]
    rows.sort(key=lambda r: r[1])
d = defaultdict(list)
for n, s, e, blocks in rows:
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    if want and not any(w in n for w in want):
        continue
    d[(n[:70], blocks)].append((e - s) / 1e3)
for (n, blocks), v in d.items():
    print("%-70s blocks=%-7d n=%-3d" % (n, blocks, len(v)), " ".join("%.1f" % x for x in v[:8]))
if rows:
    print("span of all dispatches: %.1f us" % ((rows[-1][2] - rows[0][1]) / 1e3))
