"""Print the kernel timeline of the last call in a rocprofv3 kernel-trace csv: name, start offset, duration (µs).
usage: trace_timeline.py <kernel_trace.csv> [number of kernels from the end]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = rows[-tail:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {r['Kernel_Name'][:110]}")
