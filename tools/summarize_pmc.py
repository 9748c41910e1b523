"""Sums rocprofv3 --pmc counter CSVs per kernel and divides by the number of dispatches.
usage: python tools/summarize_pmc.py gpurun_out/prof_<tag> > profiles/<name>.json"""
import csv, glob, json, re, sys
root = sys.argv[1]
out = {}
for f in sorted(glob.glob(root + "/pmc*/**/*counter_collection.csv", recursive=True)):
    per = {}
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
        if not name.startswith("sa::"):
            continue
        k = per.setdefault(name, {})
        c = k.setdefault(r["Counter_Name"], {"sum": 0.0, "ids": set()})
        c["sum"] += float(r["Counter_Value"])
        c["ids"].add(r["Dispatch_Id"])
    for name, cs in per.items():
        o = out.setdefault(name, {})
        for cn, c in cs.items():
            o["dispatches"] = len(c["ids"])
            o[cn] = c["sum"] / len(c["ids"])
print(json.dumps(out, indent=1))
