"""Sums rocprofv3 --pmc counter CSVs per kernel and divides by the number of dispatches.
usage: python tools/summarize_pmc.py gpurun_out/prof_<tag> > profiles/<name>.json
Kernel names keep their template arguments and lose the argument list and the anonymous namespace:
"sa::fill_wave_pk16<2>", "sa::min_run<sa::Machine<5u, false> >"."""
import csv, glob, json, sys


def kernel_name(raw):
    name = raw.replace("void ", "").replace("(anonymous namespace)::", "").strip()
    if name.endswith(")"):      # cut the argument list: the parenthesis group that closes the name
        depth = 0
        for i in range(len(name) - 1, -1, -1):
            depth += name[i] == ")"
            depth -= name[i] == "("
            if depth == 0:
                name = name[:i]
                break
    return name.strip()


root = sys.argv[1]
out = {}
for f in sorted(glob.glob(root + "/pmc*/**/*counter_collection.csv", recursive=True)):
    per = {}
    for r in csv.DictReader(open(f)):
        name = kernel_name(r["Kernel_Name"])
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
print(json.dumps(out, indent=1, sort_keys=True))
