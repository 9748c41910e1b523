"""Sums rocprofv3 --pmc counter CSVs per (kernel, number of workgroups) and divides by the number of dispatches.
usage: python tools/summarize_pmc.py gpurun_out/prof_<tag> [commit] > profiles/<name>.json
The directory holds one sub-directory per counter pass and run: pmc<N>* for the bench's alignment legs,
pmcreads<N>* / pmcgenome<N>* for the two minimizer legs, which run in processes of their own - so the output has
the sections "bench", "minimizers_reads" and "minimizers_genome", and a launch on the 4.6 Mbp genome is never
averaged with one on the reads.  Inside a section an entry is keyed "kernel @ N blocks".  Kernel names keep their
template arguments and lose the argument list and the anonymous namespace: "sa::fill_wave_pk16<2, 16, false>"."""
import csv, glob, json, os, sys


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


def section_of(path, root):
    top = os.path.relpath(path, root).split(os.sep)[0]
    if top.startswith("pmcreads"):
        return "minimizers_reads"
    if top.startswith("pmcgenome"):
        return "minimizers_genome"
    return "bench"


root = sys.argv[1]
out = {"_meta": {"commit": sys.argv[2] if len(sys.argv) > 2 else None,
                 "units": "FETCH_SIZE / WRITE_SIZE in KiB per dispatch as rocprofv3 reports them (FETCH_SIZE counts "
                          "half the bytes of wide reads on gfx950); SQ_* per dispatch"}}
for f in sorted(glob.glob(root + "/pmc*/**/*counter_collection.csv", recursive=True)):
    sec = out.setdefault(section_of(f, root), {})
    per = {}
    for r in csv.DictReader(open(f)):
        name = kernel_name(r["Kernel_Name"])
        if not name.startswith("sa::"):
            continue
        blocks = int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)
        k = per.setdefault((name, blocks), {})
        c = k.setdefault(r["Counter_Name"], {"sum": 0.0, "ids": set()})
        c["sum"] += float(r["Counter_Value"])
        c["ids"].add(r["Dispatch_Id"])
    for (name, blocks), cs in per.items():
        o = sec.setdefault("%s @ %d blocks" % (name, blocks), {"blocks": blocks})
        for cn, c in cs.items():
            o["dispatches"] = len(c["ids"])
            o[cn] = c["sum"] / len(c["ids"])
print(json.dumps(out, indent=1, sort_keys=True))
