#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs under gpurun_out/pmc_<tag>_*: per rhj kernel, the counter
values of the LAST dispatch of that kernel (steady state) plus its duration."""
import csv, glob, json, sys, collections
tag = sys.argv[1]
out = collections.defaultdict(dict)
for d in sorted(glob.glob("gpurun_out/pmc_%s_*/" % tag)):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        last = {}
        for r in csv.DictReader(open(f)):
            if "rhj::" not in r["Kernel_Name"]:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            last[(k, r["Counter_Name"])] = (float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for (k, c), (v, ns) in last.items():
            out[k][c] = v
            out[k]["ns"] = ns
if len(sys.argv) > 2:
    out["_collected_at_git_head"] = sys.argv[2]
json.dump(out, open("gpurun_out/pmc_%s.json" % tag, "w"), indent=1)
for k, v in out.items():
    if not isinstance(v, dict):
        continue
    print(k, {a: (round(b, 1) if isinstance(b, float) else b) for a, b in v.items()})
