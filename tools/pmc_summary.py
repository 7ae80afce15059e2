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
sys.path.insert(0, ".")
try:
    import bench
    out["_source_fingerprint"] = bench.source_fingerprint()
except Exception:
    pass
# calibrated reading (profiles/r04_calibration.json, tools/micro/calib.hip): streaming reads of 8 / 12 / 16 B a lane count exactly half in
# FETCH_SIZE, stores exactly; kernels that only stream (the partition passes, the filter) are corrected here, the join kernels —
# streams and 64-byte gather requests in one counter — in bench.py, which knows how many tuples they stream
for k, v in out.items():
    if isinstance(v, dict) and "FETCH_SIZE" in v and not any(x in k for x in ("k_join", "k_probe", "k_build", "k_lr_")):
        v["FETCH_SIZE_corrected_x2"] = v["FETCH_SIZE"] * 2.0
json.dump(out, open("gpurun_out/pmc_%s.json" % tag, "w"), indent=1)
for k, v in out.items():
    if not isinstance(v, dict):
        continue
    print(k, {a: (round(b, 1) if isinstance(b, float) else b) for a, b in v.items()})
