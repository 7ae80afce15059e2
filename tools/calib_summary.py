#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE of tools/micro/calib's kernels (gpurun_out/calib_fetch, gpurun_out/calib_write: one rocprofv3 --pmc pass
each) against the bytes each pattern moves by construction -> gpurun_out/calibration.json:
  stream patterns: true_over_raw = known bytes / counter bytes (what a raw counter value of that pattern is multiplied by);
  gather patterns: raw_bytes_per_gather = counter bytes / gathers (what one random record costs on the memory side as the counter sees it).
python3 tools/calib_summary.py [git head]"""
import csv, glob, json, sys
known = {  # bytes read, bytes written (gathers: their number)
    "calib_stream_read16": (1600 << 20, 0), "calib_stream_read12": (1_200_000_000, 0), "calib_stream_read8of12": (800_000_000, 0),
    "calib_stream_read8": (1600 << 20, 0), "calib_stream_write16": (0, 1600 << 20), "calib_stream_write12": (0, 1_200_000_000),
    "calib_gather<12>": (256 * 96 * 1024 * 4, 0), "calib_gather<4>": (256 * 96 * 1024 * 4, 0)}
raw = {}
for d, ctr in (("gpurun_out/calib_fetch", "FETCH_SIZE"), ("gpurun_out/calib_write", "WRITE_SIZE")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != ctr:
                continue
            k = r["Kernel_Name"].replace("void ", "").split("(")[0]
            raw.setdefault(k, {})[ctr] = float(r["Counter_Value"]) * 1024.0          # KB, last dispatch wins
            raw[k]["ns_" + ctr] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
out = {"_unit": "bytes per dispatch (counter values are KB x 1024)", "_collected_at_git_head": sys.argv[1] if len(sys.argv) > 1 else ""}
for k, (rd, wr) in known.items():
    v = raw.get(k, {})
    e = {"raw_FETCH_SIZE": v.get("FETCH_SIZE"), "raw_WRITE_SIZE": v.get("WRITE_SIZE"), "ns": v.get("ns_FETCH_SIZE")}
    if "gather" in k:
        e["gathers"] = rd
        if v.get("FETCH_SIZE"):
            e["raw_fetch_bytes_per_gather"] = round(v["FETCH_SIZE"] / rd, 2)
    else:
        e["known_read"], e["known_written"] = rd, wr
        if rd and v.get("FETCH_SIZE"):
            e["fetch_true_over_raw"] = round(rd / v["FETCH_SIZE"], 4)
        if wr and v.get("WRITE_SIZE"):
            e["write_true_over_raw"] = round(wr / v["WRITE_SIZE"], 4)
    out[k] = e
json.dump(out, open("gpurun_out/calibration.json", "w"), indent=1)
for k, v in out.items():
    print(k, v)
