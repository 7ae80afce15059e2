"""A/B timing of two builds of librhj.so on ONE device: tools/ab.py <workload> <libA.so> <libB.so> [rounds]
Each build runs in its OWN process (alternating A B A B ...), because inside one process the build whose
workspace is allocated second measured 2-5 % slower than an identical copy loaded first (device memory
placement; `tools/ab.py c3 X.so copy-of-X.so` showed it) — enough to hide or fake the effects being tested.
Prints median / min of the per-stage GPU times of rhj_join_device over all runs of each build."""
import json, statistics, subprocess, sys

CHILD = r'''
import ctypes as C, importlib, json, os, sys, torch
sys.path.insert(0, ".")
import bench
mod = importlib.import_module("sigmod-2018_amd")
wl, path, reps = sys.argv[1], sys.argv[2], int(sys.argv[3])
w = bench.WORKLOADS[wl]
r = mod.RHJ(device=0, lib_path=path)
r.set_bits(w["bits"])
R, S = bench.make_relations(w, r.dev, 1234)
out = torch.empty((w["nS"], 2), dtype=torch.int64, device=r.dev)
m = C.c_uint64(0)
keys = ("ms_hist", "ms_scan", "ms_scatter", "ms_plan", "ms_probe", "ms_total")
acc = {k: [] for k in keys}
for i in range(reps + 2):
    rc = r.lib.rhj_join_device(R.data_ptr(), w["nR"], S.data_ptr(), w["nS"], out.data_ptr(), w["nS"], C.byref(m))
    assert rc == 0 or os.environ.get('AB_ANYRC'), rc        # timing experiments with wrong results: AB_ANYRC=1
    if i >= 2:                      # the first two runs grow the workspace
        st = r.stats()
        for k in keys:
            acc[k].append(st[k])
print("ABJSON " + json.dumps(acc))
'''

def main():
    wl, pa, pb = sys.argv[1:4]
    rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    keys = ("ms_hist", "ms_scan", "ms_scatter", "ms_plan", "ms_probe", "ms_total")
    acc = {"A": {k: [] for k in keys}, "B": {k: [] for k in keys}}
    for blk in range(rounds):
        for name, path in (("A", pa), ("B", pb)) if blk % 2 == 0 else (("B", pb), ("A", pa)):
            res = subprocess.run([sys.executable, "-c", CHILD, wl, path, "6"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            line = [l for l in res.stdout.decode().splitlines() if l.startswith("ABJSON ")]
            if not line:
                print(res.stderr.decode()[-2000:])
                sys.exit(1)
            d = json.loads(line[0][7:])
            for k in keys:
                acc[name][k] += d[k]
    for k in keys:
        a, b = acc["A"][k], acc["B"][k]
        print("%-11s A med %.3f min %.3f | B med %.3f min %.3f | B/A %.3f" % (k, statistics.median(a), min(a), statistics.median(b), min(b),
                                                                          statistics.median(b) / max(statistics.median(a), 1e-9)))


if __name__ == "__main__":
    main()
