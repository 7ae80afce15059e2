"""Interleaved A/B timing of two builds of librhj.so in ONE process on ONE device
(guide rule 24): tools/ab.py <workload> <libA.so> <libB.so> [rounds]
Prints median / min of the per-stage GPU times of rhj_join_device."""
import ctypes as C, importlib, statistics, sys
import torch
sys.path.insert(0, ".")
import bench
mod = importlib.import_module("sigmod-2018_amd")
wl, pa, pb = sys.argv[1:4]
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 7
w = bench.WORKLOADS[wl]
libs = {"A": mod.RHJ(device=0, lib_path=pa), "B": mod.RHJ(device=0, lib_path=pb)}
for r in libs.values():
    r.set_bits(w["bits"])
R, S = bench.make_relations(w, libs["A"].dev, 1234)
out = torch.empty((w["nS"], 2), dtype=torch.int64, device=libs["A"].dev)
m = C.c_uint64(0)
keys = ("ms_hist", "ms_scan", "ms_scatter", "ms_plan", "ms_probe", "ms_total")
acc = {n: {k: [] for k in keys} for n in libs}
# blocks AAAA BBBB ...: the first run of a block (which inherits the other build's cache state) is dropped
for blk in range(rounds):
    for n, r in libs.items():
        for i in range(4):
            rc = r.lib.rhj_join_device(R.data_ptr(), w["nR"], S.data_ptr(), w["nS"], out.data_ptr(), w["nS"], C.byref(m))
            assert rc == 0
            if i:
                st = r.stats()
                for k in keys:
                    acc[n][k].append(st[k])
for k in keys:
    a, b = acc["A"][k], acc["B"][k]
    print("%-11s A med %.3f min %.3f | B med %.3f min %.3f | B/A %.3f" % (k, statistics.median(a), min(a), statistics.median(b), min(b),
                                                                      statistics.median(b) / max(statistics.median(a), 1e-9)))
