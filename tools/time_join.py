"""Stage times of rhj_join_device on a synthetic FK join WITHOUT checking the result (timing experiments with builds
whose results are wrong on purpose): python tools/time_join.py <workload> [steps]   (RHJ_LIB selects the build)"""
import importlib, sys, json, ctypes as C
sys.path.insert(0, ".")
import bench, torch
w = bench.WORKLOADS[sys.argv[1]]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
mod = importlib.import_module("sigmod-2018_amd"); rhj = mod.RHJ(device=0); rhj.set_bits(w["bits"])
R, S = bench.make_relations(w, rhj.dev, 1234)
out = torch.empty((w["nS"], 2), dtype=torch.int64, device=rhj.dev); m = C.c_uint64(0)
keys = ("ms_hist", "ms_scan", "ms_scatter", "ms_plan", "ms_build", "ms_offsets", "ms_probe", "ms_total")
acc = dict.fromkeys(keys, 0.0)
for i in range(steps + 2):
    rhj.lib.rhj_join_device(R.data_ptr(), w["nR"], S.data_ptr(), w["nS"], out.data_ptr(), w["nS"], C.byref(m))
    if i >= 2:
        st = rhj.stats()
        for k in keys: acc[k] += st[k] / steps
print(json.dumps({"path": rhj.stats()["path"], "matches": m.value, **{k: round(v, 4) for k, v in acc.items()}}))
