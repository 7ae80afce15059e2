"""C4 (100M x 1B Zipf 0.9) at several radix widths"""
import importlib, ctypes as C, torch, sys, json
sys.path.insert(0, ".")
import bench
mod = importlib.import_module("sigmod-2018_amd"); rhj = mod.RHJ(device=0)
w = dict(bench.WORKLOADS["c4"])
R, S = bench.make_relations(w, rhj.dev, 1234)
out = torch.empty((w["nS"], 2), dtype=torch.int64, device=rhj.dev)
m = C.c_uint64(0)
for bits in [int(x) for x in sys.argv[1:]]:
    rhj.set_bits(bits)
    for i in range(3):
        rhj.lib.rhj_join_device(R.data_ptr(), w["nR"], S.data_ptr(), w["nS"], out.data_ptr(), w["nS"], C.byref(m))
    st = rhj.stats()
    print(json.dumps({"bits": bits, "total_ms": round(st["ms_total"], 2), "passA": round(st["ms_hist"], 2), "hist_scan": round(st["ms_scan"], 2),
                      "passB": round(st["ms_scatter"], 2), "join": round(st["ms_probe"] + st["ms_build"] + st["ms_count"] + st["ms_offsets"], 2),
                      "matches": m.value, "units": st["units"], "max_build": st["max_build"], "Gt/s": round(w["nS"] / st["ms_total"] / 1e6, 1)}))
