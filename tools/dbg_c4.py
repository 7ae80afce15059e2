import importlib, ctypes as C, torch, sys, os
sys.path.insert(0, ".")
import bench
mod = importlib.import_module("sigmod-2018_amd"); rhj = mod.RHJ(device=0)
wl = sys.argv[1] if len(sys.argv) > 1 else "c4"
w = dict(bench.WORKLOADS[wl]); 
if len(sys.argv) > 2: w["nS"] = int(sys.argv[2])
rhj.set_bits(w["bits"])
R, S = bench.make_relations(w, rhj.dev, 1234)
m = C.c_uint64(0)
for i in range(4):
    rc = rhj.lib.rhj_join_device(R.data_ptr(), w["nR"], S.data_ptr(), w["nS"], None, 0, C.byref(m))
    st = rhj.stats()
    print("call", i, "rc", rc, "matches", m.value, "units", st["units"], "max_build", st["max_build"], "probe_ms", round(st["ms_probe"], 2))
