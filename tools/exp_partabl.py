import importlib, ctypes as C, torch, sys, os
sys.path.insert(0, ".")
import bench
mod = importlib.import_module("sigmod-2018_amd"); rhj = mod.RHJ(device=0)
w = bench.WORKLOADS["c3"]; rhj.set_bits(w["bits"])
R, S = bench.make_relations(w, rhj.dev, 1234)
out = torch.empty((w["nS"], 2), dtype=torch.int64, device=rhj.dev)
m = C.c_uint64(0)
for i in range(4):
    rhj.lib.rhj_join_device(R.data_ptr(), w["nR"], S.data_ptr(), w["nS"], out.data_ptr(), w["nS"], C.byref(m))
st = rhj.stats()
print("ABL=%s: passA %.3f hist/scan %.3f passB %.3f fused %.3f" % (os.environ.get("RHJ_ABLATE", "0"), st["ms_hist"], st["ms_scan"], st["ms_scatter"], st["ms_probe"]))
