# the slowest of the 88 joins of `small`: stage times and key multiplicities
import importlib, ctypes as C, torch, sys, os, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import helpers
mod = importlib.import_module("sigmod-2018_amd"); rhj = mod.RHJ(device=0)
g = helpers.Golden()
rhj.set_bits(4)
want = [(9013, 43131), (111, 3754), (25325, 30780), (16543, 26808), (270137, 43131), (311, 43131), (14925, 17296)]
for j in g.small["joins"]:
    R, S = g.small_join(j["idx"])
    if (len(R), len(S)) not in want:
        continue
    want.remove((len(R), len(S)))
    dR, dS = rhj.to_device(R), rhj.to_device(S)
    cap = j["matches"] + 16
    out = torch.empty((cap, 2), dtype=torch.int64, device=rhj.dev)
    m = C.c_uint64(0)
    for i in range(4):
        rhj.lib.rhj_join_device(dR.data_ptr(), len(R), dS.data_ptr(), len(S), out.data_ptr(), cap, C.byref(m))
    st = rhj.stats()
    def mult(x):
        v, c = np.unique(x["value"], return_counts=True)
        return len(v), int(c.max()), float(np.percentile(c, 99)), int((c > 16).sum())
    print("%7d x %7d -> %8d: hist %.3f scatter %.3f join %.3f total %.3f ms (%s, %d units)" % (len(R), len(S), m.value, st["ms_hist"], st["ms_scatter"], st["ms_probe"], st["ms_total"], st["path"], st["units"]))
    print("      R: %d distinct keys, max multiplicity %d, p99 %.0f, keys above 16: %d;   S: %d distinct, max %d, p99 %.0f, above 16: %d" % (mult(R) + mult(S)))
    b = (R["value"] & np.uint64(15)).astype(np.int64); bs = (S["value"] & np.uint64(15)).astype(np.int64)
    print("      bucket sizes R %s" % np.bincount(b, minlength=16).tolist())
    print("      bucket sizes S %s" % np.bincount(bs, minlength=16).tolist())
