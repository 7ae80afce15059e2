import importlib, ctypes as C, torch, sys, os
import numpy as np
sys.path.insert(0, ".")
import bench
mod = importlib.import_module("sigmod-2018_amd"); rhj = mod.RHJ(device=0)
w = dict(bench.WORKLOADS["c4"]); rhj.set_bits(14)
R, S = bench.make_relations(w, rhj.dev, 1234)
outs = []
for i in range(3):
    out, hist, psum = rhj.partition_device(S, 14)
    outs.append(hist.copy())
    print("call", i, "sum", int(hist.sum()), "zeros", int((hist == 0).sum()), "max", int(hist.max()))
    del out
d = np.nonzero(outs[0] != outs[1])[0]
print("bins differing call0 vs call1:", len(d), d[:20], outs[0][d[:10]], outs[1][d[:10]])
