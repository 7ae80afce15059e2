"""In-kernel phase stamps of k_join_spec by unit kind (diagnostics build: make -C sigmod-2018_amd instr): python tools/exp_spec_stamps.py <nR> <nS> [bits]"""
import importlib, ctypes as C, torch, sys, os
import numpy as np
sys.path.insert(0, ".")
import bench
os.environ["RHJ_STAMPS"] = "1"
os.environ.setdefault("RHJ_LIB", os.path.join("sigmod-2018_amd", "librhj_instr.so"))
mod = importlib.import_module("sigmod-2018_amd"); rhj = mod.RHJ(device=0)
nR, nS = [int(x) for x in sys.argv[1:3]]
bits = int(sys.argv[3]) if len(sys.argv) > 3 else 12
rhj.set_bits(bits)
R, S = bench.make_relations(dict(nR=nR, nS=nS, bits=bits, dist="uniform"), rhj.dev, 1234)
cap = max(nR, nS)
out = torch.empty((cap, 2), dtype=torch.int64, device=rhj.dev)
m = C.c_uint64(0)
for i in range(3):
    rhj.lib.rhj_join_device(R.data_ptr(), nR, S.data_ptr(), nS, out.data_ptr(), cap, C.byref(m))
st = rhj.stats()
units = st["units"]
print("probe stage %.3f ms, units %d, last_spec %d" % (st["ms_probe"], units, rhj.lib.rhj_last_spec()))
buf = np.zeros((units, 8), dtype=np.uint64)
rhj.lib.rhj_debug_stamps.argtypes = [C.c_void_p, C.c_uint64]
assert rhj.lib.rhj_debug_stamps(buf.ctypes.data_as(C.c_void_p), units) == 0
t = buf.astype(np.int64)
us = lambda a: a / 100.0
for name, kind in (("the hypothesis' relation probes", 1), ("the other relation probes", 2)):
    x = t[t[:, 7] == kind]
    print(name, len(x), "units")
    for nm, a, b in (("build", 0, 1), ("phase 1 (wave 0 through)", 1, 2), ("all waves through + checks", 2, 3), ("unit", 0, 3)):
        d = us(x[:, b] - x[:, a])
        print("  %-28s mean %.1f  p50 %.1f  p90 %.1f us" % (nm, d.mean(), np.median(d), np.percentile(d, 90)))
print("span %.1f us" % us(t[:, 3].max() - t[:, 0].min()))
