import importlib, ctypes as C, torch, sys, os
import numpy as np
sys.path.insert(0, ".")
import bench
os.environ["RHJ_STAMPS"] = "1"
# needs the diagnostics build: make -C sigmod-2018_amd instr
os.environ.setdefault("RHJ_LIB", os.path.join("sigmod-2018_amd", "librhj_instr.so"))
mod = importlib.import_module("sigmod-2018_amd"); rhj = mod.RHJ(device=0)
nR, nS = [int(x) for x in sys.argv[1:3]]
bits = int(sys.argv[3]) if len(sys.argv) > 3 else 12
w = dict(nR=nR, nS=nS, bits=bits, dist="uniform")
rhj.set_bits(bits)
if len(sys.argv) > 4 and sys.argv[4] == "small":      # the join of `small` with these input sizes (recorded inputs)
    sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
    import helpers
    g = helpers.Golden()
    j = [j for j in g.small["joins"] if len(g.small_join(j["idx"])[0]) == nR and len(g.small_join(j["idx"])[1]) == nS][0]
    Rh, Sh = g.small_join(j["idx"])
    R, S = rhj.to_device(Rh), rhj.to_device(Sh)
else:
    R, S = bench.make_relations(w, rhj.dev, 1234)
cap = 20000000 if len(sys.argv) > 4 else max(nR, nS)
out = torch.empty((cap, 2), dtype=torch.int64, device=rhj.dev)
m = C.c_uint64(0)
for i in range(3):
    rhj.lib.rhj_join_device(R.data_ptr(), nR, S.data_ptr(), nS, out.data_ptr(), cap, C.byref(m))
st = rhj.stats()
units = st["units"]
buf = np.zeros((units, 8), dtype=np.uint64)
rhj.lib.rhj_debug_stamps.argtypes = [C.c_void_p, C.c_uint64]
assert rhj.lib.rhj_debug_stamps(buf.ctypes.data_as(C.c_void_p), units) == 0
t = buf.astype(np.int64)
t0 = t[:, 0].min()
us = lambda a: a / 100.0
print("fused %.3f ms, units %d" % (st["ms_probe"], units))
print("span (first start -> last end): %.1f us" % us(t[:, 4].max() - t0))
for name, a, b in (("build", 0, 1), ("phase1(w0)", 1, 2), ("chain+barrier", 2, 3), ("phase2", 3, 4), ("unit total", 0, 4)):
    d = us(t[:, b] - t[:, a])
    print("%-28s mean %.1f  p50 %.1f  p90 %.1f  max %.1f us" % (name, d.mean(), np.median(d), np.percentile(d, 90), d.max()))
ok = t[:, 5] > 0
d = us(t[ok, 6] - t[ok, 5]) if ok.any() else np.zeros(1); print("%-28s mean %.1f  p50 %.1f  p90 %.1f  max %.1f us   (thread 0's own emit span)" % ("deferred emit (w0)", d.mean(), np.median(d), np.percentile(d, 90), d.max()))
ok7 = t[:, 7] > 0
d = us(t[ok7, 7] - t[ok7, 2]) if ok7.any() else np.zeros(1); print("%-28s mean %.1f  p50 %.1f  p90 %.1f  max %.1f us   (w0 after phase 1 -> all waves through the previous unit's emit)" % ("ph1 skew + prev emit", d.mean(), np.median(d), np.percentile(d, 90), d.max()))
starts = np.sort(us(t[:, 0] - t0))
print("start times of units 0,255,256,511,1024,last: ", [round(float(starts[min(i, units - 1)]), 1) for i in (0, 255, 256, 511, 1024, units - 1)])
if units <= 256:
    ends = us(t[:, 6].max() - t0) if (t[:, 6] > 0).any() else 0
    print("last emit end %.1f us; per-stamp medians from kernel start: %s" % (ends, [round(float(np.median(us(t[:, k] - t0))), 1) for k in range(7)]))
