# per-workgroup start/end of k_small_hist and k_small_scatter (diagnostics build: make -C sigmod-2018_amd instr)
import importlib, ctypes as C, torch, sys, os
import numpy as np
sys.path.insert(0, ".")
import bench
os.environ.setdefault("RHJ_LIB", os.path.join("sigmod-2018_amd", "librhj_instr.so"))
mod = importlib.import_module("sigmod-2018_amd"); rhj = mod.RHJ(device=0)
nR, nS = [int(x) for x in sys.argv[1:3]]
bits = int(sys.argv[3]) if len(sys.argv) > 3 else 8
w = dict(nR=nR, nS=nS, bits=bits, dist="uniform")
rhj.set_bits(bits)
R, S = bench.make_relations(w, rhj.dev, 1234)
cap = max(nR, nS)
out = torch.empty((cap, 2), dtype=torch.int64, device=rhj.dev)
m = C.c_uint64(0)
for i in range(5):
    rhj.lib.rhj_join_device(R.data_ptr(), nR, S.data_ptr(), nS, out.data_ptr(), cap, C.byref(m))
st = rhj.stats()
print(st["path"], "hist %.1f scatter %.1f join %.1f us" % (st["ms_hist"] * 1e3, st["ms_scatter"] * 1e3, st["ms_probe"] * 1e3))
buf = np.zeros(4 * 2048, dtype=np.uint64)
rhj.lib.rhj_debug_small_stamps.argtypes = [C.c_void_p]
assert rhj.lib.rhj_debug_small_stamps(buf.ctypes.data_as(C.c_void_p)) == 0
t = buf.astype(np.int64)[:4096].reshape(2, 1024, 2)
tiles = [(nR + 8191) // 8192, (nS + 8191) // 8192]
mt = max(tiles)
for kern, name, gx in ((0, "k_small_hist", mt), (1, "k_small_scatter", mt + 1)):
    a = t[kern, :2 * gx]
    live = a[:, 1] > 0
    t0 = a[live, 0].min()
    s, e = (a[live, 0] - t0) / 100.0, (a[live, 1] - t0) / 100.0
    print("%-16s workgroups %d: start p50 %.1f max %.1f; end p50 %.1f p90 %.1f max %.1f; duration p50 %.1f max %.1f us" %
          (name, live.sum(), np.median(s), s.max(), np.median(e), np.percentile(e, 90), e.max(), np.median(e - s), (e - s).max()))
    if kern == 1:
        p = a[gx - 1]
        print("   plan workgroup: start %.1f end %.1f us" % ((p[0] - t0) / 100.0, (p[1] - t0) / 100.0))
f = buf.astype(np.int64)[4096:4096 + 1024].reshape(128, 8)
f = f[:min(tiles[0], 128)]
names = ["tile loads issued .. column sums + bucket starts", "clear wave counters", "ranks (8 rounds of match-any + LDS counters)", "wave prefixes + digit starts",
         "stage into LDS", "write out (issue)"]
for i, nm in enumerate(names):
    d = (f[:, i + 1] - f[:, i]) / 100.0
    print("   %-55s p50 %.2f max %.2f us" % (nm, np.median(d), d.max()))
