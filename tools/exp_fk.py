import importlib, ctypes as C, torch, sys
sys.path.insert(0, ".")
import bench
mod = importlib.import_module("sigmod-2018_amd"); rhj = mod.RHJ(device=0)
for nR, nS in ((100_000_000, 100_000_000), (80_000_000, 120_000_000), (120_000_000, 80_000_000)):
    w = dict(nR=nR, nS=nS, bits=12, dist="uniform")
    rhj.set_bits(12)
    R, S = bench.make_relations(w, rhj.dev, 1234)
    out = torch.empty((nS, 2), dtype=torch.int64, device=rhj.dev)
    m = C.c_uint64(0)
    for i in range(3):
        rhj.lib.rhj_join_device(R.data_ptr(), nR, S.data_ptr(), nS, out.data_ptr(), nS, C.byref(m))
    st = rhj.stats()
    print("nR=%d nS=%d: fused %.3f ms  scatter %.3f  total %.3f matches %d units %d" % (nR, nS, st["ms_probe"], st["ms_scatter"], st["ms_total"], m.value, st["units"]))
    del R, S, out
