import importlib, sys, json, ctypes as C
sys.path.insert(0, ".")
import bench, torch
mod = importlib.import_module("sigmod-2018_amd"); rhj = mod.RHJ(device=0)
for n, bl in ((10_000_000, (9, 10, 11)), (30_000_000, (10, 11, 12)), (50_000_000, (11, 12, 13))):
    for bits in bl:
        w = dict(nR=n, nS=n, bits=bits, dist="uniform")
        rhj.set_bits(bits)
        R, S = bench.make_relations(w, rhj.dev, 1234)
        out = torch.empty((n, 2), dtype=torch.int64, device=rhj.dev); m = C.c_uint64(0)
        tot = 0.0
        for i in range(7):
            rhj.lib.rhj_join_device(R.data_ptr(), n, S.data_ptr(), n, out.data_ptr(), n, C.byref(m))
            if i >= 2: tot += rhj.stats()["ms_total"] / 5
        print(n, bits, round(tot, 4), rhj.lib.rhj_last_spec(), flush=True)
