"""Stage times of the 100M x 1B join at 14 radix bits on uniform against Zipf(0.9) foreign keys (what the skew costs each partition pass):
python tools/exp_c4_skew.py"""
import importlib, sys, json, ctypes as C
sys.path.insert(0, ".")
import bench, torch
mod = importlib.import_module("sigmod-2018_amd"); rhj = mod.RHJ(device=0)
for dist in ("uniform", "zipf"):
    w = dict(nR=100_000_000, nS=1_000_000_000, bits=14, dist=dist)
    rhj.set_bits(14)
    R, S = bench.make_relations(w, rhj.dev, 7)
    out = torch.empty((w["nS"], 2), dtype=torch.int64, device=rhj.dev); m = C.c_uint64(0)
    keys = ("ms_hist", "ms_scan", "ms_scatter", "ms_plan", "ms_probe", "ms_total")
    acc = dict.fromkeys(keys, 0.0)
    for i in range(5):
        rhj.lib.rhj_join_device(R.data_ptr(), w["nR"], S.data_ptr(), w["nS"], out.data_ptr(), w["nS"], C.byref(m))
        if i >= 2:
            st = rhj.stats()
            for k in keys: acc[k] += st[k] / 3
    print(dist, json.dumps({k: round(v, 3) for k, v in acc.items()}), flush=True)
    del R, S, out
    torch.cuda.empty_cache()
