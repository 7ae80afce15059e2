"""Partition stage times per 100 M tuples at 14 radix bits as the relations grow (n + n uniform tuples; what the SIZE costs each pass —
100M x 1B runs pass 2 18 % slower a tuple than 100M + 100M):
python tools/exp_part_scale.py        (RHJ_MSD=1 python tools/exp_part_scale.py: pass 1 on the high bits)"""
import importlib, sys, json, ctypes as C
sys.path.insert(0, ".")
import bench, torch
mod = importlib.import_module("sigmod-2018_amd"); rhj = mod.RHJ(device=0)
for n in (50_000_000, 100_000_000, 200_000_000, 400_000_000, 550_000_000):
    w = dict(nR=n, nS=n, bits=14, dist="uniform")
    rhj.set_bits(14)
    R, S = bench.make_relations(w, rhj.dev, 7)
    out = torch.empty((n, 2), dtype=torch.int64, device=rhj.dev); m = C.c_uint64(0)
    keys = ("ms_hist", "ms_scan", "ms_scatter")
    acc = dict.fromkeys(keys, 0.0)
    for i in range(4):
        rhj.lib.rhj_join_device(R.data_ptr(), n, S.data_ptr(), n, out.data_ptr(), n, C.byref(m))
        if i >= 1:
            st = rhj.stats()
            for k in keys: acc[k] += st[k] / 3
    print(2 * n, json.dumps({k: round(v * 1e8 / (2 * n), 4) for k, v in acc.items()}), "ms per 100M tuples", flush=True)
    del R, S, out
    torch.cuda.empty_cache()
