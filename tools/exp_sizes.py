import importlib, ctypes as C, torch, sys, json
sys.path.insert(0, ".")
import bench
mod = importlib.import_module("sigmod-2018_amd"); rhj = mod.RHJ(device=0)
for n, bits in ((10_000_000, 8), (10_000_000, 10), (10_000_000, 12), (30_000_000, 10), (1_000_000, 4), (300_000, 4), (50_000, 4), (5_000, 4)):
    w = dict(nR=n, nS=n, bits=bits, dist="uniform")
    rhj.set_bits(bits)
    R, S = bench.make_relations(w, rhj.dev, 7)
    out = torch.empty((n, 2), dtype=torch.int64, device=rhj.dev)
    m = C.c_uint64(0)
    import time
    for i in range(3):
        rhj.lib.rhj_join_device(R.data_ptr(), n, S.data_ptr(), n, out.data_ptr(), n, C.byref(m))
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(5):
        rhj.lib.rhj_join_device(R.data_ptr(), n, S.data_ptr(), n, out.data_ptr(), n, C.byref(m))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
    st = rhj.stats()
    print(json.dumps({"n": n, "bits": bits, "wall_ms": round(dt * 1e3, 3), "gpu_ms": round(st["ms_total"], 3), "probe_ms": round(st["ms_probe"], 3),
                      "count_ms": round(st["ms_count"], 3), "build_ms": round(st["ms_build"], 3), "Gt/s": round(n / dt / 1e9, 2), "max_build": st["max_build"], "units": st["units"]}))
