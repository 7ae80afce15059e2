import importlib, ctypes as C, torch, sys, json
import numpy as np
sys.path.insert(0, ".")
mod = importlib.import_module("sigmod-2018_amd"); rhj = mod.RHJ(device=0)
for n in (100_000_000, 400_000_000, 1_000_000_000):
    x = torch.empty((n, 2), dtype=torch.int64, device=rhj.dev)
    x[:, 0] = torch.randint(-(1 << 62), 1 << 62, (n,), device=rhj.dev)
    x[:, 1] = torch.arange(n, device=rhj.dev)
    y = torch.empty_like(x)
    for bits in (8, 12, 14):
        rhj.set_bits(bits)
        hist = np.zeros(1 << bits, dtype=np.uint64); psum = np.zeros(1 << bits, dtype=np.int64)
        for i in range(3):
            rhj.lib.rhj_partition_device(x.data_ptr(), n, y.data_ptr(), hist.ctypes.data_as(C.c_void_p), psum.ctypes.data_as(C.c_void_p))
        st = rhj.stats()
        passes = 1 if bits <= 8 else 2
        print(json.dumps({"n": n, "bits": bits, "hist_ms": round(st["ms_hist"], 3), "scatter_all_ms": round(st["ms_scatter"], 3),
                          "total_ms": round(st["ms_total"], 3), "GBps_total": round((16 * n + 32 * n * passes) / st["ms_total"] / 1e6)}))
    del x, y
    torch.cuda.empty_cache()
