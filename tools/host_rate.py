"""PCIe-inclusive rate of the host ABI: RadixHashJoin() with pageable host relations
(H2D, kernels, D2H into malloc'd nodes).  Reported in profiles/README.md; never `value`."""
import importlib, sys, time, json
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
from pyoracle import Oracle
mod = importlib.import_module("sigmod-2018_amd"); rhj = mod.RHJ(device=0)
o = Oracle()
out = {}
for n, bits in ((1_000_000, 8), (16_000_000, 12)):
    rhj.set_bits(bits)
    R = o.generate(n, 0, 0, 0.0, 42); S = o.generate(n, 1, n, 0.0, 43)
    rhj.RadixHashJoin(R, S)
    t = time.perf_counter(); pairs = rhj.RadixHashJoin(R, S); dt = time.perf_counter() - t
    st = rhj.stats()
    out["%dx%d@%d" % (n, n, bits)] = {"wall_ms_incl_python_list_walk": dt * 1e3, "ms_h2d": st["ms_h2d"], "ms_gpu": st["ms_total"],
                                       "ms_d2h": st["ms_d2h"], "pairs": len(pairs),
                                       "e9_probe_tuples_per_s_h2d_gpu_d2h": n / ((st["ms_h2d"] + st["ms_total"] + st["ms_d2h"]) * 1e-3) / 1e9}
print(json.dumps(out))
