"""device entry points of include/rhj_inter.h at size: gather of k row-id tables through a join's
match list (InsertJoinToInterResults), relation build (GetRelation), view sums (CalculateQueryResults)"""
import ctypes as C, importlib, json, sys
import numpy as np, torch
sys.path.insert(0, ".")
mod = importlib.import_module("sigmod-2018_amd"); rhj = mod.RHJ(device=0)
lib = rhj.lib
u64p = C.POINTER(C.c_uint64)
lib.rhj_build_relation_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
lib.rhj_sum_gather_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, u64p]
dev = rhj.dev
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
out = {}
n, rows = 100_000_000, 100_000_000
g = torch.Generator(device=dev); g.manual_seed(3)
pairs = torch.randint(0, rows, (n, 2), generator=g, device=dev, dtype=torch.int64)
for k in (1, 3):
    tabs = [torch.randint(0, 1 << 40, (rows,), generator=g, device=dev, dtype=torch.int64) for _ in range(k)]
    outs = [torch.empty(n, dtype=torch.int64, device=dev) for _ in range(k)]
    dst = (C.c_void_p * k)(*[o.data_ptr() for o in outs]); src = (C.c_void_p * k)(*[t.data_ptr() for t in tabs])
    ms = timeit(lambda: lib.rhj_gather_tables_device(dst, src, k, C.c_void_p(pairs.data_ptr()), 2, C.c_uint64(n)))
    alg = n * (8 + 16 * k)            # index read (8 B of each 16-B pair) + per table 8 B gathered + 8 B written
    out["gather_random_%dtables" % k] = {"n": n, "ms": ms, "algorithmic_GBps": alg / ms / 1e6, "frac_of_8TBps": alg / ms / 1e6 / 8000}
    del tabs, outs
# sorted indices (what a join of a key-ordered relation produces): the gather streams
srt = torch.sort(pairs[:, 0]).values.contiguous()
tab = torch.randint(0, 1 << 40, (rows,), generator=g, device=dev, dtype=torch.int64)
o1 = torch.empty(n, dtype=torch.int64, device=dev)
dst = (C.c_void_p * 1)(o1.data_ptr()); src = (C.c_void_p * 1)(tab.data_ptr())
ms = timeit(lambda: lib.rhj_gather_tables_device(dst, src, 1, C.c_void_p(srt.data_ptr()), 1, C.c_uint64(n)))
out["gather_sorted_1table"] = {"n": n, "ms": ms, "algorithmic_GBps": n * 24 / ms / 1e6, "frac_of_8TBps": n * 24 / ms / 1e6 / 8000}
tup = torch.empty((n, 2), dtype=torch.int64, device=dev)
ms = timeit(lambda: lib.rhj_build_relation_device(tab.data_ptr(), None, n, tup.data_ptr()))
out["build_relation_direct"] = {"n": n, "ms": ms, "algorithmic_GBps": n * 24 / ms / 1e6, "frac_of_8TBps": n * 24 / ms / 1e6 / 8000}
ms = timeit(lambda: lib.rhj_build_relation_device(tab.data_ptr(), pairs.data_ptr(), n, tup.data_ptr()))
out["build_relation_through_random_ids"] = {"n": n, "ms": ms, "algorithmic_GBps": n * 32 / ms / 1e6}
s = C.c_uint64(0)
ms = timeit(lambda: lib.rhj_sum_gather_device(tab.data_ptr(), srt.data_ptr(), n, C.byref(s)), reps=5)
out["sum_gather_sorted_ids"] = {"n": n, "ms": ms, "algorithmic_GBps": n * 16 / ms / 1e6, "note": "includes the 8-byte read-back and sync"}
print(json.dumps(out))
