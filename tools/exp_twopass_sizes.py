import importlib, ctypes as C, torch, sys, os, time
sys.path.insert(0, ".")
import bench
mod = importlib.import_module("sigmod-2018_amd"); rhj = mod.RHJ(device=0)
rhj.lib.rhj_set_timing(1)
def run(nR, nS, bits, reps=100):
    w = dict(nR=nR, nS=nS, bits=bits, dist="uniform")
    rhj.set_bits(bits)
    R, S = bench.make_relations(w, rhj.dev, 1234)
    cap = max(nR, nS) + 1024
    out = torch.empty((cap, 2), dtype=torch.int64, device=rhj.dev)
    m = C.c_uint64(0)
    for i in range(5):
        rhj.lib.rhj_join_device(R.data_ptr(), nR, S.data_ptr(), nS, out.data_ptr(), cap, C.byref(m))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        rhj.lib.rhj_join_device(R.data_ptr(), nR, S.data_ptr(), nS, out.data_ptr(), cap, C.byref(m))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, rhj.stats()
for bits in (9, 12, 15):
    for n in (1000, 100000, 1000000, 4000000, 16000000):
        a, st = run(n, n, bits)
        print("bits %2d %9d x %9d: %.4f ms (%s) gpu %.4f" % (bits, n, n, a, st["path"], st["ms_total"]), flush=True)
