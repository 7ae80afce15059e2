# small path against the path of separate launches over join sizes: ms per join, wall clock over 100 joins each
import importlib, ctypes as C, torch, sys, os, time
sys.path.insert(0, ".")
import bench
mod = importlib.import_module("sigmod-2018_amd"); rhj = mod.RHJ(device=0)
rhj.lib.rhj_set_timing(1)
def run(nR, nS, bits, small, reps=100):
    w = dict(nR=nR, nS=nS, bits=bits, dist="uniform")
    rhj.set_bits(bits)
    rhj.lib.rhj_set_small(small)
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
    return (time.perf_counter() - t0) / reps * 1e3, rhj.stats()["path"]
for bits in (4, 8):
    for n in (1000, 10000, 100000, 500000, 1000000, 2000000, 4000000):
        os.environ["RHJ_SMALL_TILES"] = "512"
        a, pa = run(n, n, bits, 1)
        b, pb = run(n, n, bits, 0)
        print("bits %d  %8d x %8d: small %.4f ms (%s)   separate launches %.4f ms (%s)   ratio %.2f" % (bits, n, n, a, pa, b, pb, a / b), flush=True)
