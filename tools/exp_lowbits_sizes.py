# big relations on few radix bits (the reference ships N_LSB 4): buckets beyond the LDS index -> tiled path with HBM tables
import importlib, ctypes as C, torch, sys, os, time
sys.path.insert(0, ".")
import bench
mod = importlib.import_module("sigmod-2018_amd"); rhj = mod.RHJ(device=0)
def run(nR, nS, bits, reps=5):
    w = dict(nR=nR, nS=nS, bits=bits, dist="uniform")
    rhj.set_bits(bits)
    R, S = bench.make_relations(w, rhj.dev, 1234)
    cap = max(nR, nS) + 1024
    out = torch.empty((cap, 2), dtype=torch.int64, device=rhj.dev)
    m = C.c_uint64(0)
    for i in range(2):
        rhj.lib.rhj_join_device(R.data_ptr(), nR, S.data_ptr(), nS, out.data_ptr(), cap, C.byref(m))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        rhj.lib.rhj_join_device(R.data_ptr(), nR, S.data_ptr(), nS, out.data_ptr(), cap, C.byref(m))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, rhj.stats()
for bits in (4, 8):
    for n in (1000000, 4000000, 16000000, 100000000):
        a, st = run(n, n, bits)
        keys = ("ms_hist", "ms_scan", "ms_scatter", "ms_plan", "ms_build", "ms_count", "ms_offsets", "ms_probe")
        print("bits %2d %9d x %9d: %.3f ms (%s, %d units, %d hbm) = %.2f G probe tuples/s   stages %s" % (bits, n, n, a, st["path"], st["units"], st["hbm_units"], n / a / 1e6,
              " ".join("%s %.3f" % (k[3:], st[k]) for k in keys)), flush=True)
