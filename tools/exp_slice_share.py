"""One device's share of a 100M x 100M join on the reference's 4 radix bits at n = 8 (two buckets a device), and half a bucket's probe
side (rhj_join_device_slice: what a device takes of a hot bucket), against the whole join on one device: python tools/exp_slice_share.py"""
import importlib, sys, json, time
sys.path.insert(0, ".")
import bench, torch
mod = importlib.import_module("sigmod-2018_amd"); rhj = mod.RHJ(device=0)
n = 100_000_000
R, S = bench.make_relations(dict(nR=n, nS=n, bits=4, dist="uniform"), rhj.dev, 7)
rhj.set_bits(4)
def timed(rng, cap):
    best = 1e9
    for i in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        t, m = rhj.join_device(R, S, capacity=cap, bucket_range=rng)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        if i: best = min(best, dt)
    return round(best, 3), int(m), {k: round(v, 3) for k, v in rhj.stats().items() if k in ("ms_hist", "ms_scatter", "ms_probe", "ms_total")}
print("whole join, one device        ", timed(None, n), flush=True)
print("buckets [0, 2) of 16 (1/8)    ", timed((0, 2), n // 4), flush=True)
half = n // 16 // 2 // 256 * 256
print("bucket 0, first half of probes", timed((0, 1, 0, half), n // 8), flush=True)
print("bucket 0, second half         ", timed((0, 1, half, 0), n // 8), flush=True)
