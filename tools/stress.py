"""Randomised stress of the device join against the oracle (CPU restatement): many shapes, key
distributions, duplicate levels, radix widths 1..15, all four device paths, wide and narrow row ids.
Prints the first mismatch (seeded: reproducible) or a summary."""
import importlib, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "oracle"); sys.path.insert(0, "tests")
from pyoracle import Oracle
mod = importlib.import_module("sigmod-2018_amd"); rhj = mod.RHJ(device=0)
o = Oracle()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.RandomState(seed)
PATHS = ["fused", "fused_gather", "tiled32", "tiled64"]
def set_path(p):
    rhj.lib.rhj_set_fused(0 if p.startswith("tiled") else 2)
    rhj.lib.rhj_set_force_hbm_table(1 if p == "tiled64" else 0)
    rhj.lib.rhj_set_resident(0 if p == "fused_gather" else 1)
t0 = time.time(); pairs = 0; ran = 0; taken = {}
for it in range(iters):
    bits = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 8, 9, 10, 11, 12, 13, 14, 15]))
    nR = int(rng.choice([1, 17, 300, 4096, 4097, 20000, 65536, 150000, 400000]))
    nS = int(rng.choice([1, 64, 1000, 8191, 50000, 131072, 300000, 600000]))
    kind = int(rng.choice([1, 2, 3, 4]))
    dom = int(rng.choice([1, 3, 50, 2000, 100000, 1 << 33]))
    R = o.generate(nR, 4 if kind == 4 else 0 if kind != 3 else 3, dom, 0.0, 10000 + it * 2 + seed * 100000)
    S = o.generate(nS, kind, min(dom, max(nR, 1)) if kind != 4 else dom, float(rng.choice([0.5, 0.9, 1.1])), 10001 + it * 2 + seed * 100000)
    r = rng.rand()
    if r < 0.2: R["row_id"] = R["row_id"] * np.uint64(0x9E3779B97F4A7C15) + np.uint64(it)          # wide everywhere
    elif r < 0.35 and nS > 5000: S["row_id"][nS // 2] += np.uint64(1 << 45)                         # one wide id in the middle
    path = PATHS[int(rng.randint(0, 4))] if rng.rand() < 0.5 else "fused"
    # bound the output before anybody materialises it (host memory): matches = sum over keys of cR * cS
    kr, cr = np.unique(R["value"], return_counts=True)
    ks, cs = np.unique(S["value"], return_counts=True)
    common, ir, is_ = np.intersect1d(kr, ks, assume_unique=True, return_indices=True)
    if int((cr[ir].astype(np.int64) * cs[is_].astype(np.int64)).sum()) > 20_000_000: continue
    want = o.join(R, S, bits)
    set_path(path); rhj.set_bits(bits)
    dR, dS = rhj.to_device(R), rhj.to_device(S)
    t, m = rhj.join_device(dR, dS, capacity=max(len(want), 1))
    got = rhj.pairs_to_numpy(t)[:m]
    ok = m == len(want) and (got == want).all()
    pairs += m; ran += 1
    pth = rhj.stats()["path"]; taken[pth] = taken.get(pth, 0) + 1
    if not ok:
        print("MISMATCH", dict(it=it, seed=seed, bits=bits, nR=nR, nS=nS, kind=kind, dom=dom, path=path, m=m, want=len(want)))
        sys.exit(1)
print("stress ok: %d of %d joins run (the rest exceed the output bound), %d pairs, %.1f s; paths taken %s" % (ran, iters, pairs, time.time() - t0, taken))
