"""Randomised stress of rhj_join_device_range / rhj_join_device_slice against the oracle: relations of 1 K .. 3 M tuples, unique / uniform /
Zipf / repeated keys, a key that dominates, radix widths 1..15, fused, tiled (HBM tables) and low-radix paths, wide row ids; the shares of a
random set of cuts in (bucket, position among the bucket's probe tuples) space must concatenate to the oracle's list, bit for bit.
python tools/stress_slices.py [joins] [seed]"""
import importlib, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "oracle"); sys.path.insert(0, "tests")
from pyoracle import Oracle
import torch
mod = importlib.import_module("sigmod-2018_amd"); rhj = mod.RHJ(device=0)
shard = importlib.import_module("sigmod-2018_amd.shard")
o = Oracle()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.RandomState(seed)
t0 = time.time(); paths = {}
for it in range(iters):
    bits = int(rng.choice([1, 2, 4, 4, 6, 8, 10, 12, 14, 15]))
    nR = int(rng.choice([1000, 30_000, 300_000, 900_000, 3_000_000])); nS = int(rng.choice([1500, 50_000, 400_000, 1_200_000, 3_000_000]))
    kindR = int(rng.choice([0, 0, 4])); kindS = int(rng.choice([1, 2, 4]))
    dom = nR if kindR == 0 else max(nR // 3, 1)
    R = o.generate(nR, kindR, dom, 0.0, 1000 + it); S = o.generate(nS, kindS, dom, 0.7, 2000 + it)
    if rng.rand() < 0.4: S["value"][rng.randint(0, nS, nS // 2)] = R["value"][rng.randint(0, nR)]          # a key that dominates
    if rng.rand() < 0.15: S["row_id"][nS // 2] += np.uint64(1 << 40)
    hbm = rng.rand() < 0.2; fused = rng.rand() > 0.15
    rhj.set_bits(bits); rhj.lib.rhj_set_force_hbm_table(1 if hbm else 0); rhj.lib.rhj_set_fused(1 if fused else 0)
    want = o.join(R, S, bits)
    if len(want) > 40_000_000: continue
    dR, dS = rhj.to_device(R), rhj.to_device(S)
    mask = np.uint64((1 << bits) - 1)
    hr = np.bincount((R["value"] & mask).astype(np.int64), minlength=1 << bits); hs = np.bincount((S["value"] & mask).astype(np.int64), minlength=1 << bits)
    if rng.rand() < 0.5:
        plan = shard.bucket_slices(hr, hs, int(rng.randint(2, 9)))
    else:
        cuts = sorted((int(b), int(rng.randint(0, max(hr[b], hs[b]) + 50)) if rng.rand() < 0.7 else 0) for b in rng.randint(0, 1 << bits, int(rng.randint(1, 6))))
        cuts = [(0, 0)] + cuts + [(1 << bits, 0)]
        plan = [(b0, b1 + 1, o0, o1) if o1 else (b0, b1, o0, 0) for (b0, o0), (b1, o1) in zip(cuts, cuts[1:])]
    parts = []
    for lo, hi, skip, end in plan:
        if lo >= hi: continue
        rng_ = (lo, hi, skip, end) if (skip or end or rng.rand() < 0.5) else (lo, hi)
        t, k = rhj.join_device(dR, dS, bucket_range=rng_)
        p = rhj.stats()["path"]; paths[p] = paths.get(p, 0) + 1
        parts.append(rhj.pairs_to_numpy(t)[:k])
    got = np.concatenate(parts) if parts else want[:0]
    info = dict(it=it, seed=seed, bits=bits, nR=nR, nS=nS, kindR=kindR, kindS=kindS, hbm=hbm, fused=fused, plan=plan[:6], m=len(want))
    if not (len(got) == len(want) and (got == want).all()):
        bad = np.nonzero(got[: min(len(got), len(want))] != want[: min(len(got), len(want))])[0]
        print("MISMATCH", info, len(got), "first at", bad[:5]); sys.exit(1)
    print("ok", info, flush=True)
rhj.lib.rhj_set_force_hbm_table(0); rhj.lib.rhj_set_fused(1)
print("stress ok: %d joins, %.0f s; shares by path %s" % (iters, time.time() - t0, paths))
