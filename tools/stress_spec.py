"""Randomised stress of the foreign-key speculation (k_join_spec) against the oracle: two-pass joins big
enough to try it, exact foreign-key inputs and inputs that break the hypothesis in one tuple or in many, either relation
the bigger one, wide row ids now and then.  Prints the first mismatch (seeded) or a summary with how the tries went."""
import importlib, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "oracle"); sys.path.insert(0, "tests")
from pyoracle import Oracle
mod = importlib.import_module("sigmod-2018_amd"); rhj = mod.RHJ(device=0)
o = Oracle()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.RandomState(seed)
t0 = time.time(); went = {0: 0, 1: 0, 2: 0}
for it in range(iters):
    bits = int(rng.choice([9, 9, 10, 11]))
    nsmall = int(rng.choice([300_000, 1_000_000, 2_500_000]))
    nbig = int(rng.choice([2_200_000, 3_000_000, 5_000_000])) if bits == 9 else int(rng.choice([4_500_000, 6_000_000])) if bits == 10 else 9_000_000
    if rng.rand() < 0.35:                                              # build sides beyond the LDS-resident size (k_join_spec<false>: both kinds of
        nsmall = int(rng.choice([8_000, 12_000, 20_000])) << bits      # units when the sizes are equal — the direct path of round 4)
        nbig = nsmall if rng.rand() < 0.6 else nsmall + (2000 << bits)
    uniq = o.generate(nsmall, 0, 0, 0.0, 777 + it)                     # unique keys
    fk = o.generate(nbig, int(rng.choice([1, 2])), nsmall, 0.9, 999 + it)   # every key has its partner (uniform or Zipf)
    how = int(rng.randint(0, 6))
    if how == 1: fk["value"][rng.randint(0, nbig)] = np.uint64(1 << 51)                       # one tuple without a partner
    elif how == 2: uniq["value"][rng.randint(1, nsmall)] = uniq["value"][0]                   # a key twice on the unique side
    elif how == 3: fk["value"][rng.randint(0, nbig, nbig // 10)] = np.uint64(3 << 50)         # many without a partner, one hot key
    elif how == 4: uniq = uniq[: nsmall - nsmall // 50]                                       # 2 % of the keys gone
    if rng.rand() < 0.25: fk["row_id"][nbig // 3] += np.uint64(1 << 44)
    R, S = (uniq, fk) if rng.rand() < 0.6 else (fk, uniq)
    want = o.join(R, S, bits)
    rhj.set_bits(bits)
    if rng.rand() < 0.7: rhj.lib.rhj_set_spec(1)                       # (resets the try-or-not score; else: whatever the history says)
    t, m = rhj.join_device(rhj.to_device(R), rhj.to_device(S), capacity=max(len(want), len(R), len(S)) + 7)
    got = rhj.pairs_to_numpy(t)[:m]
    went[int(rhj.lib.rhj_last_spec())] += 1
    if not (m == len(want) and (got == want).all()):
        print("MISMATCH", dict(it=it, seed=seed, bits=bits, nsmall=nsmall, nbig=nbig, how=how, m=m, want=len(want), spec=int(rhj.lib.rhj_last_spec())))
        sys.exit(1)
    if how == 0 and nbig > nsmall and rhj.lib.rhj_last_spec() == 2 and R is uniq:
        print("an exact foreign-key join failed its speculation", dict(it=it, seed=seed, bits=bits, nsmall=nsmall, nbig=nbig)); sys.exit(1)
print("stress ok: %d joins, %.0f s; speculation not tried %d, held %d, failed %d" % (iters, time.time() - t0, went[0], went[1], went[2]))
