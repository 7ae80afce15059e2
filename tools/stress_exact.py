"""Randomised stress of k_join_exact (rhj_join_exact.hip.h) against the oracle: two-pass joins whose buckets are the
gather class (7 K .. 28 K build tuples) at 10..12 radix bits, both relations the same size (half the buckets probed by either
side: the single-pass units and the count + emit units) or one bigger; exact foreign-key inputs, inputs that break the hypothesis,
hot keys (more than four matches a tuple: the slow emit), build sides whose row ids do not increase (handed over), wide row ids.
python tools/stress_exact.py [joins] [seed]    Prints the first mismatch (seeded) or a summary of who did the joins."""
import importlib, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "oracle"); sys.path.insert(0, "tests")
from pyoracle import Oracle
mod = importlib.import_module("sigmod-2018_amd"); rhj = mod.RHJ(device=0)
o = Oracle()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 12
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.RandomState(seed)
t0 = time.time(); went = {0: 0, 1: 0, 2: 0}
for it in range(iters):
    bits = int(rng.choice([10, 10, 10, 11]))
    per = int(rng.choice([7400, 9000, 14000]))                        # build tuples a bucket
    nsmall = per << bits
    nbig = nsmall if rng.rand() < 0.6 else nsmall + (int(rng.choice([1000, 3000])) << bits)
    uniq = o.generate(nsmall, 0, 0, 0.0, 777 + it)                     # unique keys
    kind = int(rng.choice([1, 1, 2]))
    fk = o.generate(nbig, kind, nsmall, 0.5, 999 + it)                # every key has its partner (uniform or Zipf)
    how = int(rng.randint(0, 8))
    if how == 1: fk["value"][rng.randint(0, nbig)] = np.uint64(1 << 51)                       # one tuple without a partner
    elif how == 2: uniq["value"][rng.randint(1, nsmall)] = uniq["value"][0]                   # a key twice on the unique side
    elif how == 3: fk["value"][rng.randint(0, nbig, nbig // 10)] = np.uint64(3 << 50)         # many without a partner, one hot key
    elif how == 4: uniq = uniq[: nsmall - nsmall // 50]                                       # 2 % of the keys gone
    elif how == 5: fk["row_id"] = fk["row_id"][::-1].copy()                                   # row ids that decrease: not the exact kernel's
    elif how == 6: uniq["row_id"][len(uniq) // 2] = np.uint64(0)                              # one row id out of order
    if rng.rand() < 0.15: fk["row_id"][nbig // 3] += np.uint64(1 << 44)
    R, S = (uniq, fk) if rng.rand() < 0.6 else (fk, uniq)
    want = o.join(R, S, bits)
    rhj.set_bits(bits)
    if rng.rand() < 0.8: rhj.lib.rhj_set_spec(1); rhj.lib.rhj_set_exact(1)   # (resets the try-or-not scores; else: whatever the history says)
    t, m = rhj.join_device(rhj.to_device(R), rhj.to_device(S), capacity=max(len(want), len(R), len(S)) + 7)
    got = rhj.pairs_to_numpy(t)[:m]
    ex = int(rhj.lib.rhj_last_exact())
    went[ex] += 1
    info = dict(it=it, seed=seed, bits=bits, nsmall=nsmall, nbig=nbig, kind=kind, how=how, m=m, want=len(want), spec=int(rhj.lib.rhj_last_spec()), exact=ex, R_is_uniq=R is uniq)
    if not (m == len(want) and (got == want).all()):
        bad = np.nonzero(got[: min(m, len(want))] != want[: min(m, len(want))])[0]
        print("MISMATCH", info, "first at", bad[:5], got[bad[:3]], want[bad[:3]])
        sys.exit(1)
    print("ok", info, flush=True)
    if how == 0 and ex == 2 and (nbig > nsmall or R is uniq):      # (equal sizes: the hypothesis is on S — an S of unique keys breaks it by itself)
        print("an exact foreign-key join was handed over", info); sys.exit(1)
print("stress ok: %d joins, %.0f s; k_join_exact not launched %d, did the join %d, handed over %d" % (iters, time.time() - t0, went[0], went[1], went[2]))
