import importlib, sys, numpy as np, time
sys.path.insert(0, "."); sys.path.insert(0, "oracle"); sys.path.insert(0, "tests")
from pyoracle import Oracle
mod = importlib.import_module("sigmod-2018_amd")
o = Oracle(); rhj = mod.RHJ(device=0)
def dev_join(R, S):
    t, m = rhj.join_device(rhj.to_device(R), rhj.to_device(S))
    return rhj.pairs_to_numpy(t)
for bits, nR, nS, kind, dom in ((4, 3_000_000, 4_000_000, 1, 3_000_000), (4, 4_000_000, 2_500_000, 1, 4_000_000), (6, 5_000_000, 5_000_000, 4, 1_500_000), (4, 2_000_000, 6_000_000, 2, 2_000_000), (8, 12_000_000, 16_000_000, 1, 12_000_000)):
    rhj.set_bits(bits)
    R = o.generate(nR, 0 if kind != 4 else 4, dom, 0.0, 5 + bits)
    S = o.generate(nS, kind, dom, 0.9, 6 + bits)
    t0 = time.time(); want = o.join(R, S, bits); t1 = time.time()
    got = dev_join(R, S)
    st = rhj.stats()
    ok = len(got) == len(want) and bool((got == want).all())
    print(bits, nR, nS, kind, "path", st["path"], "matches", len(want), "OK" if ok else "MISMATCH", "ms", round(st["ms_total"], 3), "oracle s", round(t1 - t0, 1), flush=True)
    if not ok:
        n = min(len(got), len(want)); bad = np.nonzero(got[:n] != want[:n])[0]
        print("  first mismatch at", bad[:5], got[bad[:3]], want[bad[:3]])
