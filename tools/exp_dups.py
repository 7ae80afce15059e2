"""Duplicate-heavy joins: device time next to the reference's own code on the host (same inputs)."""
import importlib, sys, time, json
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "oracle"); sys.path.insert(0, "tests")
from pyoracle import Oracle, Reference, ref_available
import helpers
mod = importlib.import_module("sigmod-2018_amd"); rhj = mod.RHJ(device=0)
o = Oracle(); g = helpers.Golden()
cases = []
big = max(g.small["joins"], key=lambda j: j["matches"])
R, S = g.small_join(big["idx"]); cases.append(("small join %d (%dx%d -> %d)" % (big["idx"], len(R), len(S), big["matches"]), R, S, 4))
cases.append(("1Mx1M domain 50K (20 dups/key) b8", o.generate(1000000, 4, 50000, 0, 1), o.generate(1000000, 4, 50000, 0, 2), 8))
cases.append(("200Kx200K domain 200 (1000 dups/key) b4", o.generate(200000, 4, 200, 0, 3), o.generate(200000, 4, 200, 0, 4), 4))
cases.append(("zipf both sides 2Mx2M b12", o.generate(2000000, 2, 100000, 0.9, 5), o.generate(2000000, 2, 100000, 0.9, 6), 12))
for name, R, S, bits in cases:
    rhj.set_bits(bits)
    dR, dS = rhj.to_device(R), rhj.to_device(S)
    _, m = rhj.join_device(dR, dS, count_only=True)
    t, m = rhj.join_device(dR, dS, capacity=m)
    st = rhj.stats()
    line = {"case": name, "matches": m, "gpu_ms": st["ms_total"], "probe_ms": st["ms_probe"], "units": st["units"]}
    if ref_available(bits, 1) and m < 300_000_000:
        ref = Reference(bits, 1)
        want, info = ref.join(R, S, with_info=True)
        got = rhj.pairs_to_numpy(t)
        line["cpu_ref_ms"] = info["seconds"] * 1e3
        line["equal"] = bool(len(got) == len(want) and (got == want).all())
    print(json.dumps(line))
