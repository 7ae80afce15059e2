"""Two-pass joins of small and mid-size relations, with pass 1 counting pass 2's digits (strips of tiles) and without
(RHJ_NO_COUNT_IN_PASS1): python tools/exp_midsizes.py   — GPU time of the join, median of 7 after 3 warm-ups"""
import importlib, ctypes as C, torch, sys, json, statistics
sys.path.insert(0, ".")
import bench
mod = importlib.import_module("sigmod-2018_amd"); rhj = mod.RHJ(device=0)
for n, bits in ((200_000, 9), (500_000, 9), (1_000_000, 10), (2_000_000, 12), (4_000_000, 12), (8_000_000, 12), (16_000_000, 12)):
    w = dict(nR=n, nS=n, bits=bits, dist="uniform")
    R, S = bench.make_relations(w, rhj.dev, 99)
    out = torch.empty((n, 2), dtype=torch.int64, device=rhj.dev)
    m = C.c_uint64(0)
    rhj.set_bits(bits)
    row = {"n": n, "bits": bits}
    for on in (1, 0):
        rhj.lib.rhj_set_count_in_pass1(on)
        t, h = [], []
        for i in range(10):
            rhj.lib.rhj_join_device(R.data_ptr(), n, S.data_ptr(), n, out.data_ptr(), n, C.byref(m))
            if i >= 3:
                st = rhj.stats(); t.append(st["ms_total"]); h.append(st["ms_hist"] + st["ms_scan"])
        row["count_in_pass1" if on else "digit_bytes"] = {"ms_total": round(statistics.median(t), 4), "ms_pass1_and_counts": round(statistics.median(h), 4)}
    rhj.lib.rhj_set_count_in_pass1(1)
    print(json.dumps(row), flush=True)
