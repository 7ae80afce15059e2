# the 88 RadixHashJoin calls of `small` (recorded inputs, 4 radix bits) one by one: time, path, fan-out
import importlib, ctypes as C, torch, sys, os, time
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import helpers
mod = importlib.import_module("sigmod-2018_amd"); rhj = mod.RHJ(device=0)
g = helpers.Golden()
rhj.set_bits(4)
rhj.lib.rhj_set_timing(1)
rows = []
for j in g.small["joins"]:
    R, S = g.small_join(j["idx"])
    dR, dS = rhj.to_device(R), rhj.to_device(S)
    cap = j["matches"] + 16
    out = torch.empty((cap, 2), dtype=torch.int64, device=rhj.dev)
    m = C.c_uint64(0)
    for i in range(3):
        rhj.lib.rhj_join_device(dR.data_ptr(), len(R), dS.data_ptr(), len(S), out.data_ptr(), cap, C.byref(m))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        rhj.lib.rhj_join_device(dR.data_ptr(), len(R), dS.data_ptr(), len(S), out.data_ptr(), cap, C.byref(m))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20 * 1e3
    rows.append((dt, len(R), len(S), m.value, rhj.stats()["path"]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("88 joins: %.2f ms in total; by time:" % tot)
for dt, nR, nS, m, path in rows[:12]:
    print("  %.3f ms  %7d x %7d -> %8d pairs (%.1f per probe tuple, %.2f GB/s of pairs)  %s" % (dt, nR, nS, m, m / max(nR, nS), m * 16 / dt / 1e6, path))
print("  median %.3f ms, smallest %.3f ms" % (rows[len(rows) // 2][0], rows[-1][0]))
