"""The `small` workload scaled up K times, end to end: every relation is K copies of itself, copy c with
c * STRIDE added to every value (so joins only match inside a copy and result sizes grow K-fold, not
K^2-fold).  The reference engine as shipped and the device-resident configuration run the same 50
queries on the same files and must print the same 50 lines; wall time of the whole process."""
import json, os, subprocess, sys, tempfile, time
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import helpers
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
g = helpers.Golden()
STRIDE = np.uint64(1 << 24)            # above every value of `small` (max 2^23.x)
tmp = tempfile.mkdtemp(); names = []; total = 0
for i in range(14):
    cols = g.small_relations["r%d" % i].astype("<u8")                      # [columns][tuples]
    assert int(cols.max()) < int(STRIDE)
    big = np.concatenate([cols + np.uint64(c) * STRIDE for c in range(K)], axis=1)
    with open(os.path.join(tmp, "r%d" % i), "wb") as f:
        np.array([big.shape[1], big.shape[0]], dtype="<u8").tofile(f); np.ascontiguousarray(big).tofile(f)
    names.append("r%d" % i); total += big.size * 8
stdin = ("\n".join(names) + "\nDone\n" + "\n".join(g.small["work_lines"]) + "\n").encode()
out, outputs = {"K": K, "relation_bytes": total}, {}
for exe in ("radixhash_t4", "radixhash_rhj_resident"):
    path = os.path.abspath(os.path.join("oracle", "_ref", exe))
    if not os.path.exists(path): continue
    best = None
    for rep in range(2 if exe == "radixhash_t4" else 3):
        t = time.perf_counter()
        r = subprocess.run([path], input=stdin, cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
    outputs[exe] = r.stdout.decode().splitlines()
    out[exe] = {"best_wall_s": round(best, 3), "lines": len(outputs[exe]), "rc": r.returncode}
if len(outputs) == 2:
    a, b = outputs["radixhash_t4"], outputs["radixhash_rhj_resident"]
    out["identical_output"] = a == b
    out["mismatching_lines"] = [i for i, (x, y) in enumerate(zip(a, b)) if x != y][:10]
print(json.dumps(out))
