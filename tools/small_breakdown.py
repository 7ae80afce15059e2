"""where the wall time of the `small` run goes: process + relation loading alone, then with the queries"""
import json, os, subprocess, sys, tempfile, time
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import helpers
g = helpers.Golden()
tmp = tempfile.mkdtemp(); names = []
for i in range(14):
    cols = g.small_relations["r%d" % i].astype("<u8")
    with open(os.path.join(tmp, "r%d" % i), "wb") as f:
        np.array([cols.shape[1], cols.shape[0]], dtype="<u8").tofile(f); cols.tofile(f)
    names.append("r%d" % i)
q = [l for l in g.small["work_lines"]]
out = {}
for exe in ("radixhash_t4", "radixhash_rhj", "radixhash_rhj_resident"):
    path = os.path.abspath(os.path.join("oracle", "_ref", exe))
    if not os.path.exists(path): continue
    res = {}
    for label, work in (("load_only", []), ("first_query_only", q[:2]), ("all_50", q)):
        stdin = ("\n".join(names) + "\nDone\n" + "\n".join(work) + "\n").encode()
        best = None
        for rep in range(3):
            t = time.perf_counter()
            subprocess.run([path], input=stdin, cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            dt = time.perf_counter() - t
            best = dt if best is None else min(best, dt)
        res[label] = round(best, 4)
    out[exe] = res
print(json.dumps(out))
