"""SIGMOD'18 `small` workload end to end on the GPU box: the reference engine as shipped
(oracle/_ref/radixhash_t4, THREADS 4 / radixhash_t1) next to the same engine with librhj.so doing
RadixHashJoin()/Filter() (oracle/_ref/radixhash_rhj).  Wall time of the whole process
(load + 50 queries); all three must print small.result."""
import json, os, subprocess, sys, tempfile, time
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import helpers
g = helpers.Golden()
tmp = tempfile.mkdtemp()
names = []
for i in range(14):
    cols = g.small_relations["r%d" % i].astype("<u8")
    with open(os.path.join(tmp, "r%d" % i), "wb") as f:
        np.array([cols.shape[1], cols.shape[0]], dtype="<u8").tofile(f); cols.tofile(f)
    names.append("r%d" % i)
stdin = ("\n".join(names) + "\nDone\n" + "\n".join(g.small["work_lines"]) + "\n").encode()
out = {}
for exe in ("radixhash_t1", "radixhash_t4", "radixhash_rhj", "radixhash_rhj_resident"):
    path = os.path.join("oracle", "_ref", exe)
    if not os.path.exists(path):
        continue
    best = None
    for rep in range(3):
        t = time.perf_counter()
        r = subprocess.run([os.path.abspath(path)], input=stdin, cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        dt = time.perf_counter() - t
        ok = r.stdout.decode().splitlines() == g.small["result_lines"]
        best = dt if best is None else min(best, dt)
    out[exe] = {"best_wall_s": best, "correct": ok}
print(json.dumps(out))
