"""which prefix of `small` queries makes query T come out wrong in an engine"""
import os, subprocess, sys, tempfile
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import helpers
g = helpers.Golden()
exe = sys.argv[1]; T = int(sys.argv[2])
tmp = tempfile.mkdtemp(); names = []
for i in range(14):
    cols = g.small_relations["r%d" % i].astype("<u8")
    with open(os.path.join(tmp, "r%d" % i), "wb") as f:
        np.array([cols.shape[1], cols.shape[0]], dtype="<u8").tofile(f); cols.tofile(f)
    names.append("r%d" % i)
q = [l for l in g.small["work_lines"] if "|" in l]
def run(idxs):
    stdin = ("\n".join(names) + "\nDone\n" + "\n".join(q[i] for i in idxs) + "\nF\n").encode()
    r = subprocess.run([os.path.abspath(os.path.join("oracle", "_ref", exe))], input=stdin, cwd=tmp, env=dict(os.environ),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    out = r.stdout.decode().splitlines()
    return [o == g.small["result_lines"][i] for o, i in zip(out, idxs)], r.stderr.decode()[-300:]
for k in range(T - 1, -1, -1):
    ok, err = run([k, T])
    if not all(ok):
        print("pair", k, T, ok, q[k], err)
ok, err = run(list(range(T + 1)))
print("prefix 0..%d:" % T, [i for i, o in enumerate(ok) if not o])
