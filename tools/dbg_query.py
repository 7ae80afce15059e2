"""run selected `small` queries through one of the engines with RHJ_TRACE=1"""
import os, subprocess, sys, tempfile
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import helpers
g = helpers.Golden()
exe = sys.argv[1]; which = [int(x) for x in sys.argv[2:]]
tmp = tempfile.mkdtemp(); names = []
for i in range(14):
    cols = g.small_relations["r%d" % i].astype("<u8")
    with open(os.path.join(tmp, "r%d" % i), "wb") as f:
        np.array([cols.shape[1], cols.shape[0]], dtype="<u8").tofile(f); cols.tofile(f)
    names.append("r%d" % i)
work = [l for l in g.small["work_lines"]]
q = [l for l in work if "|" in l]
sel = [q[i] for i in which]
stdin = ("\n".join(names) + "\nDone\n" + "\n".join(sel) + "\nF\n").encode()
r = subprocess.run([os.path.abspath(os.path.join("oracle", "_ref", exe))], input=stdin, cwd=tmp, env=dict(os.environ, RHJ_TRACE="1"),
                   stdout=subprocess.PIPE, stderr=subprocess.PIPE)
print("stdout:", r.stdout.decode()); print("want:", [g.small["result_lines"][i] for i in which]); print(r.stderr.decode()[-6000:])
