"""The joins of a steady-state batch of `small` by input size: how many, how long (RHJ_TRACE of the device-resident engine)."""
import os, re, subprocess, sys, tempfile, statistics
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
stdin = ("\n".join(names) + "\nDone\n" + ("\n".join(g.small["work_lines"]) + "\n") * 3).encode()
exe = os.path.abspath(os.path.join("oracle", "_ref", "radixhash_rhj_resident"))
r = subprocess.run([exe], input=stdin, cwd=tmp, env=dict(os.environ, RHJ_TRACE="1", RHJ_RADIX_BITS="4"), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
ts = []
for l in r.stderr.decode().splitlines():
    m = re.search(r"rhj-trace\s+([0-9.]+)", l)
    if m: ts.append((float(m.group(1)), l))
per = (len(ts) - 15) // 3
ts = ts[len(ts) - per:]
rows = []
for (t0, l0), (t1, _) in zip(ts, ts[1:]):
    m = re.search(r"RadixHashJoin (\d+) x (\d+)", l0)
    if m: rows.append((int(m.group(1)), int(m.group(2)), t1 - t0))
for lo, hi in ((0, 4096), (4096, 16384), (16384, 65536), (65536, 1 << 30)):
    sel = [x for x in rows if lo <= x[0] + x[1] < hi]
    if sel: print("nR + nS in [%d, %d): %d joins, %.2f ms in all, median %.3f ms" % (lo, hi, len(sel), sum(x[2] for x in sel), statistics.median(x[2] for x in sel)))
