"""RHJ_TRACE timeline of the device-resident engine on `small`: how long the 50 queries take after loading, and the
operators with the largest gaps before the next one (one trace line per librhj.so operator, stamped in ms)."""
import os, re, subprocess, sys, tempfile
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
REP = int(sys.argv[1]) if len(sys.argv) > 1 else 1      # batches: the last one is analysed (steady state: workspace grown, blocks cached)
stdin = ("\n".join(names) + "\nDone\n" + ("\n".join(g.small["work_lines"]) + "\n") * REP).encode()
exe = os.path.abspath(os.path.join("oracle", "_ref", "radixhash_rhj_resident"))
env = dict(os.environ, RHJ_TRACE="1", RHJ_RADIX_BITS="4")
r = subprocess.run([exe], input=stdin, cwd=tmp, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
lines = [l for l in r.stderr.decode().splitlines() if "rhj-trace" in l]
ts = []
for l in lines:
    m = re.search(r"rhj-trace\s+([0-9.]+)", l)
    if m: ts.append((float(m.group(1)), l))
if REP > 1:
    per = (len(ts) - 15) // REP
    ts = ts[len(ts) - per:]
print("%d operator calls; first at %.1f ms, last at %.1f ms: %.1f ms for the queries" % (len(ts), ts[0][0], ts[-1][0], ts[-1][0] - ts[0][0]))
ops = {}
for (t0, l0), (t1, _) in zip(ts, ts[1:]):
    name = l0.split()[3] if len(l0.split()) > 3 else "?"
    a = ops.setdefault(name, [0, 0.0]); a[0] += 1; a[1] += t1 - t0
for name, (n, tot) in sorted(ops.items(), key=lambda kv: -kv[1][1]):
    print("  %-34s %4d calls  %7.2f ms in all  %6.3f ms each (until the next operator starts)" % (name, n, tot, tot / n))
durs = sorted(((t1 - t0, l0) for (t0, l0), (t1, _) in zip(ts, ts[1:])), reverse=True)
print("longest operators (ms until the next one starts):")
for d, l in durs[:12]: print("   %.3f  %s" % (d, " ".join(l.split()[3:])[:110]))
j = sorted(d for d, l in durs if "RadixHashJoin" in l)
if j: print("RadixHashJoin: median %.3f ms, p90 %.3f, max %.3f, sum of the 10 longest %.2f ms" % (j[len(j) // 2], j[int(len(j) * 0.9)], j[-1], sum(j[-10:])))
misses = [l for l in r.stderr.decode().splitlines() if "hipMalloc" in l and "grows" not in l]
grows = [l for l in r.stderr.decode().splitlines() if "workspace buffer grows" in l]
print("workspace buffers grown (hipFree + hipMalloc each): %d" % len(grows))
print("device allocations that went to hipMalloc during the run: %d" % len(misses))
