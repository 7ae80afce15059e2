"""A/B/C... timing of several builds of librhj.so on ONE device: tools/abn.py <workload> <rounds> <lib1.so> <lib2.so> ...
Like tools/ab.py (each build in its OWN process, round-robin, order reversed every other round) for any number of builds.
A build may carry environment settings: path/librhj.so@RHJ_NO_RUNS12=1@RHJ_X=2.
Prints median / min of the per-stage GPU times of rhj_join_device and each build's ratio to the first."""
import json, os, statistics, subprocess, sys
sys.path.insert(0, "tools")
from ab import CHILD

wl, rounds, libs = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
keys = ("ms_hist", "ms_scan", "ms_scatter", "ms_plan", "ms_probe", "ms_total")
acc = {l: {k: [] for k in keys} for l in libs}
for blk in range(rounds):
    for path in (libs if blk % 2 == 0 else libs[::-1]):
        parts = path.split("@")
        env = dict(os.environ, **dict(kv.split("=", 1) for kv in parts[1:]))
        res = subprocess.run([sys.executable, "-c", CHILD, wl, parts[0], "6"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        line = [l for l in res.stdout.decode().splitlines() if l.startswith("ABJSON ")]
        if not line:
            print(path, "FAILED", res.stderr.decode()[-1500:])
            continue
        d = json.loads(line[0][7:])
        for k in keys:
            acc[path][k] += d[k]
print("workload", wl)
for k in keys:
    base = statistics.median(acc[libs[0]][k]) if acc[libs[0]][k] else 0.0
    print("%-11s" % k + " | ".join("%s med %.4f min %.4f (%.3f)" % (l.split("/")[-1].replace("librhj", "").replace(".so", "") or "cur", statistics.median(acc[l][k]), min(acc[l][k]),
                                    statistics.median(acc[l][k]) / max(base, 1e-9)) for l in libs if acc[l][k]))
