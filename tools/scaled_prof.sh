#!/bin/bash
# tools/scaled_prof.sh <K> <bits>: kernel-time breakdown of the device-resident engine on the scaled `small` workload
K=$1; BITS=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 - "$K" <<'PY'
import os, sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import helpers
K = int(sys.argv[1]); g = helpers.Golden(); STRIDE = np.uint64(1 << 24)
os.makedirs("/tmp/scaled", exist_ok=True); names = []
for i in range(14):
    cols = g.small_relations["r%d" % i].astype("<u8")
    big = np.concatenate([cols + np.uint64(c) * STRIDE for c in range(K)], axis=1)
    with open("/tmp/scaled/r%d" % i, "wb") as f:
        np.array([big.shape[1], big.shape[0]], dtype="<u8").tofile(f); np.ascontiguousarray(big).tofile(f)
    names.append("r%d" % i)
open("/tmp/scaled/stdin.txt", "w").write("\n".join(names) + "\nDone\n" + "\n".join(g.small["work_lines"]) + "\n")
PY
export RHJ_RADIX_BITS=$BITS
cd /tmp/scaled
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/scaled_prof -- $GRAFT_REPO_ROOT/oracle/_ref/radixhash_rhj_resident < stdin.txt > /tmp/scaled/out.txt 2> /tmp/scaled/err.txt
cd $GRAFT_REPO_ROOT
f=$(ls -t gpurun_out/scaled_prof/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms" % (tot / 1e6))
for r in rows[:12]:
    print("%-44s calls %6s total %8.1f ms  %5.1f %%" % (r["Name"].split("(")[0][-44:], r["Calls"], float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
PY
