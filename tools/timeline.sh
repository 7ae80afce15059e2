#!/bin/bash
# kernel timeline of the last joins of one bench.py run (start, duration, gap to the previous kernel's end):
# tools/timeline.sh <outdir-under-gpurun_out> <joins to print> <env assignments...> -- <bench args>
out=gpurun_out/$1; shift
last=$1; shift
envs=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do envs+=("$1"); shift; done
shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for e in "${envs[@]}"; do export "$e"; done
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/bench.json 2> $GRAFT_REPO_ROOT/$out/bench.err
cd $GRAFT_REPO_ROOT
f=$(find $out/prof -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$last" <<'PY' | tee $out/timeline.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# a join = the kernels from one histogram kernel to the next
first = [i for i, n in enumerate(names) if "k_small_hist" in n or "k_hist_tiles" in n or "k_rowid_sample" in n]
want = int(sys.argv[2])
for j in first[-want - 1:-1]:
    nxt = [i for i in first if i > j]
    end = nxt[0] if nxt else len(rows)
    t0 = int(rows[j]["Start_Timestamp"]); prev_end = None
    print("join at %d:" % j)
    for r in rows[j:end]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        print("   %-70s start %7.1f us  dur %7.1f us  gap %6.1f us" % (r["Kernel_Name"][:70], (s - t0) / 1e3, (e - s) / 1e3, gap))
        prev_end = e
    print("   span %.1f us; to next join's first kernel %.1f us" % ((prev_end - t0) / 1e3, (int(rows[end]["Start_Timestamp"]) - t0) / 1e3 if end < len(rows) else -1))
PY
