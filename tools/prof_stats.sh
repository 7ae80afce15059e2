#!/bin/bash
# rocprofv3 kernel stats of one bench.py run: tools/prof_stats.sh <outdir-under-gpurun_out> <env assignments...> -- <bench args>
out=gpurun_out/$1; shift
envs=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do envs+=("$1"); shift; done
shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for e in "${envs[@]}"; do export "$e"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/bench.json 2> $GRAFT_REPO_ROOT/$out/bench.err
cd $GRAFT_REPO_ROOT
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
cp "$f" $out/kernel_stats.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$out/kernel_stats.csv")))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:22]:
    print("%-90s calls %5s avg %10.1f us  total %8.2f ms  %5.1f%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, float(r["Percentage"])))
PY
