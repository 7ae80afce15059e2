#!/bin/bash
# kernel-trace stats for several library variants: tools/prof_kern.sh <outdir> <pattern> <bench args...> ; variants from $RHJ_VARIANTS (space separated lib names)
out=gpurun_out/$1; pat=$2; shift 2
mkdir -p $out
for v in $RHJ_VARIANTS; do
  lib=$GRAFT_REPO_ROOT/sigmod-2018_amd/$v
  ( cd /tmp && export TMPDIR=/tmp RHJ_LIB=$lib && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/$v -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/$v.json 2> $GRAFT_REPO_ROOT/$out/$v.err )
  f=$(find $out/$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"
  python3 - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows:
    if any(p in r["Name"] for p in "$pat".split(",")):
        print("  %-60s calls %4s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
