#!/bin/bash
# tools/pmc.sh <tag> <counter-set> [<counter-set> ...] -- <bench args>
# Runs one rocprofv3 --pmc pass per counter set (never combined with tracing domains other
# than --kernel-trace) and leaves CSVs under gpurun_out/pmc_<tag>_<i>/.
tag=$1; shift
sets=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do sets+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for s in "${sets[@]}"; do
  rocprofv3 --pmc $s --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_$i -- python3 bench.py "$@" > gpurun_out/pmc_${tag}_$i.log 2>&1 || { echo "pass $i failed"; tail -5 gpurun_out/pmc_${tag}_$i.log; exit 1; }
  i=$((i+1))
done
echo "pmc passes done: $i"
