#!/bin/bash
# tools/pmc_abl.sh <tag> "<counters>"  — SQ counters of the fused kernel with parts removed (RHJ_ABLATE 0/1/3)
tag=$1; ctr=$2
# needs the diagnostics build: make -C sigmod-2018_amd instr
export RHJ_LIB=${RHJ_LIB:-$GRAFT_REPO_ROOT/sigmod-2018_amd/librhj_instr.so}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for m in 0 1 3; do
  export RHJ_ABLATE=$m
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}${m}_0 -- python3 tools/exp_abl.py 100000000 100000000 > gpurun_out/pmc_${tag}${m}.log 2>&1 || { echo "pass $m failed"; tail -5 gpurun_out/pmc_${tag}${m}.log; exit 1; }
  echo "== ablate $m"; python3 tools/pmc_summary.py ${tag}${m} | grep fused
done
