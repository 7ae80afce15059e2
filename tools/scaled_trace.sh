#!/bin/bash
# tools/scaled_trace.sh <K> <bits>: RHJ_TRACE timeline of the device-resident engine on the scaled workload (files from scaled_prof.sh)
K=$1; BITS=$2
cd "$GRAFT_REPO_ROOT"
[ -f /tmp/scaled/stdin.txt ] || { echo "run tools/scaled_prof.sh first in the same call"; exit 1; }
cd /tmp/scaled
RHJ_TRACE=1 RHJ_RADIX_BITS=$BITS $GRAFT_REPO_ROOT/oracle/_ref/radixhash_rhj_resident < stdin.txt > out.txt 2> trace.txt
grep -c rhj-trace trace.txt
awk '/rhj-trace/ { t=$2+0; if (t - last > 20) print "gap", t - last, "ms before:", $0; last=t }' trace.txt | head -40
tail -3 trace.txt
