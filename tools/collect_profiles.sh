#!/bin/bash
# tools/collect_profiles.sh <tag> — bench lines, rocprofv3 kernel stats and FETCH/WRITE counters of one build
tag=$1
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 bench.py > gpurun_out/${tag}_c3_bench.json 2> gpurun_out/${tag}_c3_bench.err
echo "c3 bench done"
python3 bench.py --workload c2 --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/${tag}_c2_bench.json 2>/dev/null
python3 bench.py --workload c4 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_c4_bench.json 2>/dev/null
echo "c2 c4 bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_prof.log 2>&1
cp $(ls gpurun_out/${tag}_prof/*/*kernel_stats.csv | head -1) gpurun_out/${tag}_c3_kernel_stats.csv
echo "kernel stats done"
bash tools/pmc.sh ${tag} "FETCH_SIZE" "WRITE_SIZE" -- --steps 3 --warmup 2 --no-cpu-baseline
python3 tools/pmc_summary.py ${tag} > /dev/null
cp gpurun_out/pmc_${tag}.json gpurun_out/${tag}_c3_pmc.json
python3 tools/filter_bench.py > gpurun_out/${tag}_filter.json 2>/dev/null || echo "filter bench failed"
echo "all done"
