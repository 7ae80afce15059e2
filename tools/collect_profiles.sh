#!/bin/bash
# tools/collect_profiles.sh <tag> — bench lines, rocprofv3 kernel stats and FETCH/WRITE counters of one build
tag=$1
head=${2:-unknown}      # git head of the build (the GPU box has no .git): tools/collect_profiles.sh <tag> $(git rev-parse --short HEAD)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 bench.py > gpurun_out/${tag}_c3_bench.json 2> gpurun_out/${tag}_c3_bench.err
echo "c3 bench done"
python3 bench.py --workload c2 --steps 200 --warmup 20 > gpurun_out/${tag}_c2_bench.json 2>/dev/null
python3 bench.py --workload c4 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_c4_bench.json 2>/dev/null
python3 bench.py --workload c3 --scaling strong --steps 5 --warmup 2 > gpurun_out/${tag}_c3_strong_1gpu.json 2>/dev/null
python3 bench.py --workload c3b4 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_c3b4_bench.json 2>/dev/null
python3 bench.py --workload c3b14 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_c3b14_bench.json 2>/dev/null
python3 bench.py --workload small --steps 20 --warmup 3 > gpurun_out/${tag}_small_bench.json 2>/dev/null
echo "c2 c4 strong c3b4 c3b14 small bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_prof.log 2>&1
cp $(ls gpurun_out/${tag}_prof/*/*kernel_stats.csv | head -1) gpurun_out/${tag}_c3_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof_b4 -- python3 bench.py --workload c3b4 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_prof_b4.log 2>&1
cp $(ls gpurun_out/${tag}_prof_b4/*/*kernel_stats.csv | head -1) gpurun_out/${tag}_c3b4_kernel_stats.csv
echo "kernel stats done"
bash tools/pmc.sh ${tag} "FETCH_SIZE" "WRITE_SIZE" -- --steps 3 --warmup 2 --no-cpu-baseline
python3 tools/pmc_summary.py ${tag} $head > /dev/null
cp gpurun_out/pmc_${tag}.json gpurun_out/${tag}_c3_pmc.json
python3 tools/filter_bench.py > gpurun_out/${tag}_filter.json 2>/dev/null || echo "filter bench failed"
python3 tools/host_rate.py > gpurun_out/${tag}_host_rate.json 2>/dev/null || echo "host rate failed"
python3 tools/small_e2e.py > gpurun_out/${tag}_small_e2e.json 2>/dev/null || echo "small e2e failed"

# SQ counters of the C3 kernels (three passes; per launch, summed over the chip)
bash tools/pmc.sh ${tag}sq "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" -- --steps 3 --warmup 2 --no-cpu-baseline || echo "sq passes failed"
python3 tools/pmc_summary.py ${tag}sq $head > /dev/null && cp gpurun_out/pmc_${tag}sq.json gpurun_out/${tag}_c3_pmc_sq.json
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_c3_bench_after_pmc.json 2>/dev/null
echo "sq done"
echo "all done"
