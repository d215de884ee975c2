#!/bin/bash
# copies what tools/r02_measure.sh left under gpurun_out/ into profiles/ (run in the build container from the repo root)
set -e
for f in cfg3_bf16 cfg3_fp8 cfg5rank_bf16 cfg5rank_fp8 cfg1_vanilla cfg2_film_P1 cfg3_T300 cfg3_pad25 cfg3_graph cfg3_noprofile; do cp gpurun_out/r02/bench_$f.json profiles/r02_bench_$f.json; done
cp "$(ls -t gpurun_out/r02/trace/runc/*_kernel_stats.csv | head -1)" profiles/r02_final_kernel_stats.csv
python tools/pmc_traffic_summary.py "$(ls -t gpurun_out/pmc_fetch/runc/*_counter_collection.csv | head -1)" "$(ls -t gpurun_out/pmc_write/runc/*_counter_collection.csv | head -1)" profiles/r02_pmc_traffic.json > /dev/null
