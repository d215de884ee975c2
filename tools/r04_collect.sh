#!/bin/bash
# copies what tools/r04_measure.sh left under gpurun_out/ into profiles/ (run in the build container from the repo root)
set -e
for f in gpurun_out/r04/bench_*.json gpurun_out/r04/ab_*.json; do b=$(basename $f); cp $f profiles/r04_$b; done
cp "$(ls -t gpurun_out/r04/trace/runc/*_kernel_stats.csv | head -1)" profiles/r04_kernel_stats.csv
python tools/pmc_traffic_summary.py "$(ls -t gpurun_out/pmc_fetch/runc/*_counter_collection.csv | head -1)" "$(ls -t gpurun_out/pmc_write/runc/*_counter_collection.csv | head -1)" profiles/r04_pmc_traffic.json > /dev/null
python tools/pmc_mfma_summary.py "$(ls -t gpurun_out/pmcM1/runc/*_counter_collection.csv | head -1)" "$(ls -t gpurun_out/pmcM2/runc/*_counter_collection.csv | head -1)" "$(ls -t gpurun_out/pmcM3/runc/*_counter_collection.csv | head -1)" > profiles/r04_pmc_mfma.json
