#!/bin/bash
# Round-2 measurement batch (run on the GPU box from the repo root): bench lines for every configuration quoted in DESIGN.md,
# the rocprofv3 kernel-trace stats of the default command, PMC traffic passes.  Outputs under gpurun_out/r02/.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O
cd $R
B="--no-cpu-baseline --no-parity-mode"
python bench.py > $O/bench_cfg3_bf16.json 2> $O/bench_cfg3_bf16.err &&
python bench.py $B --precision fp8 > $O/bench_cfg3_fp8.json 2> $O/bench_cfg3_fp8.err &&
python bench.py $B --variant img --batch 128 --genes 18000 --patches 1024 --steps 8 --warmup 2 > $O/bench_cfg5rank_bf16.json 2> $O/bench_cfg5rank_bf16.err &&
python bench.py $B --variant img --batch 128 --genes 18000 --patches 1024 --steps 8 --warmup 2 --precision fp8 > $O/bench_cfg5rank_fp8.json 2> $O/bench_cfg5rank_fp8.err &&
python bench.py $B --variant vanilla --batch 64 --genes 1000 --dropout 0 --steps 200 > $O/bench_cfg1_vanilla.json 2> $O/bench_cfg1_vanilla.err &&
python bench.py $B --variant film --patches 1 --steps 100 > $O/bench_cfg2_film_P1.json 2> $O/bench_cfg2_film_P1.err &&
python bench.py $B --tokens 300 --text-dims 768 > $O/bench_cfg3_T300.json 2> $O/bench_cfg3_T300.err &&
python bench.py $B --pad-frac 0.25 > $O/bench_cfg3_pad25.json 2> $O/bench_cfg3_pad25.err &&
python bench.py $B --no-profile --graph --steps 20 > $O/bench_cfg3_graph.json 2> $O/bench_cfg3_graph.err &&
python bench.py $B --no-profile --steps 20 > $O/bench_cfg3_noprofile.json 2> $O/bench_cfg3_noprofile.err &&
cd /tmp && export TMPDIR=/tmp &&
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 5 --warmup 2 $B > $O/trace.log 2>&1 &&
cd $R && bash tools/pmc_traffic.sh
