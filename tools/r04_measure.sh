#!/bin/bash
# Round-4 measurement batch (run on the GPU box from the repo root): the driver's default bench line, the bench lines of the other
# configurations quoted in DESIGN.md, the rocprofv3 kernel-trace stats of the default command and the PMC passes (HBM traffic,
# MFMA / issue counters).  Outputs under gpurun_out/r04/.   usage: bash tools/r04_measure.sh [quick | lines]   (quick: headline + profiles only; lines: the other bench lines and the A/B lines only)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
B="--no-cpu-baseline --no-parity-mode"
if [ "$1" != "lines" ]; then
python bench.py > $O/bench_cfg3_bf16.json 2> $O/bench_cfg3_bf16.err &&
python bench.py $B --no-profile --steps 30 > $O/bench_cfg3_noprofile30.json 2> $O/bench_cfg3_noprofile30.err || exit 1
fi
if [ "$1" != "quick" ]; then
python bench.py $B --precision fp8 > $O/bench_cfg3_fp8.json 2> $O/bench_cfg3_fp8.err &&
python bench.py $B --variant img --batch 128 --genes 18000 --patches 1024 --steps 8 --warmup 2 > $O/bench_cfg5rank_bf16.json 2> $O/bench_cfg5rank_bf16.err &&
python bench.py $B --variant img --batch 128 --genes 18000 --patches 1024 --steps 8 --warmup 2 --precision fp8 > $O/bench_cfg5rank_fp8.json 2> $O/bench_cfg5rank_fp8.err &&
python bench.py $B --variant vanilla --batch 64 --genes 1000 --dropout 0 --steps 200 > $O/bench_cfg1_vanilla.json 2> $O/bench_cfg1_vanilla.err &&
python bench.py $B --variant film --patches 1 --steps 100 > $O/bench_cfg2_film_P1.json 2> $O/bench_cfg2_film_P1.err &&
python bench.py $B --tokens 300 --text-dims 768 > $O/bench_cfg3_T300.json 2> $O/bench_cfg3_T300.err &&
python bench.py $B --pad-frac 0.25 > $O/bench_cfg3_pad25.json 2> $O/bench_cfg3_pad25.err &&
GG_NO_PREFETCH_WIDE=1 python bench.py $B --no-profile > $O/ab_prefetch_1_3_1.json 2> $O/ab_prefetch_1_3_1.err &&
GG_NO_PREFETCH_WIDE=1 GG_NO_PREFETCH_PIPE=1 python bench.py $B --no-profile > $O/ab_prefetch_3_2.json 2> $O/ab_prefetch_3_2.err &&
GG_FFN2=1 python bench.py $B --no-profile > $O/ab_streamed_ffn.json 2> $O/ab_streamed_ffn.err &&
GG_ENCB=1 python bench.py $B --no-profile > $O/ab_fused_layer_backward.json 2> $O/ab_fused_layer_backward.err &&
GPU_MAX_HW_QUEUES=8 python bench.py $B --no-profile > $O/ab_hw_queues_8.json 2> $O/ab_hw_queues_8.err &&
python bench.py $B --no-profile > $O/ab_default.json 2> $O/ab_default.err || exit 1
fi
[ "$1" == "lines" ] && exit 0
cd /tmp && export TMPDIR=/tmp &&
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 5 --warmup 2 $B > $O/trace.log 2>&1 &&
cd $R && bash tools/pmc_traffic.sh && bash tools/pmc_mfma.sh > $O/pmc_mfma.log 2>&1
