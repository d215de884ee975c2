#!/bin/bash
# A/B bench lines for profiles/ (round 2, end): the round-1 attention kernels and the four-wave weight-gradient tiling against the
# defaults, on one box, defaults first and last.  Outputs gpurun_out/r02_ab/*.json
set -e
O=gpurun_out/r02_ab; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-parity-mode --steps 12 --warmup 4"
$B > $O/cfg3_default_a.json
GG_ATTN_V1=1 GG_ATTN_DKV_V1=1 $B > $O/cfg3_attention_round1_kernels.json
GG_WGRAD_W4=1 $B > $O/cfg3_wgrad_four_waves.json
GG_NO_SIDE_WGRAD=1 $B > $O/cfg3_no_side_streams.json
$B > $O/cfg3_default_b.json
C="--variant img --batch 128 --genes 18000 --patches 1024 --steps 6 --warmup 3"
$B $C > $O/cfg5rank_default.json
GG_ATTN_LONG_V1=1 $B $C > $O/cfg5rank_attention_round1_kernels.json
for f in $O/*.json; do echo "$(basename $f): $(python -c "import json;d=json.load(open('$f'));print(d['ms_per_step'],'ms')")"; done
