#!/bin/bash
# A/B of the attention kernel variants at the headline shape: per-class event timings of bench.py, interleaved.
# usage: attn_ab.sh "ENV=1 ENV2=1" "..." : one quoted environment assignment list per variant ("" = defaults)
set -e
mkdir -p gpurun_out/attn_ab
B="python bench.py --no-cpu-baseline --no-parity-mode --steps 12 --warmup 4"
for round in 1 2; do
  i=0
  for v in "$@"; do
    env $v $B > gpurun_out/attn_ab/v${i}_$round.json
    i=$((i+1))
  done
done
i=0
for v in "$@"; do
  for round in 1 2; do
    echo "== [$v] round $round"; python tools/show_bench.py gpurun_out/attn_ab/v${i}_$round.json | grep -E "ms/step|attn"
  done
  i=$((i+1))
done
