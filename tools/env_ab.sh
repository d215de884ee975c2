#!/bin/bash
# A/B of environment switches at the headline shape, interleaved twice; prints the step time and the classes matching $PAT.
# usage: PAT=wgrad env_ab.sh "GG_X=0" "GG_WGRAD_W8=1" ...
set -e
mkdir -p gpurun_out/env_ab
B="python bench.py --no-cpu-baseline --no-parity-mode --steps 12 --warmup 4"
for round in 1 2; do i=0; for v in "$@"; do env $v $B > gpurun_out/env_ab/v${i}_$round.json; i=$((i+1)); done; done
i=0; for v in "$@"; do for round in 1 2; do echo "== [$v] round $round"; python tools/show_bench.py gpurun_out/env_ab/v${i}_$round.json | grep -E "ms/step|${PAT:-xxxx}"; done; i=$((i+1)); done
