#!/bin/bash
# A/B of library variants (variants/*.so, git-ignored) at the headline shape: each is copied over the in-tree library in turn,
# interleaved twice.  usage: lib_ab.sh variants/a.so variants/b.so ...   (the last one stays in place)
set -e
mkdir -p gpurun_out/lib_ab
B="python bench.py --no-cpu-baseline --no-parity-mode --steps 12 --warmup 4"
for round in 1 2; do
  for v in "$@"; do
    cp $v gemm_gan_amd/libgemmgan.so
    $B > gpurun_out/lib_ab/$(basename $v .so)_$round.json
  done
done
for v in "$@"; do for round in 1 2; do echo "== $v round $round"; python tools/show_bench.py gpurun_out/lib_ab/$(basename $v .so)_$round.json | grep -E "ms/step|wgrad"; done; done
