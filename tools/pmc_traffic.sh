#!/bin/bash
# HBM traffic per kernel launch: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes (MI355X_MICROARCH.md, HBM section).
# run on the GPU box from the repo root:  bash tools/pmc_traffic.sh  ->  gpurun_out/pmc_fetch, gpurun_out/pmc_write
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-parity-mode"
timeout -k 10 280 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- $B > $R/gpurun_out/pmc_fetch.log 2>&1 &&
timeout -k 10 280 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- $B > $R/gpurun_out/pmc_write.log 2>&1
