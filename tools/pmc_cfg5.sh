#!/bin/bash
# PMC passes at the per-GPU shape of configs[4] (S = 1 025: the streaming attention kernels); run on the GPU box from the repo root
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python3 $R/bench.py --variant img --batch 128 --genes 18000 --patches 1024 --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-parity-mode $BENCH_EXTRA"
timeout -k 10 280 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d $R/gpurun_out/c5M1 -- $B > $R/gpurun_out/c5M1.log 2>&1 &&
timeout -k 10 280 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/c5M2 -- $B > $R/gpurun_out/c5M2.log 2>&1 &&
timeout -k 10 280 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS --output-format csv -d $R/gpurun_out/c5M3 -- $B > $R/gpurun_out/c5M3.log 2>&1
tail -2 $R/gpurun_out/c5M1.log $R/gpurun_out/c5M2.log $R/gpurun_out/c5M3.log
