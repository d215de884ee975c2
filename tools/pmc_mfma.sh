#!/bin/bash
# MFMA utilisation and issue counters per kernel (north_star: "MFMA utilisation on the cross-attention GEMMs"): separate
# rocprofv3 --pmc passes with --kernel-trace only, the program directly after `--`.
# run on the GPU box from the repo root:  bash tools/pmc_mfma.sh  ->  gpurun_out/pmcM{1,2,3}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-parity-mode"
timeout -k 10 280 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d $R/gpurun_out/pmcM1 -- $B > $R/gpurun_out/pmcM1.log 2>&1 &&
timeout -k 10 280 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmcM2 -- $B > $R/gpurun_out/pmcM2.log 2>&1 &&
timeout -k 10 280 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum --output-format csv -d $R/gpurun_out/pmcM3 -- $B > $R/gpurun_out/pmcM3.log 2>&1
ls $R/gpurun_out/pmcM1/*/ $R/gpurun_out/pmcM2/*/ $R/gpurun_out/pmcM3/*/ 2>&1 | head -20
for f in pmcM1 pmcM2 pmcM3; do tail -n 3 $R/gpurun_out/$f.log; done
