#!/bin/bash
# counters of ffn_fused_kernel / the two-launch route alone: bash tools/ffn_pmc.sh -> gpurun_out/ffnpmc{1,2}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python3 $R/tools/ffn_probe.py 197376 3"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/ffnpmc1 -- $B > $R/gpurun_out/ffnpmc1.log 2>&1 &&
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/ffnpmc2 -- $B > $R/gpurun_out/ffnpmc2.log 2>&1
tail -n 3 $R/gpurun_out/ffnpmc1.log; tail -n 3 $R/gpurun_out/ffnpmc2.log
