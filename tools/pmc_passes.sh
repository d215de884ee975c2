cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-parity-mode"
timeout -k 10 250 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM --output-format csv -d $R/gpurun_out/pmcA -- $B > $R/gpurun_out/pmcA.log 2>&1 &&
timeout -k 10 250 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM --output-format csv -d $R/gpurun_out/pmcB -- $B > $R/gpurun_out/pmcB.log 2>&1 &&
timeout -k 10 250 rocprofv3 --kernel-trace --pmc TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL TCC_EA0_WRREQ_STALL TCC_BUSY --output-format csv -d $R/gpurun_out/pmcC -- $B > $R/gpurun_out/pmcC.log 2>&1
ls $R/gpurun_out/pmcA/*/ $R/gpurun_out/pmcB/*/ $R/gpurun_out/pmcC/*/
