#!/bin/bash
# SQ / LDS / wait counters of every kernel a probe command launches: two rocprofv3 --pmc passes (kernel-trace only, program right after `--`).
# usage (GPU box, repo root): bash tools/probe_pmc.sh <tag> python3 tools/<probe>.py args...   ->  gpurun_out/<tag>_pmc{1,2} + a per-kernel table on stdout
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
args=()
for a in "$@"; do case "$a" in tools/*|tests/*|bench.py) args+=("$R/$a");; *) args+=("$a");; esac; done
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${tag}_pmc1 -- "${args[@]}" > $R/gpurun_out/${tag}_pmc1.log 2>&1 &&
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/${tag}_pmc2 -- "${args[@]}" > $R/gpurun_out/${tag}_pmc2.log 2>&1 &&
cd $R && python3 tools/probe_pmc_summary.py gpurun_out/${tag}_pmc1/*/*_counter_collection.csv gpurun_out/${tag}_pmc2/*/*_counter_collection.csv
