#!/bin/bash
# SQ counters per kernel for one config:  pmc.sh <config> <alpha> <tag>   -> gpurun_out/<tag>.csv
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
export CONFIG=$1 ALPHA=$2 IMK_SIDE_STREAMS=0
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/_pmc_$(basename $3) -- python3 $R/tests/gpu_probe/step_trace.py > /dev/null 2> $R/gpurun_out/_pmc_$(basename $3).err
python3 $R/tests/gpu_probe/pmc_summary.py $R/gpurun_out/_pmc_$(basename $3) > $R/gpurun_out/$3.csv 2>&1
rm -rf $R/gpurun_out/_pmc_$(basename $3) $R/gpurun_out/_pmc_$(basename $3).err
