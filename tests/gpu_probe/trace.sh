#!/bin/bash
# kernel timeline of one training step + one inference call:  trace.sh <config> <alpha> <tag> [single]
#   -> gpurun_out/<tag>.txt   (single: IMK_SIDE_STREAMS=0, every kernel alone on the stream = exclusive durations)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
export CONFIG=$1 ALPHA=$2
[ "$4" = "single" ] && export IMK_SIDE_STREAMS=0
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/_tr_$(basename $3) -- python3 $R/tests/gpu_probe/step_trace.py > /dev/null 2> $R/gpurun_out/_tr_$(basename $3).err
python3 $R/tests/gpu_probe/trace_summary.py $R/gpurun_out/_tr_$(basename $3) > $R/gpurun_out/$3.txt 2>&1
rm -rf $R/gpurun_out/_tr_$(basename $3) $R/gpurun_out/_tr_$(basename $3).err
