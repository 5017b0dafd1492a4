#!/bin/bash
# L2 / fabric counters per kernel for one config:  pmc_l2.sh <config> <alpha> <tag>   -> gpurun_out/<tag>.csv
# (separate passes per counter set, kernel-trace only alongside, the program directly after `--`)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
export CONFIG=$1 ALPHA=$2 IMK_SIDE_STREAMS=0
D=$R/gpurun_out/_l2_$(basename $3)
rm -rf $D; mkdir -p $D
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D/p$i -- python3 $R/tests/gpu_probe/step_trace.py > /dev/null 2> $D/p$i.err
done
python3 $R/tests/gpu_probe/pmc_l2_summary.py $D > $R/gpurun_out/$3.csv 2>&1
rm -rf $D
