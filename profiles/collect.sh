#!/bin/bash
# How the files in this directory were produced (run on the MI355X box through gpurun from the repo root):
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash profiles/collect.sh r01'
# Counters are collected in their own passes (kernel-trace only alongside), as MI355X_MICROARCH.md prescribes.
set -x
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. the default bench command, plain (the headline line) and under the kernel trace (+stats)
python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline > $OUT/bench_traced.json 2> $OUT/trace.err
# 2. HBM traffic of every kernel: FETCH_SIZE and WRITE_SIZE in separate passes, on a short variant of the same workload
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 $R/bench.py --no-cpu-baseline --no-prof --steps 1 --warmup 0 --pretrain-steps 20 --bn-settle-steps 0 > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err
done
python3 $R/profiles/summarize.py $OUT
du -sh $OUT; ls $OUT
