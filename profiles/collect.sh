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
# 2. HBM traffic of every kernel: FETCH_SIZE and WRITE_SIZE in separate passes of the same command (summarize.py keeps
#    the launches of the generations only, i.e. the launch mix of the timed region)
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 $R/bench.py --no-cpu-baseline --no-prof > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err
done
# 3. timeline of one training step and one ensemble-sized inference call (kernel by kernel), and the SQ counters that say
#    what the waves wait for; both on tests/gpu_probe/step_trace.py (6 training steps + 3 inference calls)
rocprofv3 --kernel-trace --output-format csv -d $OUT/steptrace -- python3 $R/tests/gpu_probe/step_trace.py > /dev/null 2> $OUT/steptrace.err
python3 $R/tests/gpu_probe/trace_summary.py $OUT/steptrace > $OUT/step_timeline.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --output-format csv -d $OUT/sq -- python3 $R/tests/gpu_probe/step_trace.py > /dev/null 2> $OUT/sq.err
python3 $R/tests/gpu_probe/pmc_summary.py $OUT/sq > $OUT/sq_counters.csv 2>&1
python3 $R/profiles/summarize.py $OUT
du -sh $OUT; ls $OUT
