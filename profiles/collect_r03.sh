#!/bin/bash
# Round 3: how the r03_* files in this directory were produced (on the MI355X box through gpurun, from the repo root):
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash profiles/collect_r03.sh'
# then copied from gpurun_out/r03/ into profiles/ by profiles/install_r03.py.  Counters are collected in their own passes
# (kernel-trace only alongside), the program itself directly after `--`, as MI355X_MICROARCH.md prescribes.
set -x
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ "$ONLY" = "times" ]; then    # refresh only the step / inference wall times (section 6)
cd $R
{ for C in "isic 0.5" "hela 1" "suim 1" "city 1" "city 1.25" "city 1.5" "city 1.75" "city 2" "isic 1.5"; do set -- $C
    CONFIG=$1 ALPHA=$2 python3 tests/gpu_probe/step_time.py 2>&1 | grep -E "config|train step|inference"; done
  python3 tests/gpu_probe/evalnet_time.py 2>&1 | tail -3; } > $OUT/configs_step_times_raw.txt
exit 0
fi
if [ "$ONLY" = "pmc" ]; then      # HBM traffic passes for the other configurations (bench.py reads r03_pmc_traffic_<config>.csv)
for CFG in suim cityscapes hela; do
  mkdir -p $OUT/cfg_$CFG
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/cfg_$CFG/pmc_$C -- python3 $R/bench.py --config $CFG --steps 1 --no-cpu-baseline --no-prof > /dev/null 2> $OUT/cfg_$CFG/pmc_$C.err
  done
  python3 $R/profiles/summarize.py $OUT/cfg_$CFG
done
exit 0
fi
# 1. the default bench command (BASELINE configs[1]), plain and under the kernel trace (+stats)
python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline > $OUT/bench_traced.json 2> $OUT/trace.err
# 2. HBM traffic of every kernel: FETCH_SIZE and WRITE_SIZE in separate passes of the same command
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 $R/bench.py --no-cpu-baseline --no-prof > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err
done
python3 $R/profiles/summarize.py $OUT
# 3. the other BASELINE shapes: bench lines (with their CPU baselines), kernel stats, and the counters for Cityscapes
for CFG in suim cityscapes hela; do
  python3 $R/bench.py --config $CFG --steps 2 > $OUT/bench_$CFG.json 2> $OUT/bench_$CFG.err
done
python3 $R/bench.py --config cityscapes --alpha 2 --steps 2 --no-cpu-baseline > $OUT/bench_cityscapes_a2.json 2> $OUT/bench_cityscapes_a2.err
for CFG in suim cityscapes hela; do
  mkdir -p $OUT/cfg_$CFG
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg_$CFG/trace -- python3 $R/bench.py --config $CFG --steps 1 --no-cpu-baseline > /dev/null 2> $OUT/cfg_$CFG/trace.err
  python3 $R/profiles/summarize.py $OUT/cfg_$CFG
done
mkdir -p $OUT/cfg_cityscapes_a2
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg_cityscapes_a2/trace -- python3 $R/bench.py --config cityscapes --alpha 2 --steps 1 --no-cpu-baseline > /dev/null 2> $OUT/cfg_cityscapes_a2/trace.err
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/cfg_cityscapes_a2/pmc_$C -- python3 $R/bench.py --config cityscapes --alpha 2 --steps 1 --no-cpu-baseline --no-prof > /dev/null 2> $OUT/cfg_cityscapes_a2/pmc_$C.err
done
python3 $R/profiles/summarize.py $OUT/cfg_cityscapes_a2
# 4. kernel-by-kernel timelines of one training step and one inference call
for C in "isic 0.5" "suim 1" "city 1" "city 2"; do set -- $C
  GRAFT_REPO_ROOT=$R bash $R/tests/gpu_probe/trace.sh $1 $2 r03/step_timeline_$1_a$2
  GRAFT_REPO_ROOT=$R bash $R/tests/gpu_probe/trace.sh $1 $2 r03/step_timeline_single_stream_$1_a$2 single
done
# 5. SQ counters (what the waves wait for)
GRAFT_REPO_ROOT=$R bash $R/tests/gpu_probe/pmc.sh isic 0.5 r03/sq_counters_isic
GRAFT_REPO_ROOT=$R bash $R/tests/gpu_probe/pmc.sh city 2 r03/sq_counters_city_a2
# 6. wall time of a training step / an inference call for every shape and the IM+ width schedule, and EvalNet
cd $R
{ for C in "isic 0.5" "hela 1" "suim 1" "city 1" "city 1.25" "city 1.5" "city 1.75" "city 2" "isic 1.5"; do set -- $C
    CONFIG=$1 ALPHA=$2 python3 tests/gpu_probe/step_time.py 2>&1 | grep -E "config|train step|inference"; done
  python3 tests/gpu_probe/evalnet_time.py 2>&1 | tail -3; } > $OUT/configs_step_times_raw.txt
du -sh $OUT; ls -R $OUT | head -80
