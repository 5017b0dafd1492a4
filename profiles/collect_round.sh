#!/bin/bash
# Rounds 5+: how the rNN_* files in this directory were produced (on the MI355X box through gpurun, from the repo root):
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'RND=rNN bash profiles/collect_round.sh'
# then copied from gpurun_out/${RND}/ into profiles/ by profiles/install_round.py rNN.  Counters are collected in their own passes
# (kernel-trace only alongside), the program itself directly after `--`, as MI355X_MICROARCH.md prescribes.
#   ONLY=bench   : steps 1-2 only (the default command plain / traced / counter passes)
#   ONLY=times   : step 6 only
RND=${RND:-r05}
set -x
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/${RND}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
step_times() {
cd $R
{ for C in "isic 0.5" "hela 1" "suim 1" "city 1" "city 1.25" "city 1.5" "city 1.75" "city 2" "isic 1.5"; do set -- $C
    CONFIG=$1 ALPHA=$2 python3 tests/gpu_probe/step_time.py 2>&1 | grep -E "config|train step|inference"; done
  python3 tests/gpu_probe/evalnet_time.py 2>&1 | tail -3; } > $OUT/configs_step_times_raw.txt
cd /tmp
}
if [ "$ONLY" = "times" ]; then step_times; exit 0; fi
# 1. the default bench command (BASELINE configs[1] + other_configs), plain and under the kernel trace (+stats): the traced run's
#    line carries the library's per-kernel byte totals of the timed region, the trace its marker dispatches
python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline --no-other-configs > $OUT/bench_traced.json 2> $OUT/trace.err
# 2. HBM traffic of every kernel: FETCH_SIZE and WRITE_SIZE in separate passes of the same command
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --no-prof > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err
done
python3 $R/profiles/summarize.py $OUT
if [ "$ONLY" = "bench" ]; then du -sh $OUT; exit 0; fi
# 3. the other BASELINE shapes: bench lines (with their CPU baselines), kernel stats and HBM traffic
for CFG in suim cityscapes hela; do
  python3 $R/bench.py --config $CFG --steps 2 > $OUT/bench_$CFG.json 2> $OUT/bench_$CFG.err
done
python3 $R/bench.py --config cityscapes --alpha 2 --steps 2 --no-cpu-baseline > $OUT/bench_cityscapes_a2.json 2> $OUT/bench_cityscapes_a2.err
for CFG in suim cityscapes hela cityscapes_a2; do
  A=""; N=$CFG; [ "$CFG" = "cityscapes_a2" ] && { A="--alpha 2"; N=cityscapes; }
  mkdir -p $OUT/cfg_$CFG
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg_$CFG/trace -- python3 $R/bench.py --config $N $A --steps 1 --no-cpu-baseline > $OUT/cfg_$CFG/bench_traced.json 2> $OUT/cfg_$CFG/trace.err
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/cfg_$CFG/pmc_$C -- python3 $R/bench.py --config $N $A --steps 1 --no-cpu-baseline --no-prof > /dev/null 2> $OUT/cfg_$CFG/pmc_$C.err
  done
  python3 $R/profiles/summarize.py $OUT/cfg_$CFG
done
# 4. kernel-by-kernel timelines of one training step and one inference call
for C in "isic 0.5" "suim 1" "city 1" "city 2"; do set -- $C
  GRAFT_REPO_ROOT=$R bash $R/tests/gpu_probe/trace.sh $1 $2 ${RND}/step_timeline_$1_a$2
  GRAFT_REPO_ROOT=$R bash $R/tests/gpu_probe/trace.sh $1 $2 ${RND}/step_timeline_single_stream_$1_a$2 single
done
# 5. SQ counters (what the waves wait for)
GRAFT_REPO_ROOT=$R bash $R/tests/gpu_probe/pmc.sh isic 0.5 ${RND}/sq_counters_isic
GRAFT_REPO_ROOT=$R bash $R/tests/gpu_probe/pmc.sh city 2 ${RND}/sq_counters_city_a2
# 6. wall time of a training step / an inference call for every shape and the IM+ width schedule, and EvalNet
step_times
du -sh $OUT; ls -R $OUT | head -80
