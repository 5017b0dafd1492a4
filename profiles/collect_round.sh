#!/bin/bash
# Round 6+: how the rNN_* files in this directory are produced (on the MI355X box through gpurun, from the repo root):
#   /usr/local/graft/bin/gpurun --timeout 3000 -- 'RND=r06 bash profiles/collect_round.sh'
# then copied from gpurun_out/${RND}/ into profiles/ by profiles/install_round.py rNN.  Counters are collected in their own passes
# (kernel-trace only alongside), the program itself directly after `--`, as MI355X_MICROARCH.md prescribes.
# Every configuration gets ITS OWN files (bench.py replays a value only from the files of the configuration it runs, and only when
# provenance.json -- bench.py's sha256 and the kernel sources' id at collection time -- names the running tree):
#   cfg<tag>/bench.json + bench_detail.json      the plain command (compact last line + the full record)
#   cfg<tag>/kernel_stats.csv                    rocprofv3 --kernel-trace --stats, whole process
#   cfg<tag>/timed_region_kernel_stats.csv       the trace cut at bench.py's markers, per stage and variant, with algorithmic bytes
#   cfg<tag>/excess_by_kernel.csv                per variant: floor_us = max(bytes / 8 TB/s, flops / 2.5 PFLOP/s), excess_ms, sorted
#   cfg<tag>/pmc_traffic.csv                     --pmc FETCH_SIZE / WRITE_SIZE passes (PMC=0 skips them)
# CONFIGS="isic suim ..." picks configurations (default: all six); PMC=0: traces only; ONLY=times: step 6 only; EXTRAS=0: steps 1-3 only
RND=${RND:-r06}
PMC=${PMC:-1}
EXTRAS=${EXTRAS:-1}
CONFIGS=${CONFIGS:-"isic suim cityscapes hela cityscapes_a2 cityscapes_a125"}
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
for CFG in $CONFIGS; do
  case $CFG in
    isic)            ARGS="";                                   PLAIN="";                    TAG="" ;;
    suim|cityscapes|hela) ARGS="--config $CFG";                 PLAIN="--steps 2";           TAG="_$CFG" ;;
    cityscapes_a2)   ARGS="--config cityscapes --alpha 2";      PLAIN="--steps 2 --no-cpu-baseline"; TAG="_cityscapes_a2" ;;
    cityscapes_a125) ARGS="--config cityscapes --alpha 1.25";   PLAIN="--steps 2 --no-cpu-baseline"; TAG="_cityscapes_a125" ;;
  esac
  D=$OUT/cfg$TAG; [ -z "$TAG" ] && D=$OUT/cfg_isic
  mkdir -p $D
  python3 $R/bench.py --provenance > $D/provenance.json
  # 1. the plain command (the default one carries the CPU baseline and other_configs), then the same under the kernel trace (+stats):
  #    the traced run's full record holds the library's per-kernel byte totals of the timed region, the trace its marker dispatches
  python3 $R/bench.py $ARGS $PLAIN --detail $D/bench_detail.json > $D/bench.json 2> $D/bench.err
  TR="--no-cpu-baseline --no-other-configs"; [ -n "$TAG" ] && TR="--no-cpu-baseline --steps 1"
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- python3 $R/bench.py $ARGS $TR --detail $D/bench_traced_detail.json > $D/bench_traced.json 2> $D/trace.err
  # 2. HBM traffic of every kernel: FETCH_SIZE and WRITE_SIZE in separate passes of the same command
  if [ "$PMC" = "1" ]; then
    for C in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D/pmc_$C -- python3 $R/bench.py $ARGS $TR --no-prof --detail $D/pmc_${C}_detail.json > /dev/null 2> $D/pmc_$C.err
    done
  fi
  python3 $R/profiles/summarize.py $D
done
if [ "$EXTRAS" != "1" ]; then du -sh $OUT; exit 0; fi
# 4. kernel-by-kernel timelines of one training step and one inference call
for C in "isic 0.5" "suim 1" "city 1" "city 2"; do set -- $C
  GRAFT_REPO_ROOT=$R bash $R/tests/gpu_probe/trace.sh $1 $2 ${RND}/step_timeline_$1_a$2
  GRAFT_REPO_ROOT=$R bash $R/tests/gpu_probe/trace.sh $1 $2 ${RND}/step_timeline_single_stream_$1_a$2 single
done
# 5. SQ counters (what the waves wait for)
GRAFT_REPO_ROOT=$R bash $R/tests/gpu_probe/pmc.sh isic 0.5 ${RND}/sq_counters_isic
GRAFT_REPO_ROOT=$R bash $R/tests/gpu_probe/pmc.sh city 2 ${RND}/sq_counters_cityscapes_a2
# 6. wall time of a training step / an inference call for every shape and the IM+ width schedule, and EvalNet
step_times
du -sh $OUT; ls -R $OUT | head -80
