#!/bin/bash
# After `install_round.py rNN` (the tree now holds the profiles a line may replay from): the six plain bench commands once more, so that the
# committed rNN_bench<tag>.json lines carry their replayed values (traffic, frac_rocprof, frac_rocprof_union, by_stage kernels).
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'RND=r06 bash profiles/rebench.sh'   then install_round.py again
RND=${RND:-r06}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/${RND}
for CFG in isic suim cityscapes hela cityscapes_a2 cityscapes_a125; do
  case $CFG in
    isic)            ARGS="";                                 TAG="_isic" ;;
    suim|cityscapes|hela) ARGS="--config $CFG --steps 2";     TAG="_$CFG" ;;
    cityscapes_a2)   ARGS="--config cityscapes --alpha 2 --steps 2 --no-cpu-baseline";    TAG="_cityscapes_a2" ;;
    cityscapes_a125) ARGS="--config cityscapes --alpha 1.25 --steps 2 --no-cpu-baseline"; TAG="_cityscapes_a125" ;;
  esac
  D=$OUT/cfg$TAG; mkdir -p $D
  python3 $R/bench.py $ARGS --detail $D/bench_detail.json > $D/bench.json 2> $D/bench.err
  tail -c 300 $D/bench.json; echo
done
