#!/bin/bash
# bench line + rocprofv3 kernel statistics of the Cityscapes shape at alpha 1.25 (the IM+ width schedule's second width), like step 3 of
# profiles/collect_round.sh does for alpha 1 and 2
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05/cfg_cityscapes_a125
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --config cityscapes --alpha 1.25 --steps 2 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --config cityscapes --alpha 1.25 --steps 1 --no-cpu-baseline > $OUT/bench_traced.json 2> $OUT/trace.err
python3 $R/profiles/summarize.py $OUT
python3 $R/tests/gpu_probe/kstats.py $OUT/trace 30 > $OUT/kstats.txt
rm -rf $OUT/trace
ls -la $OUT
