#!/bin/bash
# the default bench command five times in a row on one box: run-to-run spread (-> profiles/rNN_bench_repeats.txt)
RND=${RND:-r06}
mkdir -p gpurun_out/${RND}f
for i in 1 2 3 4 5; do
  python3 bench.py --no-cpu-baseline --no-other-configs 2> /dev/null | tail -1 | python3 -c "
import json, sys
d = json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['stage_ms'], d['roofline']['by_stage']['training']['step_ms'], d['config'].get('epoch_steps'), d['roofline']['frac'], d['roofline'].get('frac_rocprof_union'))"
done | tee gpurun_out/${RND}f/bench_repeats.txt
