#!/bin/bash
# the default bench command five times in a row on one box: run-to-run spread (-> profiles/rNN_bench_repeats.txt)
mkdir -p gpurun_out/r05f
for i in 1 2 3 4 5; do
  python3 bench.py --no-cpu-baseline --no-other-configs 2> /dev/null | tail -1 | python3 -c "
import json, sys
d = json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['stage_ms'], d['roofline']['step']['train_step']['ms'], d['config'].get('epoch_steps'))"
done | tee gpurun_out/r05f/bench_repeats.txt
