#!/bin/bash
mkdir -p gpurun_out/r06h
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/r06h/gpu_tests.txt 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r06h/gpu_tests.txt
python3 bench.py --no-cpu-baseline 2> /dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms'], d['roofline']['by_stage']['training'], d['other_configs'])"
