#!/bin/bash
# round 6, first GPU call: the bench line as the driver sees it, the tests that read it, the fp32-oracle gap table, per-config traces
mkdir -p gpurun_out/r06a
timeout 1500 python -m pytest tests/test_gpu_driver.py tests/test_gpu_writers.py -x -q -m gpu -k "bench or writer or golden" > gpurun_out/r06a/tests_bench.txt 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r06a/tests_bench.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06a/bench_driver.out 2> gpurun_out/r06a/bench_driver.err; echo "bench rc=$?"
python3 - <<'P'
import json
t=open('gpurun_out/r06a/bench_driver.out').read()
last=t.strip().splitlines()[-1]
print("stdout bytes", len(t), "lines", len(t.strip().splitlines()), "last line bytes", len(last))
d=json.loads(last); print(last)
P
cp gpurun_out/bench_detail.json gpurun_out/r06a/bench_driver_detail.json
timeout 900 python3 tests/gpu_probe/fp32_gap.py > gpurun_out/r06a/fp32_gap.txt 2> gpurun_out/r06a/fp32_gap.err; echo "fp32_gap rc=$?"; tail -3 gpurun_out/r06a/fp32_gap.err
RND=r06a PMC=0 EXTRAS=0 bash profiles/collect_round.sh > gpurun_out/r06a/collect.log 2>&1; echo "collect rc=$?"
ls gpurun_out/r06a/*
