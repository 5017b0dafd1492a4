#!/bin/bash
# one fresh box per setting (later runs on a box are disturbed by the previous run's files): IM_PARALLEL_CANDIDATES=$1 on the real-size ISIC generation
mkdir -p gpurun_out/r06d
IM_PARALLEL_CANDIDATES=$1 python tests/gpu_probe/full_driver_run.py /tmp/im_full_run > gpurun_out/r06d/full_driver_run_par$1.txt 2>&1
grep -hE 'pseudo-labels|side by side|candidate 4|1 run id' gpurun_out/r06d/full_driver_run_par$1.txt | sed 's/.*bo_True: //'; md5sum /tmp/im_full_run/data/csv/results_*.csv | cut -c1-12
