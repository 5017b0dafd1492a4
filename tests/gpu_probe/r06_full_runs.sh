#!/bin/bash
# round 6: the SUIM, Cityscapes, HeLa and ISIC IM++ drivers at their datasets' real sizes on the round's last build (host-side profile, 2 candidates x 10 epochs)
mkdir -p gpurun_out/r06k
for d in suim city hela impp; do
  timeout 900 python tests/gpu_probe/full_driver_run_$d.py /tmp/im_full_run_$d > gpurun_out/r06k/full_driver_run_$d.txt 2>&1
  echo "$d rc=$?: $(grep -E '^\[timing\] setup' gpurun_out/r06k/full_driver_run_$d.txt | tail -1)"
  rm -rf /tmp/im_full_run_$d
done
