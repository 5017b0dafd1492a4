#!/bin/bash
# one-box A/B of the fused 1x1 backward's staging map (this build against build/ab/libimk_base.so): training steps of the shapes that
# use bwd1x1_kernel + the EvalNet step, libraries interleaved
mkdir -p gpurun_out/r05
for rep in 1 2 3; do
  for lib in inconsistencymasks_amd/libimk.so ${OTHERS:-build/ab/libimk_base.so}; do
    for cfg in suim:1 hela:1 city:1 city:2; do
      echo "[$lib] $cfg: $(IMK_LIB_PATH=$lib CONFIG=${cfg%%:*} ALPHA=${cfg##*:} python tests/gpu_probe/step_time.py 2>&1 | grep -E 'train step' | sed 's/(.*//' | tr '\n' ' ')"
    done
    echo "[$lib] evalnet: $(IMK_LIB_PATH=$lib python tests/gpu_probe/evalnet_time.py 2>&1 | grep -E 'train step' | tr '\n' ' ')"
  done
done | tee gpurun_out/r05/ab_bwd1.txt
