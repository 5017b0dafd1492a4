#!/bin/bash
# Round 5's one-box A/B harness: this build against another library (default: round 4's, built from commit d793ff7 into
# build/ab/libimk_r04.so by compiling that commit's csrc/*.hip with build.py's flags), training step + 128-image inference call per shape,
# libraries interleaved, several repetitions.  Boxes differ by 1-3 %, so only numbers from ONE run of this script are compared.
#   bash tests/gpu_probe/ab_r05.sh [other.so] ["isic:0.5 suim:1 ..."]
# The round's successive experiments (interior-tile path, fused-3x3 rule, pooled conv on the GEMM-class kernel, persistent GEMM walk: see
# profiles/r05_notes.md section 2 and profiles/r05_ab1..6.txt) used this loop with an environment switch or a second build in place of the
# second library; those forms are in the history of this file.
OTHER=${1:-build/ab/libimk_r04.so}
CFGS=${2:-"isic:0.5 suim:1 hela:1 city:1 city:2"}
mkdir -p gpurun_out/r05
for rep in 1 2 3; do
  for lib in inconsistencymasks_amd/libimk.so $OTHER; do
    for cfg in $CFGS; do
      echo "[$lib] $cfg: $(IMK_LIB_PATH=$lib CONFIG=${cfg%%:*} ALPHA=${cfg##*:} python tests/gpu_probe/step_time.py 2>&1 | grep -E 'train step|inference' | sed 's/(.*//' | tr '\n' ' ')"
    done
  done
done | tee gpurun_out/r05/ab.txt
