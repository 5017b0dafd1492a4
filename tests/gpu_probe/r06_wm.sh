#!/bin/bash
# conv_gemm turned wave tile (128 pixels x 2 channel tiles per wave: IMK_GEMM_WM) against the 64 x 4 tiling: bit identity, then times
mkdir -p gpurun_out/r06j
L=build/ab/libimk_wm.so
for c in "city 2" "city 1" "suim 1" "isic 0.5"; do set -- $c
  CONFIG=$1 ALPHA=$2 python tests/gpu_probe/lib_ab.py > gpurun_out/r06j/bits_$1_$2_default.txt 2>&1
  for M in 2 a 1; do
    IMK_GEMM_WM=$M IMK_LIB_PATH=$L CONFIG=$1 ALPHA=$2 python tests/gpu_probe/lib_ab.py > gpurun_out/r06j/bits_$1_$2_wm$M.txt 2>&1
    cmp gpurun_out/r06j/bits_$1_$2_default.txt gpurun_out/r06j/bits_$1_$2_wm$M.txt && echo "bit-identical $1 $2 WM=$M ($(wc -l < gpurun_out/r06j/bits_$1_$2_wm$M.txt) checksums)" || echo "DIFFERENT $1 $2 WM=$M"
  done
done
CFGS="isic:0.5 suim:1 hela:1 city:1 city:1.25 city:2"
run() { for cfg in $CFGS; do echo "[$1] $cfg: $(env $2 CONFIG=${cfg%%:*} ALPHA=${cfg##*:} python tests/gpu_probe/step_time.py 2>&1 | grep -E 'train step|inference' | sed 's/(.*//' | tr '\n' ' ')"; done; echo "[$1] evalnet: $(env $2 python tests/gpu_probe/evalnet_time.py 2>&1 | grep -E 'train step|inference' | tr '\n' ' ')"; }
{ run "in-tree" "IMK_AB_DEFAULT=1"
  for M in 2 a d b 1; do run "WM=$M" "IMK_LIB_PATH=$L IMK_GEMM_WM=$M"; done
  run "in-tree" "IMK_AB_DEFAULT=1"; } > gpurun_out/r06j/ab_wm.txt 2>&1
cat gpurun_out/r06j/ab_wm.txt
