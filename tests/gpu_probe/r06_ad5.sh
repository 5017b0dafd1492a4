#!/bin/bash
# conv_gemm deep form: ring of six (five k-steps of look-ahead, two phases) for the 9-step 3x3 stage against the in-tree ring of three
mkdir -p gpurun_out/r06i
L=build/ab/libimk_ad5.so
for c in "city 2" "suim 1" "isic 0.5"; do set -- $c
  CONFIG=$1 ALPHA=$2 python tests/gpu_probe/lib_ab.py > gpurun_out/r06i/bits_$1_$2_default.txt 2>&1
  IMK_LIB_PATH=$L CONFIG=$1 ALPHA=$2 python tests/gpu_probe/lib_ab.py > gpurun_out/r06i/bits_$1_$2_ad5.txt 2>&1
  cmp gpurun_out/r06i/bits_$1_$2_default.txt gpurun_out/r06i/bits_$1_$2_ad5.txt && echo "bit-identical $1 $2 ($(wc -l < gpurun_out/r06i/bits_$1_$2_ad5.txt) checksums)" || echo "DIFFERENT $1 $2"
done
CFGS="isic:0.5 suim:1 hela:1 city:1 city:1.25 city:2"
run() { for cfg in $CFGS; do echo "[$1] $cfg: $(env $2 CONFIG=${cfg%%:*} ALPHA=${cfg##*:} python tests/gpu_probe/step_time.py 2>&1 | grep -E 'train step|inference' | sed 's/(.*//' | tr '\n' ' ')"; done; echo "[$1] evalnet: $(env $2 python tests/gpu_probe/evalnet_time.py 2>&1 | grep -E 'train step|inference' | tr '\n' ' ')"; }
{ run "ring3" "IMK_AB_DEFAULT=1"
  for T in 1024 2048 4096; do run "ring6<=$T" "IMK_LIB_PATH=$L IMK_GEMM_AD3_WGS=$T"; done
  run "ring3" "IMK_AB_DEFAULT=1"; } > gpurun_out/r06i/ab_ad5.txt 2>&1
cat gpurun_out/r06i/ab_ad5.txt
