#!/bin/bash
# Round 5, conv_gemm_kernel's RR form (round-robin channel tiles + skipped dead tile, imk_gemm.hip) against the build before it
# (build/ab/libimk_base.so): (1) do the two builds compute the same bits (inference probabilities, stored activations, every gradient,
# parameters after a step) at alpha 1.25 / 1.5; (2) training step + 128-image inference, libraries interleaved, IMK_GEMM_RR=0 as a third leg.
BASE=${1:-build/ab/libimk_base.so}
mkdir -p gpurun_out/r05
{
for a in 1.25 1.5; do
  CONFIG=city ALPHA=$a IMK_LIB_PATH=$BASE python tests/gpu_probe/lib_ab.py > gpurun_out/r05/bits_base_$a.txt 2>&1
  CONFIG=city ALPHA=$a python tests/gpu_probe/lib_ab.py > gpurun_out/r05/bits_rr_$a.txt 2>&1
  if diff -q gpurun_out/r05/bits_base_$a.txt gpurun_out/r05/bits_rr_$a.txt > /dev/null; then
    echo "alpha $a: bit-identical ($(grep -c . gpurun_out/r05/bits_rr_$a.txt) checksums; params $(grep params gpurun_out/r05/bits_rr_$a.txt))"
  else echo "alpha $a: DIFFERENT"; diff gpurun_out/r05/bits_base_$a.txt gpurun_out/r05/bits_rr_$a.txt | head -20; fi
done
for rep in 1 2; do
  for leg in rr base rr0; do
    case $leg in rr) ev="IMK_AB_DEFAULT=1";; base) ev="IMK_LIB_PATH=$BASE";; rr0) ev="IMK_GEMM_RR=0";; esac
    for cfg in city:1.25 city:1.5 city:1 city:2; do
      echo "[$leg] $cfg: $(env $ev CONFIG=${cfg%%:*} ALPHA=${cfg##*:} python tests/gpu_probe/step_time.py 2>&1 | grep -E 'train step|inference' | sed 's/(.*//' | tr '\n' ' ')"
    done
  done
done
} | tee gpurun_out/r05/ab_rr.txt
