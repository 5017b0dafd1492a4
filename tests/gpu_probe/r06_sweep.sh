#!/bin/bash
# round 6: which kernel family / tile shape a layer takes, by MEASURED step time at the IM+ widths -- every switch the library has
mkdir -p gpurun_out/r06e
bash tests/gpu_probe/ab_env.sh "city:1.25 city:2" "-" "IMK_GEMM_PN=2" "IMK_GEMM_PN=4" "IMK_GEMM_W=3" "IMK_CONV_WIDE=0" "IMK_WGRAD_GEMM_MIN=48" "IMK_WGRAD_GEMM_MIN=96" "IMK_WGRAD_NFO2=0" "IMK_BWD1X1=0" "IMK_GEMM_CHAIN_TRAIN=0" "IMK_WIDE_CHAIN_TRAIN=0" "IMK_FUSE_WGRAD=0" "IMK_SIDE_STREAMS=2" "IMK_FORK_LATE=0" "IMK_WGRAD_GEMM_PAIRS=33" "-" > gpurun_out/r06e/ab_family_sweep.txt 2>&1
cat gpurun_out/r06e/ab_family_sweep.txt
