# round 5 A/B (one box, one build): conv_gemm_kernel's persistent walk (default: 512 workgroups) against one workgroup per tile (IMK_GEMM_PERSIST=0)
mkdir -p gpurun_out/r05
{
for rep in 1 2; do
  for v in 512 0 1024; do
    for cfg in isic:0.5 suim:1 hela:1 city:1 city:1.5 city:2; do
      echo "[PERSIST=$v] $cfg: $(IMK_GEMM_PERSIST=$v CONFIG=${cfg%%:*} ALPHA=${cfg##*:} python tests/gpu_probe/step_time.py 2>&1 | grep -E 'train step|inference' | sed 's/(.*//' | tr '\n' ' ')"
    done
    echo "[PERSIST=$v] evalnet: $(IMK_GEMM_PERSIST=$v python tests/gpu_probe/evalnet_time.py 2>&1 | tail -2 | tr '\n' ' ')"
  done
done
} > gpurun_out/r05/ab6.txt 2>&1
cat gpurun_out/r05/ab6.txt
python -m pytest tests/test_gpu_unet.py tests/test_gpu_evalnet.py -q -x 2>&1 | tail -5
