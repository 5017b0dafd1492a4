# round 5 A/B (one box, one build): the pooled 16 -> 32 channel 3x3 on the GEMM-class kernel (default) against the per-tile kernel (IMK_GEMM_POOL32=0)
mkdir -p gpurun_out/r05
{
for rep in 1 2; do
  for v in 1 0; do
    IMK_GEMM_POOL32=$v INFER_B=584 python tests/gpu_probe/infer_ab.py 2>&1 | tail -1
    for cfg in suim hela city; do IMK_GEMM_POOL32=$v INFER_B=128 CONFIG=$cfg python tests/gpu_probe/infer_ab.py 2>&1 | tail -1; done
    for cfg in isic:0.5 suim:1 city:1; do
      echo "[POOL32=$v] $cfg: $(IMK_GEMM_POOL32=$v CONFIG=${cfg%%:*} ALPHA=${cfg##*:} python tests/gpu_probe/step_time.py 2>&1 | grep -E 'train step' | sed 's/(.*//' | tr '\n' ' ')"
    done
  done
done
} > gpurun_out/r05/ab5.txt 2>&1
cat gpurun_out/r05/ab5.txt
python -m pytest tests/test_gpu_unet.py tests/test_gpu_evalnet.py -q -x 2>&1 | tail -5
