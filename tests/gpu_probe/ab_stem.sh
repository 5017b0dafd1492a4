for rep in 1 2 3; do for v in 1 0; do for cfg in isic:0.5 suim:1 hela:1; do
  echo "[STEM_WGRAD=$v] $cfg: $(IMK_STEM_WGRAD=$v CONFIG=${cfg%%:*} ALPHA=${cfg##*:} python tests/gpu_probe/step_time.py 2>&1 | grep -E 'train step' | sed 's/(.*//')"
done; done; done
python -m pytest tests/test_gpu_unet.py tests/test_gpu_evalnet.py -q -x -k "train or parity or reproducible or evalnet" 2>&1 | tail -3
