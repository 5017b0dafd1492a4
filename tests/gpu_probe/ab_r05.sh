# round 5 A/B runs (one box): this build against round 4's library (build/ab/libimk_r04.so: /tmp/build_base.sh from commit d793ff7)
mkdir -p gpurun_out/r05
{
for rep in 1 2 3; do
  for lib in inconsistencymasks_amd/libimk.so build/ab/libimk_r04.so; do
    for cfg in isic:0.5 suim:1 hela:1 city:1 city:2; do
      echo "[$lib] $cfg: $(IMK_LIB_PATH=$lib CONFIG=${cfg%%:*} ALPHA=${cfg##*:} python tests/gpu_probe/step_time.py 2>&1 | grep -E 'train step|inference' | sed 's/(.*//' | tr '\n' ' ')"
    done
  done
done
} > gpurun_out/r05/ab3.txt 2>&1
cat gpurun_out/r05/ab3.txt
python -m pytest tests -m gpu -q -x 2>&1 | tail -5
