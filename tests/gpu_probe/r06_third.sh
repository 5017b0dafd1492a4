#!/bin/bash
# round 6, third GPU call: conv_gemm weight prefetch depth A/B (+ bit identity), full GPU suite, candidates side by side in clean directories
mkdir -p gpurun_out/r06c
for c in "city 2" "city 1.25" "suim 1"; do set -- $c
  CONFIG=$1 ALPHA=$2 python tests/gpu_probe/lib_ab.py > gpurun_out/r06c/bits_$1_$2_default.txt 2>&1
  IMK_LIB_PATH=build/ab/libimk_adepth2.so CONFIG=$1 ALPHA=$2 python tests/gpu_probe/lib_ab.py > gpurun_out/r06c/bits_$1_$2_adepth2.txt 2>&1
  cmp gpurun_out/r06c/bits_$1_$2_default.txt gpurun_out/r06c/bits_$1_$2_adepth2.txt && echo "bit-identical $1 $2" || echo "DIFFERENT $1 $2"
done
bash tests/gpu_probe/ab_lib.sh "city:2 city:1.25 city:1 suim:1 hela:1" - build/ab/libimk_adepth2.so > gpurun_out/r06c/ab_adepth.txt 2>&1
for lib in - build/ab/libimk_adepth2.so; do [ "$lib" = "-" ] && ev="IMK_AB_DEFAULT=1" || ev="IMK_LIB_PATH=$lib"; echo "[$lib] evalnet: $(env $ev python tests/gpu_probe/evalnet_time.py 2>&1 | grep -E 'train step|inference' | tr '\n' ' ')"; done >> gpurun_out/r06c/ab_adepth.txt 2>&1
cat gpurun_out/r06c/ab_adepth.txt
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/r06c/gpu_tests.txt 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r06c/gpu_tests.txt
for P in 5 3 4 3 5; do
  W=/tmp/im_full_run_$P_$RANDOM
  IM_PARALLEL_CANDIDATES=$P python tests/gpu_probe/full_driver_run.py $W > gpurun_out/r06c/full_driver_run_par${P}_$RANDOM.txt 2>&1
  echo "parallel $P: $(grep -hE 'pseudo-labels|side by side|1 run id' gpurun_out/r06c/full_driver_run_par${P}_*.txt | tail -3 | sed 's/.*bo_True: //' | tr '\n' ' ')"; md5sum $W/data/csv/results_*.csv | cut -c1-12
  rm -rf $W; sync; sleep 3
done
