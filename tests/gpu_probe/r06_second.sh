#!/bin/bash
# round 6, second GPU call: full GPU suite (new tests), priced excess tables of the wide configurations, wgrad split sweeps, candidates side by side 3 / 4 / 5
mkdir -p gpurun_out/r06b
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/r06b/gpu_tests.txt 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r06b/gpu_tests.txt
RND=r06b PMC=0 EXTRAS=0 CONFIGS="cityscapes_a2 cityscapes_a125 isic" bash profiles/collect_round.sh > gpurun_out/r06b/collect.log 2>&1; echo "collect rc=$?"
bash tests/gpu_probe/ab_env.sh "city:2 city:1.25 suim:1" "-" "IMK_WGRAD_GEMM_WGS=256" "IMK_WGRAD_GEMM_WGS=384" "IMK_WGRAD_GEMM_WGS=768" "IMK_WGRAD_GEMM_TILES=16" "IMK_WGRAD_GEMM_TILES=32" "-" > gpurun_out/r06b/ab_wgrad_splits.txt 2>&1
for e in "-" "IMK_WGRAD_GEMM_WGS=256" "IMK_WGRAD_GEMM_TILES=16"; do [ "$e" = "-" ] && ev="IMK_AB_DEFAULT=1" || ev="$e"; echo "[$e] evalnet: $(env $ev python tests/gpu_probe/evalnet_time.py 2>&1 | grep -E 'train step' )"; done >> gpurun_out/r06b/ab_wgrad_splits.txt 2>&1
cat gpurun_out/r06b/ab_wgrad_splits.txt
for P in 3 5 4 1; do
  rm -rf /tmp/im_full_run
  IM_PARALLEL_CANDIDATES=$P python tests/gpu_probe/full_driver_run.py > gpurun_out/r06b/full_driver_run_par$P.txt 2>&1
  echo "parallel $P: $(grep -E 'side by side|candidate 4|1 run id' gpurun_out/r06b/full_driver_run_par$P.txt | tail -2 | tr '\n' ' ')"; md5sum /tmp/im_full_run/data/csv/results_*.csv
done
