mkdir -p gpurun_out/r05f
timeout 900 python -m pytest tests/ -x -q -m gpu > gpurun_out/r05f/gpu_tests.txt 2>&1; echo "tests rc=$?" ; tail -4 gpurun_out/r05f/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05f/smoke.txt 2>&1; tail -2 gpurun_out/r05f/smoke.txt
python bench.py > gpurun_out/r05f/bench.json 2> gpurun_out/r05f/bench.err; echo "bench rc=$?"; python - <<'P'
import json
d=json.loads(open('gpurun_out/r05f/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('metric','value','ms_per_step','n_gpus','steps')}, d['roofline']['frac'], d['roofline'].get('frac_rocprof'), d['cpu_baseline']['value'], d.get('stage_ms'))
P
IMK_BENCH_ONE_GPU=1 IMK_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > gpurun_out/r05f/bench_2ranks_one_gpu_gloo.json 2> gpurun_out/r05f/bench_2ranks.err; echo "2rank rc=$?"; tail -c 600 gpurun_out/r05f/bench_2ranks_one_gpu_gloo.json
IMK_BENCH_ONE_GPU=1 IMK_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 8 --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > gpurun_out/r05f/bench_8ranks_one_gpu_gloo.json 2> gpurun_out/r05f/bench_8ranks.err; echo "8rank rc=$?"; tail -c 600 gpurun_out/r05f/bench_8ranks_one_gpu_gloo.json
