#!/bin/bash
# The round's last commit on one box: GPU suite, smoke(), the DRIVER's bench command (last stdout line parsed the way the driver does), 2- and 8-rank functional lines.
RND=${RND:-r06}
O=gpurun_out/${RND}f
mkdir -p $O
timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/gpu_tests.txt 2>&1; echo "tests rc=$?" ; tail -4 $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 --detail $O/bench_detail.json > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python - <<P
import json
t=open('$O/bench.json').read()
last=t.strip().splitlines()[-1]
d=json.loads(last)
r=d['roofline']
print("stdout", len(t), "bytes,", len(t.strip().splitlines()), "line(s); last line", len(last), "bytes")
print({k:d[k] for k in ('metric','value','ms_per_step','n_gpus','steps','warmup')}, d['stage_ms'])
print("roofline", {k:r.get(k) for k in ('kernel','bound','achieved','frac','frac_rocprof','frac_rocprof_union','traffic','traffic_over_algorithmic','exclusive_frac','replayed_from','replayed_refused')})
print("by_stage", r['by_stage'])
print("cpu_baseline", {k:d['cpu_baseline'].get(k) for k in ('value','cores','host_cpus','kind','t_infer_per_image_s','t_train_step_s')}, d['cpu_baseline']['parity_sample'])
print("other_configs", d.get('other_configs'))
P
for C in "--config suim" "--config cityscapes --alpha 2"; do python3 bench.py $C --steps 1 --no-cpu-baseline 2> /dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['config']['name'], d['config']['alpha'], d['value'], 'frac', r['frac'], 'rocprof', r.get('frac_rocprof'), 'union', r.get('frac_rocprof_union'), 'traffic', r.get('traffic'), r.get('replayed_refused'), r['by_stage']['inference'].get('kernel'), r['by_stage']['training'].get('kernel'))"; done
IMK_BENCH_ONE_GPU=1 IMK_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $O/bench_2ranks_one_gpu_gloo.json 2> $O/bench_2ranks.err; echo "2rank rc=$?"; tail -c 400 $O/bench_2ranks_one_gpu_gloo.json
IMK_BENCH_ONE_GPU=1 IMK_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 8 --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $O/bench_8ranks_one_gpu_gloo.json 2> $O/bench_8ranks.err; echo "8rank rc=$?"; tail -c 400 $O/bench_8ranks_one_gpu_gloo.json
