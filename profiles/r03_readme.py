#!/usr/bin/env python3
"""Writes the round-3 section of profiles/README.md (between the r03 markers) from the r03_* files in this directory:
   python profiles/r03_readme.py        (after profiles/collect_r03.sh + profiles/install_r03.py)"""
import csv
import json
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))


def load(name):
    p = os.path.join(HERE, name)
    return json.load(open(p)) if os.path.exists(p) else None


def step_times():
    """config/alpha -> (train ms, inference ms) from r03_configs_step_times.txt"""
    out, cur = {}, None
    p = os.path.join(HERE, "r03_configs_step_times.txt")
    if not os.path.exists(p):
        return out
    for line in open(p):
        m = re.match(r"config (\w+) alpha ([\d.]+)", line)
        if m:
            cur = (m.group(1), float(m.group(2)))
            out[cur] = [None, None]
        m = re.match(r"evalnet alpha", line)
        if m:
            cur = ("evalnet", 2.0)
            out[cur] = [None, None]
        m = re.match(r"train step B=32: ([\d.]+) ms", line)
        if m and cur:
            out[cur][0] = float(m.group(1))
        m = re.match(r"inference B=\d+: ([\d.]+) ms", line)
        if m and cur:
            out[cur][1] = float(m.group(1))
    return out


def rocprof_avg():
    rows = [x for x in csv.DictReader(open(os.path.join(HERE, "r03_bench_kernel_stats.csv")))
            if x["kernel"].startswith("conv_pipe_kernel") or x["kernel"].startswith("conv_wide_kernel")]
    return 1000.0 * sum(float(x["total_ms"]) for x in rows) / max(sum(int(x["calls"]) for x in rows), 1)


def fam_table(d, top=12):
    r = d["roofline"]
    fam = r["all_families"]
    rows = []
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])[:top]:
        rows.append(f"| `{k}` | {v['launches']} | {v['avg_us']} | {v.get('GBps', '')} | {v.get('TFLOPs', '')} | {v['share_of_sampled_time']} |")
    return ("| family | sampled launches | avg us | GB/s (algorithmic) | TFLOP/s | share of sampled kernel time |\n|---|---|---|---|---|---|\n"
            + "\n".join(rows) + "\n")


def bench_line(tag, d):
    st = d["stage_ms"]
    r = d["roofline"]
    ts = r["step"]["train_step"]
    return (f"| {tag} | {d['value']:.0f} | {d['ms_per_step']} | {st['ensemble_infer_plus_im']} | {st['train_epoch']} ({d['config']['epoch_steps']} steps) | "
            f"`{r['kernel']}` {r['achieved']} {r['unit']} = {r['frac']} ({r['bound']}) | {ts['TFLOPs']} / {ts['GBps']} | "
            f"{(d.get('cpu_baseline') or {}).get('value', '-')} |")


def main():
    b = load("r03_bench.json")
    cfgs = [("SUIM, alpha 1", load("r03_configs_bench_suim.json")), ("Cityscapes, alpha 1", load("r03_configs_bench_cityscapes.json")),
            ("HeLa, alpha 1", load("r03_configs_bench_hela.json")), ("Cityscapes, alpha 2 (last IM+ generation)", load("r03_configs_bench_cityscapes_a2.json"))]
    stt = step_times()
    r = b["roofline"]
    cb = b["cpu_baseline"]
    ps = cb["parity_sample"]
    g = lambda k: stt.get(k, [None, None])
    out = []
    out.append("## Round 3 (`r03_*`)\n")
    out.append("Commands: `collect_r03.sh` (one `gpurun` call), `install_r03.py` (copies the summaries here), this section: `r03_readme.py`.\n")
    out.append("""| file | what |
|---|---|
| `r03_bench.json`, `r03_bench_under_rocprof.json` | the default command `python bench.py` (N = 1, BASELINE configs[1]), plain and under `rocprofv3 --kernel-trace --stats` |
| `r03_rocprofv3_kernel_stats_raw.csv`, `r03_bench_kernel_stats.csv` | rocprofv3's kernel statistics of that run, raw and with shortened names |
| `r03_pmc_traffic.csv`, `r03_pmc_traffic_{suim,cityscapes,hela,cityscapes_a2}.csv` | `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes (separate runs), MB per launch per kernel, raw and corrected for gfx950; `bench.py` reads `roofline.traffic` from the file of its configuration |
| `r03_configs_bench_{suim,cityscapes,hela,cityscapes_a2}.json` | `python bench.py --config ... [--alpha 2]`: the other BASELINE shapes, same line |
| `r03_configs_kernel_stats_*.csv` | rocprofv3 kernel statistics of those runs |
| `r03_configs_step_times.txt` | `tests/gpu_probe/step_time.py` / `evalnet_time.py`: wall time of a training step (batch 32) and a 128-image inference call, all shapes and the IM+ width schedule |
| `r03_step_timeline_*.txt`, `r03_step_timeline_single_stream_*.txt` | kernel-by-kernel timeline of one step and one inference call (two streams / every kernel alone) for ISIC, SUIM, Cityscapes alpha 1 and 2 |
| `r03_sq_counters_{isic,city_a2}.csv` | SQ counters per kernel |
| `r03_bench_2ranks_one_gpu_gloo.json` | `IMK_BENCH_ONE_GPU=1 IMK_BENCH_BACKEND=gloo python bench.py --gpus 2`: the self-launching strong-scaling path with two ranks time-slicing ONE GPU (functional check: `n_gpus` 2, `sharding_check.equals_sum_over_ranks` true; its throughput means nothing).  The RCCL path on one rank (`IMK_FORCE_DIST=1`) gives 23.45 k images/s against 23.93 k without the process group |
| `r03_full_driver_run.txt` | `tests/gpu_probe/full_driver_run.py`: `ISIC_2018/09_ISIC_2018_IM.py` through PNG directories at the dataset's real size: **34.2 s** per generation of 5 candidates x 50 epochs (35.3 s in round 2, 58.8 s in round 1) |
| `r03_dp_convergence.txt` | data-parallel convergence of the ISIC toy driver at world 1 / 2 / 4 / 8 (emulated), with the BatchNorm-momentum finding |
""")
    out.append(f"""Headline (`r03_bench.json`): **{b['value']:.0f} images/s per IM generation on 1 GPU** -- {b['ms_per_step']} ms per generation =
{b['stage_ms']['ensemble_infer_plus_im']} ms (ensemble forward + fused head/IM) + {b['stage_ms']['train_epoch']} ms ({b['config']['epoch_steps']} training steps) -- next to
{cb['value']} images/s for the CPU restatement ({cb['cpu_model']}, {cb['threads_forward']} / {cb['threads_train_step']} threads, {cb['gflops_forward_batch1']} / {cb['gflops_train_step']} GFLOP/s:
not a tuned CPU library).  Round 2 ended at 23 316 (100.14 ms).  `cpu_baseline.parity_sample`: max |dp| {ps['max_abs_dp']}, {ps['decision_flip_rate']:.1e} of the
decisions flip, {ps['im_pixels_differing']} of {ps['im_pixels_total']} IM pixels differ.  `roofline.kernel` = `{r['kernel']}`: {r['achieved']} {r['unit']} = **{r['frac']} of peak** in
the timed region ({r['avg_us_per_launch']} us per launch), HBM traffic {(r['traffic'] or 0) / 1e6:.1f} MB per launch (`{(r.get('traffic_source') or '').split(' ')[0]}`);
host time to enqueue one training step {r['step']['train_step']['host_enqueue_ms_per_step']} ms (GPU: {r['step']['train_step']['ms']}).
Agreement with rocprofv3: over the whole process the family's event-bracketed average is {r['avg_us_per_launch_whole_process']} us per launch
(`roofline.avg_us_per_launch_whole_process`; {(load('r03_bench_under_rocprof.json') or b)['roofline']['avg_us_per_launch_whole_process']} in the traced run), `r03_bench_kernel_stats.csv` gives {rocprof_avg():.2f} us for the same
launches (`conv_pipe_kernel` + `conv_wide_kernel` rows): HIP events bracket the kernel AND the ~3.5-4 us launch boundary in front of
it on its stream, the tracer's begin / end timestamps only the kernel.
""")
    out.append(fam_table(b))
    out.append("""### The other BASELINE shapes (`bench.py --config`)

| shape | images/s | ms / generation | inference + IM ms | training epoch ms | dominant family, roofline | whole step TFLOP/s / GB/s (min bytes) | CPU restatement images/s |
|---|---|---|---|---|---|---|---|""")
    out.append(bench_line("ISIC, alpha 0.5 (default)", b))
    for tag, d in cfgs:
        if d:
            out.append(bench_line(tag, d))
    out.append("")
    if cfgs[3][1]:
        out.append("Cityscapes alpha 2, kernel families (`r03_configs_bench_cityscapes_a2.json`):\n")
        out.append(fam_table(cfgs[3][1], 10))
    out.append(f"""### Training step / inference call, ms (`r03_configs_step_times.txt`; round 2 in brackets; verdict targets in square brackets)

| shape | training step, batch 32 | inference call, 128 images |
|---|---|---|
| ISIC alpha 0.5 | {g(('isic', 0.5))[0]} (1.021) [<= 0.90] | {g(('isic', 0.5))[1]} (0.515) |
| HeLa alpha 1 | {g(('hela', 1.0))[0]} (1.874) | {g(('hela', 1.0))[1]} (1.204) |
| SUIM alpha 1 | {g(('suim', 1.0))[0]} (1.902) [<= 1.5] | {g(('suim', 1.0))[1]} (1.280) [<= 1.0] |
| Cityscapes alpha 1 | {g(('city', 1.0))[0]} (2.891) | {g(('city', 1.0))[1]} (2.439) |
| Cityscapes alpha 1.25 | {g(('city', 1.25))[0]} (6.534) [<= 4.0] | {g(('city', 1.25))[1]} (4.703) |
| Cityscapes alpha 1.5 | {g(('city', 1.5))[0]} (7.233) | {g(('city', 1.5))[1]} (4.954) |
| Cityscapes alpha 1.75 | {g(('city', 1.75))[0]} (9.146) | {g(('city', 1.75))[1]} (6.290) |
| Cityscapes alpha 2 | {g(('city', 2.0))[0]} (9.150) [<= 5.0] | {g(('city', 2.0))[1]} (6.384) |
| ISIC alpha 1.5 | {g(('isic', 1.5))[0]} (5.022) | {g(('isic', 1.5))[1]} (2.778) |
| EvalNet alpha 2 (batch 32 both) | {g(('evalnet', 2.0))[0]} (2.691) [<= 2.0] | {g(('evalnet', 2.0))[1]} (0.592) |

Of the verdict's targets the two Cityscapes IM+ ones are met (alpha 2 <= 5.0, alpha 1.25 <= 4.0); SUIM, EvalNet and ISIC are not.  The wide shapes are 1.3-1.9x faster than in round 2, ISIC 2 %.
""")
    out.append(open(os.path.join(HERE, "r03_notes.md")).read())
    text = "\n".join(out)
    p = os.path.join(HERE, "README.md")
    s = open(p).read()
    a, z = "<!-- r03:begin -->\n", "<!-- r03:end -->\n"
    if a in s:
        s = s[:s.index(a) + len(a)] + text + s[s.index(z):]
    else:
        s = s.replace("## Round 2 (`r02_*`)", a + text + z + "\n## Round 2 (`r02_*`)", 1)
    open(p, "w").write(s)


if __name__ == "__main__":
    main()
