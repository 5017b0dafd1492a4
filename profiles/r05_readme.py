#!/usr/bin/env python3
"""Writes the round-5 section of profiles/README.md (between the r05 markers) from the r05_* files in this directory:
   python profiles/r05_readme.py        (after `RND=r05 bash profiles/collect_round.sh` on the GPU box + `python profiles/install_round.py r05`
                                         + `python profiles/traffic_ratio.py r05`)"""
import csv
import json
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
R = "r05"


def load(name):
    p = os.path.join(HERE, name)
    return json.load(open(p)) if os.path.exists(p) else None


def step_times(rnd):
    out, cur = {}, None
    p = os.path.join(HERE, f"{rnd}_configs_step_times.txt")
    if not os.path.exists(p):
        return out
    for line in open(p):
        m = re.match(r"config (\w+) alpha ([\d.]+)", line)
        if m:
            cur = (m.group(1), float(m.group(2)))
            out[cur] = [None, None]
        if re.match(r"evalnet alpha", line):
            cur = ("evalnet", 2.0)
            out[cur] = [None, None]
        m = re.match(r"train step B=32: ([\d.]+) ms", line)
        if m and cur:
            out[cur][0] = float(m.group(1))
        m = re.match(r"inference B=\d+: ([\d.]+) ms", line)
        if m and cur:
            out[cur][1] = float(m.group(1))
    return out


def bench_line(tag, d):
    st, r = d["stage_ms"], d["roofline"]
    ts = r["step"]["train_step"]
    return (f"| {tag} | {d['value']:.0f} | {d['ms_per_step']} | {st['ensemble_infer_plus_im']} | {st['train_epoch']} ({d['config']['epoch_steps']} steps of {ts['ms']} ms) | "
            f"`{r['kernel']}` {r['achieved']} {r['unit']} = {r['frac']} ({r['bound']}) | {(d.get('cpu_baseline') or {}).get('value', '-')} |")


def main():
    b = load(f"{R}_bench.json")
    r = b["roofline"]
    bs = r.get("by_stage") or {}
    cb = b["cpu_baseline"]
    ps = cb["parity_sample"]
    cfgs = [("SUIM, alpha 1", load(f"{R}_configs_bench_suim.json")), ("Cityscapes, alpha 1", load(f"{R}_configs_bench_cityscapes.json")),
            ("HeLa, alpha 1", load(f"{R}_configs_bench_hela.json")), ("Cityscapes, alpha 2 (last IM+ generation)", load(f"{R}_configs_bench_cityscapes_a2.json"))]
    now, was = step_times(R), step_times("r04")
    g = lambda d, k: d.get(k, [None, None])
    out = []
    out.append("## Round 5 (`r05_*`)\n")
    out.append("Commands: `RND=r05 bash profiles/collect_round.sh` (one `gpurun` call), `install_round.py r05`, `traffic_ratio.py r05`, this section: `r05_readme.py`.\n")
    out.append(f"""| file | what |
|---|---|
| `r05_bench.json`, `r05_bench_under_rocprof.json` | the default command `python bench.py` (N = 1, BASELINE configs[1] + `other_configs`), plain and under `rocprofv3 --kernel-trace --stats` (traced: `--no-other-configs`) |
| `r05_timed_region_kernel_stats.csv` | the kernel trace of that command cut at bench.py's marker dispatches, NOW PER STAGE (markers at the inference / training boundary of every generation): stage, kernel variant, calls, average us from rocprofv3's own timestamps, the library's algorithmic MB / GFLOP per launch IN THAT STAGE (a variant that runs in both has other bytes per launch in each), GB/s, fraction of 8 TB/s.  `roofline.frac_rocprof` (the family's summed bytes over its summed durations, exact) and `roofline.by_stage` of the bench line are computed from it |
| `r05_traffic_vs_algorithmic.csv` | `traffic_ratio.py`: counter traffic (2 x FETCH_SIZE + WRITE_SIZE) over algorithmic bytes per variant of the dominant family, one population per variant (variants that run in one stage only; of a variant's grid sizes in the PMC pass the one with the most launches) |
| `r05_rocprofv3_kernel_stats_raw.csv`, `r05_bench_kernel_stats.csv` | rocprofv3's kernel statistics of the whole traced run, raw and with shortened names |
| `r05_pmc_traffic.csv`, `r05_pmc_traffic_{{suim,cityscapes,hela,cityscapes_a2}}.csv` | `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes (separate runs, `--kernel-trace` only beside them) for all five configurations; `bench.py` reads `roofline.traffic` from the file of its configuration |
| `r05_configs_bench_*.json`, `r05_configs_kernel_stats_*.csv` | `python bench.py --config ... [--alpha 2]` with their CPU baselines, and rocprofv3 kernel statistics of those runs |
| `r05_configs_step_times.txt` | wall time of a training step (batch 32) and a 128-image inference call, all shapes and the IM+ width schedule |
| `r05_step_timeline_*.txt`, `r05_step_timeline_single_stream_*.txt` | kernel-by-kernel timeline of one step and one inference call (two streams / every kernel alone) |
| `r05_sq_counters_{{isic,city_a2}}.csv` | SQ counters per kernel |
| `r05_bn_consumer_probe.txt` | `tests/gpu_probe/bn_consumer_probe.hip`: the finalize launch against the XCD-local ticket, the integer-atomic rows, the two-launch floor and the write-through stores (section 1 of the notes below) |
| `r05_ab1.txt` ... `r05_ab6.txt`, `r05_infer_batch.txt` | raw output of the round's one-box A/B runs (`tests/gpu_probe/ab_r05.sh` in its successive forms; what each compared: notes, section 2) |
| `r05_full_driver_run.txt` | `ISIC_2018/09_ISIC_2018_IM.py` through PNG directories at the dataset's real size, sequential candidates, native PNG codec + the reader pool / decoded-set cache: 23.8 s per generation (round 4: 35.3 s; with the codec alone: 25.5 s) |
| `r05_full_driver_run_sequential.txt`, `r05_full_driver_run_parallel3.txt` | the same driver on one (slower) box, sequential against the new one-rank default of three candidates side by side: 27.5 -> 17.9 s, identical results CSV |
| `r05_full_driver_run_five_generations.txt` | the same driver over generations 0-4 (25 candidates x 50 epochs, three side by side): 79.7 s |
| `r05_full_driver_run_hela.txt` | `HeLa/09_HeLa_IM.py` at real size (2 candidates x 10 epochs) with cProfile: 11.0 s (31.7 s with the numpy / scipy position geometry; `csrc/imk_geom.cpp`) |
| `r05_full_driver_run_impp.txt` | `ISIC_2018/12_ISIC_2018_IM++.py` at real size (2 EvalNets + 2 candidates x 10 epochs) with cProfile: 21.4 s (26.0 s before the shared reader pool, `read_png_stack` and the decoded training set kept across candidates) |
| `r05_trajectory_diag_suim.txt` | `tests/gpu_probe/trajectory_diag.py`: per tensor, how far a GPU training run and the oracle's are apart after 1 ... 30 steps |
| `r05_ab7.txt`, `r05_ab8.txt` | one-box A/B runs of the round's last two kernel changes: the staging maps of `bwd1x1_kernel` / `wgrad_gemm_kernel<.., 2, 2>` (kept) and `conv_gemm_kernel`'s RR form for 80- / 96-channel layers (`tests/gpu_probe/ab_rr.sh`: bit-identical, step-neutral, removed) |
| `r05_step_timeline_single_stream_city_a1.25.txt`, `_a1.5.txt` | the single-stream timelines of the IM+ width schedule's two middle widths (notes, section 3: kernel time by family, alpha 1 -> 2) |
| `r05_final_check_bench.json`, `r05_final_check_gpu_tests.txt` | `tests/gpu_probe/final_check.sh` on the round's last commit, one box: `pytest -m gpu` (172 passed, 3 skipped = the Keras-golden fixtures this image cannot produce), `smoke()`, the default `python bench.py` line (the test run is of the round's last commit; the line from the build one host-side commit earlier: 25 476 images/s, 91.66 ms = 14.38 + 77.22, `png_io` 9.6 k encoded / 20.8 k decoded images/s) |
| `r05_configs_bench_cityscapes_a125.json`, `r05_configs_kernel_stats_cityscapes_a125.csv` | `tests/gpu_probe/collect_a125.sh`: the Cityscapes shape at alpha 1.25 like the alpha 1 / 2 files (5 628 images/s per generation, training step 3.63 ms; no kernel above 6.1 % of the kernel time) |
| `r05_full_driver_run_rle.txt` | the real-size ISIC generation after the PNG encoder's switch to the Z_RLE strategy (notes, section 8): pseudo-label stage 0.89 s, generation 17.2 s |
| `r05_bench_repeats.txt` | the default bench command five times in a row on one box: 25.17-25.36 k images/s, +-0.4 % run to run |
| `r05_bench_2ranks_one_gpu_gloo.json`, `r05_bench_8ranks_one_gpu_gloo.json` | same script: `IMK_BENCH_ONE_GPU=1 IMK_BENCH_BACKEND=gloo python bench.py --gpus 2` / `--gpus 8` -- the self-launching strong-scaling path with 2 / 8 ranks time-slicing ONE GPU (functional: `n_gpus` 2 / 8, shards of 2 335 / 8 = 291-292 images, `sharding_check.equals_sum_over_ranks` true; the throughput means nothing) |
""")
    out.append(f"""Headline (`r05_bench.json`): **{b['value']:.0f} images/s per IM generation on 1 GPU** -- {b['ms_per_step']} ms per generation =
{b['stage_ms']['ensemble_infer_plus_im']} ms (ensemble forward + fused head / IM, calls of {b['config']['infer_batch']} images) + {b['stage_ms']['train_epoch']} ms ({b['config']['epoch_steps']} training steps of
{r['step']['train_step']['ms']} ms; {b['config'].get('kept')} of the 2 335 pseudo-labelled pairs kept) -- next to {cb['value']} images/s for the CPU restatement ({cb['cpu_model']},
{cb['threads_forward']} / {cb['threads_train_step']} threads).  Round 4 ended at 24 793 (94.18 ms); this build measured 25 006 (93.38 ms, 78 steps) and 25 177 on the round's other
boxes -- boxes differ by 1-3 % and the step count (78 / 80) follows how many pairs the ensemble keeps: the round's one-box comparisons against round 4's library show the
training step unchanged (0.985-0.991 ms either way, `r05_ab3.txt`) and the inference stage 2 % shorter (calls of 584 instead of 256 images).
`cpu_baseline.parity_sample`: max |dp| {ps['max_abs_dp']} on the trained ensemble, {ps['fixed_weights'].get('max_abs_dp')} on fixed weights, {ps['decision_flip_rate']:.1e} of the decisions
flip, {ps['im_pixels_differing']} of {ps['im_pixels_total']} IM pixels differ (SURVEY H4: IM masks are bit-identical GIVEN identical probabilities; GPU-vs-CPU probabilities differ by fp16 rounding).
`png_io`: {b['png_io']['encode_images_per_s']:.0f} / {b['png_io']['decode_images_per_s']:.0f} images/s encoded / decoded by libimk's own codec on {b['png_io']['threads']} threads (Pillow on the same pool:
{b['png_io']['pillow_encode_images_per_s']:.0f} / {b['png_io']['pillow_decode_images_per_s']:.0f}).

`roofline` of the line: kernel `{r['kernel']}`, {r['achieved']} {r['unit']} = **{r['frac']}** of peak by HIP events in the timed region, **{r['frac_rocprof']}** from rocprofv3's durations
(sum over the family's variants of calls x algorithmic bytes / sum of their durations, `r05_timed_region_kernel_stats.csv`); HBM traffic / algorithmic bytes
{r.get('traffic_over_algorithmic')} (`r05_traffic_vs_algorithmic.csv`).  The two regimes the family average hides are now in the line itself (`roofline.by_stage`):
""")
    if bs:
        i, t = bs.get("inference") or {}, bs.get("training") or {}
        out.append("| stage | dominant kernel variant (rocprofv3, timed region) | its fraction of peak | every hooked variant of the stage | bound | stage ms | of the minimum-bytes floor |\n|---|---|---|---|---|---|---|")
        if i:
            out.append(f"| inference + IM | `{i['kernel']}` ({i['avg_us']} us per call, {i['share_of_stage_kernel_time']} of the stage's kernel time) | {i['frac']} | {i['family_frac']} | {i['bound']} | {i['stage_ms']} | {i['stage_frac_of_min_bytes_floor']} |")
        if t:
            out.append(f"| training epoch | `{t['kernel']}` ({t['avg_us']} us, {t['share_of_stage_kernel_time']}) | {t['frac']} | {t['family_frac']} | {t['bound']}: T(B) = {t['chain_ms']} ms + B x {t['per_image_us']} us (steps of 8 / 16 / 32 images: {', '.join(str(v) for v in t['step_ms_by_batch'].values())} ms, measured in the run) | {t['stage_ms']} | {t['stage_frac_of_min_bytes_floor']} |")
        out.append("")
    tr = list(csv.DictReader(open(os.path.join(HERE, f"{R}_timed_region_kernel_stats.csv"))))
    for stage in ("inference", "training"):
        rows = [x for x in tr if x.get("stage") == stage][:10]
        out.append(f"Timed region, stage `{stage}`, top 10 by time (`r05_timed_region_kernel_stats.csv`):\n")
        out.append("| kernel | calls | avg us | % of the stage's kernel time | algorithmic MB / launch | GB/s | of 8 TB/s |\n|---|---|---|---|---|---|---|")
        for x in rows:
            out.append(f"| `{x['kernel'][:70]}` | {x['calls']} | {x['avg_us']} | {x['percent_of_stage_kernel_time']} | {x['algorithmic_MB_per_launch']} | {x['GBps']} | {x['frac_of_8TBps']} |")
        out.append("")
    out.append("""### The other BASELINE shapes (`bench.py --config`)

| shape | images/s | ms / generation | inference + IM ms | training epoch ms | dominant family, roofline | CPU restatement images/s |
|---|---|---|---|---|---|---|""")
    out.append(bench_line("ISIC, alpha 0.5 (default)", b))
    for tag, d in cfgs:
        if d:
            out.append(bench_line(tag, d))
    out.append("")
    rows = [("ISIC alpha 0.5", ("isic", 0.5), "[<= 0.88]"), ("HeLa alpha 1", ("hela", 1.0), ""), ("SUIM alpha 1", ("suim", 1.0), "[<= 1.5]"),
            ("Cityscapes alpha 1", ("city", 1.0), ""), ("Cityscapes alpha 1.25", ("city", 1.25), "[<= 3.3]"), ("Cityscapes alpha 1.5", ("city", 1.5), ""),
            ("Cityscapes alpha 1.75", ("city", 1.75), ""), ("Cityscapes alpha 2", ("city", 2.0), "[<= 4.3]"), ("ISIC alpha 1.5", ("isic", 1.5), ""),
            ("EvalNet alpha 2 (batch 32 both)", ("evalnet", 2.0), "[<= 2.0]")]
    out.append("### Training step / inference call, ms (`r05_configs_step_times.txt`; round 4 in brackets; the verdict's targets in square brackets)\n")
    out.append("| shape | training step, batch 32 | inference call, 128 images |\n|---|---|---|")
    for name, k, tgt in rows:
        out.append(f"| {name} | {g(now, k)[0]} ({g(was, k)[0]}) {tgt} | {g(now, k)[1]} ({g(was, k)[1]}) |")
    out.append("""
None of the targets is met, and this round says why in measurements rather than hopes: every kernel-level change that keeps the launch count
left the steps where they were (notes, section 2), and the one structural candidate -- the 28 BatchNorm-reduction launches -- was probed in its
last two forms and costs what the launch costs (section 1).
""")
    out.append(open(os.path.join(HERE, "r05_notes.md")).read())
    text = "\n".join(out)
    p = os.path.join(HERE, "README.md")
    s = open(p).read()
    a, z = "<!-- r05:begin -->\n", "<!-- r05:end -->\n"
    if a in s:
        s = s[:s.index(a) + len(a)] + text + s[s.index(z):]
    else:
        s = s.replace("<!-- r04:begin -->", a + text + z + "\n<!-- r04:begin -->", 1)
    open(p, "w").write(s)
    print("profiles/README.md: round-5 section written,", len(text), "characters")


if __name__ == "__main__":
    main()
