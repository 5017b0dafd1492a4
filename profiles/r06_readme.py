#!/usr/bin/env python3
"""Writes the round-6 section of profiles/README.md (between the r06 markers) from the r06_* files in this directory:
   python profiles/r06_readme.py        (after `RND=r06 bash profiles/collect_round.sh` on the GPU box, `python profiles/install_round.py r06`,
                                         `python profiles/traffic_ratio.py r06`)"""
import csv
import json
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
R = "r06"
TAGS = [("ISIC, alpha 0.5 (default; BASELINE configs[1])", ""), ("SUIM, alpha 1 (configs[2])", "_suim"), ("Cityscapes, alpha 1 (configs[3], first IM+ width)", "_cityscapes"),
        ("Cityscapes, alpha 1.25", "_cityscapes_a125"), ("Cityscapes, alpha 2 (last IM+ width)", "_cityscapes_a2"), ("HeLa, alpha 1 (configs[4])", "_hela")]


def load(name):
    p = os.path.join(HERE, name)
    return json.load(open(p)) if os.path.exists(p) else None


def rows_of(name):
    p = os.path.join(HERE, name)
    return list(csv.DictReader(open(p))) if os.path.exists(p) else []


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


def family_table(tag, stage, n=8):
    """training / inference kernel time by family from the excess table: launches, ms, floor ms, share"""
    rows = [r for r in rows_of(f"{R}_excess_by_kernel{tag}.csv") if r["stage"] == stage]
    fam = {}
    for r in rows:
        k = r["kernel"].split("<")[0]
        f = fam.setdefault(k, [0, 0.0, 0.0])
        f[0] += int(r["calls"]); f[1] += float(r["total_ms"]); f[2] += float(r["floor_us"] or 0) * int(r["calls"]) / 1e3
    tot = sum(f[1] for f in fam.values()) or 1.0
    out = ["| family | launches | kernel ms (summed over streams) | floor ms | floor / time | share |", "|---|---|---|---|---|---|"]
    for k, f in sorted(fam.items(), key=lambda kv: -kv[1][1])[:n]:
        out.append(f"| `{k}` | {f[0]} | {f[1]:.2f} | {f[2]:.2f} | {f[2] / max(f[1], 1e-9):.2f} | {100 * f[1] / tot:.1f} % |")
    out.append(f"| all | {sum(f[0] for f in fam.values())} | {tot:.2f} | {sum(f[2] for f in fam.values()):.2f} | {sum(f[2] for f in fam.values()) / tot:.2f} | |")
    return "\n".join(out)


def main():
    out = []
    out.append("## Round 6 (`r06_*`)\n")
    out.append("Commands: `RND=r06 bash profiles/collect_round.sh` (one `gpurun` call), `install_round.py r06`, `traffic_ratio.py r06`, this section: `r06_readme.py`.  "
               "Every configuration has ITS OWN files (`<tag>` = none for the default, `_suim`, `_cityscapes`, `_cityscapes_a125`, `_cityscapes_a2`, `_hela`) and "
               "`r06_provenance<tag>.json` records the `bench.py` sha256 and the kernel-source id they were collected with: `bench.py` replays a value only from "
               "the files of the configuration it runs, and only when both match the running tree.\n")
    out.append("""| file | what |
|---|---|
| `r06_bench<tag>.json`, `r06_bench_detail<tag>.json` | `python bench.py [--config ... [--alpha A]]`: the ONE compact stdout line (2.8 KB; the driver parses it) and the full record it points to (`detail`: all families, exclusive pass, per-stage kernel totals, thread calibration, layerwise parity, PNG rates) |
| `r06_bench_under_rocprof<tag>.json`, `r06_kernel_stats<tag>.csv`, `r06_rocprofv3_kernel_stats_raw.csv` | the same command under `rocprofv3 --kernel-trace --stats` (default: `--no-other-configs`; others: `--steps 1`), kernel statistics of the whole process with shortened names / raw |
| `r06_timed_region_kernel_stats<tag>.csv` | the kernel trace cut at `bench.py`'s marker dispatches, per stage and kernel variant: calls, average us from rocprofv3's timestamps, the library's algorithmic MB / GFLOP per launch (EVERY launched kernel is counted under its own name now), GB/s, TFLOP/s, fractions of 8 TB/s / 2.5 PFLOP/s |
| `r06_excess_by_kernel<tag>.csv` | per (stage, variant): `floor_us = max(MB / 8 TB/s, GFLOP / 2.5 PFLOP/s)`, the side that binds, `excess_ms = calls x (avg_us - floor_us)`, sorted: which launches own the gap between a stage and its roofline |
| `r06_timed_region_family_union<tag>.csv` | per family and stage: sum of the launches' durations, the UNION of their intervals (launches of a family overlap on two streams), algorithmic GB, GB/s and fraction of 8 TB/s by both |
| `r06_pmc_traffic<tag>.csv`, `r06_traffic_vs_algorithmic.csv` | `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes (separate runs, `--kernel-trace` only beside them), 2 x fetch + write per launch; counter traffic over algorithmic bytes per variant of the dominant family (`traffic_ratio.py`) |
| `r06_provenance<tag>.json` | `bench_py_sha16`, `lib_build_id` at collection time + the files installed for the configuration |
| `r06_configs_step_times.txt` | wall time of a training step (batch 32) and a 128-image inference call, all shapes, the IM+ width schedule, EvalNet |
| `r06_step_timeline_*.txt`, `r06_step_timeline_single_stream_*.txt` | kernel-by-kernel timeline of one step and one inference call (two streams / every kernel alone) |
| `r06_sq_counters_{isic,cityscapes_a2}.csv` | SQ counters per kernel (what the waves do with their cycles; `by_stage.inference.bound` quotes them) |
| `r06_ab_wgrad_splits.txt`, `r06_ab_adepth.txt`, `r06_ab_family_sweep.txt` | raw output of the round's one-box A/B runs (notes, section 3) |
| `r06_full_driver_run_par{2,3,4,5}.txt`, `r06_full_driver_run_par{3,4}_epoch_turns.txt` | the real-size ISIC generation with 2-5 candidates side by side, one fresh box each, and with all five started and epochs taking turns (notes, section 4) |
| `r06_bench_repeats.txt`, `r06_full_driver_run_five_generations.txt` | the default bench command five times in a row on one box, final build (25.29-25.56 k images/s, +-0.5 %); the real-size ISIC driver over generations 0-4 (25 candidates x 50 epochs through PNG directories, three side by side): 77.3 s (round 5: 79.7 s) |
| `r06_final_check_*` | `tests/gpu_probe/final_check.sh` on the round's last commit: `pytest -m gpu`, `smoke()`, the driver's bench command (its last stdout line, parsed), 2- and 8-rank functional lines |
""")
    b, bd = load(f"{R}_bench.json"), load(f"{R}_bench_detail.json")
    if b and bd:
        r, cb = b["roofline"], b.get("cpu_baseline") or {}
        ps = cb.get("parity_sample") or {}
        out.append(f"""Headline (`r06_bench.json`; the line is {len(json.dumps(b))} bytes): **{b['value']:.0f} images/s per IM generation on 1 GPU** -- {b['ms_per_step']} ms per generation =
{b['stage_ms']['ensemble_infer_plus_im']} ms (ensemble forward + fused head / IM, calls of {b['config']['infer_batch']} images) + {b['stage_ms']['train_epoch']} ms ({b['config']['epoch_steps']} training steps of
{r['by_stage']['training']['step_ms']} ms; {b['config'].get('kept')} of the 2 335 pseudo-labelled pairs kept) -- next to {cb.get('value')} images/s for the CPU restatement ({cb.get('cpu_model')},
{cb.get('cores')} threads of {cb.get('host_cpus')} CPUs).  `cpu_baseline.parity_sample`: max |dp| {ps.get('max_abs_dp')}, {ps.get('decision_flip_rate')} of the decisions flip,
{ps.get('im_pixels_differing')} of {ps.get('im_pixels_total')} IM pixels differ (SURVEY H4: IM masks are bit-identical GIVEN identical probabilities).

`roofline` of the line: kernel `{r['kernel']}`, {r['achieved']} {r['unit']} = **{r['frac']}** of peak by HIP events in the timed region (live), **{r.get('frac_rocprof')}** from rocprofv3's
durations (bytes over the SUM of the launches' durations), **{r.get('frac_rocprof_union')}** over the UNION of their intervals (the two ensemble members' forwards and a
step's weight gradients overlap: `r06_timed_region_family_union.csv`), {r.get('exclusive_frac')} with every kernel alone on one stream; HBM traffic / algorithmic bytes
{r.get('traffic_over_algorithmic')} (`r06_traffic_vs_algorithmic.csv`).  By stage: inference `{r['by_stage']['inference'].get('kernel')}` at {r['by_stage']['inference'].get('frac')} of peak,
bound: {r['by_stage']['inference'].get('bound')}; the stage at {r['by_stage']['inference'].get('stage_frac')} of its minimum-bytes floor.  Training: T(B) = {r['by_stage']['training'].get('chain_ms')} ms +
B x {r['by_stage']['training'].get('per_image_us')} us, the stage at {r['by_stage']['training'].get('stage_frac')} of its minimum-bytes floor ({r['by_stage']['training'].get('bound')}).
`replayed_refused`: {r.get('replayed_refused')} (the plain run of the collection precedes the installation of its own profiles; `r06_final_check_bench.json` is the same command after it).
""")
    out.append("### All six configurations, one generation each (`r06_bench<tag>.json`)\n")
    out.append("| configuration | images/s | ms / generation | inference + IM ms | training epoch ms (steps x ms) | dominant family: achieved = frac (bound) | frac by rocprofv3 sum / union | traffic / algorithmic | CPU restatement images/s |")
    out.append("|---|---|---|---|---|---|---|---|---|")
    for name, tag in TAGS:
        d = load(f"{R}_bench{tag}.json")
        if not d:
            continue
        r = d["roofline"]
        t = r["by_stage"]["training"]
        if r.get("traffic_over_algorithmic") is None:      # (the per-configuration ratio files were written after these lines were cut)
            fr = [x for x in csv.reader(open(os.path.join(HERE, f"{R}_traffic_vs_algorithmic{tag}.csv"))) if x and x[0].startswith("FAMILY")] \
                if os.path.exists(os.path.join(HERE, f"{R}_traffic_vs_algorithmic{tag}.csv")) else []
            if fr:
                r = dict(r, traffic_over_algorithmic=f"{fr[0][4]} (`{R}_traffic_vs_algorithmic{tag}.csv`)")
        out.append(f"| {name} | {d['value']:.0f} | {d['ms_per_step']} | {d['stage_ms']['ensemble_infer_plus_im']} | {d['stage_ms']['train_epoch']} ({d['config']['epoch_steps']} x {t.get('step_ms')}) | "
                   f"`{r['kernel']}` {r['achieved']} {r['unit']} = {r['frac']} ({r['bound']}) | {r.get('frac_rocprof')} / {r.get('frac_rocprof_union')} | {r.get('traffic_over_algorithmic')} | {(d.get('cpu_baseline') or {}).get('value', '-')} |")
    out.append("\n(`frac by rocprofv3` and `traffic` are `null` in the collection's own plain runs -- the files they replay are installed afterwards; the union / sum "
               "tables themselves are in `r06_timed_region_family_union<tag>.csv`.)\n")
    for name, tag in TAGS:
        if not rows_of(f"{R}_excess_by_kernel{tag}.csv"):
            continue
        out.append(f"### {name}: kernel time by family against the per-launch floors (`r06_excess_by_kernel{tag}.csv`)\n")
        for stage in ("inference", "training"):
            out.append(f"Stage `{stage}`:\n")
            out.append(family_table(tag, stage))
            out.append("")
        un = [r for r in rows_of(f"{R}_timed_region_family_union{tag}.csv") if r["stage"] in ("inference", "training") and r.get("GBps_by_union")][:6]
        if un:
            out.append("Sum of durations against union of intervals, largest families (`r06_timed_region_family_union" + tag + ".csv`):\n")
            out.append("| stage | family | calls | sum ms | union ms | overlap | GB/s by sum | GB/s by union | of 8 TB/s by union |\n|---|---|---|---|---|---|---|---|---|")
            for r in un:
                out.append(f"| {r['stage']} | `{r['family']}` | {r['calls']} | {r['sum_ms']} | {r['union_ms']} | {r['overlap_factor']} | {r['GBps_by_sum']} | {r['GBps_by_union']} | {r['frac_of_8TBps_by_union']} |")
            out.append("")
    now, was = step_times(R), step_times("r05")
    g = lambda d, k: d.get(k, [None, None])
    rows = [("ISIC alpha 0.5", ("isic", 0.5), ""), ("HeLa alpha 1", ("hela", 1.0), ""), ("SUIM alpha 1", ("suim", 1.0), "[<= 1.5]"),
            ("Cityscapes alpha 1", ("city", 1.0), ""), ("Cityscapes alpha 1.25", ("city", 1.25), "[<= 3.3]"), ("Cityscapes alpha 1.5", ("city", 1.5), ""),
            ("Cityscapes alpha 1.75", ("city", 1.75), ""), ("Cityscapes alpha 2", ("city", 2.0), "[<= 4.2]"), ("ISIC alpha 1.5", ("isic", 1.5), ""),
            ("EvalNet alpha 2 (batch 32 both)", ("evalnet", 2.0), "[<= 1.95]")]
    out.append("### Training step / inference call, ms (`r06_configs_step_times.txt`; round 5 in brackets; the verdict's targets in square brackets)\n")
    out.append("| shape | training step, batch 32 | inference call, 128 images |\n|---|---|---|")
    for name, k, tgt in rows:
        out.append(f"| {name} | {g(now, k)[0]} ({g(was, k)[0]}) {tgt} | {g(now, k)[1]} ({g(was, k)[1]}) |")
    out.append("")
    out.append(open(os.path.join(HERE, "r06_notes.md")).read())
    text = "\n".join(out)
    p = os.path.join(HERE, "README.md")
    s = open(p).read()
    a, z = "<!-- r06:begin -->\n", "<!-- r06:end -->\n"
    if a in s:
        s = s[:s.index(a) + len(a)] + text + s[s.index(z):]
    else:
        s = s.replace("<!-- r05:begin -->", a + text + z + "\n<!-- r05:begin -->", 1)
    open(p, "w").write(s)
    print("profiles/README.md: round-6 section written,", len(text), "characters")


if __name__ == "__main__":
    main()
