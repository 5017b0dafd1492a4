"""Rewrites the generated parts of profiles/README.md (headline paragraph, dominant-kernel paragraph, family table, whole-stage
view of the newest round) from profiles/rNN_bench.json.  Usage: python profiles/refresh_readme.py r02"""
import json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
here = os.path.dirname(os.path.abspath(__file__))
d = json.load(open(os.path.join(here, f"{tag}_bench.json")))
r = d["roofline"]; fam = r["all_families"]; cb = d["cpu_baseline"]; ps = cb["parity_sample"]
p = os.path.join(here, "README.md")
s = open(p).read()

def between(s, a, b, new):
    i, j = s.index(a), s.index(b)
    return s[:i] + new + s[j:]

head = f'''Headline (`{tag}_bench.json`): **{d["value"]:.0f} images/s per IM generation on 1 GPU** -- {d["ms_per_step"]} ms per generation =
{d["stage_ms"]["ensemble_infer_plus_im"]} ms (ensemble forward + fused head/IM) + {d["stage_ms"]["train_epoch"]} ms ({d["config"]["epoch_steps"]} training steps: {d["config"]["kept"]} of the 2 335 pseudo-labelled
pairs are kept with this round's ensemble, 2 255 / 78 steps with round 1's) -- next to {cb["value"]} images/s for the CPU restatement
(median of 3; {cb["threads_forward"]} threads for the batch-1 forwards, {cb["threads_train_step"]} for the training step, calibrated separately:
`cpu_baseline.thread_calibration_s`; 12.9-15.4 over the round's boxes).  Round 1 ended at 19 990 (116.8 ms).  Run-to-run
spread between boxes is +-1.5 %; A/B decisions below were taken inside one run.  `cpu_baseline.parity_sample`: on {ps["images"]} of the
bench's images and its trained ensemble the GPU probabilities differ from the fp16-emulating oracle's by at most {ps["max_abs_dp"]},
{ps["decision_flip_rate"]:.1e} of the decisions flip, {ps["im_pixels_differing"]} of {ps["im_pixels_total"]} IM pixels differ.

'''
s = between(s, "Headline (`%s_bench.json`):" % tag, "### Roofline, every kernel family", head)
para = f'''`roofline.kernel` = `{r["kernel"]}`: {r["achieved"]} GB/s = **{r["frac"]} of peak** in the timed region ({r["avg_us_per_launch"]} us per launch for
{r["avg_algorithmic_bytes_per_launch"]/1e6:.1f} MB algorithmic), {r["exclusive"]["achieved"]} GB/s = {r["exclusive"]["frac"]} alone on the chip; HBM traffic {r["traffic"]/1e6:.1f} MB per launch =
{r["traffic"]/r["avg_algorithmic_bytes_per_launch"]:.2f} x algorithmic (halo re-reads).  Round 1 reported 0.31 / 0.43 for a kernel that did less: eight weight gradients now come
out of these launches (same algorithmic bytes -- the operands are the ones the dgrad reads -- more time per launch), which is
why the whole-step figure below is the one to follow.

'''
s = between(s, "`roofline.kernel` = `", "| family | sampled launches", para)
rows = "".join(f"| `{k}` | {v['launches']} | {v['avg_us']} | {v['GBps']} | {round(v['GBps']/8000,3) if v['GBps'] else ''} | {v['share_of_sampled_time']} | {r['exclusive']['all_families_GBps'].get(k,'')} |\n"
               for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"]))
tbl = ("| family | sampled launches | avg us | GB/s (algorithmic, timed region) | of 8 TB/s | share of sampled kernel time | GB/s alone |\n"
       "|---|---|---|---|---|---|---|\n" + rows + "\n")
s = between(s, "| family | sampled launches", "Whole-stage view (`roofline.step`", tbl)
st = r["step"]
ws = f'''Whole-stage view (`roofline.step`, SURVEY 8d minimum bytes / measured stage time): inference stage {st["ensemble_infer_plus_im"]["GBps"]} GB/s =
**{st["ensemble_infer_plus_im"]["frac"]}** of the floor (round 1: 0.30), training step {st["train_step"]["GBps"]} GB/s = **{st["train_step"]["frac"]}** ({st["train_step"]["ms"]} ms per step
in the bench's epoch loop; round 1: 0.11 at 1.22 ms by the same formula).

'''
s = between(s, "Whole-stage view (`roofline.step`", "### What changed the numbers in round 2", ws)
open(p, "w").write(s)
print("refreshed", p, "from", tag, d["value"])
