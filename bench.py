#!/usr/bin/env python3
"""Benchmark of the north-star metric: images/sec per IM generation (ensemble inference + IM creation +
one U-Net training epoch) on synthetic data of a BASELINE.json shape.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling strong|weak] [--config isic|suim|cityscapes|hela] [--alpha A]

Default = the configuration the metric is quoted on (BASELINE.json configs[1]): ISIC-2018-shaped 256x256x3, 2-model
ensemble, alpha = 0.5.  --config picks one of the other shapes (configs[2..4]); --alpha overrides the width (the IM+
schedule of Cityscapes/11_Cityscapes_IM+.py:48 runs alpha 1 -> 2).

N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (the ranks read
RANK / LOCAL_RANK / WORLD_SIZE), or started plainly -- then this process spawns exactly that launcher as a child BEFORE
it touches the GPU, waits, and exits with the child's code.  `n_gpus` in the output is dist.get_world_size(); a world
size different from --gpus is an error.

One "step" = one IM generation:
  1. N-model ensemble forward over the unlabeled images (batches of --infer-batch) fused with the IM chain
     (threshold / argmax -> agreement -> IM -> blocking -> sizes)              functions.py:2844-2887, 3123-3137
  2. ISIC only: keep rule pred_size > im_size and pred_size > 0                functions.py:2878-2886
  3. one epoch (batch 32 per GPU) of a fresh U-Net on the pseudo-labels + labelled set, mse (sigmoid heads) /
     categorical cross-entropy (softmax heads), tfa-AdamW(3e-3, 1e-4)          functions.py:207-218
Inputs are resident in HBM when the timed region starts (PNG decode/encode is excluded on both the GPU and
the CPU side).

Multi-GPU, default `--scaling strong` (the north star: "shard the unlabeled image set across the GPUs"): ONE
set; rank r runs inference + IM on its contiguous block of the sorted set (functions.shard_list, no
collective) and trains on its own kept pairs + its block of the labelled set with batch 32 per GPU, one all-reduce of
the flat fp32 gradient buffer per step (RCCL), epoch steps = images // (32 * N).  value = images / time per generation.
After the timed region rank 0 recomputes the IM stage over the WHOLE set and checks that the ranks' summed IM size,
prediction size and kept count equal it (sharding does not change a single mask).
`--scaling weak`: every rank owns a full-size set (value = N * images / time).
IMK_FORCE_DIST=1 with --gpus 1: a real one-rank process group (backend nccl = RCCL), so that init, the flat all-reduce
and the barrier of the N-rank path execute on one GPU.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before the HIP runtime starts: see inconsistencymasks_amd/__init__.py

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# BASELINE.json configs; sizes from SURVEY 8d (the reference never states dataset counts: --images / --labeled override)
CONFIGS = {
    "isic": dict(h=256, w=256, c=3, k=1, alpha=0.5, n_models=2, act="sigmoid", loss=0, unlabeled=2335, labeled=259,
                 thr=0.5, cmp_ge=False, keep_rule=True,
                 workload="ISIC-2018 binary 256x256x3, 2-model IM ensemble, tiny U-Net alpha=0.5 on MI355X (configs[1])"),
    "suim": dict(h=256, w=256, c=3, k=9, alpha=1.0, n_models=3, act="softmax", loss=1, unlabeled=2468, labeled=274,
                 thr=0.5, cmp_ge=False, keep_rule=False,
                 workload="SUIM multi-class 256x256x3 (9 outputs = IM class + 8), 3-model IM ensemble, alpha=1 (configs[2])"),
    "cityscapes": dict(h=208, w=416, c=3, k=35, alpha=1.0, n_models=2, act="softmax", loss=1, unlabeled=2678, labeled=297,
                       thr=0.5, cmp_ge=False, keep_rule=False,
                       workload="Cityscapes 208x416x3 (config.ini:82-85; 35 classes), 2-model IM ensemble, alpha=1 -- the IM+ "
                                "schedule grows alpha to 2 (configs[3])"),
    "hela": dict(h=256, w=256, c=1, k=3, alpha=1.0, n_models=2, act="sigmoid", loss=0, unlabeled=1800, labeled=200,
                 thr=0.5, cmp_ge=True, keep_rule=False,
                 workload="HeLa 256x256x1 crops, 3 sigmoid maps (alive, dead, position), 2-model IM ensemble, alpha=1 (configs[4]; "
                          "set size not stated by the reference)"),
}
BATCH = 32
LR, WD = 3e-3, 1e-4
HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_PEAK_TFLOPS = 2500.0  # dense fp16 / bf16 MFMA
ROUND = "r06"            # profiles/<ROUND>_*: the only committed files a line may replay values from
LINE_BUDGET = 4096       # bytes of the LAST stdout line (the driver parses the tail of stdout; round 5's 21 KB line was cut)


def bench_py_sha16():
    import hashlib
    return hashlib.sha256(open(os.path.abspath(__file__), "rb").read()).hexdigest()[:16]


def lib_build_id():
    from inconsistencymasks_amd.build import source_id
    return source_id()


def config_tag(config_name, alpha):
    """suffix of the per-configuration files under profiles/: '' for the default (isic at its own width), '_suim',
    '_cityscapes_a2', '_cityscapes_a125', ..."""
    own = alpha is None or float(alpha) == float(CONFIGS[config_name]["alpha"])
    if config_name == "isic" and own:
        return ""
    return f"_{config_name}" + ("" if own else "_a" + f"{float(alpha):g}".replace(".", ""))


def provenance(tag=""):
    """What the committed profiles/<ROUND>_* files of configuration `tag` were collected with (profiles/collect_round.sh writes it):
    replayed values are printed only when BOTH the bench script and the kernel sources are the ones running now."""
    path = os.path.join(ROOT, "profiles", f"{ROUND}_provenance{tag}.json")
    try:
        rec = json.load(open(path))
    except Exception:
        return False, f"profiles/{ROUND}_provenance{tag}.json missing", None
    now = {"bench_py_sha16": bench_py_sha16(), "lib_build_id": lib_build_id()}
    for k, v in now.items():
        if rec.get(k) != v:
            return False, f"{k}: profiles {rec.get(k)} != running {v}", rec
    return True, None, rec


def _strip(name):
    return name.replace(", ", ",") if isinstance(name, str) else name


def compact_line(full, detail_path=None):
    """The ONE line the driver parses (last line of stdout, < LINE_BUDGET bytes): the contract's fields + roofline + cpu_baseline in
    short form.  Everything else of `full` (all families, exclusive pass, kernel totals, thread calibration, layerwise parity, PNG
    rates) goes to gpurun_out/bench_detail.json."""
    c, r = full["config"], full["roofline"]
    line = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                 "vs_baseline", "dtype", "data", "stage_ms")}
    line["config"] = {k: c.get(k) for k in ("workload", "name", "shape", "alpha", "outputs", "n_models", "unlabeled_images",
                                            "unlabeled_images_per_gpu", "labeled_images", "infer_batch", "infer_batch_rule",
                                            "train_batch_per_gpu", "global_batch", "parallelism", "process_group", "epoch_steps", "kept")}
    line["config"]["bn_momentum"] = (c.get("bn_momentum") or {}).get("value")
    rf = {k: r.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "launches", "avg_us_per_launch", "frac_rocprof",
                                "frac_rocprof_union", "rocprof_avg_us_per_launch", "traffic", "traffic_over_algorithmic", "live", "replayed_from",
                                "replayed_refused")}
    bs = r.get("by_stage") or {}
    ti, tt = bs.get("inference") or {}, bs.get("training") or {}
    rf["by_stage"] = {"inference": {"kernel": _strip(ti.get("kernel")), "frac": ti.get("frac"), "bound": ti.get("bound"),
                                    "stage_frac": ti.get("stage_frac_of_min_bytes_floor")},
                      "training": {"kernel": _strip(tt.get("kernel")), "frac": tt.get("family_frac"), "bound": tt.get("bound"),
                                   "chain_ms": tt.get("chain_ms"), "per_image_us": tt.get("per_image_us"), "step_ms": tt.get("step_ms"),
                                   "host_enqueue_ms": ((r.get("step") or {}).get("train_step") or {}).get("host_enqueue_ms_per_step"),
                                   "stage_frac": tt.get("stage_frac_of_min_bytes_floor")}}
    if r.get("exclusive"):
        rf["exclusive_frac"] = r["exclusive"].get("frac")
    line["roofline"] = rf
    if full.get("im_kernel"):
        line["im_kernel"] = {k: full["im_kernel"].get(k) for k in ("kernel", "GBps", "frac_of_hbm_peak")}
    cb = full.get("cpu_baseline")
    if cb:
        ps = cb.get("parity_sample") or {}
        line["cpu_baseline"] = {**{k: cb.get(k) for k in ("value", "unit", "cores", "host_cpus", "cpu_model", "kind", "sample",
                                                          "t_infer_per_image_s", "t_im_per_image_s", "t_train_step_s")},
                                "batched_value": (cb.get("batched_variant") or {}).get("value"),
                                "parity_sample": {k: ps.get(k) for k in ("images", "max_abs_dp", "decision_flip_rate",
                                                                         "im_pixels_differing", "im_pixels_total")}}
    oc = full.get("other_configs")
    if oc:
        line["other_configs"] = {"fields": ["value", "ms_per_step", "train_step_ms", "frac"],
                                 **{k: ([v.get("value"), v.get("ms_per_step"), v.get("train_step_ms"), (v.get("roofline") or {}).get("frac")]
                                        if "error" not in v else {"error": v["error"][:120]}) for k, v in oc.items()}}
    if full.get("sharding_check"):
        line["sharding_check"] = full["sharding_check"]
    line["detail"] = detail_path
    # the guard: a line over budget loses its optional parts (in this order) rather than its parseability
    for drop in (("cpu_baseline", "sample"), ("im_kernel",), ("other_configs",), ("roofline", "by_stage"), ("cpu_baseline", "parity_sample")):
        if len(json.dumps(line)) < LINE_BUDGET:
            break
        d = line
        for k in drop[:-1]:
            d = d.get(k, {})
        d.pop(drop[-1], None)
    assert len(json.dumps(line)) < LINE_BUDGET, len(json.dumps(line))
    return line


def write_detail(full, path=None):
    """the full record (what rounds 1-5 printed on one line) -> `path` (--detail), default gpurun_out/bench_detail[_<tag>].json (/tmp if
    the tree is read-only); returns the path written"""
    tag = config_tag(full["config"]["name"], full["config"]["alpha"])
    name = f"bench_detail{tag}" + ("" if full["n_gpus"] == 1 else f"_{full['n_gpus']}gpus") + ".json"
    for cand in ([os.path.abspath(path)] if path else []) + [os.path.join(ROOT, "gpurun_out", name), os.path.join("/tmp", name)]:
        try:
            os.makedirs(os.path.dirname(cand), exist_ok=True)
            with open(cand, "w") as f:
                json.dump(full, f, indent=1)
            return os.path.relpath(cand, ROOT) if cand.startswith(ROOT + os.sep) else cand
        except OSError:
            continue
    return None


def synth_images(cfg, n, seed, device):
    """Seeded images: smooth low-frequency field + elliptical blob + noise.  Labels: ISIC = the ellipse {0,255} [n,H,W,1];
    HeLa = 3 binary maps {0,255} [n,H,W,3] (two ellipses + small discs); multi-class = class ids [n,H,W] from the quantised
    field, the ellipse being one more class (0 = the reserved IM class is never a label)."""
    H, W, C, K = cfg["h"], cfg["w"], cfg["c"], cfg["k"]
    g = torch.Generator(device=device).manual_seed(seed)
    yy = torch.arange(H, device=device, dtype=torch.float32)[None, :, None]
    xx = torch.arange(W, device=device, dtype=torch.float32)[None, None, :]
    r = lambda *s: torch.rand(*s, device=device, generator=g)
    field = torch.zeros((n, H, W), device=device)
    for _ in range(4):
        fy, fx, ph = r(n, 1, 1) * 0.05, r(n, 1, 1) * 0.05, r(n, 1, 1) * 6.28
        field += torch.cos(yy * fy + xx * fx + ph)
    cy, cx = (0.3 + 0.4 * r(n, 1, 1)) * H, (0.3 + 0.4 * r(n, 1, 1)) * W
    ry, rx = (0.12 + 0.2 * r(n, 1, 1)) * H, (0.12 + 0.2 * r(n, 1, 1)) * W
    ell = (((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2) < 1
    base = 150 + 15 * field - ell * (40 + 50 * r(n, 1, 1))
    tint = torch.tensor([10.0, -5.0, -15.0, 0.0][:C], device=device) if C > 1 else torch.zeros(1, device=device)
    img = (base[..., None] + tint + (r(n, H, W, C) * 16 - 8)).clamp(0, 255).to(torch.uint8).contiguous()
    if cfg["act"] == "softmax":
        lab = (1 + ((field + 4.0) * 0.125 * (K - 2)).clamp(0, K - 3).to(torch.int64)).to(torch.uint8)   # 1 .. K-2
        lab = torch.where(ell, torch.full_like(lab, K - 1), lab)
        return img, lab.contiguous()
    if K == 1:
        return img, (ell.to(torch.uint8) * 255)[..., None].contiguous()
    cy2, cx2 = (0.3 + 0.4 * r(n, 1, 1)) * H, (0.3 + 0.4 * r(n, 1, 1)) * W
    ell2 = ((((yy - cy2) / (0.5 * ry)) ** 2 + ((xx - cx2) / (0.5 * rx)) ** 2) < 1) & ~ell
    pos = (((yy - cy) ** 2 + (xx - cx) ** 2) < 36) | (((yy - cy2) ** 2 + (xx - cx2) ** 2) < 36)
    return img, (torch.stack([ell, ell2, pos], -1).to(torch.uint8) * 255).contiguous()


def conv_flops_per_image(model, cfg):
    """2 x multiply-adds of one forward pass (conv layers only; SURVEY 8d: ISIC 0.625 G, SUIM 2.508 G, Cityscapes 3.384 G)"""
    lvl = {"in": 0, "e1": 0, "e2": 1, "e3": 2, "e4": 3, "b.": 4, "d6": 3, "d7": 2, "d8": 1, "d9": 0, "ou": 0}
    f = 0
    for l in model.plan.layers:
        if l["kind"] == 0:
            s = lvl[l["name"][:2]]
            f += 2 * (cfg["h"] >> s) * (cfg["w"] >> s) * l["ksize"] ** 2 * l["cin"] * l["cout"]
    return f


def forward_min_bytes_per_image(model, cfg):
    """minimum HBM bytes of one forward with one kernel per block and fp16 activations: every tensor that crosses a
    block boundary (stem, each block's output, each encoder block's pooled output) written once and read once + the uint8
    input + the fp32 output.  ISIC: 10 551 296 B."""
    ch = [int(v * cfg["alpha"]) for v in (16, 32, 64, 128, 256)]
    px = lambda s: (cfg["h"] >> s) * (cfg["w"] >> s)
    el = px(0) * ch[0]                                                   # stem
    for i in range(4):
        el += px(i) * ch[i] + px(i + 1) * ch[i]                          # encoder block output + its pooled copy
    el += px(4) * ch[3]                                                  # bottleneck output
    for j, c_out in enumerate((ch[2], ch[1], ch[0], ch[0])):
        el += px(3 - j) * c_out                                          # decoder block outputs
    return 2 * 2 * el + px(0) * cfg["c"] + px(0) * cfg["k"] * 4


def im_bytes_per_image(cfg, n_models):
    """the IM chain at the reference's boundary (fp32 probability stack in, image in / out, masks + IM out)"""
    px = cfg["h"] * cfg["w"]
    kb = cfg["k"] if cfg["act"] == "sigmoid" else 1
    return n_models * px * cfg["k"] * 4 + 2 * px * cfg["c"] + (kb + 1) * px


def _usable_cpus():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _median(v):
    v = sorted(v)
    return v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])


def _calibrate_threads(candidates, run, warm):
    """fastest thread count for `run` (tiny batch-1 convolutions get SLOWER with hundreds of threads; a batch-32 training
    step wants many more): stops once a setting is 1.5x slower than the best so far"""
    best = None
    tried = {}
    for nt in candidates:
        torch.set_num_threads(nt)
        warm()
        t0 = time.perf_counter()
        run()
        dt = time.perf_counter() - t0
        tried[nt] = round(dt, 4)
        if best is None or dt < best[1]:
            best = (nt, dt)
        if dt > 1.5 * best[1]:
            break
    return best[0], tried


def parity_sample(cfg, models, x_dev, n=4):
    """"also compare outputs" -- the bench's own ensemble on a few of its images, GPU against the oracle
    (fp16-emulating torch-CPU restatement): largest probability difference, decision flips, and the IM pixels that differ
    when the oracle's IM chain runs on the oracle's probabilities (the IM chain itself is bit-exact given equal inputs)."""
    from oracle import im_oracle, unet_oracle as U
    x = x_dev[:n].cpu().numpy()
    gpu = [m.predict_device(x_dev[:n]).cpu().numpy() for m in models]
    ref = [U.forward(m.state_dict(), x, cfg["c"], cfg["k"], cfg["alpha"], cfg["act"], emulate_fp16=True).numpy() for m in models]
    dp = max(float(np.abs(g - r).max()) for g, r in zip(gpu, ref))
    im_diff = 0
    if cfg["act"] == "sigmoid":
        dec = lambda p: (p >= cfg["thr"]) if cfg["cmp_ge"] else (p > cfg["thr"])
        flips = float(np.mean([(dec(g) != dec(r)).mean() for g, r in zip(gpu, ref)]))
        for i in range(n):
            a = im_oracle.im_binary(np.stack([g[i] for g in gpu], 0), cfg["thr"], cfg["cmp_ge"])["im"]
            b = im_oracle.im_binary(np.stack([r[i] for r in ref], 0), cfg["thr"], cfg["cmp_ge"])["im"]
            im_diff += int((a != b).sum())
    else:
        flips = float(np.mean([(g.argmax(-1) != r.argmax(-1)).mean() for g, r in zip(gpu, ref)]))
        for i in range(n):
            a = im_oracle.im_multiclass(np.stack([g[i] for g in gpu], 0))["im"]
            b = im_oracle.im_multiclass(np.stack([r[i] for r in ref], 0))["im"]
            im_diff += int((a != b).sum())
    # Where the gap opens: every stored layer of model 0 (materialized path) against the oracle's tensors, and the same sample on
    # FIXED weights (seeded init + perturbed BatchNorm parameters / statistics) -- the trained ensemble differs from run to run and
    # from round to round (its max |dp| follows how sharp that run's decision surfaces are), the fixed model does not
    rel = lambda a, b: float(np.sqrt(((a - b) ** 2).sum() / max((b ** 2).sum(), 1e-30)))
    layerwise = {}
    try:
        import torch as _t
        m0 = models[0]
        taps = {}
        U.forward(m0.state_dict(), x, cfg["c"], cfg["k"], cfg["alpha"], cfg["act"], emulate_fp16=True, taps=taps)
        m0.debug(materialize=True)
        m0.predict_device(x_dev[:n])
        for l in m0.plan.layers:
            if l["kind"] == 0 and l["name"] != "out":
                g = m0.intermediate(l["name"], n, 0).numpy()
                r = taps[l["name"]].numpy()
                layerwise[l["name"]] = [round(rel(g, r), 6), round(float(np.abs(g - r).max()), 5)]
        m0.debug(materialize=False)
        from inconsistencymasks_amd.unet import UNet as _UNet
        fx = _UNet(cfg["h"], cfg["w"], cfg["c"], cfg["k"], cfg["alpha"], cfg["act"], seed=20240, device=x_dev.device)
        sd = fx.state_dict()
        gq = _t.Generator().manual_seed(7)
        for k_, v_ in sd.items():
            if k_.endswith(".gamma"): sd[k_] = 0.8 + 0.4 * _t.rand(v_.shape, generator=gq)
            elif k_.endswith(".beta") or k_.endswith(".b"): sd[k_] = 0.1 * _t.randn(v_.shape, generator=gq)
            elif k_.endswith(".mean"): sd[k_] = 0.3 + 0.1 * _t.randn(v_.shape, generator=gq)
            elif k_.endswith(".var"): sd[k_] = 0.5 + 0.5 * _t.rand(v_.shape, generator=gq)
        fx.load_state_dict(sd)
        gf = fx.predict_device(x_dev[:n]).cpu().numpy()
        rf = U.forward(fx.state_dict(), x, cfg["c"], cfg["k"], cfg["alpha"], cfg["act"], emulate_fp16=True).numpy()
        fixed = {"max_abs_dp": round(float(np.abs(gf - rf).max()), 5), "rel_l2": round(rel(gf, rf), 6)}
    except Exception as e:       # diagnostic only: never takes the bench line down
        fixed = {"error": f"{type(e).__name__}: {e}"}
    return {"images": n, "max_abs_dp": round(dp, 5), "decision_flip_rate": round(flips, 6),
            "im_pixels_differing": im_diff, "im_pixels_total": n * cfg["h"] * cfg["w"],
            "fixed_weights": fixed, "layerwise_model0_rel_l2_maxabs": layerwise,
            "note": "GPU vs fp16-emulating oracle on the bench's trained ensemble (weights differ from run to run: compare rounds on "
                    "fixed_weights); layerwise: every stored conv output of model 0, [rel-L2, max abs]; tolerance of the parity tests: "
                    "|dp| <= 3e-2, per-layer rel-L2 <= 1e-2, flips <= 1 %"}


def cpu_baseline(cfg, n_unl, n_lab, fwd_flops):
    """The oracle (torch-CPU fp32 restatement, reference-structured: batch-1 forward per image per model, numpy IM,
    batch-32 training) on a bounded sample, extrapolated to one generation.  BASELINE.md section 3: >= 3 repetitions,
    median; thread counts calibrated SEPARATELY for the batch-1 forwards and for the batch-32 training step (up to every
    usable core), both reported, with the GFLOP/s the port reaches: it is a torch NCHW port on this host's cores, a stated
    baseline and not a tuned CPU implementation -- the GPU/CPU ratio says nothing about kernel quality."""
    from oracle import im_oracle, unet_oracle as U
    C, K, ALPHA, ACT = cfg["c"], cfg["k"], cfg["alpha"], cfg["act"]
    H, W, NM = cfg["h"], cfg["w"], cfg["n_models"]
    rng = np.random.default_rng(0)
    heavy = fwd_flops > 1.5e9
    n_img, reps = (8, 3) if heavy else (32, 3)
    x = rng.integers(0, 256, (max(n_img, BATCH), H, W, C)).astype(np.uint8)
    models = [U.init_weights(C, K, ALPHA, 1000 + j) for j in range(NM)]
    ncpu = _usable_cpus()
    cand = lambda lo: [n for n in (1, 2, 4, 8, 16, 32, 64, 128, 256) if lo <= n <= ncpu] or [1]
    fwd = lambda k: U.predict_batch1(models[0], x[:k], C, K, ALPHA, ACT)
    nt_fwd, tried_fwd = _calibrate_threads(cand(4), lambda: fwd(4), lambda: fwd(1))
    torch.set_num_threads(nt_fwd)
    t_inf_reps, t_im_reps = [], []
    for _ in range(reps):
        t0 = time.perf_counter()
        preds = [U.predict_batch1(m, x[:n_img], C, K, ALPHA, ACT) for m in models]
        t_inf_reps.append((time.perf_counter() - t0) / n_img)               # per image, all N models
        t0 = time.perf_counter()
        for i in range(n_img):
            st = np.stack([p[i] for p in preds], 0)
            if ACT == "sigmoid":
                r = im_oracle.im_binary(st, cfg["thr"], cfg["cmp_ge"])
                im_oracle.block(x[i], list(r["final"]), r["im"], True, True)
            else:
                r = im_oracle.im_multiclass(st)
                im_oracle.block(x[i], [r["final"]], r["im"], True, True)
        t_im_reps.append((time.perf_counter() - t0) / n_img)
    t_inf, t_im = _median(t_inf_reps), _median(t_im_reps)
    # the training step: its own thread count (full sweep up to every usable core)
    p = U.init_weights(C, K, ALPHA, 7)
    opt = U.new_opt_state(p)
    if ACT == "sigmoid":
        y = (rng.random((BATCH, H, W, K)) > 0.5).astype(np.float32)
        lk = "mse"
    else:
        y = np.eye(K, dtype=np.float32)[rng.integers(0, K, (BATCH, H, W))]
        lk = "cce"
    step = lambda: U.train_step(p, opt, x[:BATCH], y, C, K, ALPHA, ACT, lk)
    step()                                                                   # one-off autograd warm-up
    nt_train, tried_train = _calibrate_threads(cand(8), step, lambda: None)
    torch.set_num_threads(nt_train)
    t_step_reps = []
    for _ in range(reps if not heavy else 1):
        t0 = time.perf_counter()
        step()
        t_step_reps.append(time.perf_counter() - t0)
    t_step = _median(t_step_reps)
    # the same forwards as ONE batch per model (so that the ratio is not inflated by the reference's
    # batch-1 choice), at the training step's thread count; reported beside the reference-structured headline
    t0 = time.perf_counter()
    with torch.no_grad():
        for m in models:
            U.forward(m, x[:n_img], C, K, ALPHA, ACT)
    t_inf_batched = (time.perf_counter() - t0) / n_img
    steps = (n_unl + n_lab) // BATCH
    t_gen = n_unl * (t_inf + t_im) + steps * t_step
    return {"value": round(n_unl / t_gen, 3), "unit": "images/s", "cores": max(nt_fwd, nt_train), "kind": "port",
            "host_cpus": ncpu, "cpu_model": _cpu_model(),
            "implementation": "oracle/unet_oracle.py: torch-CPU fp32, NCHW, reference structure (batch-1 predict per image and model, "
                              "numpy IM, batch-32 autograd step); not a tuned CPU kernel library",
            "gflops_forward_batch1": round(NM * fwd_flops / t_inf / 1e9, 1),
            "gflops_train_step": round(3 * BATCH * fwd_flops / t_step / 1e9, 1),
            "sample": f"median of {reps} repetitions of: {n_img} images x {NM} models batch-1 fp32 forward + numpy IM "
                      f"({nt_fwd} threads), 1 train step of batch {BATCH} ({nt_train} threads); extrapolated to "
                      f"U={n_unl}, {steps} steps; {ncpu} CPUs usable",
            "threads_forward": nt_fwd, "threads_train_step": nt_train,
            "thread_calibration_s": {"forward_4_images": tried_fwd, "train_step": tried_train},
            "t_infer_per_image_s": round(t_inf, 5), "t_im_per_image_s": round(t_im, 6), "t_train_step_s": round(t_step, 4),
            "repetitions": {"t_infer_per_image_s": [round(v, 5) for v in t_inf_reps],
                            "t_train_step_s": [round(v, 4) for v in t_step_reps]},
            "batched_variant": {"t_infer_per_image_s": round(t_inf_batched, 5),
                                "value": round(n_unl / (n_unl * (t_inf_batched + t_im) + steps * t_step), 3),
                                "note": f"forwards as one batch of {n_img} per model instead of batch 1, {nt_train} threads"}}


def png_io_rate(images):
    """PNG encode / decode rate of the host path the directory API uses (libimk's own codec, csrc/imk_png.cpp, one file per call on a
    thread pool -- ctypes drops the interpreter lock; SURVEY H5: excluded from the headline on both the GPU and the CPU side,
    reported separately), with Pillow on the same pool beside it (what rounds 1-4 used)."""
    import ctypes
    import io
    from concurrent.futures import ThreadPoolExecutor
    from PIL import Image
    from inconsistencymasks_amd import functions as F
    from inconsistencymasks_amd._lib import lib
    threads = F._IO_THREADS
    sq = lambda a: a[..., 0] if a.shape[-1] == 1 else a
    h, w, c = images[0].shape
    def enc(a):
        a = np.ascontiguousarray(a)
        buf = np.empty(a.nbytes + h + a.nbytes // 500 + 4096, dtype=np.uint8)
        n = ctypes.c_int64()
        assert lib.imk_png_encode(a.ctypes.data, h, w, c, 1, buf.ctypes.data, buf.nbytes, ctypes.byref(n)) == 0
        return buf[:n.value].tobytes()
    def dec(b):
        out = np.empty((h, w, c), dtype=np.uint8)
        assert lib.imk_png_decode(b, len(b), c, out.ctypes.data, out.nbytes, None, None) == 0
        return out
    def enc_p(a):
        b = io.BytesIO()
        Image.fromarray(sq(a)).save(b, format="PNG", compress_level=1)
        return b.getvalue()
    def dec_p(b):
        return np.asarray(Image.open(io.BytesIO(b)))
    with ThreadPoolExecutor(max_workers=threads) as pool:
        t0 = time.perf_counter(); blobs = list(pool.map(enc, images)); t1 = time.perf_counter()
        back = list(pool.map(dec, blobs)); t2 = time.perf_counter()
        t3 = time.perf_counter(); blobs_p = list(pool.map(enc_p, images)); t4 = time.perf_counter()
        back_p = list(pool.map(dec_p, blobs_p)); t5 = time.perf_counter()
    assert all(np.array_equal(b, a) for a, b in zip(images, back)) and np.array_equal(sq(back_p[0]), sq(images[0]))
    assert np.array_equal(dec_p(blobs[0]).reshape(images[0].shape), images[0])       # Pillow reads what the native encoder wrote
    return {"encode_images_per_s": round(len(images) / (t1 - t0), 1), "decode_images_per_s": round(len(images) / (t2 - t1), 1),
            "threads": threads, "codec": "libimk (imk_png_encode / imk_png_decode: zlib level 1 Z_RLE, filters None / Sub / Up per row)",
            "pillow_encode_images_per_s": round(len(images) / (t4 - t3), 1), "pillow_decode_images_per_s": round(len(images) / (t5 - t4), 1),
            "sample": f"{len(images)} images {'x'.join(map(str, images[0].shape))}"}


FAMILIES = ["conv_mfma_kernel<16,1>", "conv_mfma_kernel<16,2>", "conv_mfma_kernel<16,4>", "conv_mfma_kernel<8,1>",
            "conv_mfma_kernel<8,2>", "conv_mfma_kernel<8,4>", "conv_pipe_kernel", "wgrad_mfma_kernel", "bn_bwd_prep_kernel",
            "bn_bwd_coef_kernel", "bn_finalize_kernel", "wgf_stage1+2_kernel", "head_kernel", "head_loss_kernel",
            "step_tail(loss_finalize,adamw,pack,fold)", "im_kernel", "conv_gemm_kernel", "wgrad_gemm_kernel"]


def self_launch(args, argv):
    """--gpus N > 1 without a launcher: start `python -m torch.distributed.run` as a CHILD (never exec: nothing in this
    process has touched the GPU yet, and it must stay that way) and return its exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd, env={**os.environ, "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong")
    ap.add_argument("--config", choices=tuple(CONFIGS), default="isic")
    ap.add_argument("--alpha", type=float, default=None, help="width multiplier (default: the config's)")
    ap.add_argument("--infer-batch", type=int, default=None, help="images per ensemble call (default 584 at alpha <= 0.5, 128 for the wider nets)")
    ap.add_argument("--images", type=int, default=None, help="unlabeled images (default: the config's set size)")
    ap.add_argument("--labeled", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="no per-launch event timing (roofline fields empty)")
    ap.add_argument("--step-events", action="store_true", help="diagnostic: one event per training step, percentiles on stderr")
    ap.add_argument("--prof-period", type=int, default=61, help="time every k-th hooked kernel launch with HIP events")
    ap.add_argument("--pretrain-steps", type=int, default=300)
    ap.add_argument("--bn-settle-steps", type=int, default=600)
    ap.add_argument("--detail", default=None, help="where the full record goes (default gpurun_out/bench_detail[_<config>].json)")
    ap.add_argument("--provenance", action="store_true",
                    help="print {bench_py_sha16, lib_build_id} (what profiles/collect_round.sh records beside the files it cuts) and exit")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="default run (isic, 1 GPU) only: skip the short runs of the other BASELINE shapes appended as other_configs")
    args = ap.parse_args()
    if args.provenance:       # before anything touches the GPU
        print(json.dumps({"bench_py_sha16": bench_py_sha16(), "lib_build_id": lib_build_id(), "round": ROUND}))
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # IMK_BENCH_ONE_GPU=1 + IMK_BENCH_BACKEND=gloo: functional test of the multi-rank path on a single-GPU box
    if os.environ.get("IMK_BENCH_ONE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    force_dist = os.environ.get("IMK_FORCE_DIST") == "1"
    use_dist = world > 1 or force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:      # the one-rank group of IMK_FORCE_DIST: no launcher set the rendezvous up
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
        backend = os.environ.get("IMK_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the process group has {dist.get_world_size()} ranks")
        w = torch.zeros(1, device=dev)
        dist.all_reduce(w)                     # communicator set up now (RCCL prints its version banner through C stdio here) ...
        torch.cuda.synchronize()
        import ctypes as _ct
        _ct.CDLL(None).fflush(None)            # ... and flushed now, so that the JSON line below stays the LAST line of stdout
    n_gpus = dist.get_world_size() if use_dist else 1
    env = dict(rank=rank, world=world, dev=dev, dist=dist, use_dist=use_dist, n_gpus=n_gpus)
    out = run_config(args, args.config, args.alpha, env, primary=True)
    # The other BASELINE shapes (configs[2..4] and the last IM+ width), two generations each, so that the one default command
    # times them too: same generation(), same line fields (value, ms_per_step, stage_ms, roofline).  Only for the plain default
    # run -- an explicit --config / --alpha / --images run stays what was asked for.
    default_run = (world == 1 and not use_dist and args.config == "isic" and args.alpha is None and args.images is None
                   and args.labeled is None and not args.no_other_configs)
    if default_run:
        others = {}
        for key, name, alpha in (("suim", "suim", None), ("cityscapes", "cityscapes", None), ("hela", "hela", None),
                                 ("cityscapes_a2", "cityscapes", 2.0)):
            a2 = argparse.Namespace(**vars(args))
            a2.steps, a2.warmup, a2.no_cpu_baseline, a2.infer_batch, a2.step_events = 2, 1, True, None, False
            t0 = time.perf_counter()
            try:
                o = run_config(a2, name, alpha, env, primary=False)
                r = o["roofline"]
                others[key] = {"value": o["value"], "unit": o["unit"], "steps": o["steps"], "ms_per_step": o["ms_per_step"],
                               "stage_ms": o["stage_ms"], "workload": o["config"]["workload"], "alpha": o["config"]["alpha"],
                               "epoch_steps": o["config"].get("epoch_steps"), "n_models": o["config"]["n_models"],
                               "train_step_ms": r["step"]["train_step"]["ms"],
                               "roofline": {k: r.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic",
                                                                  "traffic_source", "frac_rocprof", "avg_us_per_launch")},
                               "wall_s": round(time.perf_counter() - t0, 1)}
            except Exception as e:      # the headline must survive a failure here; the failure is reported, not hidden
                others[key] = {"error": f"{type(e).__name__}: {e}"}
        out["other_configs"] = others
    if rank == 0 and out is not None:
        detail = write_detail(out, args.detail)
        line = compact_line(out, detail)
        print(f"[bench] full record: {detail} ({len(json.dumps(out))} bytes); the line below: {len(json.dumps(line))} bytes",
              file=sys.stderr, flush=True)
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def run_config(args, config_name, alpha, env, primary):
    """One bench line (dict, on rank 0; None elsewhere) for BASELINE shape `config_name` at width `alpha` (None: the shape's own)."""
    rank, world, dev, dist, use_dist, n_gpus = (env[k] for k in ("rank", "world", "dev", "dist", "use_dist", "n_gpus"))
    cfg = dict(CONFIGS[config_name])
    if alpha is not None:
        cfg["alpha"] = alpha
        cfg["workload"] += f" [--alpha {alpha:g}]"
    H, W, C, K, ALPHA, N_MODELS, ACT, LOSS = (cfg[k] for k in ("h", "w", "c", "k", "alpha", "n_models", "act", "loss"))
    binary = ACT == "sigmoid"
    strong = args.scaling == "strong"

    import ctypes
    from inconsistencymasks_amd import functions as F
    from inconsistencymasks_amd import im as imk_im
    from inconsistencymasks_amd.unet import UNet

    NV = len(FAMILIES)
    # HIP events (on the launch stream) around every k-th hooked launch, from process start: the setup phase is
    # collected separately so that the whole-process average can be compared with `rocprofv3 --stats` of this command
    from inconsistencymasks_amd.prof import Profiler
    prof = Profiler(0 if args.no_prof else args.prof_period)
    prof_collect = prof.collect

    # ---- synthetic, HBM-resident inputs ---------------------------------------------------------------------------
    # strong: every rank generates THE set (seed 42) and keeps its contiguous block (functions.shard_list's rule);
    # weak:   every rank owns a full-size set of its own
    U_total = args.images if args.images is not None else cfg["unlabeled"]
    L_total = args.labeled if args.labeled is not None else cfg["labeled"]
    if strong:
        x_all, _ = synth_images(cfg, U_total, 42, dev)
        xl_all, ml_all = synth_images(cfg, L_total, 4242, dev)
        cut = lambda n: ((n * rank) // world, (n * (rank + 1)) // world)
        (u0, u1), (l0, l1) = cut(U_total), cut(L_total)
        x_unl = x_all[u0:u1].contiguous()
        x_lab, m_lab = xl_all[l0:l1].contiguous(), ml_all[l0:l1].contiguous()
        x_pre, m_pre = xl_all, ml_all                                # the ensemble is (pre)trained on the whole labelled set
        if rank != 0:
            del x_all
    else:
        x_unl, _ = synth_images(cfg, U_total, 42 + 1000 * rank, dev)
        x_lab, m_lab = synth_images(cfg, L_total, 4242 + 1000 * rank, dev)
        x_pre, m_pre = x_lab, m_lab
    # training targets as the parsers make them: ISIC 255 -> 1 (functions.py:955-977), HeLa position x 3 (functions.py:1011),
    # multi-class: the class-id map (one-hot inside the loss kernel)
    pos_w = torch.tensor([1, 1, 3], dtype=torch.uint8, device=dev)
    def targets(m):
        if not binary:
            return m
        t = m // 255
        return (t * pos_w).contiguous() if K == 3 else t.contiguous()
    y_pre = targets(m_pre)
    U = x_unl.shape[0]
    infer_batch = F.infer_batch_size(ALPHA, args.infer_batch)      # the product writers' rule (functions.infer_batch_size): 584 at alpha <= 0.5, else 128

    # ---- ensemble: seeded he_normal, briefly trained on the labelled set so that predictions are not noise (the same
    # on every rank: replicated weights)
    models = []
    for j in range(N_MODELS):
        m = UNet(H, W, C, K, ALPHA, ACT, seed=1000 + j, device=dev)
        g = torch.Generator(device=dev).manual_seed(j)
        for it in range(args.pretrain_steps + args.bn_settle_steps):
            idx = torch.randint(0, x_pre.shape[0], (BATCH,), device=dev, generator=g)
            # Keras BN momentum 0.99 needs ~500 steps before the moving statistics (what inference uses) have
            # forgotten their initial values; the reference trains 4050 steps.  Settle them with lr = wd = 0.
            lr, wd = (LR, WD) if it < args.pretrain_steps else (0.0, 0.0)
            m.train_step(x_pre[idx].contiguous(), y_pre[idx].contiguous(), LOSS, lr, wd)
        m.repack()   # fold the moving statistics for inference
        models.append(m)
    ens = F.EnsembleIM(models)
    student = UNet(H, W, C, K, ALPHA, ACT, seed=7, device=dev)
    bn_rule, bn_mom = F.dp_bn_momentum_rule(world)      # Keras' 0.99 at any world size; IMK_DP_BN_MOMENTUM=scaled: 0.99^N (functions.py)
    student.set_bn_momentum(bn_mom)
    init_params = student.params.clone()
    gen_perm = torch.Generator(device=dev).manual_seed(42 + rank)
    fwd_flops = conv_flops_per_image(student, cfg)
    fwd_min_bytes = forward_min_bytes_per_image(student, cfg)

    ev = lambda: torch.cuda.Event(enable_timing=True)
    info = {}

    # Whole-set buffers: the IM kernel writes every batch's blocked images / pseudo-labels straight into [0, U), the
    # labelled pairs sit behind them once -- the "directory" the reference builds (kept pseudo-labelled pairs + labelled
    # pairs) is then ONE gather with the composed (keep, shuffle) index instead of cat / index / cat / index passes.
    KB = K if binary else 1
    n_lab = x_lab.shape[0]
    pool_x = torch.empty((U + n_lab, H, W, C), dtype=torch.uint8, device=dev)
    pool_y = torch.empty((U + n_lab, KB, H, W), dtype=torch.uint8, device=dev)
    pool_x[U:] = x_lab
    pool_y[U:] = m_lab.permute(0, 3, 1, 2) if binary else m_lab[:, None]      # {0, 255} masks / class ids, like the written files
    lab_idx = torch.arange(U, U + n_lab, device=dev)
    steps_cap = torch.zeros(1, dtype=torch.int64, device=dev)

    infer_batches = lambda n: F.infer_batches(n, infer_batch)      # a short last call is spread over the others (functions.infer_batches)

    def im_stage(x, into_pool=True):
        ps, ims = [], []
        for i, j in infer_batches(x.shape[0]):
            r = ens.run(x[i:j], cfg["thr"], cfg["cmp_ge"], True, True,
                        out={"img_out": pool_x[i:j], "masks": pool_y[i:j]} if into_pool else None)
            ps.append(r["pred_size"].sum(1)); ims.append(r["im_size"].sum(1))
        ps, ims = torch.cat(ps), torch.cat(ims)
        keep = ((ps > ims) & (ps > 0)) if cfg["keep_rule"] else torch.ones_like(ps, dtype=torch.bool)   # functions.py:2878-2886 (ISIC only)
        return ps, ims, keep

    host_enqueue = []
    step_events = []
    last_batch = []
    stage_marks = [False]       # timed region under rocprofv3: marker dispatches at the stage boundaries (profiles/summarize.py)
    stage_snap = [None]         # one generation outside the timed region: the library's per-kernel totals after the inference stage

    def generation(record=None):
        e0, e1, e2 = ev(), ev(), ev()
        if stage_marks[0]:
            Profiler.mark(4)
        e0.record()
        ps, ims, keep = im_stage(x_unl)
        e1.record()
        if stage_marks[0]:
            Profiler.mark(3)
        if stage_snap[0] is not None:
            stage_snap[0]["inference"] = prof.totals_dump()
        # training set = kept pseudo-labelled pairs + labelled pairs (the directory the reference builds)
        if use_dist and world > 1:   # every rank must run the same number of gradient all-reduces: the smallest shard decides.
            # Issued BEFORE the keep rule's host sync below, so that the agreed count is there when the host wakes up: one
            # host round trip per generation, not two
            if cfg["keep_rule"]:
                steps_cap.copy_(((keep.sum() + n_lab) // BATCH).reshape(1))
            else:
                steps_cap.fill_((U + n_lab) // BATCH)
            dist.all_reduce(steps_cap, op=dist.ReduceOp.MIN)
        src = torch.cat([torch.nonzero(keep).squeeze(1), lab_idx]) if cfg["keep_rule"] else None
        n_train = src.shape[0] if src is not None else U + n_lab         # (the nonzero above is the keep rule's one host sync)
        steps = n_train // BATCH
        if use_dist and world > 1:
            steps = int(steps_cap.item())
        student.params.copy_(init_params)
        student._packed_ok = False
        student.init_train_state()
        perm = torch.randperm(n_train, device=dev, generator=gen_perm)
        src = src[perm] if src is not None else perm
        # the epoch's shuffle, the keep filter and the parsers' mask normalisation (255 -> 1, HeLa position x 3) as ONE gather per
        # tensor inside libimk (imk_gather_pairs); batches are contiguous views of the result
        tx, ty = F.gather_pairs(src, img=pool_x, mask_planar=pool_y, div255=binary, mul=pos_w if (binary and K == 3) else None)
        if not binary:
            ty = ty[..., 0]                  # [n, H, W, 1] class ids: a view
        for s in range(steps):
            student.fwd_bwd(tx[s * BATCH:(s + 1) * BATCH], ty[s * BATCH:(s + 1) * BATCH], LOSS)
            scale = F._grad_allreduce(student)
            student.adamw_step(LR, WD, grad_scale=scale)
            if args.step_events:
                step_events.append(ev()); step_events[-1].record()
        last_batch[:] = [tx[:BATCH], ty[:BATCH]]
        e2.record()
        if args.step_events and step_events:
            torch.cuda.synchronize()
            dt = sorted(a.elapsed_time(b) for a, b in zip(step_events[:-1], step_events[1:]))
            print(f"[step events] n={len(dt)} min {dt[0]:.3f} p10 {dt[len(dt)//10]:.3f} median {dt[len(dt)//2]:.3f} p90 {dt[9*len(dt)//10]:.3f} "
                  f"max {dt[-1]:.3f} ms; e1 -> first step end {e1.elapsed_time(step_events[0]):.3f} ms", file=sys.stderr)
            step_events.clear()
        if record is not None:
            record.append((e0, e1, e2))
        info.update(epoch_steps=int(steps), n_train=n_train)
        if use_dist and world > 1 and record is not None and len(record) == 1:      # once per run: what THIS rank did (stderr)
            print(f"[bench rank {rank}] epoch_steps={int(steps)} kept={int(keep.sum())} shard={U}", file=sys.stderr, flush=True)
        return torch.stack([ps.sum(), ims.sum(), keep.sum()]).double()      # read after the timed region

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        generation()
    barrier()
    setup_prof = prof_collect()
    rec = []
    # For rocprofv3 cross-checks (profiles/summarize.py): a marker dispatch either side of the timed region -- outside the clock --
    # and, inside it, the library's per-kernel-name sums of launches and algorithmic bytes
    if primary and not args.no_prof:
        prof.totals(True)
        Profiler.mark(1)
        torch.cuda.synchronize()
        stage_marks[0] = True      # two 64-thread marker dispatches per generation (~2 us each in a ~94 ms generation)
    t0 = time.perf_counter()
    # Sampling: an event record is a barrier packet on the stream (~5 us before the next kernel starts), so bracketing
    # every launch costs ~10 % of a training step; every 61st hooked launch (prime, coprime to the ~135 launches of a
    # training step and the 18 of a forward) costs < 1 % and still yields a few hundred samples.
    for k in range(args.steps):
        totals = generation(rec)
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_totals = None
    stage_totals = None
    if primary and not args.no_prof:
        stage_marks[0] = False
        Profiler.mark(2)
        kernel_totals = prof.totals_dump()
        prof.totals(False)
    pc, pms, pby, pfl = prof_collect()
    prof.set_period(0)
    if primary and not args.no_prof:
        # which kernel variants make up which stage (launches and algorithmic bytes per generation): one more generation, outside
        # the clock, with the library's totals read at the stage boundary (host-side sums: no synchronisation involved)
        prof.totals(True)
        stage_snap[0] = {}
        generation()
        whole = prof.totals_dump()
        prof.totals(False)
        inf = stage_snap[0].get("inference", {})
        stage_snap[0] = None
        def _fmt(d):
            return {k: {"launches": t["launches"], "MB_per_launch": round(t["bytes"] / max(t["launches"], 1) / 1e6, 3),
                        "GFLOP_per_launch": round(t["flops"] / max(t["launches"], 1) / 1e9, 4)} for k, t in sorted(d.items()) if t["launches"]}
        trn = {k: {f: t[f] - inf.get(k, {}).get(f, 0) for f in ("launches", "bytes", "flops")} for k, t in whole.items()}
        stage_totals = {"inference": _fmt(inf), "training": _fmt(trn)}
    # host cost of enqueueing a step (fwd_bwd + all-reduce + adamw), outside the timed region: from an idle device, 8 steps at
    # a time (~1000 launches: inside an epoch the host runs ahead until the launch queue back-pressures it, which would
    # time the GPU, not the host)
    for _ in range(5):
        torch.cuda.synchronize()
        t_host = time.perf_counter()
        for s in range(8):
            student.fwd_bwd(last_batch[0], last_batch[1], LOSS)
            scale = F._grad_allreduce(student)
            student.adamw_step(0.0, 0.0, grad_scale=scale)
        host_enqueue.append((time.perf_counter() - t_host) / 8)
    torch.cuda.synchronize()
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    t_inf = sum(a.elapsed_time(b) for a, b, _ in rec) / len(rec)
    t_ep = sum(b.elapsed_time(c) for _, b, c in rec) / len(rec)
    totals = torch.cat([totals, torch.tensor([float(info.pop("n_train"))], dtype=torch.float64, device=dev)])
    if use_dist:
        dist.all_reduce(totals)
    n_all = U_total * (1 if strong else world)
    info.update(kept=int(totals[2]), train_images=int(totals[3]), mean_pred_size=round(float(totals[0]) / n_all, 1),
                mean_im_size=round(float(totals[1]) / n_all, 1))

    # ---- sharding changes nothing: the whole set on one rank gives the ranks' summed sizes and kept count --------------
    sharding_check = None
    if strong and world > 1:
        ok = torch.ones(1, device=dev)
        if rank == 0:
            ps, ims, keep = im_stage(x_all, into_pool=False)
            whole = [float(ps.sum()), float(ims.sum()), float(keep.sum())]
            sharding_check = {"sum_pred_size": whole[0], "sum_im_size": whole[1], "kept": whole[2],
                              "equals_sum_over_ranks": whole == [float(v) for v in totals[:3]]}
            ok.fill_(1.0 if sharding_check["equals_sum_over_ranks"] else 0.0)
        dist.broadcast(ok, 0)            # every rank learns the verdict and leaves together (no rank hangs in a later collective)
        if float(ok.item()) == 0.0:
            if rank == 0:
                print(f"sharded IM stage differs from the single-rank one: {sharding_check} vs {totals[:3].tolist()}", file=sys.stderr)
            dist.destroy_process_group()
            raise SystemExit(3)

    # ---- roofline of the dominant kernel family: the one with the largest summed time over ALL hooked kernels ---------
    def fam_entry(i, c, ms, by, fl):
        e = {"launches": int(c[i]), "ms": round(ms[i], 3), "avg_us": round(1000 * ms[i] / c[i], 2),
             "GBps": round(by[i] / ms[i] / 1e6, 1) if ms[i] else None}
        if fl[i] and ms[i]:
            e["TFLOPs"] = round(fl[i] / ms[i] / 1e9, 1)
        return e
    v = max(range(NV), key=lambda i: pms[i])
    fam_all = {FAMILIES[i]: {**fam_entry(i, pc, pms, pby, pfl), "share_of_sampled_time": round(pms[i] / max(sum(pms), 1e-9), 3)}
               for i in range(NV) if pc[i]}
    # which side of max(bytes / 8 TB/s, flops / 2.5 PFLOP/s) binds the dominant family's launches (their live sums): SURVEY 8d -- a GEMM-class
    # family is not "mfma-bound" by name; its full-resolution launches are byte-bound
    mfma_bound = pfl[v] / (MFMA_PEAK_TFLOPS * 1e12) > pby[v] / (HBM_PEAK_GBS * 1e9)
    if mfma_bound:
        achieved, peak, unit = (pfl[v] / pms[v] / 1e9 if pms[v] else 0.0), MFMA_PEAK_TFLOPS, "TFLOP/s"
    else:
        achieved, peak, unit = (pby[v] / pms[v] / 1e6 if pms[v] else 0.0), HBM_PEAK_GBS, "GB/s"      # bytes / ms / 1e6 = GB/s
    # ---- REPLAYED values: read from the committed profiles of THIS configuration (profiles/<ROUND>_*<tag>.csv), never another
    # configuration's, and only when they were collected with this bench.py and these kernel sources (profiles/<ROUND>_provenance<tag>.json)
    import csv
    tag = config_tag(config_name, alpha)
    replay_ok, replay_refused, _prov = provenance(tag) if primary else (False, "not the primary configuration of this command", None)
    prof_file = lambda stem: os.path.join(ROOT, "profiles", f"{ROUND}_{stem}{tag}.csv")
    fam = FAMILIES[v].split("<")[0].split("+")[0].replace("wgf_stage1", "wgf_stage")
    fams = (fam, "conv_wide_kernel") if fam == "conv_pipe_kernel" else (fam,)      # one family in the hook
    # HBM traffic per launch from the PMC passes of profiles/collect_round.sh (FETCH_SIZE x 2 + WRITE_SIZE, separate passes)
    traffic, traffic_src = None, None
    try:
        if replay_ok and os.path.exists(prof_file("pmc_traffic")):
            rows = [r for r in csv.DictReader(open(prof_file("pmc_traffic"))) if r["kernel"].startswith(fams)]
            nl = sum(int(r["launches"]) for r in rows)
            if nl:
                traffic = round(1e6 * sum(int(r["launches"]) * float(r["hbm_MB_per_launch_corrected(2*fetch+write)"]) for r in rows) / nl)
                traffic_src = f"profiles/{ROUND}_pmc_traffic{tag}.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, launch-weighted mean)"
    except Exception:
        pass
    # `traffic` averages the launches of the PMC pass (the whole process: ensemble pre-training + timed region + exclusive pass), the
    # sampled algorithmic figure below the timed region only -- their quotient means nothing.  Over ONE population
    # (profiles/traffic_ratio.py: per variant, launches of the PMC pass x the library's algorithmic bytes of that variant):
    traffic_ratio, traffic_alg = None, None
    try:
        if traffic_src and os.path.exists(prof_file("traffic_vs_algorithmic")):
            fam_row = [r for r in csv.reader(open(prof_file("traffic_vs_algorithmic"))) if r and r[0].startswith("FAMILY")]
            if fam_row:
                traffic_alg, traffic_ratio = round(1e6 * float(fam_row[0][2])), float(fam_row[0][4])
    except Exception:
        pass
    # The same fraction from rocprofv3's own durations: the family's launches INSIDE the timed region of the committed kernel trace
    # of this command (profiles/summarize.py cuts the trace at the marker dispatches) against this run's algorithmic bytes / flops
    frac_rocprof, rocprof_src, rocprof_us = None, None, None
    timed_rows = []
    try:
        if replay_ok and os.path.exists(prof_file("timed_region_kernel_stats")):
            timed_rows = list(csv.DictReader(open(prof_file("timed_region_kernel_stats"))))
            rocprof_src = (f"profiles/{ROUND}_timed_region_kernel_stats{tag}.csv (rocprofv3 --kernel-trace of this command, dispatches between "
                           "the timed region's markers: sum over the family's variants of calls x algorithmic bytes / sum of their durations)")
        if timed_rows and pc[v]:
            col = "GFLOP_per_launch" if mfma_bound else "algorithmic_MB_per_launch"
            rows = [r for r in timed_rows if r["kernel"].startswith(fams) and r.get(col)]
            work = sum(int(r["calls"]) * float(r[col]) for r in rows)                        # MB or GFLOP, exact (the library's own sums)
            ms = sum(float(r["total_ms"]) for r in rows)
            if ms:
                rocprof_us = 1e3 * ms / sum(int(r["calls"]) for r in rows)
                frac_rocprof = round(work / ms / peak, 4)     # MB / ms = GB/s; GFLOP / ms = TFLOP/s
    except Exception:
        pass
    # ... and against the UNION of the family's launch intervals instead of the sum of their durations (the two ensemble members' forwards
    # overlap on two streams, the weight gradients run beside the dgrads): the bandwidth the family delivered while it was running
    frac_union, union_by_stage = None, {}
    try:
        if replay_ok and os.path.exists(prof_file("timed_region_family_union")):
            for r in csv.DictReader(open(prof_file("timed_region_family_union"))):
                if r["family"].split("+")[0] == fams[0] and r.get("GBps_by_union"):
                    f_u = float(r["frac_of_2.5PFLOPs_by_union"]) if mfma_bound else float(r["frac_of_8TBps_by_union"])
                    if r["stage"] == "all":
                        frac_union = f_u
                    else:
                        union_by_stage[r["stage"]] = {"frac_by_union": f_u, "overlap_factor": float(r["overlap_factor"])}
    except Exception:
        pass
    whole_n = setup_prof[0][v] + pc[v]
    whole_ms = setup_prof[1][v] + pms[v]
    steps_per_gen = max(info.get("epoch_steps", 1), 1)
    step_ms = t_ep / steps_per_gen
    # whole-stage view (SURVEY 8d minimum bytes and conv FLOPs / stage time): what the dependent launch chains cost beyond the kernels
    inf_bytes = U * (N_MODELS * fwd_min_bytes + im_bytes_per_image(cfg, N_MODELS))
    step_bytes = 3 * BATCH * fwd_min_bytes
    step_view = {"ensemble_infer_plus_im": {"min_bytes": inf_bytes, "ms": round(t_inf, 3),
                                            "GBps": round(inf_bytes / t_inf / 1e6, 1), "frac": round(inf_bytes / t_inf / 1e6 / HBM_PEAK_GBS, 4),
                                            "TFLOPs": round(U * N_MODELS * fwd_flops / t_inf / 1e9, 1),
                                            "frac_mfma": round(U * N_MODELS * fwd_flops / t_inf / 1e9 / MFMA_PEAK_TFLOPS, 4)},
                 "train_step": {"min_bytes": step_bytes, "ms": round(step_ms, 4),
                                "GBps": round(step_bytes / step_ms / 1e6, 1),
                                "frac": round(step_bytes / step_ms / 1e6 / HBM_PEAK_GBS, 4),
                                "TFLOPs": round(3 * BATCH * fwd_flops / step_ms / 1e9, 1),
                                "frac_mfma": round(3 * BATCH * fwd_flops / step_ms / 1e9 / MFMA_PEAK_TFLOPS, 4),
                                "host_enqueue_ms_per_step": round(1000 * _median(host_enqueue), 4) if host_enqueue else None},
                 "conv_flops_per_image_forward": fwd_flops, "min_bytes_per_image_forward": fwd_min_bytes,
                 "note": "SURVEY 8d minimum HBM bytes (every tensor that crosses a block boundary written and read once; 3x per "
                         "training image) and conv FLOPs (3x forward per training image) over the measured stage time, per rank; "
                         "host_enqueue = wall time the host needs to enqueue one step (fwd_bwd + all-reduce + adamw; 8 steps from an idle device, outside the timed region, median of 5): below the GPU step time the host is not the limiter"}
    roofline = {"bound": "mfma" if mfma_bound else "hbm", "kernel": FAMILIES[v], "achieved": round(achieved, 1), "peak": peak,
                "unit": unit, "frac": round(achieved / peak, 4), "traffic": traffic,
                "traffic_source": traffic_src,
                "traffic_over_algorithmic": traffic_ratio, "traffic_population_algorithmic_bytes_per_launch": traffic_alg,
                "frac_rocprof": frac_rocprof, "rocprof_avg_us_per_launch": round(rocprof_us, 2) if rocprof_us else None,
                "frac_rocprof_union": frac_union, "frac_rocprof_union_by_stage": union_by_stage or None,
                "frac_rocprof_source": rocprof_src,
                # what THIS run measured and what it read back from committed files (printed only when profiles/<ROUND>_provenance
                # names this bench.py and these kernel sources; otherwise null + the reason)
                "live": ["achieved", "frac", "launches", "avg_us_per_launch", "stage_ms", "by_stage.*.stage_frac", "by_stage.training.chain_ms",
                         "by_stage.training.per_image_us", "by_stage.training.step_ms", "exclusive_frac", "im_kernel"],
                "replayed": ["traffic", "traffic_over_algorithmic", "frac_rocprof", "frac_rocprof_union", "rocprof_avg_us_per_launch", "by_stage.*.kernel",
                             "by_stage.*.frac", "by_stage.inference.bound"],
                "replayed_from": f"profiles/{ROUND}_{{pmc_traffic,traffic_vs_algorithmic,timed_region_kernel_stats,timed_region_family_union,sq_counters}}{tag}.csv" if replay_ok else None,
                "replayed_refused": replay_refused,
                "avg_us_per_launch_whole_process": round(1000 * whole_ms / max(whole_n, 1), 2),
                "launches": int(pc[v]), "avg_us_per_launch": round(1000 * pms[v] / max(pc[v], 1), 2),
                "avg_algorithmic_bytes_per_launch": round(pby[v] / max(pc[v], 1)),
                "avg_flops_per_launch": round(pfl[v] / max(pc[v], 1)),
                "share_of_sampled_kernel_time": round(pms[v] / max(sum(pms), 1e-9), 3),
                "sampling": f"every {args.prof_period}th hooked kernel launch of the timed region (all families)",
                "all_families": fam_all, "step": step_view,
                "note": "timed region: the weight-gradient kernels (side stream) and the ensemble's other models run beside "
                        "the main-stream kernels, event-bracketed durations include that sharing; 'exclusive' = the same "
                        "workload with every kernel alone on one stream (the plans' single_stream switch).  bound = the side of "
                        "max(algorithmic bytes / 8 TB/s, flops / 2.5 PFLOP/s) that binds the family's launches of this run"}
    # ---- per stage: the generation has two regimes, and one family name must not hide the weaker one ---------------------
    # (a) T(B) = chain + B x per-image, fitted live on training steps of 8 / 16 / 32 images (lr = 0: the weights stay): the fixed part
    #     is the dependent launch chain, the slope the kernels' per-image work
    chain_ms = per_image_us = None
    tb = {}
    try:
        for bsz in (8, 16, BATCH):
            xb, yb = last_batch[0][:bsz], last_batch[1][:bsz]
            for _ in range(3):
                student.fwd_bwd(xb, yb, LOSS); student.adamw_step(0.0, 0.0)
            a, b = ev(), ev()
            a.record()
            for _ in range(20):
                student.fwd_bwd(xb, yb, LOSS); student.adamw_step(0.0, 0.0)
            b.record()
            torch.cuda.synchronize()
            tb[bsz] = a.elapsed_time(b) / 20
        xs, ys = np.array(sorted(tb), dtype=np.float64), np.array([tb[k] for k in sorted(tb)])
        slope, icpt = np.polyfit(xs, ys, 1)
        chain_ms, per_image_us = round(float(icpt), 4), round(1000 * float(slope), 2)
    except Exception:
        pass
    # (b) the dominant kernel variant of each stage from rocprofv3's durations inside the timed region (the committed trace of this
    #     command, cut at the stage markers by profiles/summarize.py) and the library's exact algorithmic bytes
    def stage_top(stage):
        rows = [r for r in timed_rows if r.get("stage", "") == stage]
        if not rows and stage_totals:        # a trace without stage markers (round 4's): a variant belongs to the stage that launches it
            other = "training" if stage == "inference" else "inference"
            mine = {k for k, t in stage_totals[stage].items()
                    if t["launches"] * max(t["MB_per_launch"], 1e-9) >= stage_totals[other].get(k, {"launches": 0, "MB_per_launch": 0})["launches"]
                    * max(stage_totals[other].get(k, {"MB_per_launch": 0})["MB_per_launch"], 1e-9)}
            rows = [r for r in timed_rows if r["kernel"] in mine]
        rows = [r for r in rows if r.get("algorithmic_MB_per_launch")]
        if not rows:
            return None
        r = max(rows, key=lambda q: float(q["total_ms"]))
        # which side of max(bytes / 8 TB/s, flops / 2.5 PFLOP/s) binds THIS variant's launches (not a label by family name)
        mf = float(r.get("GFLOP_per_launch") or 0) / (MFMA_PEAK_TFLOPS * 1e3) > float(r["algorithmic_MB_per_launch"]) / (HBM_PEAK_GBS * 1e3)
        return {"kernel": r["kernel"], "calls": int(r["calls"]), "avg_us": float(r["avg_us"]),
                "share_of_stage_kernel_time": round(float(r["total_ms"]) / max(sum(float(q["total_ms"]) for q in rows), 1e-9), 3),
                "frac": float(r["frac_of_2.5PFLOPs"]) if mf else float(r["frac_of_8TBps"]),
                "achieved": float(r["TFLOPs"]) if mf else float(r["GBps"]), "unit": "TFLOP/s" if mf else "GB/s",
                "family_frac": round(sum(int(q["calls"]) * float(q["algorithmic_MB_per_launch"]) for q in rows)
                                     / max(sum(float(q["total_ms"]) for q in rows), 1e-9) / HBM_PEAK_GBS, 4)}
    def issue_limiter(kernel):
        """what the waves of `kernel` do with their cycles, from the committed SQ counter pass of this configuration (replayed)"""
        try:
            path = prof_file("sq_counters") if tag else os.path.join(ROOT, "profiles", f"{ROUND}_sq_counters_isic.csv")
            for r in csv.DictReader(open(path), delimiter=";"):
                if r["kernel"].startswith(kernel[:58]):
                    wc = float(r["SQ_WAVE_CYCLES"])
                    iss, wait, mf = float(r["SQ_ACTIVE_INST_ANY"]) / wc, float(r["SQ_WAIT_ANY"]) / wc, float(r["SQ_VALU_MFMA_BUSY_CYCLES"]) / wc
                    return {"issue": round(iss, 2), "wait": round(wait, 2), "mfma_busy": round(mf, 2),
                            "source": os.path.relpath(path, ROOT)}
        except Exception:
            pass
        return None
    by_stage = {"inference": {}, "training": {}}
    ti, tt = (stage_top("inference"), stage_top("training")) if (primary and timed_rows) else (None, None)
    ti, tt = ti or {}, tt or {}
    if ti:
        lim = issue_limiter(ti["kernel"])
        # bytes floor vs flop floor decide hbm / mfma; a kernel far from both whose waves mostly wait or issue is named so, with the counters
        ti.update(bound=("mfma" if ti["unit"] == "TFLOP/s" else "hbm") if not (lim and ti["frac"] < 0.5) else
                  ("valu-issue/latency (waves issue %d %%, wait %d %% of their cycles, MFMA busy %d %%)" % (100 * lim["issue"], 100 * lim["wait"], 100 * lim["mfma_busy"])),
                  bound_evidence=lim)
    ti.update(stage_ms=round(t_inf, 3), stage_frac_of_min_bytes_floor=step_view["ensemble_infer_plus_im"]["frac"])
    tt.update(bound="launch-chain" if chain_ms and chain_ms > 0.4 * step_ms else ("mfma" if tt.get("unit") == "TFLOP/s" else "hbm"),
              chain_ms=chain_ms, per_image_us=per_image_us, step_ms=round(step_ms, 4),
              step_ms_by_batch={str(k): round(v, 4) for k, v in sorted(tb.items())},
              stage_ms=round(t_ep, 3), stage_frac_of_min_bytes_floor=step_view["train_step"]["frac"])
    by_stage = {"inference": ti, "training": tt,
                "note": "kernel = the variant with the largest summed duration among the stage's hooked kernels in the committed "
                        "rocprofv3 trace of this command and configuration (absent: no such trace committed, or refused -- replayed_refused); "
                        "frac = its algorithmic bytes (flops) over that duration over the peak; family_frac = the same over every hooked "
                        "variant of the stage; chain_ms / per_image_us = intercept / slope of the training step time over batches of "
                        "8, 16, 32 measured in THIS run; stage_frac_of_min_bytes_floor = SURVEY 8d minimum bytes / stage time / 8 TB/s, live.  "
                        "Durations in the trace are those of the product's overlap: an inference launch shares the chip with the other "
                        "ensemble member's launch on the second stream, a training-chain kernel with the weight gradients on the side stream "
                        "(roofline.exclusive = the same workload with every kernel alone on one stream)"}
    roofline["by_stage"] = by_stage
    # the same generation once more with every kernel alone on the stream: the kernels' own rates
    for m in models + [student]:
        m.debug(single_stream=True)
    prof.set_period(7 if not args.no_prof else 0)     # outside the timed region: dense sampling
    prof_collect()
    generation()
    barrier()
    xc, xms, xby, xfl = prof_collect()
    prof.set_period(0)
    for m in models + [student]:
        m.debug(single_stream=False)
    if xms[v]:
        x_ach = (xfl[v] / xms[v] / 1e9) if mfma_bound else (xby[v] / xms[v] / 1e6)
        roofline["exclusive"] = {"achieved": round(x_ach, 1), "frac": round(x_ach / peak, 4),
                                 "launches": int(xc[v]), "avg_us_per_launch": round(1000 * xms[v] / max(xc[v], 1), 2),
                                 "all_families_GBps": {FAMILIES[i]: round(xby[i] / xms[i] / 1e6, 1) for i in range(NV) if xms[i]},
                                 "all_families_TFLOPs": {FAMILIES[i]: round(xfl[i] / xms[i] / 1e9, 1) for i in range(NV) if xms[i] and xfl[i]}}

    # ---- the IM kernel alone on the same shapes, at the reference's boundary (HBM-bound; SURVEY 8d bytes per image) -----
    nb = min(infer_batch, U)
    probs = torch.stack([m.predict_device(x_unl[:nb]) for m in models], 0)
    run_im = ((lambda: imk_im.im_binary(probs, cfg["thr"], cfg["cmp_ge"], x_unl[:nb], True, True)) if binary
              else (lambda: imk_im.im_multiclass(probs, x_unl[:nb], True, True)))
    for _ in range(3):
        run_im()
    a, b = ev(), ev()
    n_rep = 20
    a.record()
    for _ in range(n_rep):
        run_im()
    b.record()
    torch.cuda.synchronize()
    im_ms = a.elapsed_time(b) / n_rep
    im_bytes = probs.shape[1] * im_bytes_per_image(cfg, N_MODELS)
    im_kernel = {"kernel": "im_binary_vec" if binary else "im_multi_kernel", "GBps": round(im_bytes / im_ms / 1e6, 1),
                 "frac_of_hbm_peak": round(im_bytes / im_ms / 1e6 / HBM_PEAK_GBS, 4),
                 "bytes_per_launch": im_bytes, "ms_per_launch": round(im_ms, 4)}

    out = None
    if rank == 0:
        n_images = U_total if strong else U_total * world
        out = {
            "metric": "images/sec per IM generation (ensemble infer + IM build + 1 train epoch), 256x256" if (H, W) == (256, 256)
                      else f"images/sec per IM generation (ensemble infer + IM build + 1 train epoch), {H}x{W}",
            "value": round(n_images * args.steps / elapsed, 2), "unit": "images/s", "n_gpus": n_gpus,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000 * elapsed / args.steps, 2),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f16",
            "data": "synthetic",
            "config": {"workload": cfg["workload"], "name": config_name, "alpha": ALPHA, "shape": [H, W, C], "outputs": K,
                       "unlabeled_images": n_images, "unlabeled_images_per_gpu": U,
                       "labeled_images": L_total if strong else L_total * world,
                       "n_models": N_MODELS, "infer_batch": infer_batch,
                       "infer_batch_rule": "functions.infer_batch_size: 584 at alpha<=0.5 else 128; short tail spread" if not args.infer_batch else "--infer-batch",
                       "train_batch_per_gpu": BATCH,
                       "global_batch": BATCH * world, "parallelism": f"dp{world}",
                       "bn_momentum": {"rule": bn_rule, "value": round(bn_mom, 6)},
                       "process_group": (os.environ.get("IMK_BENCH_BACKEND", "nccl") + (" (forced, 1 rank)" if world == 1 else "")) if use_dist else None,
                       **{k: v for k, v in info.items() if k != "n_train"}},
            "stage_ms": {"ensemble_infer_plus_im": round(t_inf, 2), "train_epoch": round(t_ep, 2)},
            "roofline": roofline,
            "im_kernel": im_kernel,
        }
        if kernel_totals:
            out["timed_region_kernel_totals"] = {k: {"launches": t["launches"], "MB_per_launch": round(t["bytes"] / max(t["launches"], 1) / 1e6, 3),
                                                     "GFLOP_per_launch": round(t["flops"] / max(t["launches"], 1) / 1e9, 4)}
                                                 for k, t in sorted(kernel_totals.items())}
        if stage_totals:
            out["stage_kernel_totals_per_generation"] = stage_totals
        if sharding_check:
            out["sharding_check"] = sharding_check
        if not args.no_cpu_baseline and world == 1:     # the CPU baseline is a 1-GPU-run item (rank 0 only)
            out["cpu_baseline"] = cpu_baseline(cfg, U_total, L_total, fwd_flops)
            out["cpu_baseline"]["parity_sample"] = parity_sample(cfg, models, x_unl)
            out["png_io"] = png_io_rate(x_unl[:512].cpu().numpy())
    prof.close()
    return out if rank == 0 else None


if __name__ == "__main__":
    main()
