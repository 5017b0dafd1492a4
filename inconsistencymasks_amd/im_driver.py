"""Generation loop of the per-dataset `*_IM.py` drivers of the reference (ISIC_2018/09_ISIC_2018_IM.py:47-153 and its
siblings HeLa/09_HeLa_IM.py, SUIM/10_SUIM_IM.py, Cityscapes/09_Cityscapes_IM.py, which are copies of one template).
Same loops, model / directory / CSV names and top-K hand-off; launch under torch.distributed.run for multi-GPU.
Environment overrides for short runs: IM_RUNIDS, IM_NS, IM_GENS, IM_CANDIDATES (comma-separated).

approach="IM_plus" is the IM+ variant (ISIC_2018/11_ISIC_2018_IM+.py:40-135, SUIM/12_SUIM_IM+.py, HeLa/11_HeLa_IM+.py,
Cityscapes/11_Cityscapes_IM+.py): the IM output goes to a `temp` directory, NUM_IMAGES_IM_PLUS augmented copies of each
pseudo-labelled pair (no originals) form the training set, augmentation strength and the U-Net width alpha grow per
generation (Noisy Student).

approach="aug_IM_plus" is AIM+ (ISIC_2018/13_ISIC_2018_aug_IM+.py:44-116, HeLa/13_HeLa_aug_IM+.py, Cityscapes/13_Cityscapes_aug_IM+.py,
SUIM/14_SUIM_aug_IM+.py): IM+ that starts from the augmented-subset baseline (`*_subset_aug_{runid}_topK_j.h5`,
subset_driver.run(..., aug=True)), keeps the un-augmented IM pairs beside their augmented copies and adds the AUGMENTED
labelled set (TRAIN_LABELED_AUG) instead of the plain one; the multi-class scripts honour FILTER_INCONSISTENT_CLASS_PRED."""
import csv
import os
import shutil

import torch

from . import functions as F
from . import paths
from .unet import get_unet

DATASETS = {
    # prefix in paths/config, kind, CSV header, index of the ranking metric in the row (after the model name)
    "ISIC_2018": dict(section="ISIC_2018", kind="isic", rank=1,
                      header=["modelname", "mIoU_val", "mIoU_test", "mIoU_train_unlabeled", "dice_score_val",
                              "dice_score_test", "dice_score_train_unlabeled"]),
    "SUIM": dict(section="SUIM", kind="multi", rank=4,
                 header=["modelname", "mPA_val", "mPA_test", "mPA_train_unlabeled", "mIoU_val", "mIoU_test",
                         "mIoU_train_unlabeled"]),
    "Cityscapes": dict(section="CITYSCAPES", kind="multi", rank=4,
                       header=["modelname", "mPA_val", "mPA_test", "mPA_train_unlabeled", "mIoU_val", "mIoU_test",
                               "mIoU_train_unlabeled"]),
    # the reference ranks HeLa candidates by tuple index 4, which is mIoU_test (HeLa/09_HeLa_IM.py:130): kept
    "HeLa": dict(section="HELA", kind="hela", rank=4,
                 header=["modelname", "mIoU_val", "mIoU_ad_val", "mean_cell_count_error_val", "mIoU_test", "mIoU_ad_test",
                         "mean_cell_count_error_test", "mIoU_unlabeled", "mIoU_ad_unlabeled",
                         "mean_cell_count_error_unlabeled"]),
}


_STRONG = dict(max_blurs=[0, 1, 1, 2, 3], max_noises=[5, 10, 15, 20, 25],
               bra=[(0.9, 1.1), (0.8, 1.2), (0.7, 1.3), (0.6, 1.4), (0.5, 1.5)],
               brb=[(-5, 5), (-10, 10), (-15, 15), (-20, 20), (-25, 25)])
IM_PLUS = {   # per-generation schedules of the IM+ scripts (lines 46-50 / 47-51 / 48-52 of each)
    "ISIC_2018": dict(alphas=[0.5, 0.75, 1, 1.25, 1.5], **_STRONG),
    "SUIM": dict(alphas=[1, 1.25, 1.5, 1.75, 2], **_STRONG),
    "HeLa": dict(alphas=[1, 1.25, 1.5, 1.75, 2], max_blurs=[0, 1, 1, 2, 3], max_noises=[5, 10, 15, 20, 25],
                 bra=[(0.9, 1.1), (0.9, 1.1), (0.8, 1.2), (0.8, 1.2), (0.7, 1.3)],
                 brb=[(-3, 3), (-6, 6), (-9, 9), (-12, 12), (-15, 15)]),
    "Cityscapes": dict(alphas=[1, 1.25, 1.5, 1.75, 2], max_blurs=[0, 0, 0, 0, 1], max_noises=[3, 6, 9, 12, 15],
                       bra=[(0.95, 1.05), (0.9, 1.1), (0.8, 1.2), (0.7, 1.3), (0.6, 1.4)],
                       brb=[(-3, 3), (-6, 6), (-9, 9), (-12, 12), (-15, 15)]),
}


def default_color_mapping(n_classes):
    """colour -> class id, deterministic palette for datasets the reference has no table for"""
    return {((37 * k) % 256, (91 * k) % 256, (173 * k) % 256): k for k in range(n_classes)}


def color_mapping(dataset, n_classes):
    """RGB colour -> class id tables of the `*_color.png` dumps (SUIM/SUIM_class_mapping.py:4-14 COLOR_TO_CLASS_MAPPING_SUIM,
    Cityscapes/Cityscapes_class_mapping.py:43-80 COLOR_TO_CLASS_MAPPING_CITYSCAPES), generated from the rules the two
    public palettes follow: SUIM = the dataset's 3-bit RGB codes shifted by one behind the light-grey IM class 0;
    Cityscapes = the PASCAL-VOC bit-interleaved colour map with blue carrying bit 0, plus the licence-plate colour."""
    if dataset == "SUIM" and n_classes == 9:
        m = {(211, 211, 211): 0}
        for k in range(8):
            m[(255 * (k >> 2 & 1), 255 * (k >> 1 & 1), 255 * (k & 1))] = k + 1
        return m
    if dataset == "Cityscapes" and n_classes == 35:
        m = {}
        for k in range(35):
            r = g = b = 0
            c = k
            for j in range(8):
                b |= (c & 1) << (7 - j)
                g |= (c >> 1 & 1) << (7 - j)
                r |= (c >> 2 & 1) << (7 - j)
                c >>= 3
            m[(r, g, b)] = k
        m[(192, 192, 192)] = -1
        return m
    return default_color_mapping(n_classes)


class _Timer:
    """wall-clock stage timer of the drivers, printed when IM_TIMING is set"""

    def __init__(self, prefix):
        import time
        self.prefix, self.clock, self.t = prefix, time.perf_counter, time.perf_counter()
        self.on = bool(os.environ.get("IM_TIMING"))

    def __call__(self, what):
        if self.on:
            torch.cuda.synchronize()
            now = self.clock()
            print(f"[timing] {self.prefix}{what}: {now - self.t:.2f} s", flush=True)
            self.t = now


def _ints(name, default):
    v = os.environ.get(name)
    return [int(x) for x in v.split(",")] if v else default


_CAND_STREAMS = {}


def _candidate_streams(n):
    """n streams per device for IM_PARALLEL_CANDIDATES, created once: a fresh Stream per candidate would leave its freed blocks in a
    dead stream's pool of the caching allocator until an out-of-memory flush."""
    dev = torch.cuda.current_device()
    pool = _CAND_STREAMS.setdefault(dev, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream())
    return pool[:n]


PARALLEL_CANDIDATES_DEFAULT = 3


def dp_mode():
    """How several ranks train a generation's candidates (IM_DP_MODE):
      'gradient'   (default, the north star's path): every candidate data-parallel over all ranks -- per-GPU batch 32, one flat gradient
                   all-reduce per step, epoch steps = images // (32 N).  Fastest per candidate; the optimisation recipe is no longer the
                   reference's (global batch 32 N, N times fewer steps: profiles/r03_dp_convergence.txt).
      'candidates' (SURVEY 8e row 3): candidate i trains WHOLE on rank i mod N with the reference's batch 32 and step count, no
                   collective; the rows are gathered for ranking / rename / CSV.  Exactly the one-rank results (byte for byte), at most
                   min(N, candidates) ranks busy during training.  Use it when the reference's recipe matters more than wall time."""
    m = os.environ.get("IM_DP_MODE", "gradient").lower()
    if m not in ("gradient", "candidates"):
        raise ValueError(f"IM_DP_MODE={m!r}: expected 'gradient' or 'candidates'")
    return m


def epoch_steps(n_files, batch, world):
    """steps_per_epoch of a candidate: files // batch on one rank and with whole candidates per rank (the reference's,
    ISIC_2018/09_ISIC_2018_IM.py:87-88); files // (batch N) data-parallel"""
    return max(n_files // batch // (1 if (world > 1 and dp_mode() == "candidates") else world), 1)


def train_candidates(cands, train_candidate, world, tick=None, parallel=None):
    """rows of `train_candidate(i, side_by_side)` for i in cands, in that order.

    The reference trains a generation's candidates one after the other (ISIC_2018/09_ISIC_2018_IM.py:90).  They are independent
    (own seed, own files), and one model's training step leaves most of the chip idle in its deep levels, so on ONE rank k of
    them train side by side -- one host thread and one stream each, without side streams of their own; 3 interleaved candidates
    reach 1.58x the model-steps per second of one (tests/gpu_probe/concurrent_candidates.py) and a real ISIC generation of 5
    candidates x 50 epochs takes 17.9 s instead of 27.5 s (profiles/r05_notes.md).  Every candidate computes exactly what it computes
    alone: CSVs, checkpoints and prediction files equal the sequential run's byte for byte (tests/test_gpu_driver.py).
    k = `parallel`, else IM_PARALLEL_CANDIDATES, else PARALLEL_CANDIDATES_DEFAULT; 1 is the reference's order.
    Several ranks, IM_DP_MODE=gradient: always 1 (collectives issued from several threads would not line up across ranks).
    Several ranks, IM_DP_MODE=candidates: this rank trains the candidates at positions rank, rank + N, ... of `cands` inside
    functions.local_rank_scope (k of them side by side as on one rank), then every rank receives all rows (all_gather_object)."""
    tick = tick or (lambda what: None)
    par = parallel if parallel is not None else int(os.environ.get("IM_PARALLEL_CANDIDATES", PARALLEL_CANDIDATES_DEFAULT))
    by_candidate = world > 1 and dp_mode() == "candidates"
    rank = F._rank_world()[0]
    mine = [i for pos, i in enumerate(cands) if pos % world == rank] if by_candidate else list(cands)
    if (world > 1 and not by_candidate) or len(mine) < 2:
        par = 1
    par = min(par, max(len(mine), 1))

    def one(i, side_by_side):
        if by_candidate:
            with F.local_rank_scope():
                return train_candidate(i, side_by_side)
        return train_candidate(i, side_by_side)

    if par <= 1:
        rows = []
        for i in mine:
            rows.append(one(i, False))
            tick(f"candidate {i}: training + 3 benchmarks")
    else:
        import queue
        from concurrent.futures import ThreadPoolExecutor
        dev_index = torch.cuda.current_device()
        free = queue.SimpleQueue()                    # a fixed set of streams, reused over candidates and generations;
        for st in _candidate_streams(par):            # a worker holds one for the whole candidate
            free.put(st)

        def worker(i):
            torch.cuda.set_device(dev_index)          # the current device is per thread
            st = free.get()
            try:
                with torch.cuda.stream(st):
                    row = one(i, True)
                    st.synchronize()
            finally:
                free.put(st)
            return row
        with ThreadPoolExecutor(max_workers=par) as pool:
            rows = list(pool.map(worker, mine))
        F.flush_writes(all_threads=True)              # every candidate's prediction PNGs are on disk
        tick(f"{len(mine)} candidates, {par} side by side: training + 3 benchmarks each")
    if by_candidate:
        F.flush_writes(all_threads=True)
        got = [None] * world
        torch.distributed.all_gather_object(got, list(zip(mine, rows)))      # also the barrier behind the checkpoints on disk
        by_i = {i: row for part in got for i, row in part}
        rows = [by_i[i] for i in cands]
        tick(f"{len(cands)} candidates, one whole candidate per rank ({world} ranks): training + 3 benchmarks each")
    return rows


def run(dataset, approach="IM", parallel_candidates=None):
    ds = DATASETS[dataset]
    S = F.config[ds["section"]]
    H, W, C = int(S["IMAGE_HEIGHT"]), int(S["IMAGE_WIDTH"]), int(S["IMAGE_CHANNELS"])
    K, alpha = int(S["NUM_CLASSES"]), float(S["ALPHA"])
    actifu, actifu_out = S["ACTIFU"], S["ACTIFU_OUTPUT"]
    batch, top_k = int(F.config["DEFAULT"]["BATCH_SIZE"]), int(F.config["DEFAULT"]["TOP_Ks"])
    EK, DK = int(S["ERODE_KERNEL"]), int(S["DILATE_KERNEL"])
    if ds["kind"] == "isic":     # :38-39 -- bool(str) is always True in the reference's ISIC script; kept
        BI, BO = bool(S["BLOCK_INPUT"]), bool(S["BLOCK_OUTPUT"])
    else:
        BI, BO = S["BLOCK_INPUT"].lower() == "true", S["BLOCK_OUTPUT"].lower() == "true"
    filt = S.get("FILTER_INCONSISTENT_CLASS_PRED", "false").lower() == "true"
    aim = approach == "aug_IM_plus"
    plus = IM_PLUS[dataset] if approach in ("IM_plus", "aug_IM_plus") else None
    if plus:    # the IM+ scripts parse the blocking flags properly for every dataset (ISIC_2018/11_...IM+.py:38-39)
        BI, BO = S["BLOCK_INPUT"].lower() == "true", S["BLOCK_OUTPUT"].lower() == "true"
        filt = filt if (aim and ds["kind"] == "multi") else False      # Cityscapes/13_Cityscapes_aug_IM+.py:42, 68-71
        free_rot = S.get("FREE_ROTATION", "false").lower() == "true"
        n_plus = int(S.get("NUM_IMAGES_IM_PLUS", 1))
    P = lambda name: getattr(paths, f"{dataset.upper() if dataset != 'Cityscapes' else 'CITYSCAPES'}_{name}")
    base, model_dir, csv_dir = P("BASE_DIR"), P("MODEL_DIR"), P("CSV_DIR")
    F.init_distributed()
    rank, world = F._rank_world()
    tag = {"HeLa": "HELA", "Cityscapes": "CITYSCAPES"}.get(dataset, dataset)   # name prefix of models / CSVs (HeLa/09_HeLa_IM.py:61)

    for runid in _ints("IM_RUNIDS", [1, 2, 3]):
        for n in _ints("IM_NS", [2, 3, 4]):
            for gen in _ints("IM_GENS", [0, 1, 2, 3, 4]):
                name_of = lambda g: f"{tag}_{approach}_{runid}_n{n}_gen{g}_e{EK}_d{DK}_bi_{BI}_bo_{BO}" + \
                    ("_filtered" if (filt and ds["kind"] == "multi") else "")
                modelname = name_of(gen)
                out = {k: os.path.join(base, f"{k}_predictions", approach, *(["temp"] if plus else []), modelname)
                       for k in ("val", "test", "train_unlabeled")}
                if gen == 0:
                    files = [os.path.join(model_dir, f"{tag}_subset{'_aug' if aim else ''}_{runid}_topK_{j}.h5") for j in range(1, n + 1)]
                else:
                    files = [os.path.join(model_dir, f"{name_of(gen - 1)}_topK_{j}.h5") for j in range(1, n + 1)]
                best_models = [F.load_model(f) for f in files]
                tick = _Timer(f"{modelname}: ")     # IM_TIMING=1 prints the wall time of every stage

                means = []
                for split, key in (("VAL", "val"), ("TEST", "test"), ("TRAIN_UNLABELED", "train_unlabeled")):
                    if ds["kind"] == "isic":
                        means.append(F.create_pseudo_labels_im_ISIC_2018(best_models, H, W, C, P(f"{split}_IMAGES_DIR"), out[key],
                                                                         True, EK, DK, BI, BO, True))
                    elif ds["kind"] == "multi":
                        means.append(F.create_pseudo_labels_im_multiclass(best_models, H, W, C, P(f"{split}_IMAGES_DIR"), out[key],
                                                                          True, EK, DK, BI, BO, filt))
                    else:
                        means.append(F.create_pseudo_labels_im_hela(best_models, H, W, C, os.path.join(P(f"{split}_DIR"), "brightfield"),
                                                                    out[key], EK, DK, BI, BO))
                tick("pseudo-labels (val, test, unlabeled): ensemble inference + IM + PNG I/O")
                unl = out["train_unlabeled"]
                if plus:     # augmented copies only (copy_org False) form the training set
                    src, unl = unl, os.path.join(base, "train_unlabeled_predictions", approach, modelname)
                    kw = dict(brightness_range_alpha=plus["bra"][gen], brightness_range_beta=plus["brb"][gen],
                              max_blur=plus["max_blurs"][gen], max_noise=plus["max_noises"][gen])
                    if ds["kind"] == "isic":
                        F.create_augment_images_and_masks_ISIC_2018(os.path.join(src, "images"), os.path.join(src, "masks"),
                                                                    unl, n_plus, False, free_rotation=free_rot, **kw)
                    elif ds["kind"] == "multi":
                        F.create_augment_images_and_masks_multiclass(os.path.join(src, "images"), os.path.join(src, "masks"),
                                                                     unl, n_plus, False, free_rot, **kw)
                    else:
                        F.create_augment_images_and_masks_hela(src, unl, n_plus, False, free_rot, **kw)
                    alpha = plus["alphas"][gen]
                if rank == 0:    # labelled pairs join the pseudo-labelled directory
                    subs = ("brightfield", "alive", "dead", "mod_position") if ds["kind"] == "hela" else ("images", "masks")
                    if aim:      # AIM+: the un-augmented IM pairs too (ISIC_2018/13_ISIC_2018_aug_IM+.py:110-112)
                        for name in os.listdir(os.path.join(out["train_unlabeled"], subs[0])):
                            for sub in subs:
                                shutil.copy(os.path.join(out["train_unlabeled"], sub, name), os.path.join(unl, sub, name))
                    lab = P("TRAIN_LABELED_AUG_DIR" if aim else "TRAIN_LABELED_DIR")      # :114-116 / 11_...IM+.py:110-112
                    for name in os.listdir(os.path.join(lab, subs[0])):
                        for sub in subs:
                            shutil.copy(os.path.join(lab, sub, name), os.path.join(unl, sub, name))
                if torch.distributed.is_initialized():
                    torch.distributed.barrier()
                train_dir = os.path.join(unl, "brightfield" if ds["kind"] == "hela" else "images")
                steps = epoch_steps(len(os.listdir(train_dir)), batch, world)

                def train_candidate(i, side_by_side=False):
                    name_i = f"{modelname}_{i}"
                    h5 = os.path.join(model_dir, name_i + ".h5")
                    preds = [os.path.join(base, f"{k}_predictions", approach, name_i) for k in ("val", "test", "train_unlabeled")]
                    model = get_unet(H, W, C, K, alpha, actifu, actifu_out, seed=1000 * runid + 100 * gen + i)
                    if side_by_side:      # the other candidates' streams fill this one's gaps: no side stream of its own (results identical)
                        model.debug(single_stream=True)
                    if ds["kind"] == "isic":
                        res = F.train_ISIC_2018(train_dir, P("VAL_IMAGES_DIR"), P("VAL_MASKS_DIR"), P("TEST_IMAGES_DIR"),
                                                P("TEST_MASKS_DIR"), P("TRAIN_UNLABELED_IMAGES_DIR"), P("TRAIN_UNLABELED_MASKS_DIR"),
                                                name_i, h5, model, "mse", steps, H, W, C, *preds)
                    elif ds["kind"] == "multi":
                        res = F.train_multiclass(train_dir, P("VAL_IMAGES_DIR"), P("VAL_MASKS_DIR"), P("TEST_IMAGES_DIR"),
                                                 P("TEST_MASKS_DIR"), P("TRAIN_UNLABELED_IMAGES_DIR"), P("TRAIN_UNLABELED_MASKS_DIR"),
                                                 name_i, h5, model, "categorical_crossentropy", steps, H, W, C, K,
                                                 color_mapping(dataset, K), *preds)
                    else:
                        res = F.train_hela(train_dir, os.path.join(P("VAL_DIR"), "brightfield"), P("VAL_DIR"), P("TEST_DIR"),
                                           P("TRAIN_UNLABELED_DIR"), name_i, h5, model, "mse", steps, H, W, C, *preds)
                    del model
                    return (name_i,) + tuple(res)

                rows = train_candidates(_ints("IM_CANDIDATES", [0, 1, 2, 3, 4]), train_candidate, world, tick, parallel_candidates)

                if rank == 0:
                    top = sorted(rows, key=lambda r: r[ds["rank"]], reverse=True)[:top_k]
                    print(top)
                    for i, row in enumerate(top, start=1):
                        os.rename(os.path.join(model_dir, f"{row[0]}.h5"), os.path.join(model_dir, f"{row[0][:-2]}_topK_{i}.h5"))
                    os.makedirs(csv_dir, exist_ok=True)
                    with open(os.path.join(csv_dir, f"results_{modelname}.csv"), "w", encoding="utf-8", newline="") as f:
                        wr = csv.writer(f, delimiter=";")
                        wr.writerow(ds["header"])
                        wr.writerows(rows)
                    if world > 1:       # the CSV stays byte-compatible with the reference's; what a data-parallel run did differently
                        import json     # goes into a sidecar file
                        rule, mom = F.dp_bn_momentum_rule(world)
                        with open(os.path.join(csv_dir, f"results_{modelname}.meta.json"), "w", encoding="utf-8") as f:
                            json.dump({"data_parallel_ranks": world, "dp_mode": dp_mode(), "batch_per_rank": batch, "bn_momentum_rule": rule,
                                       "bn_momentum": round(mom, 6), "env": "IMK_DP_BN_MOMENTUM"}, f)
                    with open(os.path.join(csv_dir, f"mean_im_size_{modelname}.csv"), "w", encoding="utf-8", newline="") as f:
                        wr = csv.writer(f, delimiter=";")
                        wr.writerow(["val_mean_im_size", "test_mean_im_size", "unlabeled_mean_im_size"])
                        wr.writerow(means)
                if torch.distributed.is_initialized():
                    torch.distributed.barrier()
