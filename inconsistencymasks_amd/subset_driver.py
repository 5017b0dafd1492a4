"""The labelled-subset baseline that seeds generation 0 of every IM driver: counterpart of the reference's
ISIC_2018/03_ISIC_2018_subset.py:41-104, HeLa/03_HeLa_subset.py:41-104, SUIM/04_SUIM_subset.py:41-107 and
Cityscapes/03_Cityscapes_subset.py (copies of one template).  Ten candidates per run id are trained on the labelled
subset only, ranked (ISIC: mIoU_val; SUIM / Cityscapes: mIoU_val; HeLa: tuple index 6 ascending, as in the reference),
and the best TOP_Ks are renamed `{TAG}_subset_{runid}_topK_{j}.h5` -- the files `im_driver.run` loads for gen 0.
Environment overrides for short runs: IM_RUNIDS, IM_CANDIDATES (comma-separated).

run(dataset, aug=True) is the augmented-subset baseline (ALDT: ISIC_2018/04_ISIC_2018_subset_aug.py:34-43, HeLa/04_HeLa_subset_aug.py,
Cityscapes/04_Cityscapes_subset_aug.py, SUIM/05_SUIM_subset_aug.py): the labelled set is first augmented into
TRAIN_LABELED_AUG with the functions' default arguments, the candidates train on that, names carry `subset_aug` -- the
generation-0 ensemble of the AIM+ drivers (im_driver.run(..., approach="aug_IM_plus"))."""
import csv
import os

import torch

from . import functions as F
from . import paths
from .im_driver import DATASETS, _ints, color_mapping, epoch_steps, train_candidates
from .unet import get_unet

# (index of the ranking value in the row, descending?) -- ISIC_2018/03_ISIC_2018_subset.py:82, SUIM/04_SUIM_subset.py:84,
# HeLa/03_HeLa_subset.py:82 (`key=lambda x: x[6], reverse=False`)
_RANK = {"isic": (1, True), "multi": (4, True), "hela": (6, False)}


def train_candidate(ds, dataset, train_dir, name_i, h5, model, steps, H, W, C, K, preds):
    """One `train_*` call with the directory arguments of the reference's scripts."""
    P = lambda name: getattr(paths, f"{dataset.upper() if dataset != 'Cityscapes' else 'CITYSCAPES'}_{name}")
    if ds["kind"] == "isic":
        return F.train_ISIC_2018(train_dir, P("VAL_IMAGES_DIR"), P("VAL_MASKS_DIR"), P("TEST_IMAGES_DIR"),
                                 P("TEST_MASKS_DIR"), P("TRAIN_UNLABELED_IMAGES_DIR"), P("TRAIN_UNLABELED_MASKS_DIR"),
                                 name_i, h5, model, "mse", steps, H, W, C, *preds)
    if ds["kind"] == "multi":
        return F.train_multiclass(train_dir, P("VAL_IMAGES_DIR"), P("VAL_MASKS_DIR"), P("TEST_IMAGES_DIR"),
                                  P("TEST_MASKS_DIR"), P("TRAIN_UNLABELED_IMAGES_DIR"), P("TRAIN_UNLABELED_MASKS_DIR"),
                                  name_i, h5, model, "categorical_crossentropy", steps, H, W, C, K,
                                  color_mapping(dataset, K), *preds)
    return F.train_hela(train_dir, os.path.join(P("VAL_DIR"), "brightfield"), P("VAL_DIR"), P("TEST_DIR"),
                        P("TRAIN_UNLABELED_DIR"), name_i, h5, model, "mse", steps, H, W, C, *preds)


def run(dataset, aug=False):
    ds = DATASETS[dataset]
    S = F.config[ds["section"]]
    H, W, C = int(S["IMAGE_HEIGHT"]), int(S["IMAGE_WIDTH"]), int(S["IMAGE_CHANNELS"])
    K, alpha = int(S["NUM_CLASSES"]), float(S["ALPHA"])
    batch, top_k = int(F.config["DEFAULT"]["BATCH_SIZE"]), int(F.config["DEFAULT"]["TOP_Ks"])
    P = lambda name: getattr(paths, f"{dataset.upper() if dataset != 'Cityscapes' else 'CITYSCAPES'}_{name}")
    base, model_dir, csv_dir = P("BASE_DIR"), P("MODEL_DIR"), P("CSV_DIR")
    F.init_distributed()
    rank, world = F._rank_world()
    tag = {"HeLa": "HELA", "Cityscapes": "CITYSCAPES"}.get(dataset, dataset)
    approach = "subset_aug" if aug else "subset"
    if aug:     # default arguments, as the reference's scripts call them
        if ds["kind"] == "isic":
            F.create_augment_images_and_masks_ISIC_2018(P("TRAIN_LABELED_IMAGES_DIR"), P("TRAIN_LABELED_MASKS_DIR"), P("TRAIN_LABELED_AUG_MAIN_DIR"))
        elif ds["kind"] == "multi":
            F.create_augment_images_and_masks_multiclass(P("TRAIN_LABELED_IMAGES_DIR"), P("TRAIN_LABELED_MASKS_DIR"), P("TRAIN_LABELED_AUG_MAIN_DIR"))
        else:
            F.create_augment_images_and_masks_hela(P("TRAIN_LABELED_DIR"), P("TRAIN_LABELED_AUG_DIR"))
        if torch.distributed.is_initialized():
            torch.distributed.barrier()
    lab = "TRAIN_LABELED_AUG" if aug else "TRAIN_LABELED"
    train_dir = os.path.join(P(f"{lab}_DIR"), "brightfield") if ds["kind"] == "hela" else P(f"{lab}_IMAGES_DIR")
    steps = epoch_steps(len(os.listdir(train_dir)), batch, world)
    os.makedirs(model_dir, exist_ok=True)
    idx, desc = _RANK[ds["kind"]]
    for runid in _ints("IM_RUNIDS", [1, 2, 3]):
        modelname = f"{tag}_{approach}_{runid}"
        def one(i, side_by_side=False):
            name_i = f"{modelname}_{i}"
            h5 = os.path.join(model_dir, name_i + ".h5")
            preds = [os.path.join(base, f"{k}_predictions", approach, name_i) for k in ("val", "test", "train_unlabeled")]
            model = get_unet(H, W, C, K, alpha, S["ACTIFU"], S["ACTIFU_OUTPUT"], seed=7000 * runid + i)
            if side_by_side:
                model.debug(single_stream=True)
            row = (name_i,) + tuple(train_candidate(ds, dataset, train_dir, name_i, h5, model, steps, H, W, C, K, preds))
            del model
            return row
        # one rank: IM_PARALLEL_CANDIDATES (default 3) of the ten side by side, results identical (im_driver.train_candidates)
        rows = train_candidates(_ints("IM_CANDIDATES", list(range(10))), one, world)
        if rank == 0:
            top = sorted(rows, key=lambda r: r[idx], reverse=desc)[:top_k]
            print(top)
            for j, row in enumerate(top, start=1):
                os.rename(os.path.join(model_dir, f"{row[0]}.h5"), os.path.join(model_dir, f"{row[0][:-2]}_topK_{j}.h5"))
            os.makedirs(csv_dir, exist_ok=True)
            with open(os.path.join(csv_dir, f"results_{modelname}.csv"), "w", encoding="utf-8", newline="") as f:
                wr = csv.writer(f, delimiter=";")
                wr.writerow(ds["header"])
                wr.writerows(rows)
        if torch.distributed.is_initialized():
            torch.distributed.barrier()
