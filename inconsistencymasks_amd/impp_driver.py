"""Generation loop of the HeLa IM++ / AIM++ drivers of the reference (HeLa/12_HeLa_IM++.py, HeLa/14_HeLa_aug_IM++.py --
the two differ in names and in which labelled set joins the training directory): per run an ensemble of mIoU EvalNets is
trained on IM predictions of the labelled set (5 candidates, top-K by iou_mae), then per generation the IM
pseudo-labels of the unlabeled set get 1..5 augmented copies each, weighted by the EvalNets' predicted IoU, and 5
U-Net candidates of growing width are trained on them.  Same loops, schedules, model / directory / CSV names.
Environment overrides for short runs: IM_RUNIDS, IM_GENS, IM_CANDIDATES, IM_EVALNET_CANDIDATES (comma-separated)."""
import csv
import os
import shutil

import torch

from . import functions as F
from . import paths
from .evalnet import get_evalnet_miou
from .im_driver import DATASETS, _ints
from .unet import get_unet

# HeLa/14_HeLa_aug_IM++.py:53-57 (identical in 12_HeLa_IM++.py)
SCHEDULE = dict(alphas=[1, 1.25, 1.5, 1.75, 2], max_blurs=[0, 1, 1, 2, 3], max_noises=[5, 10, 15, 20, 25],
                bra=[(0.9, 1.1), (0.9, 1.1), (0.8, 1.2), (0.8, 1.2), (0.7, 1.3)],
                brb=[(-3, 3), (-6, 6), (-9, 9), (-12, 12), (-15, 15)])
SUBS = ("brightfield", "alive", "dead", "mod_position")


def run_hela(aug=True, train_new_evalnet=True):
    S, D = F.config["HELA"], F.config["DEFAULT"]
    H, W, C, K = int(S["IMAGE_HEIGHT"]), int(S["IMAGE_WIDTH"]), int(S["IMAGE_CHANNELS"]), int(S["NUM_CLASSES"])
    alpha_evalnet = float(S["ALPHA_EVALNET"])
    actifu, actifu_out = S["ACTIFU"], S["ACTIFU_OUTPUT"]
    bs_evalnet, ep_evalnet = int(D["BATCH_SIZE_EVALNET"]), int(D["NUM_EPOCHS_EVALNET"])
    loops_train, loops_val = int(D["NUM_LOOPS_TRAIN"]), int(D["NUM_LOOPS_VAL"])
    batch, top_k = int(D["BATCH_SIZE"]), int(D["TOP_Ks"])
    EK, DK = int(S["ERODE_KERNEL"]), int(S["DILATE_KERNEL"])
    BI, BO = S["BLOCK_INPUT"].lower() == "true", S["BLOCK_OUTPUT"].lower() == "true"
    t_min, t_max = float(S["MIN_THRESHOLD"]), float(S["MAX_THRESHOLD"])
    free_rot = S["FREE_ROTATION"].lower() == "true"
    approach = "aug_IM_plus_plus" if aug else "IM_plus_plus"
    subset_tag = "HELA_subset_aug" if aug else "HELA_subset"
    evalnet_tag = "HELA_evalnet_miou_aug_im" if aug else "HELA_evalnet_miou_im"
    base, model_dir, csv_dir = paths.HELA_BASE_DIR, paths.HELA_MODEL_DIR, paths.HELA_CSV_DIR
    labeled_dir = os.path.join(base, "train_labeled_aug") if aug else paths.HELA_TRAIN_LABELED_DIR
    if int(os.environ.get("WORLD_SIZE", 1)) > 1 and not torch.distributed.is_initialized():
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
        torch.distributed.init_process_group("nccl")
    rank, world = F._rank_world()
    barrier = lambda: torch.distributed.barrier() if torch.distributed.is_initialized() else None

    for runid in _ints("IM_RUNIDS", [1, 2, 3]):
        if train_new_evalnet:
            ev_dir = os.path.join(base, "evalnet_aug_im" if aug else "evalnet_im", f"run_{runid}")
            subset_models = [F.load_model(os.path.join(model_dir, n)) for n in sorted(os.listdir(model_dir))
                             if f"{subset_tag}_{runid}" in n]
            if rank == 0:     # the labelled / validation sets are small: one rank writes, all ranks read
                F.create_training_data_evalnet_miou_im_hela(subset_models, H, W, C, paths.HELA_TRAIN_LABELED_DIR,
                                                            os.path.join(ev_dir, "train"), loops_train)
                F.create_training_data_evalnet_miou_im_hela(subset_models, H, W, C, paths.HELA_VAL_DIR,
                                                            os.path.join(ev_dir, "val"), loops_val)
            barrier()
            del subset_models
            rows = []
            for i in _ints("IM_EVALNET_CANDIDATES", [0, 1, 2, 3, 4]):
                name = f"{evalnet_tag}_{runid}_{i}"
                evalnet = get_evalnet_miou(H, W, C, K, alpha_evalnet, seed=7000 * runid + i)
                res = F.train_evalnet_miou_model_hela(evalnet, os.path.join(ev_dir, "train"), os.path.join(ev_dir, "val"),
                                                      os.path.join(model_dir, name + ".h5"), bs_evalnet, ep_evalnet)
                rows.append((name,) + tuple(res))
                del evalnet
            if rank == 0:
                top = sorted(rows, key=lambda r: r[4])[:top_k]        # by iou_mae, ascending
                print(top)
                for i, row in enumerate(top, start=1):
                    os.rename(os.path.join(model_dir, f"{row[0]}.h5"), os.path.join(model_dir, f"{row[0][:-2]}_topK_{i}.h5"))
                os.makedirs(csv_dir, exist_ok=True)
                with open(os.path.join(csv_dir, f"results_{rows[-1][0]}.csv"), "w", encoding="utf-8", newline="") as f:
                    wr = csv.writer(f, delimiter=";")
                    wr.writerow(["modelname", "total_loss", "iou_loss", "detection_loss", "iou_mae", "detection_mae"])
                    wr.writerows(rows)
            barrier()

        n = 2
        for gen in _ints("IM_GENS", [0, 1, 2, 3, 4]):
            name_of = lambda g: f"HELA_{approach}_{runid}_n{n}_gen{g}_e{EK}_d{DK}_bi_{BI}_bo_{BO}"
            modelname = name_of(gen)
            tmp = {k: os.path.join(base, f"{k}_predictions", approach, "temp", modelname) for k in ("val", "test", "train_unlabeled")}
            unl = os.path.join(base, "train_unlabeled_predictions", approach, modelname)
            if gen == 0:
                files = [os.path.join(model_dir, f"{subset_tag}_{runid}_topK_{j}.h5") for j in range(1, n + 1)]
            else:
                files = [os.path.join(model_dir, f"{name_of(gen - 1)}_topK_{j}.h5") for j in range(1, n + 1)]
            best_models = [F.load_model(f) for f in files]
            means = [F.create_pseudo_labels_im_hela(best_models, H, W, C, os.path.join(d, "brightfield"), tmp[k], EK, DK, BI, BO)
                     for d, k in ((paths.HELA_VAL_DIR, "val"), (paths.HELA_TEST_DIR, "test"),
                                  (paths.HELA_TRAIN_UNLABELED_DIR, "train_unlabeled"))]
            best_evalnets = [F.load_evalnet(os.path.join(model_dir, f"{evalnet_tag}_{runid}_topK_{j}.h5")) for j in range(1, n + 1)]
            F.create_augment_images_and_masks_with_evalnet_ensemble_hela(
                best_evalnets, H, W, C, t_min, t_max, tmp["train_unlabeled"], unl, SCHEDULE["bra"][gen], SCHEDULE["brb"][gen],
                SCHEDULE["max_blurs"][gen], SCHEDULE["max_noises"][gen], free_rot)
            del best_evalnets
            if rank == 0:
                srcs = [tmp["train_unlabeled"], labeled_dir] if aug else [labeled_dir]   # 14_...:218-228 / 12_...:218-223
                for src in srcs:
                    for name in os.listdir(os.path.join(src, "brightfield")):
                        for sub in SUBS:
                            shutil.copy(os.path.join(src, sub, name), os.path.join(unl, sub, name))
            barrier()
            train_dir = os.path.join(unl, "brightfield")
            steps = max(len(os.listdir(train_dir)) // batch // world, 1)
            rows = []
            for i in _ints("IM_CANDIDATES", [0, 1, 2, 3, 4]):
                name_i = f"{modelname}_{i}"
                h5 = os.path.join(model_dir, name_i + ".h5")
                preds = [os.path.join(base, f"{k}_predictions", approach, name_i) for k in ("val", "test", "train_unlabeled")]
                model = get_unet(H, W, C, K, SCHEDULE["alphas"][gen], actifu, actifu_out, seed=1000 * runid + 100 * gen + i)
                res = F.train_hela(train_dir, os.path.join(paths.HELA_VAL_DIR, "brightfield"), paths.HELA_VAL_DIR,
                                   paths.HELA_TEST_DIR, paths.HELA_TRAIN_UNLABELED_DIR, name_i, h5, model, "mse", steps, H, W, C,
                                   *preds)
                rows.append((name_i,) + tuple(res))
                del model
            if rank == 0:
                top = sorted(rows, key=lambda r: r[4], reverse=True)[:top_k]   # tuple index 4 = mIoU_test, as the reference
                print(top)
                for i, row in enumerate(top, start=1):
                    os.rename(os.path.join(model_dir, f"{row[0]}.h5"), os.path.join(model_dir, f"{row[0][:-2]}_topK_{i}.h5"))
                os.makedirs(csv_dir, exist_ok=True)
                with open(os.path.join(csv_dir, f"results_{modelname}.csv"), "w", encoding="utf-8", newline="") as f:
                    wr = csv.writer(f, delimiter=";")
                    wr.writerow(DATASETS["HeLa"]["header"])
                    wr.writerows(rows)
                with open(os.path.join(csv_dir, f"mean_im_size_{modelname}.csv"), "w", encoding="utf-8", newline="") as f:
                    wr = csv.writer(f, delimiter=";")
                    wr.writerow(["val_mean_im_size", "test_mean_im_size", "unlabeled_mean_im_size"])
                    wr.writerow(means)
            barrier()
