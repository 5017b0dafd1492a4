"""Generation loop of the IM++ / AIM++ drivers of the reference: ISIC_2018/12_ISIC_2018_IM++.py,
ISIC_2018/14_ISIC_2018_aug_IM++.py, HeLa/12_HeLa_IM++.py, HeLa/14_HeLa_aug_IM++.py, SUIM/13_SUIM_IM++.py,
SUIM/15_SUIM_aug_IBAs++.py, Cityscapes/12_Cityscapes_IM++.py, Cityscapes/14_Cityscapes_aug_IM++.py (copies of one template; the `aug` variants differ in names and in which sets join the training directory).  Per run an ensemble of EvalNets is trained
on IM predictions of the labelled set (5 candidates, top-K by mean absolute error), then per generation the IM
pseudo-labels of the unlabeled set get 1..5 augmented copies each, weighted by the IoU the EvalNets predict, and 5 U-Net
candidates of growing width are trained on them.  Same loops, schedules, model / directory / CSV names.
Environment overrides for short runs: IM_RUNIDS, IM_NS, IM_GENS, IM_CANDIDATES, IM_EVALNET_CANDIDATES (comma-separated).

run(dataset, gt=True) is SUIM/16_SUIM_GT_IM++.py ("GT_IM_plus_plus": an upper bound for IM++): no EvalNet -- the number of
augmented copies follows the pseudo-label's IoU against the ground truth of the unlabeled set
(create_augment_images_and_masks_with_gt), and an epoch has at least as many steps as one over TRAIN_FULL (:126-132)."""
import csv
import os
import shutil

import torch

from . import functions as F
from . import paths
from .evalnet import get_evalnet, get_evalnet_miou
from .im_driver import DATASETS, _ints, color_mapping, epoch_steps, train_candidates
from .unet import get_unet

_HELA = dict(   # HeLa/14_HeLa_aug_IM++.py:53-57 (identical in 12_HeLa_IM++.py)
    alphas=[1, 1.25, 1.5, 1.75, 2], max_blurs=[0, 1, 1, 2, 3], max_noises=[5, 10, 15, 20, 25],
    bra=[(0.9, 1.1), (0.9, 1.1), (0.8, 1.2), (0.8, 1.2), (0.7, 1.3)], brb=[(-3, 3), (-6, 6), (-9, 9), (-12, 12), (-15, 15)])
_ISIC = dict(   # ISIC_2018/12_ISIC_2018_IM++.py:52-56
    alphas=[0.5, 0.75, 1, 1.25, 1.5], max_blurs=[0, 1, 1, 2, 3], max_noises=[5, 10, 15, 20, 25],
    bra=[(0.9, 1.1), (0.8, 1.2), (0.7, 1.3), (0.6, 1.4), (0.5, 1.5)], brb=[(-5, 5), (-10, 10), (-15, 15), (-20, 20), (-25, 25)])
SCHEDULE = {"HeLa": _HELA, "ISIC_2018": _ISIC,
            "SUIM": dict(_ISIC, alphas=[1, 1.25, 1.5, 1.75, 2]),   # SUIM/13_SUIM_IM++.py:54-58
            "Cityscapes": dict(alphas=[1, 1.25, 1.5, 1.75, 2], max_blurs=[0, 0, 0, 0, 1], max_noises=[3, 6, 9, 12, 15],   # 12_...:54-58
                               bra=[(0.95, 1.05), (0.9, 1.1), (0.8, 1.2), (0.7, 1.3), (0.6, 1.4)],
                               brb=[(-3, 3), (-6, 6), (-9, 9), (-12, 12), (-15, 15)])}


def run(dataset, aug=False, train_new_evalnet=True, gt=False):
    kind = DATASETS[dataset]["kind"]            # isic | hela | multi
    hela, multi = kind == "hela", kind == "multi"
    tag = {"HeLa": "HELA", "ISIC_2018": "ISIC_2018", "SUIM": "SUIM", "Cityscapes": "CITYSCAPES"}[dataset]
    S, D, sch = F.config[tag], F.config["DEFAULT"], SCHEDULE[dataset]
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
    approach = "GT_IM_plus_plus" if gt else ("aug_IM_plus_plus" if aug else "IM_plus_plus")
    if gt:
        if not multi or aug:
            raise ValueError("the ground-truth variant exists for the multi-class datasets only (SUIM/16_SUIM_GT_IM++.py)")
        train_new_evalnet = False
    subset_tag = f"{tag}_subset_aug" if aug else f"{tag}_subset"
    evalnet_tag = f"{tag}_evalnet_{'miou_' if (hela or multi) else ''}{'aug_' if aug else ''}im"
    P = lambda name: getattr(paths, f"{tag}_{name}")
    base, model_dir, csv_dir = P("BASE_DIR"), P("MODEL_DIR"), P("CSV_DIR")
    labeled_dir = os.path.join(base, "train_labeled_aug") if aug else P("TRAIN_LABELED_DIR")
    subs = ("brightfield", "alive", "dead", "mod_position") if hela else ("images", "masks")
    F.init_distributed()
    rank, world = F._rank_world()
    barrier = lambda: torch.distributed.barrier() if torch.distributed.is_initialized() else None

    for runid in _ints("IM_RUNIDS", [1, 2, 3]):
        if train_new_evalnet:
            ev_dir = os.path.join(base, "evalnet_aug_im" if aug else "evalnet_im", f"run_{runid}")
            subset_models = [F.load_model(os.path.join(model_dir, n)) for n in sorted(os.listdir(model_dir))
                             if f"{subset_tag}_{runid}" in n]
            if rank == 0:     # the labelled / validation sets are small: one rank writes, all ranks read
                for split, loops, sub in (("TRAIN_LABELED", loops_train, "train"), ("VAL", loops_val, "val")):
                    if hela:
                        F.create_training_data_evalnet_miou_im_hela(subset_models, H, W, C, P(f"{split}_DIR"),
                                                                    os.path.join(ev_dir, sub), loops)
                    elif multi:
                        F.create_training_data_evalnet_miou_im_multiclass(subset_models, H, W, C, K, P(f"{split}_IMAGES_DIR"),
                                                                          P(f"{split}_MASKS_DIR"), os.path.join(ev_dir, sub), loops)
                    else:
                        F.create_training_data_evalnet_im_binary(subset_models, H, W, C, P(f"{split}_IMAGES_DIR"),
                                                                 P(f"{split}_MASKS_DIR"), os.path.join(ev_dir, sub), loops)
            barrier()
            del subset_models
            def train_evalnet_candidate(i, side_by_side=False):
                name = f"{evalnet_tag}_{runid}_{i}"
                h5 = os.path.join(model_dir, name + ".h5")
                if hela:
                    evalnet = get_evalnet_miou(H, W, C, K, alpha_evalnet, seed=7000 * runid + i)
                elif multi:
                    evalnet = get_evalnet_miou(H, W, C, K, alpha_evalnet, seed=7000 * runid + i, onehot_B=True)
                else:
                    evalnet = get_evalnet(H, W, C, K, alpha_evalnet, normalize_B=True, seed=7000 * runid + i)
                if side_by_side:      # no side stream of its own: the other candidates fill the gaps (results identical)
                    evalnet.debug(single_stream=True)
                if hela:
                    res = F.train_evalnet_miou_model_hela(evalnet, os.path.join(ev_dir, "train"), os.path.join(ev_dir, "val"), h5,
                                                          bs_evalnet, ep_evalnet)
                elif multi:
                    res = F.train_evalnet_miou_model_multiclass(evalnet, H, W, os.path.join(ev_dir, "train"),
                                                                os.path.join(ev_dir, "val"), h5, bs_evalnet, K, ep_evalnet)
                else:
                    res = F.train_evalnet_ISIC_2018(evalnet, os.path.join(ev_dir, "train"), os.path.join(ev_dir, "val"), h5,
                                                    bs_evalnet, ep_evalnet)
                del evalnet
                return (name,) + tuple(res)
            rows = train_candidates(_ints("IM_EVALNET_CANDIDATES", [0, 1, 2, 3, 4]), train_evalnet_candidate, world)
            if rank == 0:
                # ascending: HeLa by iou_mae (14_HeLa...:131), SUIM by total_loss (13_SUIM...:129), ISIC by mae (12_ISIC...:124)
                top = sorted(rows, key=lambda r: r[4 if hela else (1 if multi else 2)])[:top_k]
                print(top)
                for i, row in enumerate(top, start=1):
                    os.rename(os.path.join(model_dir, f"{row[0]}.h5"), os.path.join(model_dir, f"{row[0][:-2]}_topK_{i}.h5"))
                os.makedirs(csv_dir, exist_ok=True)
                with open(os.path.join(csv_dir, f"results_{rows[-1][0]}.csv"), "w", encoding="utf-8", newline="") as f:
                    wr = csv.writer(f, delimiter=";")
                    wr.writerow(["modelname", "total_loss", "iou_loss", "detection_loss", "iou_mae", "detection_mae"] if hela
                                else ["modelname", "total_loss", "iou_loss", "conf_loss", "iou_mae", "conf_mae"] if multi
                                else ["modelname", "mse", "mae"])
                    wr.writerows(rows)
            barrier()

        for n in _ints("IM_NS", [2] if (hela or multi or aug) else [2, 3, 4]):
            for gen in _ints("IM_GENS", [0, 1, 2, 3, 4]):
                name_of = lambda g: f"{tag}_{approach}_{runid}_n{n}_gen{g}_e{EK}_d{DK}_bi_{BI}_bo_{BO}"
                modelname = name_of(gen)
                tmp = {k: os.path.join(base, f"{k}_predictions", approach, "temp", modelname) for k in ("val", "test", "train_unlabeled")}
                unl = os.path.join(base, "train_unlabeled_predictions", approach, modelname)
                if gen == 0:
                    files = [os.path.join(model_dir, f"{subset_tag}_{runid}_topK_{j}.h5") for j in range(1, n + 1)]
                else:
                    files = [os.path.join(model_dir, f"{name_of(gen - 1)}_topK_{j}.h5") for j in range(1, n + 1)]
                best_models = [F.load_model(f) for f in files]
                means = []
                for split, key in (("VAL", "val"), ("TEST", "test"), ("TRAIN_UNLABELED", "train_unlabeled")):
                    if hela:
                        means.append(F.create_pseudo_labels_im_hela(best_models, H, W, C, os.path.join(P(f"{split}_DIR"), "brightfield"),
                                                                    tmp[key], EK, DK, BI, BO))
                    elif multi:
                        means.append(F.create_pseudo_labels_im_multiclass(best_models, H, W, C, P(f"{split}_IMAGES_DIR"), tmp[key],
                                                                          True, EK, DK, BI, BO))
                    else:
                        means.append(F.create_pseudo_labels_im_ISIC_2018(best_models, H, W, C, P(f"{split}_IMAGES_DIR"), tmp[key],
                                                                         True, EK, DK, BI, BO))
                best_evalnets = [] if gt else [F.load_evalnet(os.path.join(model_dir, f"{evalnet_tag}_{runid}_topK_{j}.h5"))
                                               for j in range(1, n + 1)]
                aug_args = (t_min, t_max, tmp["train_unlabeled"], unl, sch["bra"][gen], sch["brb"][gen], sch["max_blurs"][gen],
                            sch["max_noises"][gen], free_rot)
                if gt:      # SUIM/16_SUIM_GT_IM++.py:110-120
                    F.create_augment_images_and_masks_with_gt(P("TRAIN_UNLABELED_MASKS_DIR"), *aug_args, True)
                elif hela:
                    F.create_augment_images_and_masks_with_evalnet_ensemble_hela(best_evalnets, H, W, C, *aug_args)
                elif multi:
                    F.create_augment_images_and_masks_with_evalnet_ensemble_multiclass(best_evalnets, H, W, C, K, *aug_args, True)
                else:
                    F.create_augment_images_and_masks_with_evalnet_ensemble_binary(best_evalnets, H, W, C, *aug_args)
                del best_evalnets
                if rank == 0:   # aug: the plain pseudo-labels + the augmented labelled set; else the labelled set (lines 215-228)
                    for src in ([tmp["train_unlabeled"], labeled_dir] if aug else [labeled_dir]):
                        for name in os.listdir(os.path.join(src, subs[0])):
                            if all(os.path.exists(os.path.join(src, sub, name)) for sub in subs):
                                for sub in subs:
                                    shutil.copy(os.path.join(src, sub, name), os.path.join(unl, sub, name))
                barrier()
                train_dir = os.path.join(unl, subs[0])
                steps = epoch_steps(len(os.listdir(train_dir)), batch, world)
                if gt:      # :126-132: never fewer steps than an epoch over the full training set
                    steps = max(steps, epoch_steps(len(os.listdir(P("TRAIN_FULL_IMAGES_DIR"))), batch, world))
                def train_candidate(i, side_by_side=False):
                    name_i = f"{modelname}_{i}"
                    h5 = os.path.join(model_dir, name_i + ".h5")
                    preds = [os.path.join(base, f"{k}_predictions", approach, name_i) for k in ("val", "test", "train_unlabeled")]
                    model = get_unet(H, W, C, K, sch["alphas"][gen], actifu, actifu_out, seed=1000 * runid + 100 * gen + i)
                    if side_by_side:      # the other candidates' streams fill this one's gaps: no side stream of its own (results identical)
                        model.debug(single_stream=True)
                    if hela:
                        res = F.train_hela(train_dir, os.path.join(P("VAL_DIR"), "brightfield"), P("VAL_DIR"), P("TEST_DIR"),
                                           P("TRAIN_UNLABELED_DIR"), name_i, h5, model, "mse", steps, H, W, C, *preds)
                    elif multi:
                        res = F.train_multiclass(train_dir, P("VAL_IMAGES_DIR"), P("VAL_MASKS_DIR"), P("TEST_IMAGES_DIR"),
                                                 P("TEST_MASKS_DIR"), P("TRAIN_UNLABELED_IMAGES_DIR"), P("TRAIN_UNLABELED_MASKS_DIR"),
                                                 name_i, h5, model, "categorical_crossentropy", steps, H, W, C, K,
                                                 color_mapping(dataset, K), *preds)
                    else:
                        res = F.train_ISIC_2018(train_dir, P("VAL_IMAGES_DIR"), P("VAL_MASKS_DIR"), P("TEST_IMAGES_DIR"),
                                                P("TEST_MASKS_DIR"), P("TRAIN_UNLABELED_IMAGES_DIR"), P("TRAIN_UNLABELED_MASKS_DIR"),
                                                name_i, h5, model, "mse", steps, H, W, C, *preds)
                    del model
                    return (name_i,) + tuple(res)
                # one rank: IM_PARALLEL_CANDIDATES (default 3) of them side by side, results identical (im_driver.train_candidates)
                rows = train_candidates(_ints("IM_CANDIDATES", [0, 1, 2, 3, 4]), train_candidate, world)
                if rank == 0:
                    rank_col = DATASETS[dataset]["rank"]
                    top = sorted(rows, key=lambda r: r[rank_col], reverse=True)[:top_k]
                    print(top)
                    for i, row in enumerate(top, start=1):
                        os.rename(os.path.join(model_dir, f"{row[0]}.h5"), os.path.join(model_dir, f"{row[0][:-2]}_topK_{i}.h5"))
                    os.makedirs(csv_dir, exist_ok=True)
                    with open(os.path.join(csv_dir, f"results_{modelname}.csv"), "w", encoding="utf-8", newline="") as f:
                        wr = csv.writer(f, delimiter=";")
                        wr.writerow(DATASETS[dataset]["header"])
                        wr.writerows(rows)
                    with open(os.path.join(csv_dir, f"mean_im_size_{modelname}.csv"), "w", encoding="utf-8", newline="") as f:
                        wr = csv.writer(f, delimiter=";")
                        wr.writerow(["val_mean_im_size", "test_mean_im_size", "unlabeled_mean_im_size"])
                        wr.writerow(means)
                barrier()


def run_hela(aug=True, train_new_evalnet=True):
    run("HeLa", aug, train_new_evalnet)
