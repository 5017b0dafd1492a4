"""ISIC-2018 Inconsistency-Mask generations on MI355X -- counterpart of the reference driver
ISIC_2018/09_ISIC_2018_IM.py (line numbers below refer to it).  Same loops, model / directory / CSV names
and top-K hand-off; the bodies run on libimk.so.  Launch with torch.distributed.run for multi-GPU.
Environment overrides for short runs: IM_RUNIDS, IM_NS, IM_GENS, IM_CANDIDATES (e.g. IM_GENS=0,1)."""
import csv
import os
import shutil
import sys

sys.path.append(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from inconsistencymasks_amd import functions as F  # noqa: E402
from inconsistencymasks_amd import paths  # noqa: E402
from inconsistencymasks_amd.functions import create_pseudo_labels_im_ISIC_2018, train_ISIC_2018  # noqa: E402
from inconsistencymasks_amd.unet import get_unet  # noqa: E402

config = F.config
S = config["ISIC_2018"]
IMAGE_WIDTH, IMAGE_HEIGHT, IMAGE_CHANNELS = int(S["IMAGE_WIDTH"]), int(S["IMAGE_HEIGHT"]), int(S["IMAGE_CHANNELS"])
NUM_CLASSES, ALPHA = int(S["NUM_CLASSES"]), float(S["ALPHA"])
ACTIFU, ACTIFU_OUTPUT = S["ACTIFU"], S["ACTIFU_OUTPUT"]
BATCH_SIZE, TOP_Ks = int(config["DEFAULT"]["BATCH_SIZE"]), int(config["DEFAULT"]["TOP_Ks"])
EK, DK = int(S["ERODE_KERNEL"]), int(S["DILATE_KERNEL"])
BI, BO = bool(S["BLOCK_INPUT"]), bool(S["BLOCK_OUTPUT"])   # :38-39 -- bool(str) is always True in the reference too

approach = "IM"
rgb = True
filter_bad_predictions = True


def _ints(name, default):
    v = os.environ.get(name)
    return [int(x) for x in v.split(",")] if v else default


def main():
    if int(os.environ.get("WORLD_SIZE", 1)) > 1 and not torch.distributed.is_initialized():
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
        torch.distributed.init_process_group("nccl")
    rank = F._rank_world()[0]
    for runid in _ints("IM_RUNIDS", [1, 2, 3]):                                     # :49
        for n in _ints("IM_NS", [2, 3, 4]):                                          # :51
            for gen in _ints("IM_GENS", [0, 1, 2, 3, 4]):                            # :53
                modelname = f"ISIC_2018_{approach}_{runid}_n{n}_gen{gen}_e{EK}_d{DK}_bi_{BI}_bo_{BO}"
                base = paths.ISIC_2018_BASE_DIR
                val_dir = os.path.join(base, "val_predictions", approach, modelname)
                test_dir = os.path.join(base, "test_predictions", approach, modelname)
                unl_dir = os.path.join(base, "train_unlabeled_predictions", approach, modelname)
                if gen == 0:                                                           # :67-72
                    files = [os.path.join(paths.ISIC_2018_MODEL_DIR, f"ISIC_2018_subset_{runid}_topK_{j}.h5")
                             for j in range(1, n + 1)]
                else:
                    prev = f"ISIC_2018_{approach}_{runid}_n{n}_gen{gen - 1}_e{EK}_d{DK}_bi_{BI}_bo_{BO}"
                    files = [os.path.join(paths.ISIC_2018_MODEL_DIR, f"{prev}_topK_{j}.h5") for j in range(1, n + 1)]
                best_models = [F.load_model(f, custom_objects={"dice_loss": F.dice_loss}) for f in files]

                args = (rgb, EK, DK, BI, BO, filter_bad_predictions)
                val_mean = create_pseudo_labels_im_ISIC_2018(best_models, IMAGE_HEIGHT, IMAGE_WIDTH, IMAGE_CHANNELS,
                                                             paths.ISIC_2018_VAL_IMAGES_DIR, val_dir, *args)
                test_mean = create_pseudo_labels_im_ISIC_2018(best_models, IMAGE_HEIGHT, IMAGE_WIDTH, IMAGE_CHANNELS,
                                                              paths.ISIC_2018_TEST_IMAGES_DIR, test_dir, *args)
                unl_mean = create_pseudo_labels_im_ISIC_2018(best_models, IMAGE_HEIGHT, IMAGE_WIDTH, IMAGE_CHANNELS,
                                                             paths.ISIC_2018_TRAIN_UNLABELED_IMAGES_DIR, unl_dir, *args)
                unl_images, unl_masks = os.path.join(unl_dir, "images"), os.path.join(unl_dir, "masks")
                if rank == 0:                                                          # :83-85
                    for name in os.listdir(paths.ISIC_2018_TRAIN_LABELED_IMAGES_DIR):
                        shutil.copy(os.path.join(paths.ISIC_2018_TRAIN_LABELED_IMAGES_DIR, name), os.path.join(unl_images, name))
                        shutil.copy(os.path.join(paths.ISIC_2018_TRAIN_LABELED_MASKS_DIR, name), os.path.join(unl_masks, name))
                if torch.distributed.is_initialized():
                    torch.distributed.barrier()
                steps_per_epoch = len(os.listdir(unl_images)) // BATCH_SIZE / max(F._rank_world()[1], 1)
                steps_per_epoch = max(int(steps_per_epoch), 1)

                benchmarks = []
                for i in _ints("IM_CANDIDATES", [0, 1, 2, 3, 4]):                    # :90
                    name_i = f"{modelname}_{i}"
                    h5 = os.path.join(paths.ISIC_2018_MODEL_DIR, name_i + ".h5")
                    model = get_unet(IMAGE_HEIGHT, IMAGE_WIDTH, IMAGE_CHANNELS, NUM_CLASSES, ALPHA, ACTIFU, ACTIFU_OUTPUT,
                                     seed=1000 * runid + 100 * gen + i)
                    res = train_ISIC_2018(unl_images, paths.ISIC_2018_VAL_IMAGES_DIR, paths.ISIC_2018_VAL_MASKS_DIR,
                                          paths.ISIC_2018_TEST_IMAGES_DIR, paths.ISIC_2018_TEST_MASKS_DIR,
                                          paths.ISIC_2018_TRAIN_UNLABELED_IMAGES_DIR, paths.ISIC_2018_TRAIN_UNLABELED_MASKS_DIR,
                                          name_i, h5, model, "mse", steps_per_epoch, IMAGE_HEIGHT, IMAGE_WIDTH, IMAGE_CHANNELS,
                                          os.path.join(base, "val_predictions", approach, name_i),
                                          os.path.join(base, "test_predictions", approach, name_i),
                                          os.path.join(base, "train_unlabeled_predictions", approach, name_i))
                    benchmarks.append((name_i,) + tuple(res))
                    del model

                if rank == 0:
                    top = sorted(benchmarks, key=lambda r: r[1], reverse=True)[:TOP_Ks]   # :124-126
                    print(top)
                    for i, row in enumerate(top, start=1):                              # :131-135
                        os.rename(os.path.join(paths.ISIC_2018_MODEL_DIR, f"{row[0]}.h5"),
                                  os.path.join(paths.ISIC_2018_MODEL_DIR, f"{row[0][:-2]}_topK_{i}.h5"))
                    os.makedirs(paths.ISIC_2018_CSV_DIR, exist_ok=True)
                    header = ["modelname", "mIoU_val", "mIoU_test", "mIoU_train_unlabeled", "dice_score_val",
                              "dice_score_test", "dice_score_train_unlabeled"]
                    with open(os.path.join(paths.ISIC_2018_CSV_DIR, f"results_{modelname}.csv"), "w", encoding="utf-8",
                              newline="") as f:
                        w = csv.writer(f, delimiter=";")
                        w.writerow(header)
                        w.writerows(benchmarks)
                    with open(os.path.join(paths.ISIC_2018_CSV_DIR, f"mean_im_size_{modelname}.csv"), "w", encoding="utf-8",
                              newline="") as f:
                        w = csv.writer(f, delimiter=";")
                        w.writerow(["val_mean_im_size", "test_mean_im_size", "unlabeled_mean_im_size"])
                        w.writerow([val_mean, test_mean, unl_mean])
                if torch.distributed.is_initialized():
                    torch.distributed.barrier()


if __name__ == "__main__":
    main()
