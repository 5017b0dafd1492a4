"""End-to-end run of the ISIC IM driver on a toy dataset (1 run id, n = 2, generations 0 and 1, 2 candidates):
the file/model/CSV naming and the top-K hand-off of ISIC_2018/09_ISIC_2018_IM.py:59-153."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CONFIG = """[DEFAULT]
SEED = 42
NUM_EPOCHS = 2
BATCH_SIZE = 8
LR = 0.003
WD = 1e-4
THRESHOLD = 0.5
TOP_Ks = 2

[ISIC_2018]
IMAGE_HEIGHT = 64
IMAGE_WIDTH = 64
IMAGE_CHANNELS = 3
NUM_CLASSES = 1
BASE_DIR = {base}/
ALPHA = 0.5
ACTIFU = relu
ACTIFU_OUTPUT = sigmoid
ERODE_KERNEL = 0
DILATE_KERNEL = 0
BLOCK_INPUT = True
BLOCK_OUTPUT = True
"""

SETUP = """
import os, sys
import numpy as np
sys.path.insert(0, {root!r})
from inconsistencymasks_amd import functions as F, paths
from inconsistencymasks_amd.unet import get_unet
rng = np.random.default_rng(0)
def sample(n, d_img, d_mask):
    os.makedirs(d_img, exist_ok=True); os.makedirs(d_mask, exist_ok=True)
    yy, xx = np.mgrid[0:64, 0:64]
    for i in range(n):
        cy, cx, r = rng.integers(20, 44), rng.integers(20, 44), rng.integers(8, 18)
        ell = (yy - cy) ** 2 + (xx - cx) ** 2 < r * r
        img = (170 + rng.integers(-10, 10, (64, 64, 3)) - ell[..., None] * 90).clip(0, 255).astype(np.uint8)
        F.write_png(os.path.join(d_img, f"ISIC_{{i:05d}}.png"), img)
        F.write_png(os.path.join(d_mask, f"ISIC_{{i:05d}}.png"), (ell * 255).astype(np.uint8))
sample(16, paths.ISIC_2018_TRAIN_LABELED_IMAGES_DIR, paths.ISIC_2018_TRAIN_LABELED_MASKS_DIR)
sample(24, paths.ISIC_2018_TRAIN_UNLABELED_IMAGES_DIR, paths.ISIC_2018_TRAIN_UNLABELED_MASKS_DIR)
sample(8, paths.ISIC_2018_VAL_IMAGES_DIR, paths.ISIC_2018_VAL_MASKS_DIR)
sample(8, paths.ISIC_2018_TEST_IMAGES_DIR, paths.ISIC_2018_TEST_MASKS_DIR)
os.makedirs(paths.ISIC_2018_MODEL_DIR, exist_ok=True)
import torch
x = torch.from_numpy(np.stack([F.read_png(os.path.join(paths.ISIC_2018_TRAIN_LABELED_IMAGES_DIR, n), 3) for n in sorted(os.listdir(paths.ISIC_2018_TRAIN_LABELED_IMAGES_DIR))])).cuda()
y = torch.from_numpy(np.stack([F.read_png(os.path.join(paths.ISIC_2018_TRAIN_LABELED_MASKS_DIR, n), 1) // 255 for n in sorted(os.listdir(paths.ISIC_2018_TRAIN_LABELED_MASKS_DIR))])).cuda()
for j in (1, 2):      # the gen-0 ensemble (03_ISIC_2018_subset.py's product), trained long enough for the BN statistics
    m = get_unet(64, 64, 3, 1, 0.5, "relu", "sigmoid", seed=j)
    for it in range(700):
        m.train_step(x, y, 0, 3e-3 if it < 200 else 0.0, 1e-4 if it < 200 else 0.0)
    m.repack()
    F.save_model(m, os.path.join(paths.ISIC_2018_MODEL_DIR, f"ISIC_2018_subset_1_topK_{{j}}.h5"))
"""


def test_isic_driver_toy_run(tmp_path):
    base = tmp_path / "data"
    cfg = tmp_path / "config.ini"
    cfg.write_text(CONFIG.format(base=base))
    env = {**os.environ, "IM_CONFIG": str(cfg), "IM_RUNIDS": "1", "IM_NS": "2", "IM_GENS": "0,1", "IM_CANDIDATES": "0,1"}
    subprocess.run([sys.executable, "-c", SETUP.format(root=ROOT)], env=env, check=True, cwd=tmp_path)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "ISIC_2018", "09_ISIC_2018_IM.py")], env=env, cwd=tmp_path,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    models = sorted(os.listdir(base / "models"))
    csvs = sorted(os.listdir(base / "csv"))
    stem = "ISIC_2018_IM_1_n2_gen{g}_e0_d0_bi_True_bo_True"
    expect_models = ["ISIC_2018_subset_1_topK_1.h5", "ISIC_2018_subset_1_topK_2.h5"]
    for g in (0, 1):          # both candidates survive (TOP_Ks = 2) and are renamed _topK_1 / _topK_2
        expect_models += [stem.format(g=g) + f"_topK_{i}.h5" for i in (1, 2)]
    assert models == sorted(expect_models)
    assert csvs == sorted([f"{k}_{stem.format(g=g)}.csv" for g in (0, 1) for k in ("results", "mean_im_size")])
    rows = (base / "csv" / f"results_{stem.format(g=1)}.csv").read_text().strip().splitlines()
    assert rows[0] == "modelname;mIoU_val;mIoU_test;mIoU_train_unlabeled;dice_score_val;dice_score_test;dice_score_train_unlabeled"
    assert len(rows) == 3 and rows[1].startswith(stem.format(g=1) + "_0;")
    im_rows = (base / "csv" / f"mean_im_size_{stem.format(g=0)}.csv").read_text().strip().splitlines()
    assert im_rows[0] == "val_mean_im_size;test_mean_im_size;unlabeled_mean_im_size" and len(im_rows[1].split(";")) == 3
    # pseudo-label directory: im/ for every unlabeled image, images/ + masks/ for kept + copied labelled ones
    unl = base / "train_unlabeled_predictions" / "IM" / stem.format(g=0)
    assert len(os.listdir(unl / "im")) == 24
    n_img = len(os.listdir(unl / "images"))
    assert 16 <= n_img <= 40 and n_img == len(os.listdir(unl / "masks"))
    miou_val = float(rows[1].split(";")[1])
    assert 0.0 <= miou_val <= 1.0


def test_isic_im_plus_toy_run(tmp_path):
    """ISIC_2018/11_ISIC_2018_IM+.py:59-135: IM output under temp/, augmented copies (+ labelled pairs) as the training
    set, alpha growing per generation (gen 0: 0.5, gen 1: 0.75 -> the gen-1 top-K models are wider)."""
    base = tmp_path / "data"
    cfg = tmp_path / "config.ini"
    cfg.write_text(CONFIG.format(base=base) + "FREE_ROTATION = True\nNUM_IMAGES_IM_PLUS = 2\n")
    env = {**os.environ, "IM_CONFIG": str(cfg), "IM_RUNIDS": "1", "IM_NS": "2", "IM_GENS": "0,1", "IM_CANDIDATES": "0,1"}
    subprocess.run([sys.executable, "-c", SETUP.format(root=ROOT)], env=env, check=True, cwd=tmp_path)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "ISIC_2018", "11_ISIC_2018_IM+.py")], env=env, cwd=tmp_path,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    stem = "ISIC_2018_IM_plus_1_n2_gen{g}_e0_d0_bi_True_bo_True"
    models = sorted(os.listdir(base / "models"))
    for g in (0, 1):
        for i in (1, 2):
            assert stem.format(g=g) + f"_topK_{i}.h5" in models
    temp = base / "train_unlabeled_predictions" / "IM_plus" / "temp" / stem.format(g=0)
    plus = base / "train_unlabeled_predictions" / "IM_plus" / stem.format(g=0)
    assert len(os.listdir(temp / "im")) == 24
    kept = sorted(os.listdir(temp / "images"))
    got = sorted(os.listdir(plus / "images"))
    want = sorted([f"{k[:-4]}_aug_{n}.png" for k in kept for n in (0, 1)] + [f"ISIC_{i:05d}.png" for i in range(16)])
    assert got == want and sorted(os.listdir(plus / "masks")) == want
    # masks of the augmented pairs are a flip/rotation of the temp mask: same foreground pixel count
    sys.path.insert(0, ROOT)
    from inconsistencymasks_amd import functions as F
    k = kept[0]
    src = F.read_png(str(temp / "masks" / k), 1)
    aug = F.read_png(str(plus / "masks" / f"{k[:-4]}_aug_0.png"), 1)
    assert aug.shape == src.shape and int((aug > 0).sum()) == int((src > 0).sum())
    # the width schedule: gen-0 models alpha 0.5, gen-1 models alpha 0.75
    m0 = F.load_model(str(base / "models" / (stem.format(g=0) + "_topK_1.h5")))
    m1 = F.load_model(str(base / "models" / (stem.format(g=1) + "_topK_1.h5")))
    assert (m0.plan.alpha, m1.plan.alpha) == (0.5, 0.75)


MULTI_CONFIG = """[DEFAULT]
SEED = 42
NUM_EPOCHS = 2
BATCH_SIZE = 8
LR = 0.003
WD = 1e-4
THRESHOLD = 0.5
TOP_Ks = 2

[SUIM]
IMAGE_HEIGHT = 64
IMAGE_WIDTH = 64
IMAGE_CHANNELS = 3
NUM_CLASSES = 3
BASE_DIR = {base}/
ALPHA = 0.5
ACTIFU = relu
ACTIFU_OUTPUT = softmax
ERODE_KERNEL = 0
DILATE_KERNEL = 0
BLOCK_INPUT = True
BLOCK_OUTPUT = True
FILTER_INCONSISTENT_CLASS_PRED = False
FREE_ROTATION = False
NUM_IMAGES_IM_PLUS = 1
"""

MULTI_SETUP = """
import os, sys
import numpy as np
sys.path.insert(0, {root!r})
from inconsistencymasks_amd import functions as F, paths
from inconsistencymasks_amd.unet import get_unet
rng = np.random.default_rng(0)
def sample(n, d_img, d_mask):
    os.makedirs(d_img, exist_ok=True); os.makedirs(d_mask, exist_ok=True)
    yy, xx = np.mgrid[0:64, 0:64]
    for i in range(n):
        cy, cx, r = rng.integers(20, 44), rng.integers(20, 44), rng.integers(8, 16)
        cls = np.zeros((64, 64), np.uint8)
        cls[yy > 40] = 1                                        # "sea floor"
        cls[(yy - cy) ** 2 + (xx - cx) ** 2 < r * r] = 2        # "object"
        img = np.stack([60 + 70 * (cls == 1) + 150 * (cls == 2), 90 + 60 * (cls == 2), 200 - 80 * (cls == 1)], -1)
        img = (img + rng.integers(-10, 10, (64, 64, 3))).clip(0, 255).astype(np.uint8)
        F.write_png(os.path.join(d_img, f"s_{{i:04d}}.png"), img)
        F.write_png(os.path.join(d_mask, f"s_{{i:04d}}.png"), cls)
sample(16, paths.SUIM_TRAIN_LABELED_IMAGES_DIR, paths.SUIM_TRAIN_LABELED_MASKS_DIR)
sample(24, paths.SUIM_TRAIN_UNLABELED_IMAGES_DIR, paths.SUIM_TRAIN_UNLABELED_MASKS_DIR)
sample(8, paths.SUIM_VAL_IMAGES_DIR, paths.SUIM_VAL_MASKS_DIR)
sample(8, paths.SUIM_TEST_IMAGES_DIR, paths.SUIM_TEST_MASKS_DIR)
os.makedirs(paths.SUIM_MODEL_DIR, exist_ok=True)
import torch
names = sorted(os.listdir(paths.SUIM_TRAIN_LABELED_IMAGES_DIR))
x = torch.from_numpy(np.stack([F.read_png(os.path.join(paths.SUIM_TRAIN_LABELED_IMAGES_DIR, n), 3) for n in names])).cuda()
y = torch.from_numpy(np.stack([F.read_png(os.path.join(paths.SUIM_TRAIN_LABELED_MASKS_DIR, n), 1)[..., 0] for n in names])).cuda()
for j in (1, 2):
    m = get_unet(64, 64, 3, 3, 0.5, "relu", "softmax", seed=j)
    for it in range(700):
        m.train_step(x, y, 1, 3e-3 if it < 200 else 0.0, 1e-4 if it < 200 else 0.0)
    m.repack()
    F.save_model(m, os.path.join(paths.SUIM_MODEL_DIR, f"SUIM_subset_1_topK_{{j}}.h5"))
"""


@pytest.mark.parametrize("script,approach", [("10_SUIM_IM.py", "IM"), ("12_SUIM_IM+.py", "IM_plus")])
def test_suim_driver_toy_run(tmp_path, script, approach):
    """SUIM/10_SUIM_IM.py and SUIM/12_SUIM_IM+.py on a 3-class toy set: argmax IM, CCE training, multiclass benchmark."""
    base = tmp_path / "data"
    cfg = tmp_path / "config.ini"
    cfg.write_text(MULTI_CONFIG.format(base=base))
    env = {**os.environ, "IM_CONFIG": str(cfg), "IM_RUNIDS": "1", "IM_NS": "2", "IM_GENS": "0", "IM_CANDIDATES": "0,1"}
    subprocess.run([sys.executable, "-c", MULTI_SETUP.format(root=ROOT)], env=env, check=True, cwd=tmp_path)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "SUIM", script)], env=env, cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    stem = f"SUIM_{approach}_1_n2_gen0_e0_d0_bi_True_bo_True"
    models = sorted(os.listdir(base / "models"))
    assert stem + "_topK_1.h5" in models and stem + "_topK_2.h5" in models
    rows = (base / "csv" / f"results_{stem}.csv").read_text().strip().splitlines()
    assert rows[0] == "modelname;mPA_val;mPA_test;mPA_train_unlabeled;mIoU_val;mIoU_test;mIoU_train_unlabeled"
    vals = [float(v) for v in rows[1].split(";")[1:]]
    assert len(rows) == 3 and all(0.0 <= v <= 1.0 for v in vals)
    # (candidates train for 2 short epochs: their BN moving statistics are far from converged, so no accuracy bar here;
    #  the 700-step gen-0 ensemble must agree on most pixels, though)
    im_rows = (base / "csv" / f"mean_im_size_{stem}.csv").read_text().strip().splitlines()
    assert all(float(v) < 0.5 * 64 * 64 for v in im_rows[1].split(";"))
    sub = ("temp",) if approach == "IM_plus" else ()
    unl = base.joinpath("train_unlabeled_predictions", approach, *sub, stem)
    assert len(os.listdir(unl / "im")) == 24 and len(os.listdir(unl / "masks")) == len(os.listdir(unl / "images"))
    sys.path.insert(0, ROOT)
    from inconsistencymasks_amd import functions as F
    m = F.read_png(str(unl / "masks" / sorted(os.listdir(unl / "masks"))[0]), 1)
    assert set(np.unique(m)) <= {0, 1, 2}                  # class ids, not x255
    pred_dir = base / "val_predictions" / approach / (stem + "_0")
    assert any(n.endswith("_color.png") for n in os.listdir(pred_dir))


HELA_CONFIG = """[DEFAULT]
SEED = 42
NUM_EPOCHS = 2
BATCH_SIZE = 8
LR = 0.003
WD = 1e-4
THRESHOLD = 0.5
TOP_Ks = 2

[HELA]
IMAGE_HEIGHT = 64
IMAGE_WIDTH = 64
IMAGE_CHANNELS = 1
NUM_CLASSES = 3
BASE_DIR = {base}/
ALPHA = 0.5
ACTIFU = relu
ACTIFU_OUTPUT = sigmoid
ERODE_KERNEL = 0
DILATE_KERNEL = 0
BLOCK_INPUT = True
BLOCK_OUTPUT = True
FREE_ROTATION = True
NUM_IMAGES_IM_PLUS = 1
"""

HELA_SETUP = """
import os, sys
import numpy as np
sys.path.insert(0, {root!r})
from inconsistencymasks_amd import functions as F, paths
from inconsistencymasks_amd.unet import get_unet
rng = np.random.default_rng(0)
yy, xx = np.mgrid[0:64, 0:64]
def sample(n, d, tag):
    for k in ("brightfield", "alive", "dead", "mod_position"):
        os.makedirs(os.path.join(d, k), exist_ok=True)
    for i in range(n):
        bf = np.full((64, 64), 120, np.int64) + rng.integers(-8, 8, (64, 64))
        alive = np.zeros((64, 64), np.uint8); dead = np.zeros((64, 64), np.uint8); pos = np.zeros((64, 64), np.uint8)
        for cy, cx, is_dead in ((16, 16, 0), (16, 46, 1), (46, 30, int(rng.integers(0, 2)))):
            cy += int(rng.integers(-4, 5)); cx += int(rng.integers(-4, 5))
            cell = (yy - cy) ** 2 + (xx - cx) ** 2 < 64
            bf[cell] += 70 if is_dead else -60
            (dead if is_dead else alive)[cell] = 255
            pos[(yy - cy) ** 2 + (xx - cx) ** 2 < 9] = 255
        name = f"{{tag}}_{{i:04d}}.png"
        F.write_png(os.path.join(d, "brightfield", name), bf.clip(0, 255).astype(np.uint8))
        F.write_png(os.path.join(d, "alive", name), alive); F.write_png(os.path.join(d, "dead", name), dead)
        F.write_png(os.path.join(d, "mod_position", name), pos)
sample(16, paths.HELA_TRAIN_LABELED_DIR, "lab"); sample(24, paths.HELA_TRAIN_UNLABELED_DIR, "unl")
sample(8, paths.HELA_VAL_DIR, "val"); sample(8, paths.HELA_TEST_DIR, "tst")
os.makedirs(paths.HELA_MODEL_DIR, exist_ok=True)
import torch
bfd = os.path.join(paths.HELA_TRAIN_LABELED_DIR, "brightfield")
items = [F.parse_image_hela(os.path.join(bfd, n), 1) for n in sorted(os.listdir(bfd))]
x = torch.from_numpy(np.stack([it[0] for it in items])).cuda()
y = torch.from_numpy(np.stack([it[1] for it in items])).cuda()
for j in (1, 2):
    m = get_unet(64, 64, 1, 3, 0.5, "relu", "sigmoid", seed=j)
    for it in range(700):
        m.train_step(x, y, 0, 3e-3 if it < 200 else 0.0, 1e-4 if it < 200 else 0.0)
    m.repack()
    F.save_model(m, os.path.join(paths.HELA_MODEL_DIR, f"HELA_subset_1_topK_{{j}}.h5"))
"""


def test_hela_driver_toy_run(tmp_path):
    """HeLa/09_HeLa_IM.py on a toy set: three binary IMs (>=) per image, position discs, 5 output directories, the
    9-value result tuple, ranking by tuple index 4."""
    base = tmp_path / "data"
    cfg = tmp_path / "config.ini"
    cfg.write_text(HELA_CONFIG.format(base=base))
    env = {**os.environ, "IM_CONFIG": str(cfg), "IM_RUNIDS": "1", "IM_NS": "2", "IM_GENS": "0", "IM_CANDIDATES": "0,1"}
    subprocess.run([sys.executable, "-c", HELA_SETUP.format(root=ROOT)], env=env, check=True, cwd=tmp_path)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "HeLa", "09_HeLa_IM.py")], env=env, cwd=tmp_path,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    stem = "HELA_IM_1_n2_gen0_e0_d0_bi_True_bo_True"
    models = sorted(os.listdir(base / "models"))
    assert stem + "_topK_1.h5" in models and stem + "_topK_2.h5" in models
    rows = (base / "csv" / f"results_{stem}.csv").read_text().strip().splitlines()
    assert rows[0].split(";") == ["modelname", "mIoU_val", "mIoU_ad_val", "mean_cell_count_error_val", "mIoU_test",
                                  "mIoU_ad_test", "mean_cell_count_error_test", "mIoU_unlabeled", "mIoU_ad_unlabeled",
                                  "mean_cell_count_error_unlabeled"]
    assert len(rows) == 3 and len(rows[1].split(";")) == 10
    unl = base / "train_unlabeled_predictions" / "IM" / stem
    for k in ("brightfield", "alive", "dead", "mod_position"):
        assert len(os.listdir(unl / k)) == 24 + 16                 # every unlabeled image + the labelled pairs
    assert len(os.listdir(unl / "im")) == 24
    # blocking: where the combined IM is set, brightfield and masks are zero
    sys.path.insert(0, ROOT)
    from inconsistencymasks_amd import functions as F
    n = sorted(os.listdir(unl / "im"))[0]
    im = F.read_png(str(unl / "im" / n), 1)[..., 0] > 0
    for k in ("brightfield", "alive", "dead"):
        assert not F.read_png(str(unl / k / n), 1)[..., 0][im].any()


HELA_PP_SETUP_EXTRA = """
import shutil
for j in (1, 2):   # the IM++ drivers start from the `HELA_subset_aug_{{runid}}` ensemble (HeLa/14_HeLa_aug_IM++.py:75,180)
    shutil.copy(os.path.join(paths.HELA_MODEL_DIR, f"HELA_subset_1_topK_{{j}}.h5"),
                os.path.join(paths.HELA_MODEL_DIR, f"HELA_subset_aug_1_topK_{{j}}.h5"))
sample(8, os.path.join(paths.HELA_BASE_DIR, "train_labeled_aug"), "laug")
"""


def test_hela_aug_im_plus_plus_toy_run(tmp_path):
    """HeLa/14_HeLa_aug_IM++.py on a toy set: EvalNet training data from sub-ensemble IMs (labels.csv, {0,1} masks),
    EvalNet candidates + top-K by iou_mae, EvalNet-weighted augmentation (1..5 copies `___j`), one U-Net generation."""
    base = tmp_path / "data"
    cfg = tmp_path / "config.ini"
    extra = "NUM_EPOCHS_EVALNET = 2\nBATCH_SIZE_EVALNET = 8\nNUM_LOOPS_TRAIN = 2\nNUM_LOOPS_VAL = 1\n"
    text = HELA_CONFIG.format(base=base).replace("TOP_Ks = 2\n", "TOP_Ks = 2\n" + extra)
    text += "ALPHA_EVALNET = 0.5\nMIN_THRESHOLD = 0.3\nMAX_THRESHOLD = 0.8\n"
    cfg.write_text(text)
    env = {**os.environ, "IM_CONFIG": str(cfg), "IM_RUNIDS": "1", "IM_GENS": "0", "IM_CANDIDATES": "0,1",
           "IM_EVALNET_CANDIDATES": "0,1,2"}
    subprocess.run([sys.executable, "-c", HELA_SETUP.format(root=ROOT) + HELA_PP_SETUP_EXTRA.format()], env=env, check=True,
                   cwd=tmp_path)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "HeLa", "14_HeLa_aug_IM++.py")], env=env, cwd=tmp_path,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    sys.path.insert(0, ROOT)
    from inconsistencymasks_amd import functions as F
    # EvalNet training data: 2 loops x 16 labelled images, label rows = name + 3 IoUs + 3 detection flags
    ev = base / "evalnet_aug_im" / "run_1"
    rows = [l.split(";") for l in (ev / "train" / "labels.csv").read_text().strip().splitlines()]
    assert len(rows) == 32 and all(len(r) == 7 for r in rows)
    assert all(0.0 <= float(v) <= 1.0 for r in rows for v in r[1:4]) and all(v in ("0", "1") for r in rows for v in r[4:])
    assert len(os.listdir(ev / "train" / "alive")) == 32 and len(os.listdir(ev / "val" / "alive")) == 8
    m = F.read_png(str(ev / "train" / "alive" / rows[0][0]), 1)
    assert set(np.unique(m)) <= {0, 1}                        # the uint8 wrap of `mask * 255` (functions.py:3939)
    # the cells are larger than 1 % of the image: detection flags of alive/dead present in every sample's GT
    assert any(r[4] == "1" for r in rows)
    models = sorted(os.listdir(base / "models"))
    assert "HELA_evalnet_miou_aug_im_1_topK_1.h5" in models and "HELA_evalnet_miou_aug_im_1_topK_2.h5" in models
    ev_rows = (base / "csv" / "results_HELA_evalnet_miou_aug_im_1_2.csv").read_text().strip().splitlines()
    assert ev_rows[0].split(";") == ["modelname", "total_loss", "iou_loss", "detection_loss", "iou_mae", "detection_mae"]
    assert len(ev_rows) == 4
    stem = "HELA_aug_IM_plus_plus_1_n2_gen0_e0_d0_bi_True_bo_True"
    assert stem + "_topK_1.h5" in models and stem + "_topK_2.h5" in models
    unl = base / "train_unlabeled_predictions" / "aug_IM_plus_plus" / stem
    names = os.listdir(unl / "brightfield")
    n_aug = [n for n in names if "___" in n]
    per_image = {}
    for n in n_aug:
        per_image.setdefault(n.split("___")[0], []).append(int(n.split("___")[1][:-4]))
    assert len(per_image) == 24 and all(sorted(v) == list(range(len(v))) and 1 <= len(v) <= 5 for v in per_image.values())
    assert len(names) == len(n_aug) + 24 + 8                  # + the plain IM pseudo-labels + train_labeled_aug
    for k in ("alive", "dead", "mod_position"):
        assert sorted(os.listdir(unl / k)) == sorted(names)
    res = (base / "csv" / f"results_{stem}.csv").read_text().strip().splitlines()
    assert len(res) == 3 and len(res[1].split(";")) == 10


def test_isic_im_plus_plus_toy_run(tmp_path):
    """ISIC_2018/12_ISIC_2018_IM++.py on a toy set: EvalNet training data (half of it augmented, IoU labels rounded to 4
    decimals), 3 EvalNet candidates ranked by mae, EvalNet-weighted augmentation, one U-Net generation at n = 2."""
    base = tmp_path / "data"
    cfg = tmp_path / "config.ini"
    extra = "NUM_EPOCHS_EVALNET = 2\nBATCH_SIZE_EVALNET = 8\nNUM_LOOPS_TRAIN = 2\nNUM_LOOPS_VAL = 1\n"
    text = CONFIG.format(base=base).replace("TOP_Ks = 2\n", "TOP_Ks = 2\n" + extra)
    text += "FREE_ROTATION = True\nALPHA_EVALNET = 0.5\nMIN_THRESHOLD = 0.3\nMAX_THRESHOLD = 0.8\n"
    cfg.write_text(text)
    env = {**os.environ, "IM_CONFIG": str(cfg), "IM_RUNIDS": "1", "IM_NS": "2", "IM_GENS": "0", "IM_CANDIDATES": "0,1",
           "IM_EVALNET_CANDIDATES": "0,1,2"}
    subprocess.run([sys.executable, "-c", SETUP.format(root=ROOT)], env=env, check=True, cwd=tmp_path)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "ISIC_2018", "12_ISIC_2018_IM++.py")], env=env, cwd=tmp_path,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    ev = base / "evalnet_im" / "run_1"
    rows = [l.split(";") for l in (ev / "train" / "labels.csv").read_text().strip().splitlines()]
    assert len(rows) == 32 and all(len(r) == 2 and 0.0 <= float(r[1]) <= 1.0 for r in rows)
    assert all(len(r[1].split(".")[-1]) <= 4 for r in rows)             # round(iou, 4), functions.py:3644
    assert np.mean([float(r[1]) for r in rows]) > 0.3                    # the toy ensemble segments the lesions
    assert sorted(os.listdir(ev / "train" / "images")) == sorted(r[0] for r in rows) == sorted(os.listdir(ev / "train" / "masks"))
    models = sorted(os.listdir(base / "models"))
    assert "ISIC_2018_evalnet_im_1_topK_1.h5" in models and "ISIC_2018_evalnet_im_1_topK_2.h5" in models
    ev_rows = (base / "csv" / "results_ISIC_2018_evalnet_im_1_2.csv").read_text().strip().splitlines()
    assert ev_rows[0] == "modelname;mse;mae" and len(ev_rows) == 4
    stem = "ISIC_2018_IM_plus_plus_1_n2_gen0_e0_d0_bi_True_bo_True"
    assert stem + "_topK_1.h5" in models and stem + "_topK_2.h5" in models
    unl = base / "train_unlabeled_predictions" / "IM_plus_plus" / stem
    tmp_imgs = os.listdir(base / "train_unlabeled_predictions" / "IM_plus_plus" / "temp" / stem / "images")
    names = os.listdir(unl / "images")
    per_image = {}
    for n in names:
        if "___" in n:
            per_image.setdefault(n.split("___")[0], []).append(int(n.split("___")[1][:-4]))
    assert sorted(per_image) == sorted(n[:-4] for n in tmp_imgs)         # every kept pseudo-label gets 1..5 copies
    assert all(sorted(v) == list(range(len(v))) and 1 <= len(v) <= 5 for v in per_image.values())
    assert len(names) == sum(len(v) for v in per_image.values()) + 16    # + the labelled pairs (12_...IM++.py:221-223)
    assert sorted(os.listdir(unl / "masks")) == sorted(names)
    res = (base / "csv" / f"results_{stem}.csv").read_text().strip().splitlines()
    assert len(res) == 3 and len(res[1].split(";")) == 7


def test_suim_im_plus_plus_toy_run(tmp_path):
    """SUIM/13_SUIM_IM++.py on a 3-class toy set: the EvalNet takes the label map as a one-hot stack; label rows = name +
    K class-wise IoUs + K detection flags; EvalNet candidates ranked by total loss."""
    base = tmp_path / "data"
    cfg = tmp_path / "config.ini"
    extra = "NUM_EPOCHS_EVALNET = 2\nBATCH_SIZE_EVALNET = 8\nNUM_LOOPS_TRAIN = 2\nNUM_LOOPS_VAL = 1\n"
    text = MULTI_CONFIG.format(base=base).replace("TOP_Ks = 2\n", "TOP_Ks = 2\n" + extra)
    text += "ALPHA_EVALNET = 0.5\nMIN_THRESHOLD = 0.3\nMAX_THRESHOLD = 0.8\n"
    cfg.write_text(text)
    env = {**os.environ, "IM_CONFIG": str(cfg), "IM_RUNIDS": "1", "IM_GENS": "0", "IM_CANDIDATES": "0,1",
           "IM_EVALNET_CANDIDATES": "0,1,2"}
    subprocess.run([sys.executable, "-c", MULTI_SETUP.format(root=ROOT)], env=env, check=True, cwd=tmp_path)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "SUIM", "13_SUIM_IM++.py")], env=env, cwd=tmp_path,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    ev = base / "evalnet_im" / "run_1"
    rows = [l.split(";") for l in (ev / "train" / "labels.csv").read_text().strip().splitlines()]
    assert len(rows) == 32 and all(len(r) == 1 + 2 * 3 for r in rows)
    assert all(0.0 <= float(v) <= 1.0 for r in rows for v in r[1:4]) and all(v in ("0", "1") for r in rows for v in r[4:])
    assert np.mean([float(r[3]) for r in rows]) > 0.3                   # the toy ensemble finds the "object" class
    models = sorted(os.listdir(base / "models"))
    assert "SUIM_evalnet_miou_im_1_topK_1.h5" in models and "SUIM_evalnet_miou_im_1_topK_2.h5" in models
    ev_rows = (base / "csv" / "results_SUIM_evalnet_miou_im_1_2.csv").read_text().strip().splitlines()
    assert ev_rows[0] == "modelname;total_loss;iou_loss;conf_loss;iou_mae;conf_mae" and len(ev_rows) == 4
    stem = "SUIM_IM_plus_plus_1_n2_gen0_e0_d0_bi_True_bo_True"
    assert stem + "_topK_1.h5" in models and stem + "_topK_2.h5" in models
    unl = base / "train_unlabeled_predictions" / "IM_plus_plus" / stem
    names = os.listdir(unl / "images")
    per_image = {}
    for n in names:
        if "___" in n:
            per_image.setdefault(n.split("___")[0], []).append(int(n.split("___")[1][:-4]))
    assert len(per_image) == 24 and all(sorted(v) == list(range(len(v))) and 1 <= len(v) <= 5 for v in per_image.values())
    assert len(names) == sum(len(v) for v in per_image.values()) + 16
    sys.path.insert(0, ROOT)
    from inconsistencymasks_amd import functions as F
    m = F.read_png(str(unl / "masks" / [n for n in names if "___" in n][0]), 1)
    assert set(np.unique(m)) <= {0, 1, 2}
    res = (base / "csv" / f"results_{stem}.csv").read_text().strip().splitlines()
    assert len(res) == 3 and len(res[1].split(";")) == 7


CITY_CONFIG = MULTI_CONFIG.replace("[SUIM]", "[CITYSCAPES]").replace("IMAGE_HEIGHT = 64", "IMAGE_HEIGHT = 48") \
    .replace("IMAGE_WIDTH = 64", "IMAGE_WIDTH = 96").replace("NUM_CLASSES = 3", "NUM_CLASSES = 5").replace("ALPHA = 0.5", "ALPHA = 1")

CITY_SETUP = """
import os, sys
import numpy as np
sys.path.insert(0, {root!r})
from inconsistencymasks_amd import functions as F, paths
from inconsistencymasks_amd.unet import get_unet
rng = np.random.default_rng(0)
H, W = 48, 96
def sample(n, d_img, d_mask):
    os.makedirs(d_img, exist_ok=True); os.makedirs(d_mask, exist_ok=True)
    yy, xx = np.mgrid[0:H, 0:W]
    for i in range(n):
        horizon, cx = rng.integers(16, 30), rng.integers(20, 76)
        cls = np.full((H, W), 1, np.uint8)                       # "sky"
        cls[yy > horizon] = 2                                     # "road"
        cls[(yy > horizon - 8) & (abs(xx - cx) < 9) & (yy < horizon + 6)] = 3      # "car"
        cls[(xx < 6) & (yy > 8)] = 4                              # "pole"
        img = np.stack([40 * cls + 20, 250 - 45 * cls, 30 + 50 * (cls == 3) + 20 * cls], -1)
        img = (img + rng.integers(-10, 10, (H, W, 3))).clip(0, 255).astype(np.uint8)
        F.write_png(os.path.join(d_img, f"c_{{i:04d}}.png"), img)
        F.write_png(os.path.join(d_mask, f"c_{{i:04d}}.png"), cls)
sample(16, paths.CITYSCAPES_TRAIN_LABELED_IMAGES_DIR, paths.CITYSCAPES_TRAIN_LABELED_MASKS_DIR)
sample(24, paths.CITYSCAPES_TRAIN_UNLABELED_IMAGES_DIR, paths.CITYSCAPES_TRAIN_UNLABELED_MASKS_DIR)
sample(8, paths.CITYSCAPES_VAL_IMAGES_DIR, paths.CITYSCAPES_VAL_MASKS_DIR)
sample(8, paths.CITYSCAPES_TEST_IMAGES_DIR, paths.CITYSCAPES_TEST_MASKS_DIR)
"""


@pytest.mark.parametrize("script,approach", [("09_Cityscapes_IM.py", "IM"), ("11_Cityscapes_IM+.py", "IM_plus")])
def test_cityscapes_driver_toy_run(tmp_path, script, approach):
    """Cityscapes/03_Cityscapes_subset.py (the labelled-subset baseline that seeds generation 0) followed by
    Cityscapes/09_Cityscapes_IM.py / 11_Cityscapes_IM+.py on a 5-class 48 x 96 toy set: non-square images, the CITYSCAPES
    model / CSV prefix, and for IM+ the width schedule alpha = 1 -> 1.25 over two generations."""
    base = tmp_path / "data"
    cfg = tmp_path / "config.ini"
    cfg.write_text(CITY_CONFIG.format(base=base))
    gens = "0,1" if approach == "IM_plus" else "0"
    env = {**os.environ, "IM_CONFIG": str(cfg), "IM_RUNIDS": "1", "IM_NS": "2", "IM_GENS": gens, "IM_CANDIDATES": "0,1,2"}
    subprocess.run([sys.executable, "-c", CITY_SETUP.format(root=ROOT)], env=env, check=True, cwd=tmp_path)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "Cityscapes", "03_Cityscapes_subset.py")], env=env, cwd=tmp_path,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    models = sorted(os.listdir(base / "models"))
    # 3 candidates, TOP_Ks = 2: two renamed, the third keeps its candidate name (03_Cityscapes_subset.py top-K rename)
    assert [m for m in models if "topK" in m] == ["CITYSCAPES_subset_1_topK_1.h5", "CITYSCAPES_subset_1_topK_2.h5"]
    assert len(models) == 3
    rows = (base / "csv" / "results_CITYSCAPES_subset_1.csv").read_text().strip().splitlines()
    assert rows[0] == "modelname;mPA_val;mPA_test;mPA_train_unlabeled;mIoU_val;mIoU_test;mIoU_train_unlabeled" and len(rows) == 4
    r = subprocess.run([sys.executable, os.path.join(ROOT, "Cityscapes", script)], env=env, cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    models = sorted(os.listdir(base / "models"))
    for g in gens.split(","):
        stem = f"CITYSCAPES_{approach}_1_n2_gen{g}_e0_d0_bi_True_bo_True"
        assert stem + "_topK_1.h5" in models and stem + "_topK_2.h5" in models
        res = (base / "csv" / f"results_{stem}.csv").read_text().strip().splitlines()
        assert len(res) == 4 and all(0.0 <= float(v) <= 1.0 for v in res[1].split(";")[1:])
    sys.path.insert(0, ROOT)
    from inconsistencymasks_amd import functions as F
    stem0 = f"CITYSCAPES_{approach}_1_n2_gen0_e0_d0_bi_True_bo_True"
    sub = ("temp",) if approach == "IM_plus" else ()
    unl = base.joinpath("train_unlabeled_predictions", approach, *sub, stem0)
    assert len(os.listdir(unl / "im")) == 24
    m = F.read_png(str(unl / "masks" / sorted(os.listdir(unl / "masks"))[0]), 1)
    assert m.shape[:2] == (48, 96) and set(np.unique(m)) <= {0, 1, 2, 3, 4}
    if approach == "IM_plus":
        m0 = F.load_model(str(base / "models" / (stem0 + "_topK_1.h5")))
        m1 = F.load_model(str(base / "models" / (stem0.replace("gen0", "gen1") + "_topK_1.h5")))
        assert (m0.plan.alpha, m1.plan.alpha) == (1.0, 1.25)


REF_STYLE_SCRIPT = """
# a per-dataset script written the way the reference's are (ISIC_2018/09_ISIC_2018_IM.py:1-153): top-level module names,
# TensorFlow touched through tf.device / load_model / clear_session / set_global_policy only, positional calls
import sys, os, gc
from functions import train_ISIC_2018, create_pseudo_labels_im_ISIC_2018, dice_loss
from unet import get_unet
import tensorflow as tf
from tensorflow.keras import mixed_precision
import paths
mixed_precision.set_global_policy('mixed_float16')
H = W = 64
with tf.device('/gpu:0'):
    files = [os.path.join(paths.ISIC_2018_MODEL_DIR, f'ISIC_2018_subset_1_topK_{j}.h5') for j in (1, 2)]
    best_models = [tf.keras.models.load_model(f, custom_objects={'dice_loss': dice_loss}) for f in files]
    out = os.path.join(paths.ISIC_2018_BASE_DIR, 'train_unlabeled_predictions', 'IM', 'ref_style')
    mean_im = create_pseudo_labels_im_ISIC_2018(best_models, H, W, 3, paths.ISIC_2018_TRAIN_UNLABELED_IMAGES_DIR, out, True, 0, 0, True, True, False)
    model = get_unet(H, W, 3, 1, 0.5, 'relu', 'sigmoid')
    res = train_ISIC_2018(os.path.join(out, 'images'), paths.ISIC_2018_VAL_IMAGES_DIR, paths.ISIC_2018_VAL_MASKS_DIR,
                          paths.ISIC_2018_TEST_IMAGES_DIR, paths.ISIC_2018_TEST_MASKS_DIR,
                          paths.ISIC_2018_TRAIN_UNLABELED_IMAGES_DIR, paths.ISIC_2018_TRAIN_UNLABELED_MASKS_DIR,
                          'ref_style_0', os.path.join(paths.ISIC_2018_MODEL_DIR, 'ref_style_0.h5'), model, 'mse',
                          max(len(os.listdir(os.path.join(out, 'images'))) // 8, 1), H, W, 3,
                          os.path.join(out, 'val_pred'), os.path.join(out, 'test_pred'), os.path.join(out, 'unl_pred'))
    del model
    tf.keras.backend.clear_session()
    gc.collect()
print('RESULT', mean_im, len(res))
"""


def test_reference_style_script_with_compat_namespace(tmp_path):
    """ISIC_2018/03_ISIC_2018_subset.py produces the gen-0 ensemble from the labelled subset; then a script in the
    reference's own style (`from functions import ...`, `import tensorflow as tf`) runs against the top-level shims and
    inconsistencymasks_amd/compat."""
    base = tmp_path / "data"
    cfg = tmp_path / "config.ini"
    cfg.write_text(CONFIG.format(base=base))
    env = {**os.environ, "IM_CONFIG": str(cfg), "IM_RUNIDS": "1", "IM_CANDIDATES": "0,1,2",
           "PYTHONPATH": os.pathsep.join([ROOT, os.path.join(ROOT, "inconsistencymasks_amd", "compat")])}
    setup = SETUP.format(root=ROOT).split("import torch\nx = torch.from_numpy")[0]      # the data only, no fabricated models
    subprocess.run([sys.executable, "-c", setup], env=env, check=True, cwd=tmp_path)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "ISIC_2018", "03_ISIC_2018_subset.py")], env=env, cwd=tmp_path,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert {"ISIC_2018_subset_1_topK_1.h5", "ISIC_2018_subset_1_topK_2.h5"} <= set(os.listdir(base / "models"))
    script = tmp_path / "ref_style.py"
    script.write_text(REF_STYLE_SCRIPT)
    r = subprocess.run([sys.executable, str(script)], env=env, cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][0].split()
    assert float(line[1]) >= 0 and int(line[2]) == 6
    assert os.path.exists(base / "models" / "ref_style_0.h5")


def _run_one_and_two_ranks(tmp_path, config, setup, script, extra_env=None, worlds=(1, 2)):
    """the same toy driver run with one rank and with two (or `worlds[1]`) ranks time-slicing one GPU over gloo; returns the data dirs"""
    import socket
    outs = {}
    for world in worlds:
        work = tmp_path / f"w{world}"
        base = work / "data"
        work.mkdir()
        cfg = work / "config.ini"
        cfg.write_text(config.format(base=base))
        env = {**os.environ, "IM_CONFIG": str(cfg), "IM_RUNIDS": "1", "IM_NS": "2", "IM_GENS": "0", "IM_CANDIDATES": "0,1",
               **(extra_env or {})}
        subprocess.run([sys.executable, "-c", setup.format(root=ROOT)], env=env, check=True, cwd=work)
        if world == 1:
            cmd = [sys.executable, script]
        else:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            env.update(IMK_DIST_BACKEND="gloo", IMK_ONE_GPU="1")
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                   "--master-port", str(port), script]
        r = subprocess.run(cmd, env=env, cwd=work, capture_output=True, text=True, timeout=1500)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        outs[world] = base
    return outs


def _same_png_tree(a, b, subs, channels):
    sys.path.insert(0, ROOT)
    from inconsistencymasks_amd import functions as F
    for sub in subs:
        assert sorted(os.listdir(a / sub)) == sorted(os.listdir(b / sub)), sub
        for n in sorted(os.listdir(a / sub)):
            ch = channels.get(sub, 1)
            assert np.array_equal(F.read_png(str(a / sub / n), ch), F.read_png(str(b / sub / n), ch)), (sub, n)


def test_isic_driver_two_ranks_on_one_gpu(tmp_path):
    """The multi-rank path of the IM driver end to end: `torch.distributed.run --nproc-per-node 2` of
    ISIC_2018/09_ISIC_2018_IM.py with the two ranks time-slicing ONE GPU over gloo (IMK_DIST_BACKEND / IMK_ONE_GPU; RCCL
    refuses two ranks on one device).  Sharded ensemble inference + IM must write exactly the files and pixels of the
    single-rank run (same mean IM sizes), training runs data-parallel (gradient all-reduce, moving statistics averaged,
    sharded benchmarks with gathered metric lists), rank 0 saves and ranks the candidates."""
    outs = _run_one_and_two_ranks(tmp_path, CONFIG, SETUP, os.path.join(ROOT, "ISIC_2018", "09_ISIC_2018_IM.py"))
    stem = "ISIC_2018_IM_1_n2_gen0_e0_d0_bi_True_bo_True"
    a, b = (outs[w] / "train_unlabeled_predictions" / "IM" / stem for w in (1, 2))
    _same_png_tree(a, b, ("im", "images", "masks"), {"images": 3})
    im1, im2 = ((outs[w] / "csv" / f"mean_im_size_{stem}.csv").read_text() for w in (1, 2))
    assert im1 == im2
    models = sorted(os.listdir(outs[2] / "models"))
    assert stem + "_topK_1.h5" in models and stem + "_topK_2.h5" in models
    rows = (outs[2] / "csv" / f"results_{stem}.csv").read_text().strip().splitlines()
    # the CSV is the reference's format at any world size (header on line 1); a data-parallel run names its BatchNorm momentum rule
    # (default: the reference's 0.99 per step) in a sidecar file
    assert not rows[0].startswith("#")
    assert rows[0] == (outs[1] / "csv" / f"results_{stem}.csv").read_text().strip().splitlines()[0]
    meta = json.loads((outs[2] / "csv" / f"results_{stem}.meta.json").read_text())
    assert meta["data_parallel_ranks"] == 2 and meta["bn_momentum_rule"] == "reference" and meta["bn_momentum"] == 0.99
    assert not (outs[1] / "csv" / f"results_{stem}.meta.json").exists()
    assert len(rows) == 3 and all(0.0 <= float(v) <= 1.0 for v in rows[1].split(";")[1:])
    # every validation / test prediction was written exactly once although the benchmarks were sharded
    assert len(os.listdir(outs[2] / "val_predictions" / "IM" / (stem + "_0"))) == 8
    assert len(os.listdir(outs[2] / "test_predictions" / "IM" / (stem + "_0"))) == 8


def test_suim_and_hela_drivers_two_ranks_on_one_gpu(tmp_path):
    """the same for the multiclass (argmax IM, CCE, MeanIoU monitor, sharded benchmark_multiclass) and the HeLa (three `>=`
    IMs, position discs, val-loss monitor, sharded benchmark_hela) drivers"""
    (tmp_path / "suim").mkdir(); (tmp_path / "hela").mkdir()
    outs = _run_one_and_two_ranks(tmp_path / "suim", MULTI_CONFIG, MULTI_SETUP, os.path.join(ROOT, "SUIM", "10_SUIM_IM.py"))
    stem = "SUIM_IM_1_n2_gen0_e0_d0_bi_True_bo_True"
    a, b = (outs[w] / "train_unlabeled_predictions" / "IM" / stem for w in (1, 2))
    _same_png_tree(a, b, ("im", "images", "masks"), {"images": 3})
    assert (outs[1] / "csv" / f"mean_im_size_{stem}.csv").read_text() == (outs[2] / "csv" / f"mean_im_size_{stem}.csv").read_text()
    rows = (outs[2] / "csv" / f"results_{stem}.csv").read_text().strip().splitlines()
    assert len(rows) == 3 and all(0.0 <= float(v) <= 1.0 for v in rows[1].split(";")[1:])
    outs = _run_one_and_two_ranks(tmp_path / "hela", HELA_CONFIG, HELA_SETUP, os.path.join(ROOT, "HeLa", "09_HeLa_IM.py"))
    stem = "HELA_IM_1_n2_gen0_e0_d0_bi_True_bo_True"
    a, b = (outs[w] / "train_unlabeled_predictions" / "IM" / stem for w in (1, 2))
    _same_png_tree(a, b, ("im", "brightfield", "alive", "dead", "mod_position"), {"mod_position": 3})
    rows = (outs[2] / "csv" / f"results_{stem}.csv").read_text().strip().splitlines()
    assert len(rows) == 3 and len(rows[1].split(";")) == 10


def test_isic_im_plus_plus_two_ranks_on_one_gpu(tmp_path):
    """ISIC_2018/12_ISIC_2018_IM++.py with two ranks: EvalNet training data, EvalNet candidates, EvalNet-weighted
    augmentation and the U-Net generation all run under torch.distributed; the temp IM directory equals the single-rank one."""
    extra = "NUM_EPOCHS_EVALNET = 2\nBATCH_SIZE_EVALNET = 8\nNUM_LOOPS_TRAIN = 2\nNUM_LOOPS_VAL = 1\n"
    text = CONFIG.replace("TOP_Ks = 2\n", "TOP_Ks = 2\n" + extra) + "FREE_ROTATION = True\nALPHA_EVALNET = 0.5\nMIN_THRESHOLD = 0.3\nMAX_THRESHOLD = 0.8\n"
    outs = _run_one_and_two_ranks(tmp_path, text, SETUP, os.path.join(ROOT, "ISIC_2018", "12_ISIC_2018_IM++.py"),
                                  {"IM_EVALNET_CANDIDATES": "0,1,2"})
    stem = "ISIC_2018_IM_plus_plus_1_n2_gen0_e0_d0_bi_True_bo_True"
    a, b = (outs[w] / "train_unlabeled_predictions" / "IM_plus_plus" / "temp" / stem for w in (1, 2))
    _same_png_tree(a, b, ("im", "images", "masks"), {"images": 3})
    models = sorted(os.listdir(outs[2] / "models"))
    assert "ISIC_2018_evalnet_im_1_topK_1.h5" in models and stem + "_topK_1.h5" in models
    res = (outs[2] / "csv" / f"results_{stem}.csv").read_text().strip().splitlines()
    assert len(res) == 3 and len(res[1].split(";")) == 7


def test_isic_aug_subset_and_aim_plus_toy_run(tmp_path):
    """ISIC_2018/04_ISIC_2018_subset_aug.py (ALDT: the labelled set augmented into train_labeled_aug, candidates trained on it,
    `ISIC_2018_subset_aug_{runid}_topK_j.h5`) followed by ISIC_2018/13_ISIC_2018_aug_IM+.py:44-116 (AIM+: gen 0 loads those
    models; training set = augmented copies + the un-augmented IM pairs + the AUGMENTED labelled set)."""
    base = tmp_path / "data"
    cfg = tmp_path / "config.ini"
    cfg.write_text(CONFIG.format(base=base) + "FREE_ROTATION = True\nNUM_IMAGES_IM_PLUS = 1\n")
    env = {**os.environ, "IM_CONFIG": str(cfg), "IM_RUNIDS": "1", "IM_NS": "2", "IM_GENS": "0", "IM_CANDIDATES": "0,1"}
    subprocess.run([sys.executable, "-c", SETUP.format(root=ROOT)], env=env, check=True, cwd=tmp_path)
    for script in ("04_ISIC_2018_subset_aug.py", "13_ISIC_2018_aug_IM+.py"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "ISIC_2018", script)], env=env, cwd=tmp_path, capture_output=True, text=True)
        assert r.returncode == 0, script + r.stdout[-2000:] + r.stderr[-3000:]
    aug = sorted(os.listdir(base / "train_labeled_aug" / "images"))
    assert len(aug) == 16 * 10 and aug == sorted(os.listdir(base / "train_labeled_aug" / "masks"))     # 9 copies + the original
    models = sorted(os.listdir(base / "models"))
    assert "ISIC_2018_subset_aug_1_topK_1.h5" in models and "ISIC_2018_subset_aug_1_topK_2.h5" in models
    stem = "ISIC_2018_aug_IM_plus_1_n2_gen0_e0_d0_bi_True_bo_True"
    assert stem + "_topK_1.h5" in models and stem + "_topK_2.h5" in models
    temp = base / "train_unlabeled_predictions" / "aug_IM_plus" / "temp" / stem
    plus = base / "train_unlabeled_predictions" / "aug_IM_plus" / stem
    kept = sorted(os.listdir(temp / "images"))
    # (a set: the toy splits share file names, so an augmented copy of a kept pair can carry the name of a labelled copy; which
    #  pairs are kept varies from run to run with the unseeded augmentation draws of the first script)
    want = sorted(set([f"{k[:-4]}_aug_0.png" for k in kept] + kept + aug))
    for sub in ("images", "masks"):
        have = sorted(os.listdir(plus / sub))
        assert have == want, (sub, sorted(set(want) - set(have))[:6], sorted(set(have) - set(want))[:6], len(have), len(want))
    assert (base / "csv" / f"results_{stem}.csv").exists() and (base / "csv" / "results_ISIC_2018_subset_aug_1.csv").exists()


def test_suim_gt_im_plus_plus_toy_run(tmp_path):
    """SUIM/16_SUIM_GT_IM++.py: IM++ without an EvalNet -- 1..5 augmented copies per pseudo-labelled pair by its IoU against
    the ground truth (IM pixels blanked), at least an epoch over TRAIN_FULL worth of steps."""
    base = tmp_path / "data"
    cfg = tmp_path / "config.ini"
    extra = "NUM_EPOCHS_EVALNET = 1\nBATCH_SIZE_EVALNET = 8\nNUM_LOOPS_TRAIN = 1\nNUM_LOOPS_VAL = 1\n"
    cfg.write_text(MULTI_CONFIG.format(base=base).replace("TOP_Ks = 2\n", "TOP_Ks = 2\n" + extra)
                   + "ALPHA_EVALNET = 0.5\nMIN_THRESHOLD = 0.3\nMAX_THRESHOLD = 0.8\n")
    env = {**os.environ, "IM_CONFIG": str(cfg), "IM_RUNIDS": "1", "IM_GENS": "0", "IM_CANDIDATES": "0,1"}
    setup = MULTI_SETUP.format(root=ROOT) + "\nsample(4, paths.SUIM_TRAIN_FULL_IMAGES_DIR, paths.SUIM_TRAIN_FULL_MASKS_DIR)\n"
    subprocess.run([sys.executable, "-c", setup], env=env, check=True, cwd=tmp_path)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "SUIM", "16_SUIM_GT_IM++.py")], env=env, cwd=tmp_path,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    stem = "SUIM_GT_IM_plus_plus_1_n2_gen0_e0_d0_bi_True_bo_True"
    models = sorted(os.listdir(base / "models"))
    assert stem + "_topK_1.h5" in models and not any("evalnet" in m for m in models)
    unl = base / "train_unlabeled_predictions" / "GT_IM_plus_plus" / stem
    names = os.listdir(unl / "images")
    per_image = {}
    for n in names:
        if "___" in n:
            per_image.setdefault(n.split("___")[0], []).append(int(n.split("___")[1][:-4]))
    assert len(per_image) == 24 and all(sorted(v) == list(range(len(v))) and 1 <= len(v) <= 5 for v in per_image.values())
    assert len(names) == sum(len(v) for v in per_image.values()) + 16
    assert max(len(v) for v in per_image.values()) >= 3          # the toy ensemble's pseudo-labels are good: many copies


def test_writer_rgb_false_feeds_the_file_order_to_the_nets(tmp_path):
    """rgb=False (functions.py:2847-2850: the nets get the channels as OpenCV decodes the file, BGR): equals the rgb=True run
    on the channel-flipped files, the written images keeping each file's own channel order."""
    sys.path.insert(0, ROOT)
    from inconsistencymasks_amd import functions as F
    from inconsistencymasks_amd.unet import get_unet
    rng = np.random.default_rng(1)
    d1, d2 = tmp_path / "a", tmp_path / "b"
    os.makedirs(d1); os.makedirs(d2)
    for i in range(5):
        img = rng.integers(0, 256, (32, 48, 3)).astype(np.uint8)
        F.write_png(str(d1 / f"i{i}.png"), img)
        F.write_png(str(d2 / f"i{i}.png"), img[..., ::-1].copy())
    models = [get_unet(32, 48, 3, 1, 0.5, "relu", "sigmoid", seed=s) for s in (1, 2)]
    for m in models:
        m.repack()
    F.create_pseudo_labels_im_ISIC_2018(models, 32, 48, 3, str(d1), str(tmp_path / "o1"), False, 0, 0, True, True, False)
    F.create_pseudo_labels_im_ISIC_2018(models, 32, 48, 3, str(d2), str(tmp_path / "o2"), True, 0, 0, True, True, False)
    for i in range(5):
        for sub in ("masks", "im"):
            assert np.array_equal(F.read_png(str(tmp_path / "o1" / sub / f"i{i}.png"), 1), F.read_png(str(tmp_path / "o2" / sub / f"i{i}.png"), 1))
        a, b = F.read_png(str(tmp_path / "o1" / "images" / f"i{i}.png"), 3), F.read_png(str(tmp_path / "o2" / "images" / f"i{i}.png"), 3)
        assert np.array_equal(a, b[..., ::-1])


def test_bench_one_rank_process_group_rccl():
    """bench.py --gpus 1 with IMK_FORCE_DIST=1: a real one-rank `nccl` (= RCCL) process group, so that init_process_group with
    device_id, the flat gradient all-reduce of every step, the MAX / SUM reductions of the timing and dist.barrier execute on
    the GPU box (the multi-rank path otherwise only runs over gloo in this repository's tests).  The JSON line must be the
    last line of stdout (RCCL prints a version banner through C stdio)."""
    import json
    env = {**os.environ, "IMK_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1"}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--images", "96", "--labeled", "32", "--steps", "1",
                        "--pretrain-steps", "10", "--bn-settle-steps", "10", "--no-cpu-baseline"], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    from test_cpu_api import check_bench_line
    out = check_bench_line(r.stdout, expect_cpu_baseline=False)        # the last stdout line: < 4 KB, the contract's keys
    assert out["n_gpus"] == 1 and out["config"]["process_group"].startswith("nccl") and out["value"] > 0
    assert out["roofline"]["by_stage"]["training"]["host_enqueue_ms"] > 0
    full = json.load(open(os.path.join(ROOT, out["detail"]) if not os.path.isabs(out["detail"]) else out["detail"]))
    assert full["value"] == out["value"] and full["roofline"]["step"]["train_step"]["host_enqueue_ms_per_step"] > 0
    assert "all_families" in full["roofline"] and "timed_region_kernel_totals" in full


def test_bench_two_ranks_driver_command_form():
    """The driver's N > 1 command, literally (`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1
    --master-port P bench.py --gpus 2 ...`), with both ranks time-slicing this box's one GPU over gloo (IMK_BENCH_ONE_GPU /
    IMK_BENCH_BACKEND): the unlabeled set sharded in contiguous blocks, the per-step gradient all-reduce, the barrier-bracketed timing
    with its MAX over ranks and the single JSON line from rank 0 all execute.  (RCCL with two ranks needs two GPUs.)"""
    import json
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {**os.environ, "IMK_BENCH_ONE_GPU": "1", "IMK_BENCH_BACKEND": "gloo"}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--images", "96", "--labeled", "32", "--steps", "1",
           "--warmup", "1", "--pretrain-steps", "10", "--bn-settle-steps", "10", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 1 and out["warmup"] == 1 and out["value"] > 0
    c = out["config"]
    assert c["unlabeled_images"] == 96 and c["unlabeled_images_per_gpu"] == 48 and c["parallelism"] == "dp2"
    assert c["global_batch"] == 2 * c["train_batch_per_gpu"] and c["process_group"].startswith("gloo")
    assert out["scaling"] in ("strong", "weak") and "other_configs" not in out     # (the extra shapes ride on the default N = 1 line only)


def test_isic_driver_eight_ranks_on_one_gpu(tmp_path):
    """`torch.distributed.run --nproc-per-node 8` of ISIC_2018/09_ISIC_2018_IM.py with eight ranks time-slicing
    ONE GPU over gloo -- shards of 3 unlabeled / 2 labelled / 1 validation file per rank.  The pseudo-label directories must equal the
    single-rank run's file by file (functions.py:2889: the mean IM size must survive the sharding), the candidates train
    data-parallel and rank 0 writes the reference's CSV plus the sidecar.  (The 1 -> 8 curve itself needs eight GPUs: unmeasured.)"""
    outs = _run_one_and_two_ranks(tmp_path, CONFIG, SETUP, os.path.join(ROOT, "ISIC_2018", "09_ISIC_2018_IM.py"), worlds=(1, 8))
    stem = "ISIC_2018_IM_1_n2_gen0_e0_d0_bi_True_bo_True"
    a, b = (outs[w] / "train_unlabeled_predictions" / "IM" / stem for w in (1, 8))
    _same_png_tree(a, b, ("im", "images", "masks"), {"images": 3})
    assert (outs[1] / "csv" / f"mean_im_size_{stem}.csv").read_text() == (outs[8] / "csv" / f"mean_im_size_{stem}.csv").read_text()
    rows = (outs[8] / "csv" / f"results_{stem}.csv").read_text().strip().splitlines()
    assert len(rows) == 3 and rows[0] == (outs[1] / "csv" / f"results_{stem}.csv").read_text().strip().splitlines()[0]
    meta = json.loads((outs[8] / "csv" / f"results_{stem}.meta.json").read_text())
    assert meta["data_parallel_ranks"] == 8 and meta["bn_momentum_rule"] == "reference"
    assert len(os.listdir(outs[8] / "val_predictions" / "IM" / (stem + "_0"))) == 8      # every prediction exactly once
    models = sorted(os.listdir(outs[8] / "models"))
    assert stem + "_topK_1.h5" in models and stem + "_topK_2.h5" in models


def test_bench_eight_ranks_full_set_on_one_gpu():
    """The driver's N = 8 command on the full ISIC-shaped set (2 335 unlabeled + 259 labelled: shards of 291 / 292 and 32 / 33), eight
    ranks time-slicing this box's one GPU over gloo: the sharded IM stage must sum to the single-rank one
    (sharding_check.equals_sum_over_ranks), every rank must run the same number of steps (the MIN all-reduce; otherwise the
    gradient all-reduces would not line up and the run would hang), and all ranks must leave cleanly."""
    import json
    import re
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {**os.environ, "IMK_BENCH_ONE_GPU": "1", "IMK_BENCH_BACKEND": "gloo"}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--images", "2335", "--labeled", "259", "--steps", "1",
           "--warmup", "0", "--pretrain-steps", "10", "--bn-settle-steps", "10", "--no-cpu-baseline", "--no-prof"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1700)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    c = out["config"]
    assert out["n_gpus"] == 8 and c["parallelism"] == "dp8" and c["unlabeled_images"] == 2335 and c["global_batch"] == 256
    assert c["unlabeled_images_per_gpu"] in (291, 292)
    assert out["sharding_check"]["equals_sum_over_ranks"] is True
    per_rank = {int(m.group(1)): (int(m.group(2)), int(m.group(3))) for m in re.finditer(r"\[bench rank (\d+)\] epoch_steps=(\d+) kept=(\d+)", r.stderr)}
    assert sorted(per_rank) == list(range(8)), r.stderr[-3000:]
    assert len({v[0] for v in per_rank.values()}) == 1 and per_rank[0][0] == c["epoch_steps"]       # one step count on every rank
    assert sum(v[1] for v in per_rank.values()) == c["kept"]


def test_isic_driver_candidates_side_by_side(tmp_path):
    """IM_PARALLEL_CANDIDATES=3: the generation's candidates train on three host threads with a stream each (VERDICT round 3, item
    7; the reference trains them one after the other, ISIC_2018/09_ISIC_2018_IM.py:90).  Every candidate must compute exactly
    what it computes alone: results CSV, mean-IM-size CSV, the surviving checkpoints and their prediction PNGs equal the
    sequential run's byte for byte."""
    outs = {}
    for par in (1, 3):
        work = tmp_path / f"p{par}"
        base = work / "data"
        work.mkdir()
        cfg = work / "config.ini"
        cfg.write_text(CONFIG.format(base=base))
        env = {**os.environ, "IM_CONFIG": str(cfg), "IM_RUNIDS": "1", "IM_NS": "2", "IM_GENS": "0", "IM_CANDIDATES": "0,1,2,3",
               "IM_PARALLEL_CANDIDATES": str(par), "IM_TIMING": "1"}
        subprocess.run([sys.executable, "-c", SETUP.format(root=ROOT)], env=env, check=True, cwd=work)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "ISIC_2018", "09_ISIC_2018_IM.py")], env=env, cwd=work,
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        assert ("3 side by side" in r.stdout) == (par > 1)      # IM_PARALLEL_CANDIDATES=1 is the reference's order
        outs[par] = base
    stem = "ISIC_2018_IM_1_n2_gen0_e0_d0_bi_True_bo_True"
    for name in (f"results_{stem}.csv", f"mean_im_size_{stem}.csv"):
        assert (outs[1] / "csv" / name).read_text() == (outs[3] / "csv" / name).read_text(), name
    from safetensors import safe_open
    for j in (1, 2):      # (the files' JSON headers may order their keys differently: compare the tensors)
        sd = []
        for p in (1, 3):
            with safe_open(str(outs[p] / "models" / f"{stem}_topK_{j}.h5"), framework="np") as f:
                sd.append({k: f.get_tensor(k) for k in f.keys()})
        assert sd[0].keys() == sd[1].keys() and all(np.array_equal(sd[0][k], sd[1][k]) for k in sd[0]), j
    for i in range(4):
        a, b = (outs[p] / "test_predictions" / "IM" / f"{stem}_{i}" for p in (1, 3))
        assert sorted(os.listdir(a)) == sorted(os.listdir(b)) and len(os.listdir(a)) == 8
        for n in os.listdir(a):
            assert (a / n).read_bytes() == (b / n).read_bytes(), (i, n)


def test_isic_driver_whole_candidates_per_rank(tmp_path):
    """IM_DP_MODE=candidates (the reference's five independent candidates, ISIC_2018/09_ISIC_2018_IM.py:90-135): inference
    + IM sharded over all ranks as always, then candidate i trained WHOLE on rank i mod N with the reference's batch 32 and step count
    and no collective, the rows gathered for ranking / rename / CSV.  2 and 8 ranks time-slicing this box's one GPU over gloo must
    reproduce the ONE-rank run byte for byte: results CSV, mean-IM-size CSV, the surviving checkpoints' tensors, every prediction PNG."""
    from safetensors import safe_open
    outs = _run_one_and_two_ranks(tmp_path, CONFIG, SETUP, os.path.join(ROOT, "ISIC_2018", "09_ISIC_2018_IM.py"), worlds=(1, 2, 8),
                                  extra_env={"IM_DP_MODE": "candidates", "IM_CANDIDATES": "0,1,2", "IM_PARALLEL_CANDIDATES": "1"})
    stem = "ISIC_2018_IM_1_n2_gen0_e0_d0_bi_True_bo_True"
    for w in (2, 8):
        for name in (f"results_{stem}.csv", f"mean_im_size_{stem}.csv"):
            assert (outs[1] / "csv" / name).read_text() == (outs[w] / "csv" / name).read_text(), (w, name)
        meta = json.loads((outs[w] / "csv" / f"results_{stem}.meta.json").read_text())
        assert meta["data_parallel_ranks"] == w and meta["dp_mode"] == "candidates"
        assert sorted(os.listdir(outs[1] / "models")) == sorted(os.listdir(outs[w] / "models"))
        kept = sorted(n for n in os.listdir(outs[1] / "models") if n.startswith(stem + "_topK_"))
        assert len(kept) >= 2                                   # TOP_Ks of the toy config
        for name in kept:
            sd = []
            for q in (1, w):
                with safe_open(str(outs[q] / "models" / name), framework="np") as f:
                    sd.append({k: f.get_tensor(k) for k in f.keys()})
            assert sd[0].keys() == sd[1].keys() and all(np.array_equal(sd[0][k], sd[1][k]) for k in sd[0]), (w, name)
        for i in range(3):
            for split in ("val", "test", "train_unlabeled"):
                a, b = (outs[q] / f"{split}_predictions" / "IM" / f"{stem}_{i}" for q in (1, w))
                assert sorted(os.listdir(a)) == sorted(os.listdir(b)) and len(os.listdir(a)) > 0
                for n in os.listdir(a):
                    assert (a / n).read_bytes() == (b / n).read_bytes(), (w, i, split, n)


@pytest.mark.parametrize("which", ["hela_im", "suim_im", "isic_subset", "isic_impp_evalnets"])
def test_other_drivers_candidates_side_by_side(tmp_path, which):
    """the default on one rank (three candidates side by side: im_driver.train_candidates) against IM_PARALLEL_CANDIDATES=1 (the
    reference's order) for the drivers whose runs are seeded end to end: HeLa/09_HeLa_IM.py (host geometry + benchmark_hela on the
    candidates' threads), SUIM/10_SUIM_IM.py (colour masks on the writer pool) and ISIC_2018/03_ISIC_2018_subset.py (ten
    candidates): every CSV and every surviving checkpoint equal"""
    config, setup, script, cands = {
        "hela_im": (HELA_CONFIG, HELA_SETUP, os.path.join("HeLa", "09_HeLa_IM.py"), "0,1,2"),
        "suim_im": (MULTI_CONFIG, MULTI_SETUP, os.path.join("SUIM", "10_SUIM_IM.py"), "0,1,2"),
        "isic_subset": (CONFIG, SETUP, os.path.join("ISIC_2018", "03_ISIC_2018_subset.py"), "0,1,2,3"),
        # IM++: the EvalNet stage is seeded end to end (training data, 3 EvalNet candidates); the U-Net stage behind it draws its
        # augmentations unseeded, so only the EvalNet CSV and checkpoints are compared
        "isic_impp_evalnets": (CONFIG.replace("TOP_Ks = 2\n", "TOP_Ks = 2\nNUM_EPOCHS_EVALNET = 2\nBATCH_SIZE_EVALNET = 8\nNUM_LOOPS_TRAIN = 2\n"
                                                                 "NUM_LOOPS_VAL = 1\n")
                               + "FREE_ROTATION = True\nALPHA_EVALNET = 0.5\nMIN_THRESHOLD = 0.3\nMAX_THRESHOLD = 0.8\n",
                               SETUP, os.path.join("ISIC_2018", "12_ISIC_2018_IM++.py"), "0,1"),
    }[which]
    only = (lambda n: "evalnet" in n) if which == "isic_impp_evalnets" else (lambda n: True)
    outs = {}
    for par in ("1", None):
        work = tmp_path / f"p{par}"
        base = work / "data"
        work.mkdir()
        cfg = work / "config.ini"
        cfg.write_text(config.format(base=base))
        env = {**os.environ, "IM_CONFIG": str(cfg), "IM_RUNIDS": "1", "IM_NS": "2", "IM_GENS": "0", "IM_CANDIDATES": cands,
               "IM_EVALNET_CANDIDATES": "0,1,2"}
        env.pop("IM_PARALLEL_CANDIDATES", None)
        if par:
            env["IM_PARALLEL_CANDIDATES"] = par
        subprocess.run([sys.executable, "-c", setup.format(root=ROOT)], env=env, check=True, cwd=work)
        if which == "isic_subset":      # the set-up script's stand-in ensemble would collide with the subset driver's own names
            for f in os.listdir(base / "models"):
                os.remove(base / "models" / f)
        r = subprocess.run([sys.executable, os.path.join(ROOT, script)], env=env, cwd=work, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        outs[par] = base
    csvs = sorted(n for n in os.listdir(outs["1"] / "csv") if only(n))
    assert csvs and csvs == sorted(n for n in os.listdir(outs[None] / "csv") if only(n))
    for name in csvs:
        assert (outs["1"] / "csv" / name).read_text() == (outs[None] / "csv" / name).read_text(), name
    from safetensors import safe_open
    tops = sorted(n for n in os.listdir(outs["1"] / "models")
                  if "_topK_" in n and {"isic_subset": True, "isic_impp_evalnets": "evalnet" in n}.get(which, "_IM_" in n))
    assert tops and all((outs[None] / "models" / n).exists() for n in tops)
    for n in tops:
        sd = []
        for p in ("1", None):
            with safe_open(str(outs[p] / "models" / n), framework="np") as f:
                sd.append({k: f.get_tensor(k) for k in f.keys()})
        assert sd[0].keys() == sd[1].keys() and all(np.array_equal(sd[0][k], sd[1][k]) for k in sd[0]), n


@pytest.mark.parametrize("which", ["hela_im", "suim_im", "isic_subset"])
def test_other_drivers_whole_candidates_per_rank(tmp_path, which):
    """IM_DP_MODE=candidates behind the HeLa and SUIM IM drivers and the ISIC subset driver (im_driver.train_candidates serves all
    of them; subset_driver / impp_driver size their epochs with im_driver.epoch_steps): two ranks time-slicing one GPU over gloo
    reproduce the one-rank run -- every CSV and every surviving checkpoint's tensors equal."""
    from safetensors import safe_open
    config, setup, script, cands = {
        "hela_im": (HELA_CONFIG, HELA_SETUP, os.path.join("HeLa", "09_HeLa_IM.py"), "0,1,2"),
        "suim_im": (MULTI_CONFIG, MULTI_SETUP, os.path.join("SUIM", "10_SUIM_IM.py"), "0,1,2"),
        "isic_subset": (CONFIG, SETUP, os.path.join("ISIC_2018", "03_ISIC_2018_subset.py"), "0,1,2"),
    }[which]
    import socket
    outs = {}
    for world in (1, 2):
        work = tmp_path / f"w{world}"
        base = work / "data"
        work.mkdir()
        cfg = work / "config.ini"
        cfg.write_text(config.format(base=base))
        env = {**os.environ, "IM_CONFIG": str(cfg), "IM_RUNIDS": "1", "IM_NS": "2", "IM_GENS": "0", "IM_CANDIDATES": cands,
               "IM_PARALLEL_CANDIDATES": "1", "IM_DP_MODE": "candidates"}
        subprocess.run([sys.executable, "-c", setup.format(root=ROOT)], env=env, check=True, cwd=work)
        if which == "isic_subset":
            for f in os.listdir(base / "models"):
                os.remove(base / "models" / f)
        if world == 1:
            cmd = [sys.executable, os.path.join(ROOT, script)]
        else:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            env.update(IMK_DIST_BACKEND="gloo", IMK_ONE_GPU="1")
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                   "--master-port", str(port), os.path.join(ROOT, script)]
        r = subprocess.run(cmd, env=env, cwd=work, capture_output=True, text=True, timeout=1500)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        outs[world] = base
    csvs = sorted(n for n in os.listdir(outs[1] / "csv") if n.endswith(".csv"))
    assert csvs and csvs == sorted(n for n in os.listdir(outs[2] / "csv") if n.endswith(".csv"))
    for name in csvs:
        assert (outs[1] / "csv" / name).read_text() == (outs[2] / "csv" / name).read_text(), name
    tops = sorted(n for n in os.listdir(outs[1] / "models") if "_topK_" in n and (which == "isic_subset" or "_IM_" in n))
    assert tops and all((outs[2] / "models" / n).exists() for n in tops)
    for n in tops:
        sd = []
        for w in (1, 2):
            with safe_open(str(outs[w] / "models" / n), framework="np") as f:
                sd.append({k: f.get_tensor(k) for k in f.keys()})
        assert sd[0].keys() == sd[1].keys() and all(np.array_equal(sd[0][k], sd[1][k]) for k in sd[0]), n
