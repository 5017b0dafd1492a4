"""GPU: the ISIC IM++ driver (ISIC_2018/12_ISIC_2018_IM++.py) at the dataset's real size on synthetic images through the PNG directories:
EvalNet training data (10 + 3 loops over 259 labelled / 100 validation pairs), IM_EVALNET_CANDIDATES x NUM_EPOCHS_EVALNET EvalNets, pseudo-labels,
EvalNet-weighted augmentation of 2 335 unlabeled pairs, IM_CANDIDATES x NUM_EPOCHS U-Nets (defaults 2 x 10 and 2 x 10: a host-side profile).
cProfile of the driver by cumulative time.  Usage: python tests/gpu_probe/full_driver_run_impp.py [workdir]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
work = sys.argv[1] if len(sys.argv) > 1 else "/tmp/im_full_run_impp"
os.makedirs(work, exist_ok=True)
cfg = os.path.join(work, "config.ini")
text = open(os.path.join(ROOT, "config.ini")).read().replace("./data/ISIC_2018/", os.path.join(work, "data") + "/")
text = text.replace("NUM_EPOCHS = 50", "NUM_EPOCHS = " + os.environ.get("EPOCHS", "10")).replace("NUM_EPOCHS_EVALNET = 50", "NUM_EPOCHS_EVALNET = " + os.environ.get("EPOCHS_EVALNET", "10"))
open(cfg, "w").write(text)
DRIVER = os.environ.get("DRIVER", "12_ISIC_2018_IM++.py")      # DRIVER=11_ISIC_2018_IM+.py / 13_ISIC_2018_aug_IM+.py / 14_ISIC_2018_aug_IM++.py
env = {**os.environ, "IM_CONFIG": cfg, "IM_RUNIDS": "1", "IM_NS": "2", "IM_GENS": "0", "IM_TIMING": "1", "IM_CANDIDATES": os.environ.get("IM_CANDIDATES", "0,1"),
       "IM_EVALNET_CANDIDATES": os.environ.get("IM_EVALNET_CANDIDATES", "0,1")}
SETUP = f"""
import os, sys, time
import numpy as np
sys.path.insert(0, {ROOT!r})
import torch
from concurrent.futures import ThreadPoolExecutor
from inconsistencymasks_amd import functions as F, paths
from inconsistencymasks_amd.unet import get_unet
t0 = time.perf_counter()
yy, xx = np.mgrid[0:256, 0:256].astype(np.float32)
def sample(n, d_img, d_mask, seed):
    os.makedirs(d_img, exist_ok=True); os.makedirs(d_mask, exist_ok=True)
    rng = np.random.default_rng(seed)
    def one(i):
        r = np.random.default_rng(seed * 100003 + i)
        f = sum(r.uniform(10, 30) * np.cos(r.uniform(0.01, 0.05) * xx + r.uniform(0.01, 0.05) * yy + r.uniform(0, 6)) for _ in range(4))
        cy, cx, a, b = r.uniform(80, 176), r.uniform(80, 176), r.uniform(25, 70), r.uniform(25, 70)
        ell = ((yy - cy) / a) ** 2 + ((xx - cx) / b) ** 2 < 1
        img = (150 + f)[..., None] + r.uniform(-8, 8, (256, 256, 3)) - ell[..., None] * r.uniform(40, 90)
        F.write_png(os.path.join(d_img, f"ISIC_{{i:07d}}.png"), img.clip(0, 255).astype(np.uint8))
        F.write_png(os.path.join(d_mask, f"ISIC_{{i:07d}}.png"), (ell * 255).astype(np.uint8))
    with ThreadPoolExecutor(8) as pool:
        list(pool.map(one, range(n)))
sample(259, paths.ISIC_2018_TRAIN_LABELED_IMAGES_DIR, paths.ISIC_2018_TRAIN_LABELED_MASKS_DIR, 1)
sample(2335, paths.ISIC_2018_TRAIN_UNLABELED_IMAGES_DIR, paths.ISIC_2018_TRAIN_UNLABELED_MASKS_DIR, 2)
sample(100, paths.ISIC_2018_VAL_IMAGES_DIR, paths.ISIC_2018_VAL_MASKS_DIR, 3)
sample(1000, paths.ISIC_2018_TEST_IMAGES_DIR, paths.ISIC_2018_TEST_MASKS_DIR, 4)
print(f"[timing] synthetic dataset written (3 694 image / mask pairs): {{time.perf_counter() - t0:.2f}} s", flush=True)
t0 = time.perf_counter()
os.makedirs(paths.ISIC_2018_MODEL_DIR, exist_ok=True)
d = paths.ISIC_2018_TRAIN_LABELED_IMAGES_DIR
names = sorted(os.listdir(d))
x = torch.from_numpy(np.stack([F.read_png(os.path.join(d, n), 3) for n in names])).cuda()
y = torch.from_numpy(np.stack([F.read_png(os.path.join(paths.ISIC_2018_TRAIN_LABELED_MASKS_DIR, n), 1) // 255 for n in names])).cuda()
g = torch.Generator(device="cuda").manual_seed(0)
for j in (1, 2):      # stand-in for 03_ISIC_2018_subset.py's product: the generation-0 ensemble
    m = get_unet(256, 256, 3, 1, 0.5, "relu", "sigmoid", seed=j)
    for it in range(1200):
        idx = torch.randint(0, len(names), (32,), device="cuda", generator=g)
        m.train_step(x[idx].contiguous(), y[idx].contiguous(), 0, 3e-3 if it < 600 else 0.0, 1e-4 if it < 600 else 0.0)
    m.repack()
    F.save_model(m, os.path.join(paths.ISIC_2018_MODEL_DIR, f"ISIC_2018_subset_1_topK_{{j}}.h5"))
print(f"[timing] generation-0 ensemble (2 models x 1 200 steps): {{time.perf_counter() - t0:.2f}} s", flush=True)
"""
t0 = time.perf_counter()
subprocess.run([sys.executable, "-c", SETUP], env=env, check=True, cwd=work)
t1 = time.perf_counter()
prof = os.path.join(work, "driver.prof")
subprocess.run([sys.executable, "-m", "cProfile", "-o", prof, os.path.join(ROOT, "ISIC_2018", DRIVER)], env=env, check=True, cwd=work)
t2 = time.perf_counter()
print(f"[timing] setup {t1 - t0:.1f} s; ISIC_2018/{DRIVER} (n = 2, generation 0, EvalNets {env['IM_EVALNET_CANDIDATES']}, candidates {env['IM_CANDIDATES']}): {t2 - t1:.1f} s")
import pstats
pstats.Stats(prof).sort_stats("cumulative").print_stats(45)
