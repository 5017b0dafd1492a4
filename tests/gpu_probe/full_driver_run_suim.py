"""GPU: the SUIM IM driver (SUIM/10_SUIM_IM.py) at the dataset's real size on synthetic images through the PNG directories: 2 468
unlabeled + 274 labelled + 152 validation + 110 test images of 256 x 256 x 3, 9 outputs, alpha 1, n = 2, generation 0,
IM_CANDIDATES x NUM_EPOCHS as given (default 2 x 10: this is a host-side profile, not a training run).  Prints the wall time of every
stage (IM_TIMING) and the driver's top host functions by cumulative time (cProfile).
Usage: python tests/gpu_probe/full_driver_run_suim.py [workdir]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
work = sys.argv[1] if len(sys.argv) > 1 else "/tmp/im_full_run_suim"
os.makedirs(work, exist_ok=True)
cfg = os.path.join(work, "config.ini")
text = open(os.path.join(ROOT, "config.ini")).read().replace("./data/SUIM/", os.path.join(work, "data") + "/")
text = text.replace("NUM_EPOCHS = 50", "NUM_EPOCHS = " + os.environ.get("EPOCHS", "10"))
text = text.replace("NUM_EPOCHS_EVALNET = 50", "NUM_EPOCHS_EVALNET = " + os.environ.get("EPOCHS_EVALNET", "10"))
open(cfg, "w").write(text)
DRIVER = os.environ.get("DRIVER", "10_SUIM_IM.py")      # e.g. DRIVER=12_HeLa_IM++.py / 13_SUIM_IM++.py: the IM++ driver on the same data
env = {**os.environ, "IM_CONFIG": cfg, "IM_RUNIDS": "1", "IM_NS": "2", "IM_GENS": os.environ.get("IM_GENS", "0"), "IM_TIMING": "1", "IM_CANDIDATES": os.environ.get("IM_CANDIDATES", "0,1"),
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
    def one(i):
        r = np.random.default_rng(seed * 100003 + i)
        f = sum(np.cos(r.uniform(0.01, 0.05) * xx + r.uniform(0.01, 0.05) * yy + r.uniform(0, 6)) for _ in range(4))
        cls = (1 + np.clip((f + 4) / 8 * 7, 0, 6)).astype(np.uint8)                 # classes 1 .. 7 from the smooth field
        cy, cx, a, b = r.uniform(80, 176), r.uniform(80, 176), r.uniform(25, 70), r.uniform(25, 70)
        cls[((yy - cy) / a) ** 2 + ((xx - cx) / b) ** 2 < 1] = 8
        img = np.stack([40 + 25 * cls, 220 - 20 * cls, 90 + 10 * ((cls * 5) % 9)], -1) + r.uniform(-8, 8, (256, 256, 3))
        F.write_png(os.path.join(d_img, f"s_{{i:05d}}.png"), img.clip(0, 255).astype(np.uint8))
        F.write_png(os.path.join(d_mask, f"s_{{i:05d}}.png"), cls)
    with ThreadPoolExecutor(16) as pool:
        list(pool.map(one, range(n)))
sample(274, paths.SUIM_TRAIN_LABELED_IMAGES_DIR, paths.SUIM_TRAIN_LABELED_MASKS_DIR, 1)
sample(2468, paths.SUIM_TRAIN_UNLABELED_IMAGES_DIR, paths.SUIM_TRAIN_UNLABELED_MASKS_DIR, 2)
sample(152, paths.SUIM_VAL_IMAGES_DIR, paths.SUIM_VAL_MASKS_DIR, 3)
sample(110, paths.SUIM_TEST_IMAGES_DIR, paths.SUIM_TEST_MASKS_DIR, 4)
print(f"[timing] synthetic dataset written: {{time.perf_counter() - t0:.2f}} s", flush=True)
t0 = time.perf_counter()
os.makedirs(paths.SUIM_MODEL_DIR, exist_ok=True)
d = paths.SUIM_TRAIN_LABELED_IMAGES_DIR
names = sorted(os.listdir(d))
x = torch.from_numpy(np.stack([F.read_png(os.path.join(d, n), 3) for n in names])).cuda()
y = torch.from_numpy(np.stack([F.read_png(os.path.join(paths.SUIM_TRAIN_LABELED_MASKS_DIR, n), 1)[..., 0] for n in names])).cuda()
g = torch.Generator(device="cuda").manual_seed(0)
for j in (1, 2):
    m = get_unet(256, 256, 3, 9, 1.0, "relu", "softmax", seed=j)
    for it in range(900):
        idx = torch.randint(0, len(names), (32,), device="cuda", generator=g)
        m.train_step(x[idx].contiguous(), y[idx].contiguous(), 1, 3e-3 if it < 500 else 0.0, 1e-4 if it < 500 else 0.0)
    m.repack()
    F.save_model(m, os.path.join(paths.SUIM_MODEL_DIR, f"SUIM_subset_1_topK_{{j}}.h5"))
print(f"[timing] generation-0 ensemble (2 models x 900 steps): {{time.perf_counter() - t0:.2f}} s", flush=True)
"""
t0 = time.perf_counter()
subprocess.run([sys.executable, "-c", SETUP], env=env, check=True, cwd=work)
t1 = time.perf_counter()
prof = os.path.join(work, "driver.prof")
subprocess.run([sys.executable, "-m", "cProfile", "-o", prof, os.path.join(ROOT, "SUIM", DRIVER)], env=env, check=True, cwd=work)
t2 = time.perf_counter()
print(f"[timing] setup {t1 - t0:.1f} s; SUIM/{DRIVER} (n = 2, generation 0, candidates {env['IM_CANDIDATES']} x {os.environ.get('EPOCHS', '10')} epochs): {t2 - t1:.1f} s")
import pstats
st = pstats.Stats(prof)
st.sort_stats("cumulative").print_stats(28)
