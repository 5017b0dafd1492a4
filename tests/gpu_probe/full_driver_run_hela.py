"""GPU: the HeLa IM driver (HeLa/09_HeLa_IM.py) at a realistic size on synthetic images through the PNG directories: 1 800 unlabeled +
200 labelled + 100 validation + 200 test crops of 256 x 256 x 1 with ~12 cells each, 3 sigmoid maps, alpha 1, n = 2, generation 0,
IM_CANDIDATES x EPOCHS (default 2 x 10: a host-side profile).  Stage times (IM_TIMING) + the driver's top host functions (cProfile).
Usage: python tests/gpu_probe/full_driver_run_hela.py [workdir]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
work = sys.argv[1] if len(sys.argv) > 1 else "/tmp/im_full_run_hela"
os.makedirs(work, exist_ok=True)
cfg = os.path.join(work, "config.ini")
text = open(os.path.join(ROOT, "config.ini")).read().replace("./data/HeLa/", os.path.join(work, "data") + "/")
text = text.replace("NUM_EPOCHS = 50", "NUM_EPOCHS = " + os.environ.get("EPOCHS", "10"))
text = text.replace("NUM_EPOCHS_EVALNET = 50", "NUM_EPOCHS_EVALNET = " + os.environ.get("EPOCHS_EVALNET", "10"))
open(cfg, "w").write(text)
DRIVER = os.environ.get("DRIVER", "09_HeLa_IM.py")      # e.g. DRIVER=12_HeLa_IM++.py / 13_SUIM_IM++.py: the IM++ driver on the same data
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
yy, xx = np.mgrid[0:256, 0:256]
def sample(n, d, tag, seed):
    for k in ("brightfield", "alive", "dead", "mod_position"):
        os.makedirs(os.path.join(d, k), exist_ok=True)
    def one(i):
        r = np.random.default_rng(seed * 100003 + i)
        bf = np.full((256, 256), 120, np.int64) + r.integers(-8, 8, (256, 256))
        alive = np.zeros((256, 256), np.uint8); dead = np.zeros((256, 256), np.uint8); pos = np.zeros((256, 256), np.uint8)
        for _ in range(12):
            cy, cx, rad, is_dead = int(r.integers(20, 236)), int(r.integers(20, 236)), int(r.integers(8, 16)), int(r.integers(0, 2))
            cell = (yy - cy) ** 2 + (xx - cx) ** 2 < rad * rad
            bf[cell] += 70 if is_dead else -60
            (dead if is_dead else alive)[cell] = 255
            pos[(yy - cy) ** 2 + (xx - cx) ** 2 < 16] = 255
        name = f"{{tag}}_{{i:05d}}.png"
        F.write_png(os.path.join(d, "brightfield", name), bf.clip(0, 255).astype(np.uint8))
        F.write_png(os.path.join(d, "alive", name), alive); F.write_png(os.path.join(d, "dead", name), dead)
        F.write_png(os.path.join(d, "mod_position", name), pos)
    with ThreadPoolExecutor(16) as pool:
        list(pool.map(one, range(n)))
sample(200, paths.HELA_TRAIN_LABELED_DIR, "lab", 1); sample(1800, paths.HELA_TRAIN_UNLABELED_DIR, "unl", 2)
sample(100, paths.HELA_VAL_DIR, "val", 3); sample(200, paths.HELA_TEST_DIR, "tst", 4)
print(f"[timing] synthetic dataset written: {{time.perf_counter() - t0:.2f}} s", flush=True)
t0 = time.perf_counter()
os.makedirs(paths.HELA_MODEL_DIR, exist_ok=True)
bfd = os.path.join(paths.HELA_TRAIN_LABELED_DIR, "brightfield")
items = [F.parse_image_hela(os.path.join(bfd, n), 1) for n in sorted(os.listdir(bfd))]
x = torch.from_numpy(np.stack([it[0] for it in items])).cuda()
y = torch.from_numpy(np.stack([it[1] for it in items])).cuda()
g = torch.Generator(device="cuda").manual_seed(0)
for j in (1, 2):
    m = get_unet(256, 256, 1, 3, 1.0, "relu", "sigmoid", seed=j)
    for it in range(900):
        idx = torch.randint(0, x.shape[0], (32,), device="cuda", generator=g)
        m.train_step(x[idx].contiguous(), y[idx].contiguous(), 0, 3e-3 if it < 500 else 0.0, 1e-4 if it < 500 else 0.0)
    m.repack()
    F.save_model(m, os.path.join(paths.HELA_MODEL_DIR, f"HELA_subset_1_topK_{{j}}.h5"))
print(f"[timing] generation-0 ensemble (2 models x 900 steps): {{time.perf_counter() - t0:.2f}} s", flush=True)
"""
t0 = time.perf_counter()
subprocess.run([sys.executable, "-c", SETUP], env=env, check=True, cwd=work)
t1 = time.perf_counter()
prof = os.path.join(work, "driver.prof")
subprocess.run([sys.executable, "-m", "cProfile", "-o", prof, os.path.join(ROOT, "HeLa", DRIVER)], env=env, check=True, cwd=work)
t2 = time.perf_counter()
print(f"[timing] setup {t1 - t0:.1f} s; HeLa/{DRIVER} (n = 2, generation 0, candidates {env['IM_CANDIDATES']} x {os.environ.get('EPOCHS', '10')} epochs): {t2 - t1:.1f} s")
import pstats
pstats.Stats(prof).sort_stats("cumulative").print_stats(32)
