import os, sys, time, io
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from PIL import Image
rng = np.random.default_rng(0)
yy, xx = np.mgrid[0:256, 0:256]
imgs = [(150 + 30 * np.sin(xx / 17.0 + i)[..., None] + rng.integers(-8, 8, (256, 256, 3))).clip(0, 255).astype(np.uint8) for i in range(512)]
def enc(a):
    b = io.BytesIO(); Image.fromarray(a).save(b, format="PNG", compress_level=1); return b.getvalue()
def dec(b): return np.asarray(Image.open(io.BytesIO(b)))
print("cpus", os.cpu_count())
for th in (8, 16, 32, 64, 128):
    with ThreadPoolExecutor(max_workers=th) as pool:
        t0 = time.perf_counter(); blobs = list(pool.map(enc, imgs)); t1 = time.perf_counter()
        back = list(pool.map(dec, blobs)); t2 = time.perf_counter()
    print(f"threads {th:4d}: encode {len(imgs) / (t1 - t0):8.1f} images/s   decode {len(imgs) / (t2 - t1):8.1f} images/s")
