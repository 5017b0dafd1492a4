"""CPU: encode rate and file size of libimk's PNG encoder on bench-like images (no GPU call).  IMK_LIB_PATH picks the build:
    python tests/gpu_probe/png_encode_ab.py [threads]"""
import ctypes, os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = ctypes.CDLL(os.environ.get("IMK_LIB_PATH", os.path.join(ROOT, "inconsistencymasks_amd", "libimk.so")))
lib.imk_png_encode.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
lib.imk_png_decode.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
threads = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = np.random.default_rng(0)
H = W = 256
yy, xx = np.mgrid[0:H, 0:W]
def image(k):       # a smooth field + an ellipse + sensor-like noise (what bench.synth_images draws), and its IM-blocked variant
    base = (120 + 50 * np.sin(xx / (9.0 + k % 7)) * np.cos(yy / (13.0 + k % 5)))[..., None] + np.array([10, 0, -10])
    img = base + 40 * (((xx - 128 - k % 30) / 60.0) ** 2 + ((yy - 120) / 40.0) ** 2 < 1)[..., None] + rng.integers(-6, 7, (H, W, 3))
    img = np.clip(img, 0, 255).astype(np.uint8)
    if k % 2:
        img[(rng.random((H, W)) > 0.995)] = 0
    return np.ascontiguousarray(img)
imgs = [image(k) for k in range(256)]
masks = [np.ascontiguousarray(((((xx - 128) / (40.0 + k % 20)) ** 2 + ((yy - 128) / 50.0) ** 2 < 1) * 255).astype(np.uint8)[..., None]) for k in range(256)]
def enc(a):
    h, w, c = a.shape
    buf = np.empty(a.nbytes + h + a.nbytes // 500 + 4096, np.uint8)
    n = ctypes.c_int64()
    assert lib.imk_png_encode(a.ctypes.data, h, w, c, 1, buf.ctypes.data, buf.nbytes, ctypes.byref(n)) == 0
    return buf[:n.value].copy()
def dec(b, shape):
    out = np.empty(shape, np.uint8)
    assert lib.imk_png_decode(b.ctypes.data, b.size, shape[2], out.ctypes.data, out.nbytes, None, None) == 0
    return out
for name, data in (("rgb 256x256x3", imgs), ("mask 256x256x1", masks)):
    with ThreadPoolExecutor(threads) as pool:
        list(pool.map(enc, data[:16]))
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); blobs = list(pool.map(enc, data)); best = min(best, time.perf_counter() - t0)
        t0 = time.perf_counter(); back = list(pool.map(lambda bs: dec(bs[0], bs[1].shape), zip(blobs, data))); td = time.perf_counter() - t0
    assert all(np.array_equal(a, b) for a, b in zip(data, back))
    print(f"{name}: encode {len(data) / best:8.0f} images/s on {threads} threads, {sum(b.size for b in blobs) / len(blobs) / 1024:6.1f} KB per file; decode {len(data) / td:8.0f} images/s")
