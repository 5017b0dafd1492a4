"""Host-side mirror of the hot-path part of the reference's functions.py, on top of libimk.so.

Same function names, positional arguments, return values, output directory layout and file names as the
reference (file:line cited per function), so that the per-dataset `*_IM.py` drivers run unchanged from
the user's point of view.  What differs is underneath: images are processed in batches on the GPU
(ensemble forward + fused IM kernel, training step kernels), PNG I/O uses Pillow on a host thread pool
(OpenCV is not available), and model files are safetensors under the reference's `.h5` names by default -- `load_model` also takes
the reference's own Keras HDF5 checkpoints directly (keras_h5.py / h5lite.py: a pure-numpy HDF5 reader), and
IMK_MODEL_FORMAT=keras_h5 makes `save_model` write Keras `save_weights` HDF5 files instead.

Multi-GPU: if torch.distributed is initialised, every `create_pseudo_labels_im_*` call shards the
sorted file list in contiguous blocks over the ranks (no data collective; one 3-number all-reduce for the
mean IM size) and `train_*` averages gradients with one all-reduce per step (RCCL over xGMI).
"""
import configparser
import glob
import os
import re
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch
from PIL import Image

from . import evaluate as _ev
from . import im as _im
from ._lib import check, lib
from .unet import UNet, _stream

_CFG_PATH = os.environ.get("IM_CONFIG", "config.ini")   # the reference reads ./config.ini (functions.py:23-33)
config = configparser.ConfigParser()
config.read(_CFG_PATH)
_D = config["DEFAULT"] if "DEFAULT" in config else {}
SEED = int(_D.get("SEED", 42))
BATCH_SIZE = int(_D.get("BATCH_SIZE", 32))
LR = float(_D.get("LR", 0.003))
WD = float(_D.get("WD", 1e-4))
THRESHOLD = float(_D.get("THRESHOLD", 0.5))
NUM_EPOCHS = int(_D.get("NUM_EPOCHS", 50))
NUM_EPOCHS_CS = int(_D.get("NUM_EPOCHS_CS", 100))

INFER_BATCH = int(os.environ.get("IMK_INFER_BATCH", 256))      # chunk of the augmentation / EvalNet loops (not the ensemble's)


def infer_batch_size(alpha, override=None):
    """Images per ensemble call of the pseudo-label writers AND of bench.py (one rule, here): 584 at alpha <= 0.5 (ISIC's 2335 images =
    4 calls per model; measured 256: 14.8 ms, 584: 14.5, 1168: 14.4 per 2335-image stage), 128 for the wider nets (their workspace is
    4-16x larger per image and the launch chain is already amortised).  IMK_INFER_BATCH / `override` replace it.  The reference's loop
    being batched: functions.py:2844-2854 (one image per predict call)."""
    if override:
        return int(override)
    env = os.environ.get("IMK_INFER_BATCH")
    if env:
        return int(env)
    return 584 if float(alpha) <= 0.5 else 128


def infer_batches(n, batch):
    """[(begin, end)] over n images in calls of `batch`; a last call under a quarter of that is spread over the others instead (at 8
    ranks an ISIC shard is 292 images: one call, not 256 + 36 -- the deep levels of a forward cost the same for 36 images as for 256)"""
    if n <= 0:
        return []
    k, b = -(-n // batch), batch
    if k > 1 and n - (k - 1) * b < b // 4:
        k -= 1
        b = -(-n // k)
    return [(i, min(i + b, n)) for i in range(0, n, b)]


def _models_alpha(models):
    m = models[0] if isinstance(models, (list, tuple)) else models
    return getattr(getattr(m, "plan", None), "alpha", 1.0)
# PNG decode / encode threads.  Pillow on a thread pool (rounds 1-4) stopped scaling at ~32 threads on the 256-CPU MI355X host
# (256 x 256 x 3 images: 8 / 16 / 32 / 64 / 128 threads -> 1.4 / 2.7 / 3.5 / 3.3 / 3.1 k encoded images/s: the interpreter lock);
# the native codec (csrc/imk_png.cpp) holds no lock, so the pool is as wide as half the host's CPUs, at most 64.
_IO_THREADS = int(os.environ.get("IMK_IO_THREADS", min(64, max(4, (os.cpu_count() or 8) // 2))))      # native codec: scales with cores


# ---------------------------------------------------------------------------------------------------
# distributed helpers
# ---------------------------------------------------------------------------------------------------
_LOCAL = __import__("threading").local()


class local_rank_scope:
    """Inside this block (on the entering THREAD) the package behaves as a one-rank run although a process group exists: file lists
    are not sharded, no gradient all-reduce, no barrier, evaluations see every file.  im_driver's IM_DP_MODE=candidates trains whole
    candidates per rank this way (independent models, no collective; ISIC_2018/09_ISIC_2018_IM.py:90)."""

    def __enter__(self):
        _LOCAL.depth = getattr(_LOCAL, "depth", 0) + 1
        return self

    def __exit__(self, *exc):
        _LOCAL.depth -= 1
        return False


def _dist():
    if getattr(_LOCAL, "depth", 0):
        return None
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


def init_distributed():
    """One process per GPU under torch.distributed.run: pick the rank's device and join the process group (RCCL, i.e. the
    "nccl" backend).  IMK_DIST_BACKEND=gloo + IMK_ONE_GPU=1 run several ranks on ONE GPU (functional testing of the
    multi-rank path on a single-GPU box: RCCL refuses two ranks on one device)."""
    import torch.distributed as dist
    if int(os.environ.get("WORLD_SIZE", 1)) <= 1 or dist.is_initialized():
        return
    local = 0 if os.environ.get("IMK_ONE_GPU") == "1" else int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(os.environ.get("IMK_DIST_BACKEND", "nccl"))


def _rank_world():
    d = _dist()
    return (d.get_rank(), d.get_world_size()) if d else (0, 1)


def shard_list(items, rank=None, world=None):
    """Contiguous block of the SORTED list for this rank."""
    if rank is None:
        rank, world = _rank_world()
    items = sorted(items)
    n = len(items)
    lo, hi = (n * rank) // world, (n * (rank + 1)) // world
    return items[lo:hi]


# ---------------------------------------------------------------------------------------------------
# PNG I/O (host)
# ---------------------------------------------------------------------------------------------------
# PNG files go through libimk's own host codec (csrc/imk_png.cpp: zlib + the PNG container, one file per call, no interpreter lock
# held) -- Pillow on a thread pool stopped scaling at 2.7 k encoded images/s, an eighth of what the GPU stages produce.  Pillow
# stays the decoder of what the native one declines (16-bit, 1/2/4-bit, interlaced files) and can be forced for both directions
# with IMK_PNG=pillow.
_NATIVE_PNG = os.environ.get("IMK_PNG", "native").lower() != "pillow"


def _read_png_pillow(path, channels):
    with Image.open(path) as im:
        im = im.convert("RGB" if channels == 3 else "L")
        a = np.asarray(im, dtype=np.uint8)
    return a if a.ndim == 3 else a[..., None]


def read_png(path, channels):
    """uint8 [H,W,C] in RGB order (C=3) or [H,W,1] (C=1).  The reference reads BGR with cv2 and converts
    to RGB for the net (functions.py:2846-2848); on-disk channel order is preserved either way."""
    if _NATIVE_PNG:
        import ctypes
        from ._lib import lib
        h, w = ctypes.c_int(), ctypes.c_int()
        bpath = os.fsencode(path)
        if lib.imk_png_info(bpath, ctypes.byref(h), ctypes.byref(w), None, None) == 0:
            out = np.empty((h.value, w.value, 3 if channels == 3 else 1), dtype=np.uint8)
            if lib.imk_png_read_file(bpath, 3 if channels == 3 else 1, out.ctypes.data, out.nbytes, None, None) == 0:
                return out
    return _read_png_pillow(path, channels)


def _read_png_into(path, channels, row):
    """decode `path` into `row` (uint8 [H,W,C], C-contiguous): one open, no intermediate array when the native codec takes the file"""
    if _NATIVE_PNG:
        import ctypes
        from ._lib import lib
        h, w = ctypes.c_int(), ctypes.c_int()
        if lib.imk_png_read_file(os.fsencode(path), 3 if channels == 3 else 1, row.ctypes.data, row.nbytes, ctypes.byref(h),
                                 ctypes.byref(w)) == 0 and (h.value, w.value) == row.shape[:2]:
            return
    row[...] = read_png(path, channels)      # Pillow's formats; a file of another size raises here


def read_png_stack(pool, paths, channels):
    """uint8 [N,H,W,C] of N same-sized PNG files: every pool thread decodes its files straight into their rows (no per-file array,
    no np.stack pass over the set)"""
    paths = list(paths)
    first = read_png(paths[0], channels)
    out = np.empty((len(paths),) + first.shape, np.uint8)
    out[0] = first
    list(pool.map(lambda i: _read_png_into(paths[i], channels, out[i]), range(1, len(paths))))      # consumed: a plain executor's map is lazy
    return out


def write_png(path, arr):
    arr = np.asarray(arr, dtype=np.uint8)
    if arr.ndim == 3 and arr.shape[2] == 1:
        arr = arr[..., 0]
    if _NATIVE_PNG and (arr.ndim == 2 or (arr.ndim == 3 and arr.shape[2] == 3)):
        from ._lib import check, lib
        a = np.ascontiguousarray(arr)
        check(lib.imk_png_write_file(os.fsencode(path), a.ctypes.data, a.shape[0], a.shape[1], 1 if a.ndim == 2 else 3, 1),
              f"imk_png_write_file({path})")
        return
    Image.fromarray(arr).save(path, format="PNG", compress_level=1)


class _ReaderPool:
    """The package's ONE pool of reader threads (PNG decode, per-image host geometry: work that runs outside the interpreter lock).
    `with _pool() as pool:` hands it out and leaving the block leaves it running -- a real generation used to start 850 threads for
    its ~130 short-lived executors.  map() keeps the order and, for long lists, hands every thread a few contiguous slices instead of
    one future per item (a 12 000-file training set was 24 000 submits of ~45 us each on the calling thread).  Not for use from
    inside one of its own tasks (a task waiting for the pool it runs on can starve it)."""

    def __init__(self):
        self._ex = None
        self._lock = __import__("threading").Lock()

    def _executor(self):
        with self._lock:
            if self._ex is None:
                self._ex = ThreadPoolExecutor(max_workers=_IO_THREADS, thread_name_prefix="imk-read")
            return self._ex

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    @staticmethod
    def _not_from_a_reader():
        # a task that waits for the pool it runs on can starve it (every thread waiting for a queued task): refuse instead of hanging
        assert not __import__("threading").current_thread().name.startswith("imk-read"), \
            "the reader pool was used from one of its own tasks"

    def submit(self, fn, *args, **kw):
        self._not_from_a_reader()
        return self._executor().submit(fn, *args, **kw)

    def map(self, fn, items):
        self._not_from_a_reader()
        items = list(items)
        ex = self._executor()
        if len(items) <= 4 * _IO_THREADS:
            futures = [ex.submit(fn, it) for it in items]
            return [f.result() for f in futures]
        step = -(-len(items) // (4 * _IO_THREADS))
        futures = [ex.submit(lambda lo=lo: [fn(it) for it in items[lo:lo + step]]) for lo in range(0, len(items), step)]
        return [r for f in futures for r in f.result()]


_READER_POOL = _ReaderPool()


def _pool():
    return _READER_POOL


# PNG encoding is the slowest part of a real generation (1.6 k images/s against 20 k images/s for the GPU stages), so file
# writes are queued on a background pool and overlap with whatever comes next: the next batch's decode + GPU work inside a
# writer, and -- for the prediction dumps of benchmark_*, which nothing in the pipeline reads back -- the next candidate's
# training.  Writers flush before they return (callers list their output directories); everything is flushed at exit.
_WRITE_POOL = None
_PENDING = {}                                       # thread id -> futures that thread queued
_PENDING_LOCK = __import__("threading").Lock()      # candidates training on threads (im_driver) queue writes concurrently


_MAX_PENDING_WRITES = int(os.environ.get("IMK_MAX_PENDING_WRITES", 512))      # tasks (8 files each from write_pngs_async) per calling thread


def _submit_write(job):
    """queue `job()` (a PNG write, possibly with its last host-side preparation) on the writer pool, on the calling thread's account.
    Back-pressure: every queued task pins its arrays, so a thread with more than IMK_MAX_PENDING_WRITES tasks outstanding waits for
    its OLDEST half before it queues more (re-raising a failed write) -- the host backlog stays bounded when the GPU stages outrun the
    encoders on large images or sets."""
    me = __import__("threading").get_ident()
    with _PENDING_LOCK:
        _ensure_write_pool()
        mine = _PENDING.setdefault(me, [])
        mine.append(_WRITE_POOL.submit(job))
        oldest = []
        if len(mine) > _MAX_PENDING_WRITES:
            oldest, _PENDING[me] = mine[:len(mine) // 2], mine[len(mine) // 2:]
    for f in oldest:
        f.result()


def write_png_async(path, arr):
    _submit_write(lambda: write_png(path, arr))


def write_pngs_async(jobs, per_task=8):
    """queue a batch of (path, array) writes, `per_task` files to a task: a writer's batch is ~1 500 files and one submit per file
    (~35 us on the calling thread) was a tenth of an IM++ generation's host time"""
    jobs = list(jobs)
    for lo in range(0, len(jobs), per_task):
        _submit_write(lambda part=jobs[lo:lo + per_task]: [write_png(*j) for j in part])


def _ensure_write_pool():
    global _WRITE_POOL
    if _WRITE_POOL is None:
        import atexit
        _WRITE_POOL = ThreadPoolExecutor(max_workers=_IO_THREADS)
        atexit.register(_flush_at_exit)


def _flush_at_exit():
    """the interpreter ignores exceptions of atexit handlers (the process would still exit 0 with files missing): a failed write
    that only surfaces here ends the process with status 1"""
    try:
        flush_writes(all_threads=True)
    except BaseException:      # noqa: BLE001 -- report and fail, whatever it was
        import sys
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        os._exit(1)


def flush_writes(all_threads=False):
    """wait for the PNG writes THIS thread queued (re-raising the first failure): a candidate training on a thread of its own
    (IM_PARALLEL_CANDIDATES) waits for and reports its own files, not another candidate's.  all_threads=True (the driver after it
    has joined its candidate threads; interpreter exit) waits for everything."""
    me = __import__("threading").get_ident()
    with _PENDING_LOCK:
        keys = list(_PENDING) if all_threads else [me]
        pending = [f for k in keys for f in _PENDING.pop(k, [])]
    for f in pending:
        f.result()


# ---------------------------------------------------------------------------------------------------
# IM cores (functions.py:3104-3238)
# ---------------------------------------------------------------------------------------------------
def pred_masks_to_im_binary(pred_masks):
    """functions.py:3104-3120.  pred_masks: list of N int arrays [H,W,1] (or [H,W]) of 0/1 votes.
    Returns (final u8 [H,W] {0,255}, im u8 [H,W] {0,255}, im_size, pred_size)."""
    votes = np.stack([np.asarray(m) for m in pred_masks], 0).astype(np.float32)
    if votes.ndim == 3:
        votes = votes[..., None]
    preds = torch.from_numpy(votes[:, None]).cuda()                      # [N,1,H,W,1], vote>0.5 == vote
    r = _im.im_binary(preds, 0.5, False, None, False, False)
    return (r["masks"][0, 0].cpu().numpy(), r["im"][0].cpu().numpy(),
            np.int64(r["im_size"][0, 0].item()), np.int64(r["pred_size"][0, 0].item()))


def pred_masks_to_im_multiclass(pred_masks):
    """functions.py:3123-3137.  pred_masks: list of N integer label maps [1,H,W] / [H,W]."""
    labels = np.stack([np.asarray(m).reshape(np.asarray(m).shape[-2:]) for m in pred_masks], 0)
    k = int(labels.max()) + 1
    onehot = np.eye(k, dtype=np.float32)[labels]                          # argmax(onehot) == label
    r = _im.im_multiclass(torch.from_numpy(onehot[:, None]).cuda(), None, False, False, want_presence=False)
    return r["final"][0].cpu().numpy(), r["im"][0].cpu().numpy(), np.int64(r["im_size"][0].item())


def _is_native(models):
    return all(isinstance(m, UNet) for m in models)


def _stack_predictions(models, x_u8):
    """[N,B,H,W,K] float32 device tensor from any objects exposing predict_device / predict."""
    outs = []
    for m in models:
        if hasattr(m, "predict_device"):
            outs.append(m.predict_device(x_u8))
        else:                                                              # duck-typed Keras-like model
            p = m.predict([x_u8.cpu().numpy()])
            outs.append(torch.as_tensor(np.asarray(p, dtype=np.float32)).cuda())
    return torch.stack(outs, 0).contiguous()


def _prep(prepared_image):
    x = torch.as_tensor(np.asarray(prepared_image))
    if x.dim() == 3:
        x = x[None]
    return x.to(torch.uint8).cuda().contiguous()


def get_im_prediction_binary(models, prepared_image, threshold=0.5):
    """functions.py:3140-3162.  Returns (final_pred_mask, final_inconsistency_mask, inconsistency_size, pred_size)."""
    preds = _stack_predictions(models, _prep(prepared_image))
    r = _im.im_binary(preds[:, :1], threshold, False, None, False, False)
    return (r["masks"][0, 0].cpu().numpy(), r["im"][0].cpu().numpy(),
            np.int64(r["im_size"][0, 0].item()), np.int64(r["pred_size"][0, 0].item()))


def get_im_prediction_hela(models, prepared_image, threshold=0.5):
    """functions.py:3165-3202: `>=` comparison, three independent IMs, combined = max, size = sum."""
    preds = _stack_predictions(models, _prep(prepared_image))
    r = _im.im_binary(preds[:, :1], threshold, True, None, False, False)
    m = r["masks"][0].cpu().numpy()
    return m[0], m[1], m[2], r["im"][0].cpu().numpy(), np.int64(r["im_size"][0].sum().item())


def get_im_prediction_multiclass(models, prepared_image, filter_unequal_class_pred=False):
    """functions.py:3206-3238.  Returns (final_pred_mask, im, im_size, lists_equal)."""
    probs = _stack_predictions(models, _prep(prepared_image))
    r = _im.im_multiclass(probs[:, :1], None, False, False, want_presence=True)
    pres = r["presence"][:, 0]
    lists_equal = bool(torch.all(pres == pres[0:1]).item()) if filter_unequal_class_pred else True
    return r["final"][0].cpu().numpy(), r["im"][0].cpu().numpy(), np.int64(r["im_size"][0].item()), lists_equal


# ---------------------------------------------------------------------------------------------------
# batched ensemble + IM on device (the body of the create_pseudo_labels_im_* writers)
# ---------------------------------------------------------------------------------------------------
class EnsembleIM:
    """N models of one architecture + the workspace of imk_unet_forward_im."""

    def __init__(self, models):
        if not _is_native(models):
            raise TypeError("EnsembleIM needs inconsistencymasks_amd.unet.UNet models")
        self.models = list(models)
        self.plan = models[0].plan
        for m in models:
            m.ready_for_inference()
        import ctypes
        n = len(models)
        self._params = (ctypes.c_void_p * n)(*[m.params.data_ptr() for m in models])
        self._packed = (ctypes.c_void_p * n)(*[m.packed.data_ptr() for m in models])
        self._ws = None
        self._ws_batch = 0

    def run(self, x_u8, thr=0.5, cmp_ge=False, block_in=True, block_out=True, want_presence=False, out=None):
        """x_u8 [B,H,W,C] uint8 device -> dict(img_out, masks [B,Kb,H,W] | final [B,H,W], im, im_size, pred_size, presence).
        out: optional dict with preallocated contiguous `img_out` [B,H,W,C] and / or `masks` [B,Kb,H,W] uint8 tensors (e.g.
        slices of a whole-set buffer) that the kernel writes into directly."""
        p = self.plan
        b = x_u8.shape[0]
        n = len(self.models)
        dev = x_u8.device
        if self._ws is None or self._ws_batch < b:
            # one activation workspace per concurrently running model (imk_unet_forward_im puts up to 3 on streams)
            nbytes = lib.imk_unet_forward_im_workspace_bytes(p.ptr, n, b, min(n, 3))
            if nbytes < 0:
                check(int(nbytes), "imk_unet_forward_im_workspace_bytes")
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            self._ws_batch = b
        binary = p.act_out == "sigmoid"
        kb = p.n_out if binary else 1
        masks = out["masks"] if out and "masks" in out else torch.empty((b, kb, p.h, p.w), dtype=torch.uint8, device=dev)
        im = torch.empty((b, p.h, p.w), dtype=torch.uint8, device=dev)
        im_size = torch.empty((b, kb), dtype=torch.int64, device=dev)
        pred_size = torch.zeros((b, kb), dtype=torch.int64, device=dev)
        presence = torch.empty((n, b, p.n_out), dtype=torch.uint8, device=dev) if (want_presence and not binary) else None
        img_out = out["img_out"] if out and "img_out" in out else torch.empty_like(x_u8)
        assert img_out.is_contiguous() and masks.is_contiguous() and img_out.shape == x_u8.shape and masks.shape == (b, kb, p.h, p.w)
        check(lib.imk_unet_forward_im(p.ptr, n, self._params, self._packed, x_u8.data_ptr(), b, float(thr),
                                      int(bool(cmp_ge)), x_u8.data_ptr(), int(bool(block_in)), int(bool(block_out)),
                                      img_out.data_ptr(), masks.data_ptr(), im.data_ptr(), im_size.data_ptr(),
                                      pred_size.data_ptr(), 0 if presence is None else presence.data_ptr(),
                                      self._ws.data_ptr(), self._ws.numel(), _stream()), "imk_unet_forward_im")
        return {"img_out": img_out, "masks": masks, "im": im, "im_size": im_size, "pred_size": pred_size,
                "presence": presence}


class StackIM:
    """The same `.run()` as EnsembleIM for duck-typed models (anything with `.predict(x)` like a Keras model, called once
    per image with a [1,H,W,C] batch exactly as functions.py:3157 / :3224 do): the probability stack goes through
    imk_im_binary / imk_im_multiclass.  Native UNet models take the fused EnsembleIM path instead."""

    def __init__(self, models, binary):
        self.models, self.binary = list(models), binary

    def run(self, x_u8, thr=0.5, cmp_ge=False, block_in=True, block_out=True, want_presence=False):
        outs = []
        for m in self.models:
            if hasattr(m, "predict_device"):
                outs.append(m.predict_device(x_u8))
            else:
                xs = x_u8.cpu().numpy()
                p = np.concatenate([np.asarray(m.predict(xs[i:i + 1]), dtype=np.float32) for i in range(len(xs))], 0)
                outs.append(torch.from_numpy(p).cuda())
        preds = torch.stack(outs, 0).contiguous()
        if self.binary:
            r = _im.im_binary(preds, thr, cmp_ge, x_u8, block_in, block_out)
            r["presence"] = None
            return r
        r = _im.im_multiclass(preds, x_u8, block_in, block_out, want_presence=want_presence)
        b = x_u8.shape[0]
        return {"img_out": r["img_out"], "masks": r["final"][:, None], "im": r["im"], "im_size": r["im_size"][:, None],
                "pred_size": torch.zeros((b, 1), dtype=torch.int64, device=x_u8.device), "presence": r["presence"]}


def _ensemble(models, binary):
    return EnsembleIM(models) if _is_native(models) else StackIM(models, binary)


def dilate_mask(mask, kernel_size=3, iterations=1):
    """functions.py:3075-3100: every non-zero class dilated separately in ascending order, later classes overwriting
    earlier ones -- i.e. the grey-level dilation (neighbourhood maximum, out-of-image taps ignored) of the class-id map,
    which is what imk_morph computes.  Accepts [H,W] / [B,H,W] uint8 (numpy or device tensor)."""
    t = torch.as_tensor(mask)
    squeeze = t.dim() == 2
    d = (t[None] if squeeze else t).to(torch.uint8).cuda().contiguous()
    for _ in range(iterations):
        d = _im.morph(d, kernel_size, "dilate")
    d = d[0] if squeeze else d
    return d if isinstance(mask, torch.Tensor) else d.cpu().numpy()


def _morph_then_block(r, erode_kernel, dilate_kernel, block_input, block_output, x_u8, dilate_masks=False):
    """EK / DK > 0 (the reference's DEFAULT arguments; 0 in every shipped config): morphology on the IM, with
    erode_kernel > 0 also dilate_mask on the label maps where the reference does it (multiclass functions.py:3041-3047,
    HeLa alive / dead :2940-2946), then blocking with the modified IM (functions.py:2858-2874, 3049-3062 order)."""
    im = r["im"]
    masks = r["masks"]
    if erode_kernel > 0:
        im = _im.morph(im, erode_kernel, "erode")
        if dilate_masks:
            b, m, h, w = masks.shape
            masks = _im.morph(masks.reshape(b * m, h, w), 3, "dilate").reshape(b, m, h, w)
    if dilate_kernel > 0:
        im = _im.morph(im, dilate_kernel, "dilate")
    img = x_u8.clone()
    masks = masks.clone() if masks is r["masks"] else masks
    _im.block_apply(im, img if block_input else None, masks if block_output else None)
    return img, masks, im


def _all_reduce_sum(vals):
    d = _dist()
    dev = "cuda" if (d is None or d.get_backend() == "nccl") and torch.cuda.is_available() else "cpu"
    t = torch.tensor(vals, dtype=torch.float64, device=dev)
    if d:
        d.all_reduce(t)
    return t.cpu().tolist()


def _run_writer(models, h, w, c, images_path, out_dirs, kind, erode_kernel, dilate_kernel, block_input, block_output,
                flag, flip_channels=False):
    """Shared body of the ISIC and multiclass writers.  kind in {'isic', 'multi'}.
    flip_channels: rgb=False of the reference (functions.py:2847-2850) -- the nets see the channel order of the file as
    OpenCV decodes it (BGR) instead of RGB; the written image keeps the file's order either way, so the flip is applied to
    the nets' input and undone on the blocked image."""
    names = os.listdir(images_path)
    mine = shard_list(names)
    ens = _ensemble(models, kind == "isic")
    fused_block = (erode_kernel <= 0 and dilate_kernel <= 0)
    sum_im, count = 0, 0
    with _pool() as pool:
        for i, j in infer_batches(len(mine), infer_batch_size(_models_alpha(models))):
            chunk = mine[i:j]
            imgs = read_png_stack(pool, [os.path.join(images_path, n) for n in chunk], c)
            x = torch.from_numpy(imgs).cuda()
            if flip_channels:
                x = x.flip(-1).contiguous()
            r = ens.run(x, THRESHOLD, False, block_input and fused_block, block_output and fused_block,
                        want_presence=(kind == "multi" and flag))
            if fused_block:
                img_out, masks, im = r["img_out"], r["masks"], r["im"]
            else:
                img_out, masks, im = _morph_then_block(r, erode_kernel, dilate_kernel, block_input, block_output, x,
                                                       dilate_masks=(kind == "multi"))
            if flip_channels:
                img_out = img_out.flip(-1)
            img_np, m_np, im_np = img_out.cpu().numpy(), masks.cpu().numpy(), im.cpu().numpy()
            ims = r["im_size"].sum(1).cpu().numpy()
            pss = r["pred_size"].sum(1).cpu().numpy()
            pres = None if r["presence"] is None else r["presence"].cpu().numpy()
            jobs = []
            for j, name in enumerate(chunk):
                sum_im += int(ims[j])
                count += 1
                if kind == "isic":   # functions.py:2878-2886
                    keep = (pss[j] > ims[j] and pss[j] > 0) if flag else True
                else:                # functions.py:3029-3035
                    keep = bool(np.all(pres[:, j] == pres[0:1, j])) if flag else True
                if keep:
                    jobs.append((os.path.join(out_dirs["images"], name), img_np[j]))
                    jobs.append((os.path.join(out_dirs["masks"], name), m_np[j, 0]))
                jobs.append((os.path.join(out_dirs["im"], name), im_np[j]))
            write_pngs_async(jobs)      # encoded while the next batch is decoded and run
    flush_writes()
    tot_im, tot_n = _all_reduce_sum([sum_im, count])
    return round(tot_im / tot_n, 0) if tot_n else 0.0


def create_pseudo_labels_im_ISIC_2018(models, h, w, c, images_path, main_output_path, rgb=True, erode_kernel=5,
                                      dilate_kernel=5, block_input=True, block_output=True,
                                      filter_bad_predictions=True):
    """functions.py:2832-2891.  Writes images/ masks/ im/ under main_output_path, returns mean_im_size."""
    out = {k: os.path.join(main_output_path, k) for k in ("images", "masks", "im")}
    for d in out.values():
        os.makedirs(d, exist_ok=True)
    return _run_writer(models, h, w, c, images_path, out, "isic", erode_kernel, dilate_kernel, block_input,
                       block_output, filter_bad_predictions, flip_channels=(not rgb and c == 3))


def create_pseudo_labels_im_multiclass(models, h, w, c, images_path, main_output_path, rgb=True, erode_kernel=5,
                                       dilate_kernel=5, block_input=True, block_output=True,
                                       filter_unequal_class_pred=False):
    """functions.py:2988-3070.  With erode_kernel > 0 the label map is also dilated per class (functions.py:3047,
    dilate_mask) before blocking."""
    out = {k: os.path.join(main_output_path, k) for k in ("images", "masks", "im")}
    for d in out.values():
        os.makedirs(d, exist_ok=True)
    return _run_writer(models, h, w, c, images_path, out, "multi", erode_kernel, dilate_kernel, block_input,
                       block_output, filter_unequal_class_pred, flip_channels=(not rgb and c == 3))


# ---------------------------------------------------------------------------------------------------
# Noisy-Student augmentation of a pseudo-label directory (IM+ drivers; functions.py:2567-2722)
# ---------------------------------------------------------------------------------------------------
from .augment import augment_batch, augment_image_and_mask, augment_image_and_masks, draw_params  # noqa: E402,F401


def _augment_dirs(image_dir, mask_dirs, image_out, mask_outs, image_channels, num_images, copy_org, draw_kw):
    """Shared body: every file of image_dir (+ same-named masks) -> num_images augmented copies `{stem}_aug_{n}.png`.
    The file list is sharded over the ranks; batches are formed from consecutive files of equal size."""
    import shutil
    for d in [image_out] + list(mask_outs):
        os.makedirs(d, exist_ok=True)
    names = os.listdir(image_dir)
    mine = shard_list(names)
    if copy_org:
        for name in mine:
            shutil.copy(os.path.join(image_dir, name), os.path.join(image_out, name))
            for src, dst in zip(mask_dirs, mask_outs):
                shutil.copy(os.path.join(src, name), os.path.join(dst, name))
    with _pool() as pool:
        for i in range(0, len(mine), INFER_BATCH):
            chunk = mine[i:i + INFER_BATCH]
            imgs = list(pool.map(lambda n: read_png(os.path.join(image_dir, n), image_channels), chunk))
            msks = list(pool.map(lambda n: np.concatenate([read_png(os.path.join(d, n), 1) for d in mask_dirs], 2), chunk))
            by_shape = {}
            for j, a in enumerate(imgs):
                by_shape.setdefault(a.shape, []).append(j)
            jobs = []
            for idx in by_shape.values():
                x = torch.from_numpy(np.stack([imgs[j] for j in idx], 0)).cuda()
                m = torch.from_numpy(np.stack([msks[j] for j in idx], 0)).cuda()
                for n in range(num_images):
                    o, om = augment_batch(x, m, draw_params(len(idx), **draw_kw))
                    o, om = o.cpu().numpy(), om.cpu().numpy()
                    for row, j in enumerate(idx):
                        stem = chunk[j][:-4]
                        jobs.append((os.path.join(image_out, f"{stem}_aug_{n}.png"), o[row]))
                        for k, dst in enumerate(mask_outs):
                            jobs.append((os.path.join(dst, f"{stem}_aug_{n}.png"), om[row, :, :, k]))
            list(pool.map(lambda a: write_png(*a), jobs))
    if _dist():
        _dist().barrier()


def create_augment_images_and_masks_ISIC_2018(images_path, masks_path, main_output_path, num_images=9, copy_org=True,
                                              brightness_range_alpha=(0.5, 1.5), brightness_range_beta=(-25, 25),
                                              max_blur=3, max_noise=25, free_rotation=True):
    """functions.py:2567-2609."""
    _augment_dirs(images_path, [masks_path], os.path.join(main_output_path, "images"),
                  [os.path.join(main_output_path, "masks")], 3, num_images, copy_org,
                  dict(brightness_range_alpha=brightness_range_alpha, brightness_range_beta=brightness_range_beta,
                       max_blur=max_blur, max_noise=max_noise, free_rotation=free_rotation))


def create_augment_images_and_masks_multiclass(images_path, masks_path, main_output_path, num_images=9, copy_org=True,
                                               free_rotation=False, brightness_range_alpha=(0.5, 1.5),
                                               brightness_range_beta=(-25, 25), max_blur=3, max_noise=25):
    """functions.py:2678-2720 (note the position of free_rotation, as in the reference)."""
    _augment_dirs(images_path, [masks_path], os.path.join(main_output_path, "images"),
                  [os.path.join(main_output_path, "masks")], 3, num_images, copy_org,
                  dict(brightness_range_alpha=brightness_range_alpha, brightness_range_beta=brightness_range_beta,
                       max_blur=max_blur, max_noise=max_noise, free_rotation=free_rotation))


def create_augment_images_and_masks_hela(main_input_path, main_output_path, num_images=9, copy_org=True,
                                         free_rotation=True, brightness_range_alpha=(0.7, 1.3),
                                         brightness_range_beta=(-15, 15), max_blur=3, max_noise=25):
    """functions.py:2613-2674.  The brightfield image is processed as 3 channels (cv2.imread's default), so that the
    noise of the 3 channels is independent exactly as in the reference; the parser's grey conversion follows later."""
    subs = ("alive", "dead", "mod_position")
    _augment_dirs(os.path.join(main_input_path, "brightfield"), [os.path.join(main_input_path, s) for s in subs],
                  os.path.join(main_output_path, "brightfield"), [os.path.join(main_output_path, s) for s in subs],
                  3, num_images, copy_org,
                  dict(brightness_range_alpha=brightness_range_alpha, brightness_range_beta=brightness_range_beta,
                       max_blur=max_blur, max_noise=max_noise, free_rotation=free_rotation))


# ---------------------------------------------------------------------------------------------------
# metrics (functions.py:162-184, 1767-1861)
# ---------------------------------------------------------------------------------------------------
def dice_loss(y_true, y_pred, smooth=1):
    """functions.py:162-184 on arrays [B,H,W,K] (the `custom_objects` entry of the scripts' load_model calls; no shipped
    script trains with it)."""
    y_true = np.asarray(y_true, dtype=np.float32)
    y_pred = np.asarray(y_pred, dtype=np.float32)
    inter = (y_true * y_pred).sum(axis=(1, 2, 3))
    union = y_true.sum(axis=(1, 2, 3)) + y_pred.sum(axis=(1, 2, 3))
    return 1 - ((2 * inter + smooth) / (union + smooth)).mean()


class ignore_im_categorical_crossentropy:
    """functions.py:105-124 -- imported (never passed to a trainer) by Cityscapes/13_Cityscapes_aug_IM+.py:6, HeLa/12_HeLa_IM++.py:5,
    HeLa/14_HeLa_aug_IM++.py:5.  A host-side (numpy) restatement of the loss VALUE, literally as written there: per-pixel categorical
    cross-entropy (Keras clips probabilities to [1e-7, 1 - 1e-7] and renormalises), times `1 - y_true[:, 0]` -- which indexes the second
    AXIS at 0, not class 0, and therefore only broadcasts for a few shapes -- then the mean.  No fused training kernel implements it:
    `train_*` given an instance raises NotImplementedError (no reference script trains with it)."""

    def __init__(self, **kwargs):
        self.kwargs = kwargs

    def __call__(self, y_true, y_pred):
        return self.call(y_true, y_pred)

    def call(self, y_true, y_pred):
        y_true = np.asarray(y_true, dtype=np.float32)
        y_pred = np.asarray(y_pred, dtype=np.float32)
        p = y_pred / y_pred.sum(-1, keepdims=True)
        p = np.clip(p, 1e-7, 1 - 1e-7)
        loss = -(y_true * np.log(p)).sum(-1)
        loss = loss * (1.0 - y_true[:, 0])
        return np.float32(loss.mean())


class ignore_im_dice_loss_multiclass:
    """functions.py:128-160 (same three importing scripts, never passed to a trainer): numpy restatement, literally -- `[:, :, 1:]`
    drops index 0 of the THIRD axis, the sums run over axes 1 and 2.  `train_*` given an instance raises NotImplementedError."""

    def __init__(self, **kwargs):
        self.kwargs = kwargs

    def __call__(self, y_true, y_pred):
        return self.call(y_true, y_pred)

    def call(self, y_true, y_pred):
        y_true = np.asarray(y_true, dtype=np.float32)[:, :, 1:]
        y_pred = np.asarray(y_pred, dtype=np.float32)[:, :, 1:]
        inter = (y_true * y_pred).sum(axis=(1, 2))
        dice = (2.0 * inter + 1e-7) / (y_true.sum(axis=(1, 2)) + y_pred.sum(axis=(1, 2)) + 1e-7)
        return np.float32((1 - dice).mean())


_UNTRAINABLE_LOSSES = (ignore_im_categorical_crossentropy, ignore_im_dice_loss_multiclass)


def _reject_untrainable(loss_func, who):
    """the losses no reference script trains with have no fused kernel: say so instead of silently training with another loss"""
    if isinstance(loss_func, _UNTRAINABLE_LOSSES) or loss_func in _UNTRAINABLE_LOSSES or loss_func is dice_loss:
        name = getattr(loss_func, "__name__", type(loss_func).__name__)
        raise NotImplementedError(f"{who}: no fused training kernel for {name} (imported by some reference scripts, passed by none); "
                                  "the scripts train with 'mse' / CategoricalCrossentropy()")


def get_IoU_binary(gt, pred):
    """functions.py:1767-1788."""
    gt, pred = np.asarray(gt), np.asarray(pred)
    inter = np.logical_and(gt, pred).sum()
    union = np.logical_or(gt, pred).sum()
    return inter / (union + 1e-7)


def dice_score_numpy_binary(gt, pred, smooth=1, threshold=128):
    """functions.py:1837-1861."""
    g = (np.asarray(gt) >= threshold).astype(np.float32)
    p = (np.asarray(pred) >= threshold).astype(np.float32)
    inter = np.sum(g * p)
    union = np.sum(g) + np.sum(p)
    return (2.0 * inter + smooth) / (union + smooth)


# ---------------------------------------------------------------------------------------------------
# training (functions.py:189-316)
# ---------------------------------------------------------------------------------------------------
def _mask_path(image_path):
    return re.sub("images", "masks", image_path)      # tf.strings.regex_replace(image_dir, 'images', 'masks')


def parse_image_ISIC_2018(image_path, IMG_CHANNELS=3):
    """functions.py:955-977: image u8 [H,W,C]; mask = uint8(mask/255) => 255 -> 1, everything else 0."""
    image = read_png(image_path, IMG_CHANNELS)
    mask = read_png(_mask_path(image_path), 1)
    return image, (mask // 255).astype(np.uint8)


def parse_image_multiclass(image_path, n_classes, image_channels=3):
    """functions.py:1021-1048: mask stays a class-id map here; the one-hot is formed inside the loss kernel."""
    image = read_png(image_path, image_channels)
    mask = read_png(_mask_path(image_path), 1)[..., 0]
    return image, mask


def gather_pairs(idx, img=None, mask_planar=None, div255=False, mul=None):
    """Rows idx of a device-resident set in one libimk launch per tensor (imk_gather_pairs): what list_files(shuffle).map(parse)
    .batch() delivers per epoch (functions.py:207-209) when the files already sit in HBM.  img [N, ...] uint8 -> [n, ...];
    mask_planar [N, P, H, W] uint8 (as the IM kernels write label maps) -> [n, H, W, P] with v -> (v / 255 if div255) * mul[p]
    (parse_image_ISIC_2018 / parse_image_hela, functions.py:975, 1001-1011).  Returns (img_out, mask_out); absent ones are None."""
    idx = idx.to(torch.int64).contiguous()
    n = int(idx.shape[0])
    # the kernel copies bytes: rows of uint8, densely laid out (a float or strided set would be gathered silently wrong)
    for name, t in (("img", img), ("mask_planar", mask_planar), ("mul", mul)):
        if t is not None and (t.dtype != torch.uint8 or not t.is_contiguous()):
            raise ValueError(f"gather_pairs: {name} must be a contiguous uint8 tensor, got {t.dtype}, contiguous={t.is_contiguous()}")
    if mask_planar is not None and mask_planar.dim() != 4:
        raise ValueError("gather_pairs: mask_planar must be [N, P, H, W]")
    out_i = out_m = None
    if n == 0:      # correctly shaped empty results
        if img is not None:
            out_i = torch.empty((0,) + tuple(img.shape[1:]), dtype=torch.uint8, device=img.device)
        if mask_planar is not None:
            N, P, H, W = mask_planar.shape
            out_m = torch.empty((0, H, W, P), dtype=torch.uint8, device=mask_planar.device)
        return out_i, out_m
    for lo in range(0, n, 65535):          # the row index rides on gridDim.y
        hi = min(n, lo + 65535)
        if img is not None:
            if out_i is None:
                out_i = torch.empty((n,) + tuple(img.shape[1:]), dtype=torch.uint8, device=img.device)
            row = int(img[0].numel())
        if mask_planar is not None:
            if out_m is None:
                N, P, H, W = mask_planar.shape
                out_m = torch.empty((n, H, W, P), dtype=torch.uint8, device=mask_planar.device)
        check(lib.imk_gather_pairs(img.data_ptr() if img is not None else None, row if img is not None else 0,
                                   mask_planar.data_ptr() if mask_planar is not None else None,
                                   int(mask_planar.shape[1]) if mask_planar is not None else 0,
                                   int(mask_planar.shape[2] * mask_planar.shape[3]) if mask_planar is not None else 0,
                                   1 if div255 else 0, mul.data_ptr() if mul is not None else None, idx[lo:hi].data_ptr(), hi - lo,
                                   out_i[lo:hi].data_ptr() if out_i is not None else None,
                                   out_m[lo:hi].data_ptr() if out_m is not None else None, _stream()), "imk_gather_pairs")
    return out_i, out_m


class _EpochLoader:
    """list_files(seed).map(parse).batch(B).repeat(): seeded shuffle per pass, one short batch per pass
    (functions.py:207-209), PNG decode on a thread pool, whole set cached on the device after the first pass."""

    def __init__(self, files, parse, batch, seed, parse_key=None):
        self.files = sorted(files)
        self.parse = parse
        self.parse_key = parse_key      # names what `parse` does to a file: with it the decoded set is shared between loaders
        self.batch = batch
        self.rng = np.random.default_rng(seed)
        self.x = self.y = None
        self._order = []

    def _load(self):
        """decode the whole set once: every pool thread parses a slice of the files straight into its rows of two host arrays
        (no per-item list, no np.stack pass).  The candidates of a generation train on the same directory, so the device copy is
        kept in the decode cache under (parse_key, names, hash of every file's size and mtime) and the next candidate starts without decoding."""
        key = None
        if self.parse_key is not None and self.files:
            st = [os.stat(f) for f in self.files]
            key = ("train", self.parse_key, tuple(self.files), _stat_sig(st))
        with _CACHE_LOCK:
            hit = _DECODE_CACHE.get(key) if key is not None else None
            if hit is not None:
                self.x, self.y = _used_here(hit)
                return
            x0, y0 = self.parse(self.files[0])
            n = len(self.files)
            xs, ys = np.empty((n,) + x0.shape, x0.dtype), np.empty((n,) + y0.shape, y0.dtype)
            xs[0], ys[0] = x0, y0

            def fill(i):
                xs[i], ys[i] = self.parse(self.files[i])
            with _pool() as pool:
                pool.map(fill, range(1, n))
            self.x, self.y = torch.from_numpy(xs).cuda(), torch.from_numpy(ys).cuda()
            if key is not None:
                _uploaded()
                _decode_cache_put(key, [self.x, self.y])

    def next_batch(self):
        if self.x is None:
            self._load()
        if not self._order:      # new pass: one shuffled copy of the set, batches are contiguous views of it
            perm = torch.as_tensor(self.rng.permutation(len(self.files)), device="cuda")
            self._px, _ = gather_pairs(perm, img=self.x)           # the epoch's shuffle: one libimk launch per tensor
            self._py, _ = gather_pairs(perm, img=self.y)
            self._order = [(i, min(i + self.batch, len(perm))) for i in range(0, len(perm), self.batch)]
        lo, hi = self._order.pop(0)
        return self._px[lo:hi], self._py[lo:hi]


def _train_shard(files):
    """This rank's block of the training file list; with fewer files than ranks every rank keeps the whole list (the
    averaged gradient is then the single-GPU gradient) instead of an empty shard."""
    files = sorted(files)
    _, world = _rank_world()
    return files if len(files) < world else shard_list(files)


def _sync_moving_stats(model):
    """Data-parallel replicas normalise with their own batch statistics, so their BatchNorm moving
    statistics drift apart; before a checkpoint decision they are averaged, so that every rank evaluates -- and rank 0
    saves -- the same model."""
    d = _dist()
    if not d:
        return
    tail = model.params[model.plan.n_trainable:]
    d.all_reduce(tail)
    tail.mul_(1.0 / d.get_world_size())
    model._fold_ok = False


def _gather_lists(*lists):
    """Per-image metric lists of a sharded benchmark -> the complete lists on every rank (rank order = sorted file order)."""
    d = _dist()
    if not d:
        return lists
    out = [None] * d.get_world_size()
    d.all_gather_object(out, lists)
    return tuple(sum((o[i] for o in out), []) for i in range(len(lists)))


def _bench_names(directory):
    """os.listdir order on one GPU (as the reference); this rank's contiguous block of the sorted list otherwise"""
    names = os.listdir(directory)
    return shard_list(names) if _dist() else names


def _grad_allreduce(model):
    d = _dist()
    if not d:
        return 1.0
    d.all_reduce(model.grads_and_stats)   # one flat fp32 bucket per step (RCCL over xGMI); loss, found_inf ride along
    return 1.0 / d.get_world_size()


def dp_bn_momentum_rule(world=None):
    """(rule name, momentum) of the BatchNorm moving statistics for a run of `world` ranks.

    Default at ANY world size: the reference's recipe, Keras' momentum 0.99 per optimizer step (unet.py:7 leaves the layer's
    default) -- a data-parallel run trains the same model definition as one rank, and 1-rank and N-rank results are comparable
    under one rule.  What that costs at 8 ranks is measured (profiles/r03_dp_convergence.txt): the per-GPU batch stays 32, an epoch
    has N times fewer steps, the moving statistics need ~500 steps to forget their initial values, and the validation metric that
    picks the checkpoint lags (val IoU 0.989 instead of 0.9998 after 50 epochs).  IMK_DP_BN_MOMENTUM=scaled opts into momentum
    0.99^N -- the same memory in SAMPLES as the single-GPU recipe (0.9994 at 8 ranks) -- a stated deviation from the reference;
    fit() prints one line per process group naming the rule in force.  One rank is always 0.99."""
    if world is None:
        _, world = _rank_world()
    mode = os.environ.get("IMK_DP_BN_MOMENTUM", "").lower()
    if mode not in ("", "scaled", "reference"):
        raise ValueError(f"IMK_DP_BN_MOMENTUM={mode!r}: expected 'scaled' or 'reference'")
    if world > 1 and mode == "scaled":
        return "scaled", 0.99 ** world
    return "reference", 0.99


_DP_RULE_SAID = [False]


def _dp_bn_momentum(model):
    rule, mom = dp_bn_momentum_rule()
    model.set_bn_momentum(mom)
    rank, world = _rank_world()
    if world > 1 and rank == 0 and not _DP_RULE_SAID[0]:
        _DP_RULE_SAID[0] = True
        print(f"[imk] data parallel, {world} ranks: BatchNorm momentum rule '{rule}' = {mom:.6f} per step"
              + ("" if rule == "scaled" else " (the reference's; IMK_DP_BN_MOMENTUM=scaled keeps the moving statistics' memory in samples)"),
              flush=True)


def fit(model, loader, steps_per_epoch, epochs, loss_kind, on_epoch_end=None, lr=None, wd=None):
    """model.fit(train_dataset, epochs, steps_per_epoch) of functions.py:218."""
    lr = LR if lr is None else lr
    wd = WD if wd is None else wd
    model.init_train_state()
    _dp_bn_momentum(model)
    history = []
    for ep in range(epochs):
        loss_acc = torch.zeros((), device="cuda")
        for _ in range(steps_per_epoch):
            x, y = loader.next_batch()
            model.fwd_bwd(x, y, loss_kind)
            scale = _grad_allreduce(model)
            model.adamw_step(lr, wd, grad_scale=scale)
            loss_acc += model.stats[0] * scale
        history.append(float(loss_acc.item()) / max(steps_per_epoch, 1))
        _sync_moving_stats(model)
        if on_epoch_end:
            on_epoch_end(ep, history[-1])
    return history


def save_model(model, path):
    """ModelCheckpoint's model.save(path) (functions.py:217).  Default: safetensors (weights + geometry in the metadata) under the same
    name; IMK_MODEL_FORMAT=keras_h5: an HDF5 file in Keras' save_weights layout, which `get_unet(...).load_weights(path)` restores in
    TensorFlow and load_model below reads back."""
    if os.environ.get("IMK_MODEL_FORMAT", "safetensors") == "keras_h5":
        from .keras_h5 import save_keras_weights
        return save_keras_weights(model, path)
    from safetensors.torch import save_file
    sd = {k: v.contiguous() for k, v in model.state_dict().items()}
    meta = {"h": str(model.plan.h), "w": str(model.plan.w), "c_in": str(model.plan.c_in),
            "n_out": str(model.plan.n_out), "alpha": repr(model.plan.alpha), "act_out": model.plan.act_out}
    save_file(sd, path, metadata=meta)


def load_model(path, custom_objects=None, device="cuda"):
    """Counterpart of tf.keras.models.load_model(path, custom_objects=...) (ISIC_2018/09_ISIC_2018_IM.py:75).  Takes this package's
    safetensors files and Keras HDF5 checkpoints of the reference's get_unet (full model, or weights-only written by save_model /
    keras_h5.save_keras_weights); the optimizer state of a Keras file is ignored, like load_model(..., compile=False)."""
    from . import h5lite
    if h5lite.is_hdf5(path):
        from .keras_h5 import load_keras_model
        return load_keras_model(path, device=device)
    from safetensors import safe_open
    with safe_open(path, framework="pt") as f:
        meta = f.metadata()
        sd = {k: f.get_tensor(k) for k in f.keys()}
    # a fixed seed: the initialisation is overwritten below, and an unseeded UNet() under data parallelism broadcasts its seed
    # (a collective) -- a checkpoint load must not block on the other ranks
    m = UNet(int(meta["h"]), int(meta["w"]), int(meta["c_in"]), int(meta["n_out"]), float(meta["alpha"]),
             meta["act_out"], seed=0, device=device)
    m.load_state_dict(sd)
    return m


_VAL_CACHE = {}   # (directories, file list) -> decoded validation set on the device: the monitor runs once per epoch
# Candidates of a generation may train side by side on threads with a stream each (im_driver.run, IM_PARALLEL_CANDIDATES): the
# shared caches are filled under a lock, and a filled entry is published only once its uploads have completed on the filling
# thread's stream (another thread reads it on ITS stream).
import threading as _threading
_CACHE_LOCK = _threading.RLock()


def _uploaded(*tensors):
    torch.cuda.current_stream().synchronize()
    return tensors


def _used_here(tensors):
    """Cached device tensors are allocated on the filling thread's stream and read on every candidate thread's stream
    (IM_PARALLEL_CANDIDATES): tell the caching allocator, so that an evicted block is not handed out again on its original
    stream while kernels queued on another one still read it."""
    cur = torch.cuda.current_stream()
    for t in tensors:
        t.record_stream(cur)
    return tensors


# Decoded benchmark sets (images + ground truth of the val / test / unlabeled splits) stay on the device across the
# candidates of a generation: every candidate is evaluated on the same three directories (functions.py:221-226), and
# decoding them again costs more than evaluating them.  Keyed by the files' names, sizes and modification times; bounded
# by IMK_DECODE_CACHE_GB (default 16 of the 288 GB; 0 switches it off).
_DECODE_CACHE = {}
_DECODE_CACHE_BYTES = int(float(os.environ.get("IMK_DECODE_CACHE_GB", 16)) * 2 ** 30)


def _decode_cache_put(key, tensors):
    """keep `tensors` under `key` if they fit the cache's budget, dropping the oldest entries to make room (callers hold _CACHE_LOCK)"""
    nbytes = sum(t.numel() * t.element_size() for t in tensors)
    if 0 < nbytes <= _DECODE_CACHE_BYTES:
        held = lambda: sum(sum(t.numel() * t.element_size() for t in v) for v in _DECODE_CACHE.values())
        while _DECODE_CACHE and held() + nbytes > _DECODE_CACHE_BYTES:
            _DECODE_CACHE.pop(next(iter(_DECODE_CACHE)))
        _DECODE_CACHE[key] = tensors


def _stat_sig(stats):
    """one hash over EVERY file's (size, mtime_ns): a same-sized file replaced by an older one changes it (a sum of sizes + newest
    mtime did not)"""
    import hashlib
    h = hashlib.blake2b(digest_size=16)
    for x in stats:
        h.update(x.st_size.to_bytes(8, "little", signed=False))
        h.update(x.st_mtime_ns.to_bytes(8, "little", signed=True))
    return h.hexdigest()


def _decoded_set(pool, dirs_and_channels, names):
    """[(directory, channels)] x names -> list of uint8 device tensors [N,H,W,C] (one per directory), cached"""
    def sig(d):
        st = [os.stat(os.path.join(d, n)) for n in names]
        return (d, tuple(names), _stat_sig(st))
    key = tuple((sig(d), c) for d, c in dirs_and_channels)
    with _CACHE_LOCK:       # (candidates on threads: one decodes, the others wait for it instead of decoding the same files again)
        hit = _DECODE_CACHE.get(key)
        if hit is not None:
            return _used_here(hit)
        out = []
        for d, c in dirs_and_channels:
            out.append(torch.from_numpy(read_png_stack(pool, [os.path.join(d, n) for n in names], c)).cuda())
        _uploaded()
        _decode_cache_put(key, out)
        return out


def _binary_iou_dataset(model, images_dir, masks_dir, c, batch=64):
    """Keras BinaryIoU(target_class_ids=[1], threshold=0.5) accumulated over the whole directory (the
    val_binary_io_u monitor of functions.py:216-217); pixel counts from imk_eval_binary."""
    files = sorted(glob.glob(os.path.join(images_dir, "*.png")))
    key = (images_dir, masks_dir, c, tuple(files))
    with _CACHE_LOCK:
        if key not in _VAL_CACHE:
            _VAL_CACHE.clear()
            with _pool() as pool:
                items = list(pool.map(lambda p: parse_image_ISIC_2018(p, c), files))
            _VAL_CACHE[key] = _uploaded(torch.from_numpy(np.stack([it[0] for it in items], 0)).cuda(),
                                        torch.from_numpy(np.stack([it[1] for it in items], 0)[..., 0]).cuda())
        xs, ys = _used_here(_VAL_CACHE[key])
    inter = union = 0
    for i in range(0, len(files), batch):
        _, cnt = _ev.eval_binary(model.predict_device(xs[i:i + batch]), ys[i:i + batch], 0.5, True, want_pred=False)
        inter += int(cnt[:, 0].sum()); union += int(cnt[:, 1].sum())
    return inter / max(union, 1)


def benchmark_ISIC2018(model, images_dir, masks_dir, pred_path, h, w, c, batch_size=64, create_images=True,
                       print_results=False):
    """functions.py:1078-1151: batch-64 predict, > 0.5, optional PNG dump, per-image IoU/Dice rounded to 4,
    means rounded to 3.  Threshold + per-image pixel counts run in imk_eval_binary; the ratios are formed on the
    host with the reference's float expressions (evaluate.iou_dice_from_counts)."""
    os.makedirs(pred_path, exist_ok=True)
    names = _bench_names(images_dir)
    ious, dices = [], []
    with _pool() as pool:
        xs, gs = _decoded_set(pool, [(images_dir, c), (masks_dir, 1)], names) if names else (None, None)
        for i in range(0, len(names), batch_size):
            chunk = names[i:i + batch_size]
            probs = model.predict_device(xs[i:i + batch_size])
            pred_d, counts = _ev.eval_binary(probs, gs[i:i + batch_size, ..., 0], 0.5, False, want_pred=create_images)
            if create_images:      # prediction dumps: nothing reads them back, they are written in the background
                pred = pred_d.cpu().numpy()
                for j, n in enumerate(chunk):
                    write_png_async(os.path.join(pred_path, n), pred[j])
            for j, n in enumerate(chunk):
                iou, dice = _ev.iou_dice_from_counts(counts[j])
                d = round(float(dice), 4)
                u = round(float(iou), 4)
                dices.append(d); ious.append(u)
                if print_results:
                    print(f"{n} IoU: {u}    DS: {d}")
    ious, dices = _gather_lists(ious, dices)
    mIoU = round(float(np.sum(ious) / len(ious)), 3)
    mdice = round(float(np.sum(dices) / len(dices)), 3)
    print(f"------------------------------------------------------------  mIoU: {mIoU}    mdice score: {mdice}  "
          "------------------------------------------------------------")
    return mIoU, mdice


def train_ISIC_2018(train_images_dir, val_images_dir, val_masks_dir, test_images_dir, test_masks_dir,
                    unlabeled_images_dir, unlabeled_masks_dir, modelname, filepath_h5, model, loss_func,
                    steps_per_epoch, h, w, c, val_pred_dir, test_pred_dir, unlabeled_pred_dir, print_results=False):
    """functions.py:189-228.  loss_func must be 'mse' (what every ISIC script passes)."""
    if loss_func != "mse":
        raise NotImplementedError("the ISIC scripts train with 'mse'")
    files = _train_shard(glob.glob(os.path.join(train_images_dir, "*.png")))
    loader = _EpochLoader(files, lambda p: parse_image_ISIC_2018(p, c), BATCH_SIZE, SEED, ("isic", c))
    best = {"iou": -1.0}

    def on_epoch_end(ep, loss):      # ModelCheckpoint(save_best_only, monitor='val_binary_io_u', mode='max')
        iou = _binary_iou_dataset(model, val_images_dir, val_masks_dir, c)
        if iou > best["iou"]:
            best["iou"] = iou
            if _rank_world()[0] == 0:
                save_model(model, filepath_h5)

    fit(model, loader, steps_per_epoch, NUM_EPOCHS, 0, on_epoch_end)
    d = _dist()
    if d:
        d.barrier()
    best_model = load_model(filepath_h5)
    mIoU_val, dice_val = benchmark_ISIC2018(best_model, val_images_dir, val_masks_dir, val_pred_dir, h, w, c,
                                            print_results=print_results)
    mIoU_test, dice_test = benchmark_ISIC2018(best_model, test_images_dir, test_masks_dir, test_pred_dir, h, w, c,
                                              print_results=print_results)
    mIoU_unl, dice_unl = benchmark_ISIC2018(best_model, unlabeled_images_dir, unlabeled_masks_dir, unlabeled_pred_dir,
                                            h, w, c, print_results=print_results)
    print(f"{modelname} mIoU_val: {mIoU_val}")
    flush_writes()      # the benchmarks' prediction PNGs are on disk (a failed write raises here, not at interpreter exit)
    return mIoU_val, mIoU_test, mIoU_unl, dice_val, dice_test, dice_unl


# ---------------------------------------------------------------------------------------------------
# multi-class training / evaluation (functions.py:275-316, 1265-1339, 51-102, 1791-1834)
# ---------------------------------------------------------------------------------------------------
def pixel_accuracy(pred_mask, gt_mask):
    """functions.py:1820-1834."""
    return np.sum(pred_mask == gt_mask) / np.prod(np.asarray(gt_mask).shape)


def get_IoU_multi_unique(pred, gt):
    """functions.py:1791-1816: mean IoU over the classes PRESENT in the ground truth."""
    classes = np.unique(gt)
    total = 0.0
    for i in classes:
        g, p = gt == i, pred == i
        total += np.logical_and(g, p).sum() / (np.logical_or(g, p).sum() + 1e-7)
    return total / len(classes)


def _color_lut(class_to_color_mapping):
    """class id -> RGB as a 256-row table: what the reference's loop `color[class_mask == cls] = col` over the colour -> class dict
    computes (functions.py:6127-6149; for a class listed twice the LAST colour wins, classes that are not listed stay black)"""
    lut = np.zeros((256, 3), dtype=np.uint8)
    for col, cls in class_to_color_mapping.items():
        lut[int(cls) & 255] = col
    return lut


def convert_class_to_color_mask(class_mask, output_path, class_to_color_mapping):
    """functions.py:6127-6149 (the mapping is colour -> class value; file written in RGB order on disk).  One table gather instead of
    a masked assignment per class: the loop cost 1 ms per 256 x 256 map on the caller's thread -- half of a multiclass candidate's
    wall time in a real SUIM run (tests/gpu_probe/full_driver_run_suim.py) -- and the gather runs on the writer pool with the encode."""
    lut = _color_lut(class_to_color_mapping)
    cm = np.asarray(class_mask, dtype=np.uint8)
    _submit_write(lambda: write_png(output_path, lut[cm]))


class MeanIoU:
    """functions.py:51-102: running mean over batches of the per-class SOFT IoU between one-hot targets and
    probabilities (the `val_mean_io_u` monitor of train_multiclass)."""

    def __init__(self, num_classes):
        self.num_classes = num_classes
        self.reset_state()

    def reset_state(self):
        self.total, self.count = 0.0, 0.0

    def update_state(self, y_true_ids, probs):
        """y_true_ids [B,H,W] uint8 class ids (device), probs [B,H,W,K] float32 (device): the per-class sums come from
        imk_eval_soft_sums, the ratios are formed like compute_iou (functions.py:75-80)."""
        s = _ev.soft_sums(probs, y_true_ids.to(torch.uint8), 0)
        inter, n_true, n_pred = s[0], s[1], s[2]
        with np.errstate(divide="ignore", invalid="ignore"):
            iou = inter / (n_true + n_pred - inter)            # 0/0 -> nan propagates exactly like the reference
        self.total += float(np.mean(iou.astype(np.float32)))
        self.count += 1.0

    def result(self):
        return self.total / self.count


def benchmark_multiclass(model, image_path, gt_path, pred_path, h, w, c, class_to_color_mapping, batch_size=64,
                         create_images=True, print_results=True):
    """functions.py:1265-1339: batch-64 predict, argmax, PNG dumps, per-image PA / IoU rounded to 4, means to 3.
    argmax + per-image class histograms run in imk_eval_multiclass (evaluate.pa_iou_from_counts forms the ratios)."""
    os.makedirs(pred_path, exist_ok=True)
    names = _bench_names(image_path)
    ious, pas = [], []
    with _pool() as pool:
        xs, gs = _decoded_set(pool, [(image_path, c), (gt_path, 1)], names) if names else (None, None)
        for i in range(0, len(names), batch_size):
            chunk = names[i:i + batch_size]
            gts = gs[i:i + batch_size, ..., 0]
            probs = model.predict_device(xs[i:i + batch_size])
            pred_d, counts = _ev.eval_multiclass(probs, gts, want_pred=create_images)
            pred = pred_d.cpu().numpy() if create_images else None
            n_px = int(gts.shape[1] * gts.shape[2])
            for j, n in enumerate(chunk):
                if create_images:
                    write_png_async(os.path.join(pred_path, n), pred[j])
                    convert_class_to_color_mask(pred[j], os.path.join(pred_path, f"{n[:-4]}_color.png"), class_to_color_mapping)
                pa_f, iou_f = _ev.pa_iou_from_counts(counts[j], n_px)
                pa = round(float(pa_f), 4)
                iou = round(float(iou_f), 4)
                pas.append(pa); ious.append(iou)
                if print_results:
                    print(f"{n} IoU: {iou}    PA: {pa}")
    ious, pas = _gather_lists(ious, pas)
    mPA = round(float(np.sum(pas) / len(pas)), 3)
    mIoU = round(float(np.sum(ious) / len(ious)), 3)
    print(f"------------------------------------------------------------   mPA: {mPA}      mIoU: {mIoU}  "
          "------------------------------------------------------------")
    return mPA, mIoU


def train_multiclass(train_images_dir, val_images_dir, val_masks_dir, test_images_dir, test_masks_dir,
                     unlabeled_images_dir, unlabeled_masks_dir, modelname, filepath_h5, model, loss_func, steps_per_epoch,
                     h, w, c, n_classes, class_to_color_mapping, val_pred_dir, test_pred_dir, unlabeled_pred_dir,
                     print_results=False):
    """functions.py:275-316.  loss_func: anything (the SUIM / Cityscapes scripts pass CategoricalCrossentropy());
    the fused loss kernel implements exactly that loss on class-id masks."""
    _reject_untrainable(loss_func, "train_multiclass")
    files = _train_shard(glob.glob(os.path.join(train_images_dir, "*.png")))
    loader = _EpochLoader(files, lambda p: parse_image_multiclass(p, n_classes, c), BATCH_SIZE, SEED, ("multi", n_classes, c))
    val_files = sorted(glob.glob(os.path.join(val_images_dir, "*.png")))
    best = {"miou": -1.0}

    def on_epoch_end(ep, loss):      # ModelCheckpoint(monitor='val_mean_io_u', mode='max')
        metric = MeanIoU(n_classes)
        key = ("multi", val_images_dir, c, tuple(val_files))
        with _CACHE_LOCK:
            if key not in _VAL_CACHE:      # the validation set is decoded once and stays on the device over the epochs
                _VAL_CACHE.clear()
                with _pool() as pool:
                    items = list(pool.map(lambda p: parse_image_multiclass(p, n_classes, c), val_files))
                _VAL_CACHE[key] = _uploaded(torch.from_numpy(np.stack([it[0] for it in items], 0)).cuda(),
                                            torch.from_numpy(np.stack([it[1] for it in items], 0)).cuda())
            xs, ys = _used_here(_VAL_CACHE[key])
        for i in range(0, len(val_files), BATCH_SIZE):      # per-batch IoU, averaged over the batches (functions.py:82-90)
            metric.update_state(ys[i:i + BATCH_SIZE], model.predict_device(xs[i:i + BATCH_SIZE]))
        v = metric.result()
        if v > best["miou"]:
            best["miou"] = v
            if _rank_world()[0] == 0:
                save_model(model, filepath_h5)

    fit(model, loader, steps_per_epoch, NUM_EPOCHS, 1, on_epoch_end)
    d = _dist()
    if d:
        d.barrier()
    best_model = load_model(filepath_h5, custom_objects={"MeanIoU": MeanIoU})
    mPA_val, mIoU_val = benchmark_multiclass(best_model, val_images_dir, val_masks_dir, val_pred_dir, h, w, c,
                                             class_to_color_mapping, print_results=print_results)
    mPA_test, mIoU_test = benchmark_multiclass(best_model, test_images_dir, test_masks_dir, test_pred_dir, h, w, c,
                                               class_to_color_mapping, print_results=print_results)
    mPA_unl, mIoU_unl = benchmark_multiclass(best_model, unlabeled_images_dir, unlabeled_masks_dir, unlabeled_pred_dir,
                                             h, w, c, class_to_color_mapping, print_results=print_results)
    print(f"{modelname} mIoU_val: {mIoU_val}")
    flush_writes()
    return mPA_val, mPA_test, mPA_unl, mIoU_val, mIoU_test, mIoU_unl


# ---------------------------------------------------------------------------------------------------
# HeLa (functions.py:232-269, 980-1018, 1155-1260, 2895-2984, 6181-6371).  The position post-processing is
# contour tracing + circle drawing in OpenCV in the reference; here scipy.ndimage connected components and an
# explicit disc rasteriser stand in (host side, unpinned -- SURVEY 8a' marks this row so).
# ---------------------------------------------------------------------------------------------------
def _erode3(mask):
    from scipy import ndimage
    return ndimage.grey_erosion(mask, size=(3, 3), mode="constant", cval=255)


def _u8_plane(img):
    """one contiguous uint8 plane of a 2-D / [h, w, 1] / [h, w, 3] image (functions.py:6186-6194: BGR -> grey)"""
    a = np.asarray(img)
    assert a.ndim in (2, 3), "Invalid image dimensions."
    if a.ndim == 3:
        a = a[..., 0] if a.shape[2] == 1 else (0.114 * a[..., 0] + 0.587 * a[..., 1] + 0.299 * a[..., 2]).astype(np.uint8)
    return np.ascontiguousarray(a, dtype=np.uint8)


def get_pos_contours(img, erode_kernel=3):
    """functions.py:6181-6218: centres (x, y) of the blobs of a position mask.  The reference erodes, thresholds at 10,
    takes cv2.findContours + cv2.moments of every contour and reports (int(m10 / m00) + 1, int(m01 / m00) + 1), skipping
    contours whose polygon area m00 is zero (single pixels, one-pixel-wide lines); RETR_TREE also reports every HOLE of a blob
    as a contour of its own and the reference's loop adds a position for each.  Host C++ (csrc/imk_geom.cpp imk_pos_contours:
    connected components, Moore-neighbour border tracing, Green's-theorem polygon moments; the interpreter lock is released
    around the call), checked bit for bit against the numpy / scipy restatement in oracle/hela_geometry.py.  Unpinned: OpenCV
    is not available to the reference in this environment."""
    a = _u8_plane(img)
    k = int(erode_kernel)
    if k > 1 and k % 2 == 0:          # an even window has no centre: scipy's placement, then the native path without erosion
        from scipy import ndimage
        a, k = np.ascontiguousarray(ndimage.grey_erosion(a, size=(k, k), mode="constant", cval=255)), 0
    cap = 256
    while True:
        xy = np.empty((cap, 2), np.int32)
        n = lib.imk_pos_contours(a.ctypes.data, a.shape[0], a.shape[1], max(k, 0), xy.ctypes.data, cap)
        if n < 0:
            check(n, "imk_pos_contours")
        if n <= cap:
            return [(int(x), int(y)) for x, y in xy[:n]]
        cap = n


def get_min_dist(xy, positions):
    """functions.py:6221-6252."""
    d = np.linalg.norm(np.array(positions) - np.array(xy), axis=1)
    d = d[d > 0]
    return 0 if d.size == 0 else float(np.min(d))


def _redraw_positions(gray_img, max_r, min_r, lone_dist, blur2):
    a = _u8_plane(gray_img)
    out = np.empty(a.shape, np.uint8)
    check(lib.imk_mod_pos_size(a.ctypes.data, a.shape[0], a.shape[1], int(max_r), int(min_r), int(lone_dist), int(blur2),
                               out.ctypes.data), "imk_mod_pos_size")
    return out


def mod_pos_size(gray_img, max_pos_circle_size=8, min_pos_circle_size=3):
    """functions.py:6255-6292: redraw every position blob as a filled circle of radius clamp(min_dist // 4, 3, 8), then
    `cv2.blur(out, (2, 2))` and `out[out < 254] = 0`: a pixel survives iff its whole 2x2 window (itself, left, upper,
    upper-left neighbour; BORDER_REFLECT_101 at the image edge) is set.  Host C++ (imk_mod_pos_size)."""
    return _redraw_positions(gray_img, max_pos_circle_size, min_pos_circle_size, 0, 1)


def get_cell_count(positions, img_alive, img_dead, measuring_range=3):
    """functions.py:6298-6371.  Host C++ (imk_cell_count) on one grey plane per image: [h, w], [h, w, 1] or BGR [h, w, 3] (converted as
    the reference does, functions.py:6321-6338).  A position whose clamped window (functions.py:6346-6356) still leaves the image --
    where the reference's slices would wrap or come out empty -- is IMK_EINVAL here, as is an image smaller than the window."""
    a, d = _u8_plane(img_alive), _u8_plane(img_dead)
    if a.shape != d.shape:
        raise ValueError(f"get_cell_count: alive {a.shape} and dead {d.shape} masks differ in size")
    xy = np.ascontiguousarray(np.asarray(positions, np.int32).reshape(-1, 2))
    counts = np.zeros(3, np.int32)
    check(lib.imk_cell_count(xy.ctypes.data, xy.shape[0], a.ctypes.data, d.ctypes.data, a.shape[0], a.shape[1],
                                   int(measuring_range), counts.ctypes.data), "imk_cell_count")
    return int(counts[0]), int(counts[1]), int(counts[2])



def parse_image_hela(path_brightfield, IMG_CHANNELS=1, Position_weight=3):
    """functions.py:980-1018: targets alive/dead in {0,1}, position in {0, Position_weight}."""
    bf = read_png(path_brightfield, IMG_CHANNELS)
    chans = []
    for name, wgt in (("alive", 1), ("dead", 1), ("mod_position", Position_weight)):
        m = read_png(re.sub("brightfield", name, path_brightfield), 1)[..., 0]
        chans.append(((m // 255) * wgt).astype(np.uint8))
    return bf, np.stack(chans, -1)


def create_pseudo_labels_im_hela(models, h, w, c, images_path, main_output_path, erode_kernel=5, dilate_kernel=5,
                                 block_input=True, block_output=True, max_pos_circle_size=8, min_pos_circle_size=3):
    """functions.py:2895-2984: three binary IMs (>=), combined IM = max; position mask re-drawn as discs on the host;
    brightfield / alive / dead / mod_position / im written."""
    out = {k: os.path.join(main_output_path, k) for k in ("brightfield", "alive", "dead", "mod_position", "im")}
    for d in out.values():
        os.makedirs(d, exist_ok=True)
    mine = shard_list(os.listdir(images_path))
    ens = _ensemble(models, True)
    sum_im = count = 0
    with _pool() as pool:
        for i, j in infer_batches(len(mine), infer_batch_size(_models_alpha(models))):
            chunk = mine[i:j]
            imgs = read_png_stack(pool, [os.path.join(images_path, n) for n in chunk], c)
            x = torch.from_numpy(imgs).cuda()
            r = ens.run(x, 0.5, True, False, False)          # blocking happens after the host-side position step
            im, masks = r["im"], r["masks"]
            if erode_kernel > 0:       # functions.py:2940-2946: erode the combined IM, dilate_mask on alive and dead
                im = _im.morph(im, erode_kernel, "erode")
                ad = _im.morph(masks[:, :2].reshape(-1, h, w).contiguous(), 3, "dilate").reshape(-1, 2, h, w)
                masks = torch.cat([ad, masks[:, 2:]], 1)     # contours come from the RAW position mask (:2952)
            if dilate_kernel > 0:
                im = _im.morph(im, dilate_kernel, "dilate")
            masks_np, im_np = masks.cpu().numpy(), im.cpu().numpy()
            ims = r["im_size"].sum(1).cpu().numpy()
            sum_im += int(ims.sum()); count += len(chunk)

            def one(j):      # functions.py:2952-2966: circles from the RAW position mask, a lone cell drawn with distance 99
                pos = _redraw_positions(masks_np[j, 2], max_pos_circle_size, min_pos_circle_size, 99, 0)
                pos = np.repeat(pos[..., None], 3, 2)
                bf, alive, dead = imgs[j][..., 0].copy(), masks_np[j, 0].copy(), masks_np[j, 1].copy()
                hit = im_np[j] > 0
                if block_input:
                    bf[hit] = 0
                if block_output:
                    alive[hit] = 0; dead[hit] = 0; pos[hit] = 0
                name = chunk[j]
                return [(os.path.join(out["brightfield"], name), bf), (os.path.join(out["alive"], name), alive),
                        (os.path.join(out["dead"], name), dead), (os.path.join(out["mod_position"], name), pos),
                        (os.path.join(out["im"], name), im_np[j])]
            jobs = [job for per_image in pool.map(one, range(len(chunk))) for job in per_image]
            write_pngs_async(jobs)      # encoded while the next batch is decoded and run
    flush_writes()
    tot_im, tot_n = _all_reduce_sum([sum_im, count])
    return round(tot_im / tot_n, 0) if tot_n else 0.0


_HELA_GT_COUNTS = {}      # id(position-mask tensor of a cached decoded set) -> (weak reference to it, [(alive, dead)] ground-truth cell counts)


def _hela_gt_cell_counts(pool, gp_dev, gp, ga, gd):
    """get_cell_count(get_pos_contours(gt position), gt alive, gt dead) of every image of a benchmark directory: a property of
    the files, computed once per decoded set (the candidates of a generation score against the same three directories)"""
    import weakref
    with _CACHE_LOCK:
        for k in [k for k, (ref, _) in _HELA_GT_COUNTS.items() if ref() is None]:
            del _HELA_GT_COUNTS[k]
        hit = _HELA_GT_COUNTS.get(id(gp_dev))
        if hit is None or hit[0]() is not gp_dev:
            hit = (weakref.ref(gp_dev),
                   list(pool.map(lambda j: get_cell_count(get_pos_contours(gp[j]), ga[j], gd[j])[:2], range(len(gp)))))
            _HELA_GT_COUNTS[id(gp_dev)] = hit
        return hit[1]


def benchmark_hela(model, gt_main_dir, pred_dir, h, w, c, threshold=0.5, batch_size=64, save_output=True, benchmark=True,
                   mod_position=True):
    """functions.py:1155-1260: returns (mIoU, mIoU_ad, mean_cell_count_error).  Thresholds and the alive / dead pixel counts run
    in imk_eval_binary (the ratios are formed on the host with the reference's float expression); the position map goes through
    the host geometry (mod_pos_size, get_pos_contours, get_cell_count: libimk, one image per pool thread)."""
    sub = "mod_position" if mod_position else "position"
    for k in ("alive", "dead", sub):
        os.makedirs(os.path.join(pred_dir, k), exist_ok=True)
    names = _bench_names(os.path.join(gt_main_dir, "brightfield"))
    mious, mious_ad, delta = [], [], 0
    iou = lambda cnt: round(float(np.int64(cnt[0]) / (np.int64(cnt[1]) + 1e-7)), 4)      # get_IoU_binary (functions.py:1767-1788)
    with _pool() as pool:
        dirs = [("brightfield", c)] + ([("alive", 1), ("dead", 1), ("mod_position", 1)] if benchmark else [])
        sets = _decoded_set(pool, [(os.path.join(gt_main_dir, k), ch) for k, ch in dirs], names) if names else None
        if names and benchmark:
            ga_all, gd_all, gp_all = (t[..., 0].cpu().numpy() for t in sets[1:])
            gt_counts = _hela_gt_cell_counts(pool, sets[3], gp_all, ga_all, gd_all)
        for i in range(0, len(names), batch_size):
            chunk = names[i:i + batch_size]
            probs = model.predict_device(sets[0][i:i + batch_size])
            preds, counts = [], []
            for k in range(3):
                gt_k = sets[1 + k][i:i + batch_size, ..., 0] if benchmark else torch.zeros_like(probs[..., k], dtype=torch.uint8)
                pr, cnt = _ev.eval_binary(probs[..., k].contiguous(), gt_k, threshold, False, want_pred=True)
                preds.append(pr.cpu().numpy()); counts.append(cnt)

            def one(j):
                a_u, d_u, p_u = preds[0][j], preds[1][j], preds[2][j]
                if mod_position:
                    p_u = mod_pos_size(p_u)
                score = None
                if benchmark:
                    ia, idd = iou(counts[0][j]), iou(counts[1][j])
                    ip = round(float(get_IoU_binary(gp_all[i + j], p_u)), 4) if mod_position else iou(counts[2][j])
                    pa, pd, _ = get_cell_count(get_pos_contours(p_u), a_u, d_u)
                    qa, qd = gt_counts[i + j]
                    score = ((ia + idd + ip) / 3, (ia + idd) / 2, abs(pa - qa) + abs(pd - qd))
                return score, (a_u, d_u, p_u)
            for n, (score, (a_u, d_u, p_u)) in zip(chunk, pool.map(one, range(len(chunk)))):
                if score is not None:
                    mious.append(score[0]); mious_ad.append(score[1]); delta += score[2]
                if save_output:
                    write_png_async(os.path.join(pred_dir, "alive", n), a_u)
                    write_png_async(os.path.join(pred_dir, "dead", n), d_u)
                    write_png_async(os.path.join(pred_dir, sub, n), p_u)
    mious, mious_ad, deltas = _gather_lists(mious, mious_ad, [delta])
    delta = sum(deltas)
    return (round(float(np.sum(mious) / len(mious)), 3), round(float(np.sum(mious_ad) / len(mious_ad)), 3),
            round(delta / len(mious), 3))


def train_hela(train_images_dir, val_images_dir, val_gt_dir, test_gt_dir, unlabeled_gt_dir, modelname, filepath_h5, model,
               loss_func, steps_per_epoch, h, w, c, val_pred_dir, test_pred_dir, unlabeled_pred_dir):
    """functions.py:232-269: 'mse' on (alive, dead, 3 x position) targets, best epoch by val_loss (min)."""
    if loss_func != "mse":
        raise NotImplementedError("the HeLa scripts train with 'mse'")
    files = _train_shard(glob.glob(os.path.join(train_images_dir, "*.png")))
    loader = _EpochLoader(files, lambda p: parse_image_hela(p, c), BATCH_SIZE, SEED, ("hela", c))
    val_files = sorted(glob.glob(os.path.join(val_images_dir, "*.png")))
    best = {"loss": float("inf")}

    def on_epoch_end(ep, loss):      # ModelCheckpoint(monitor='val_loss', mode='min'): Keras' sample-weighted mean of the mse
        key = ("hela", val_images_dir, c, tuple(val_files))
        with _CACHE_LOCK:
            if key not in _VAL_CACHE:
                _VAL_CACHE.clear()
                with _pool() as pool:
                    items = list(pool.map(lambda p: parse_image_hela(p, c), val_files))
                _VAL_CACHE[key] = _uploaded(torch.from_numpy(np.stack([it[0] for it in items], 0)).cuda(),
                                            torch.from_numpy(np.stack([it[1] for it in items], 0)).cuda())
            xs, ys = _used_here(_VAL_CACHE[key])
        sq, n = 0.0, 0
        for i in range(0, len(val_files), BATCH_SIZE):
            y = ys[i:i + BATCH_SIZE]
            sq += _ev.soft_sums(model.predict_device(xs[i:i + BATCH_SIZE]), y, 1)
            n += y.numel()
        val_loss = sq / max(n, 1)
        if val_loss < best["loss"]:
            best["loss"] = val_loss
            if _rank_world()[0] == 0:
                save_model(model, filepath_h5)

    fit(model, loader, steps_per_epoch, NUM_EPOCHS, 0, on_epoch_end)
    d = _dist()
    if d:
        d.barrier()
    best_model = load_model(filepath_h5)
    res = []
    for gt, pd in ((val_gt_dir, val_pred_dir), (test_gt_dir, test_pred_dir), (unlabeled_gt_dir, unlabeled_pred_dir)):
        res += list(benchmark_hela(best_model, gt, pd, h, w, c))
    print(f"{modelname} mIoU_val: {res[0]}   mean_cell_count_error_val: {res[2]}")
    flush_writes()
    return tuple(res)


# ---------------------------------------------------------------------------------------------------
# EvalNet call sites of the IM++ / AIM++ drivers (functions.py:3572-3670, 3881-4006, 4464-4506, 4673-4722, 5684-5757,
# 5837-5941)
# ---------------------------------------------------------------------------------------------------
from .evalnet_functions import (compute_classwise_detection, compute_classwise_detection_im,  # noqa: E402,F401
                                compute_classwise_IoU, create_augment_images_and_masks_with_evalnet_ensemble_binary,
                                create_augment_images_and_masks_with_evalnet_ensemble_hela,
                                create_augment_images_and_masks_with_evalnet_ensemble_multiclass,
                                create_augment_images_and_masks_with_gt,
                                create_training_data_evalnet_im_binary, create_training_data_evalnet_miou_im_hela,
                                create_training_data_evalnet_miou_im_multiclass, load_evalnet, num_augs_from_miou,
                                save_evalnet, train_evalnet_ISIC_2018, train_evalnet_miou_model_hela,
                                train_evalnet_miou_model_multiclass)
