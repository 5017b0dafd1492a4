"""Host-side mirror of the hot-path part of the reference's functions.py, on top of libimk.so.

Same function names, positional arguments, return values, output directory layout and file names as the
reference (file:line cited per function), so that the per-dataset `*_IM.py` drivers run unchanged from
the user's point of view.  What differs is underneath: images are processed in batches on the GPU
(ensemble forward + fused IM kernel, training step kernels), PNG I/O uses Pillow on a host thread pool
(OpenCV is not available), and model files are safetensors (h5py/Keras are not available).

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

INFER_BATCH = int(os.environ.get("IMK_INFER_BATCH", 128))
_IO_THREADS = int(os.environ.get("IMK_IO_THREADS", 8))


# ---------------------------------------------------------------------------------------------------
# distributed helpers
# ---------------------------------------------------------------------------------------------------
def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


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
def read_png(path, channels):
    """uint8 [H,W,C] in RGB order (C=3) or [H,W,1] (C=1).  The reference reads BGR with cv2 and converts
    to RGB for the net (functions.py:2846-2848); on-disk channel order is preserved either way."""
    with Image.open(path) as im:
        im = im.convert("RGB" if channels == 3 else "L")
        a = np.asarray(im, dtype=np.uint8)
    return a if a.ndim == 3 else a[..., None]


def write_png(path, arr):
    arr = np.asarray(arr, dtype=np.uint8)
    if arr.ndim == 3 and arr.shape[2] == 1:
        arr = arr[..., 0]
    Image.fromarray(arr).save(path, format="PNG", compress_level=1)


def _pool():
    return ThreadPoolExecutor(max_workers=_IO_THREADS)


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
            if not m._packed_ok:
                m.repack()
        import ctypes
        n = len(models)
        self._params = (ctypes.c_void_p * n)(*[m.params.data_ptr() for m in models])
        self._packed = (ctypes.c_void_p * n)(*[m.packed.data_ptr() for m in models])
        self._ws = None
        self._ws_batch = 0

    def run(self, x_u8, thr=0.5, cmp_ge=False, block_in=True, block_out=True, want_presence=False):
        """x_u8 [B,H,W,C] uint8 device -> dict(img_out, masks [B,Kb,H,W] | final [B,H,W], im, im_size, pred_size, presence)"""
        p = self.plan
        b = x_u8.shape[0]
        n = len(self.models)
        dev = x_u8.device
        if self._ws is None or self._ws_batch < b:
            per = ((b * p.h * p.w * p.n_out * 4 + 255) // 256) * 256
            self._ws = torch.empty(per * n + p.workspace_bytes(b, 0), dtype=torch.uint8, device=dev)
            self._ws_batch = b
        binary = p.act_out == "sigmoid"
        kb = p.n_out if binary else 1
        masks = torch.empty((b, kb, p.h, p.w), dtype=torch.uint8, device=dev)
        im = torch.empty((b, p.h, p.w), dtype=torch.uint8, device=dev)
        im_size = torch.empty((b, kb), dtype=torch.int64, device=dev)
        pred_size = torch.zeros((b, kb), dtype=torch.int64, device=dev)
        presence = torch.empty((n, b, p.n_out), dtype=torch.uint8, device=dev) if (want_presence and not binary) else None
        img_out = torch.empty_like(x_u8)
        check(lib.imk_unet_forward_im(p.ptr, n, self._params, self._packed, x_u8.data_ptr(), b, float(thr),
                                      int(bool(cmp_ge)), x_u8.data_ptr(), int(bool(block_in)), int(bool(block_out)),
                                      img_out.data_ptr(), masks.data_ptr(), im.data_ptr(), im_size.data_ptr(),
                                      pred_size.data_ptr(), 0 if presence is None else presence.data_ptr(),
                                      self._ws.data_ptr(), self._ws.numel(), _stream()), "imk_unet_forward_im")
        return {"img_out": img_out, "masks": masks, "im": im, "im_size": im_size, "pred_size": pred_size,
                "presence": presence}


def _morph_then_block(r, erode_kernel, dilate_kernel, block_input, block_output, x_u8, dilate_masks=False):
    """Cold path (EK/DK > 0, dead in every shipped config): morphology on the IM, then blocking with the
    modified IM (functions.py:2858-2874 order)."""
    im = r["im"]
    masks = r["masks"]
    if erode_kernel > 0:
        im = _im.morph(im, erode_kernel, "erode")
    if dilate_kernel > 0:
        im = _im.morph(im, dilate_kernel, "dilate")
    img = x_u8.clone()
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
                flag):
    """Shared body of the three writers.  kind in {'isic', 'multi'}."""
    names = os.listdir(images_path)
    mine = shard_list(names)
    ens = EnsembleIM(models)
    fused_block = (erode_kernel <= 0 and dilate_kernel <= 0)
    sum_im, count = 0, 0
    with _pool() as pool:
        for i in range(0, len(mine), INFER_BATCH):
            chunk = mine[i:i + INFER_BATCH]
            imgs = list(pool.map(lambda n: read_png(os.path.join(images_path, n), c), chunk))
            x = torch.from_numpy(np.stack(imgs, 0)).cuda()
            r = ens.run(x, THRESHOLD, False, block_input and fused_block, block_output and fused_block,
                        want_presence=(kind == "multi" and flag))
            if fused_block:
                img_out, masks, im = r["img_out"], r["masks"], r["im"]
            else:
                img_out, masks, im = _morph_then_block(r, erode_kernel, dilate_kernel, block_input, block_output, x)
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
            list(pool.map(lambda a: write_png(*a), jobs))
    tot_im, tot_n = _all_reduce_sum([sum_im, count])
    return round(tot_im / tot_n, 0) if tot_n else 0.0


def create_pseudo_labels_im_ISIC_2018(models, h, w, c, images_path, main_output_path, rgb=True, erode_kernel=5,
                                      dilate_kernel=5, block_input=True, block_output=True,
                                      filter_bad_predictions=True):
    """functions.py:2832-2891.  Writes images/ masks/ im/ under main_output_path, returns mean_im_size."""
    out = {k: os.path.join(main_output_path, k) for k in ("images", "masks", "im")}
    for d in out.values():
        os.makedirs(d, exist_ok=True)
    if not rgb and c == 3:
        raise NotImplementedError("rgb=False (feeding BGR to the net) is not used by any reference script")
    return _run_writer(models, h, w, c, images_path, out, "isic", erode_kernel, dilate_kernel, block_input,
                       block_output, filter_bad_predictions)


def create_pseudo_labels_im_multiclass(models, h, w, c, images_path, main_output_path, rgb=True, erode_kernel=5,
                                       dilate_kernel=5, block_input=True, block_output=True,
                                       filter_unequal_class_pred=False):
    """functions.py:2988-3070.  With erode_kernel > 0 the reference also dilates the label map per class
    (functions.py:3047, dilate_mask) -- not implemented on the GPU path (EK = 0 in every shipped config)."""
    if erode_kernel > 0:
        raise NotImplementedError("per-class dilate_mask (erode_kernel > 0) is dead in the shipped configs")
    out = {k: os.path.join(main_output_path, k) for k in ("images", "masks", "im")}
    for d in out.values():
        os.makedirs(d, exist_ok=True)
    if not rgb and c == 3:
        raise NotImplementedError("rgb=False is not used by any reference script")
    return _run_writer(models, h, w, c, images_path, out, "multi", erode_kernel, dilate_kernel, block_input,
                       block_output, filter_unequal_class_pred)


# ---------------------------------------------------------------------------------------------------
# metrics (functions.py:162-184, 1767-1861)
# ---------------------------------------------------------------------------------------------------
def dice_loss(y_true, y_pred, smooth=1):
    """functions.py:162-184 on torch tensors [B,H,W,K]."""
    y_true = torch.as_tensor(y_true).float()
    y_pred = torch.as_tensor(y_pred).float()
    inter = (y_true * y_pred).sum(dim=(1, 2, 3))
    union = y_true.sum(dim=(1, 2, 3)) + y_pred.sum(dim=(1, 2, 3))
    return 1 - ((2 * inter + smooth) / (union + smooth)).mean()


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


class _EpochLoader:
    """list_files(seed).map(parse).batch(B).repeat(): seeded shuffle per pass, one short batch per pass
    (functions.py:207-209), PNG decode on a thread pool, whole set cached on the device after the first pass."""

    def __init__(self, files, parse, batch, seed):
        self.files = sorted(files)
        self.parse = parse
        self.batch = batch
        self.rng = np.random.default_rng(seed)
        self.x = self.y = None
        self._order = []

    def _load(self):
        with _pool() as pool:
            items = list(pool.map(self.parse, self.files))
        self.x = torch.from_numpy(np.stack([i[0] for i in items], 0)).cuda()
        self.y = torch.from_numpy(np.stack([i[1] for i in items], 0)).cuda()

    def next_batch(self):
        if self.x is None:
            self._load()
        if not self._order:
            perm = self.rng.permutation(len(self.files))
            self._order = [perm[i:i + self.batch] for i in range(0, len(perm), self.batch)]
        idx = torch.as_tensor(self._order.pop(0), device="cuda")
        return self.x[idx].contiguous(), self.y[idx].contiguous()


def _grad_allreduce(model):
    d = _dist()
    if not d:
        return 1.0
    d.all_reduce(model.grads)       # one flat fp32 bucket per step (RCCL over xGMI)
    d.all_reduce(model.stats)       # loss and found_inf ride along (4 floats)
    return 1.0 / d.get_world_size()


def fit(model, loader, steps_per_epoch, epochs, loss_kind, on_epoch_end=None, lr=None, wd=None):
    """model.fit(train_dataset, epochs, steps_per_epoch) of functions.py:218."""
    lr = LR if lr is None else lr
    wd = WD if wd is None else wd
    model.init_train_state()
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
        if on_epoch_end:
            on_epoch_end(ep, history[-1])
    return history


def save_model(model, path):
    from safetensors.torch import save_file
    sd = {k: v.contiguous() for k, v in model.state_dict().items()}
    meta = {"h": str(model.plan.h), "w": str(model.plan.w), "c_in": str(model.plan.c_in),
            "n_out": str(model.plan.n_out), "alpha": repr(model.plan.alpha), "act_out": model.plan.act_out}
    save_file(sd, path, metadata=meta)


def load_model(path, custom_objects=None, device="cuda"):
    """Counterpart of tf.keras.models.load_model(path, custom_objects=...) (ISIC_2018/09_ISIC_2018_IM.py:75)."""
    from safetensors import safe_open
    with safe_open(path, framework="pt") as f:
        meta = f.metadata()
        sd = {k: f.get_tensor(k) for k in f.keys()}
    m = UNet(int(meta["h"]), int(meta["w"]), int(meta["c_in"]), int(meta["n_out"]), float(meta["alpha"]),
             meta["act_out"], device=device)
    m.load_state_dict(sd)
    return m


def _binary_iou_dataset(model, images_dir, masks_dir, c, batch=64):
    """Keras BinaryIoU(target_class_ids=[1], threshold=0.5) accumulated over the whole directory (the
    val_binary_io_u monitor of functions.py:216-217)."""
    files = sorted(glob.glob(os.path.join(images_dir, "*.png")))
    tp = fp = fn = 0
    with _pool() as pool:
        for i in range(0, len(files), batch):
            chunk = files[i:i + batch]
            items = list(pool.map(lambda p: parse_image_ISIC_2018(p, c), chunk))
            x = torch.from_numpy(np.stack([it[0] for it in items], 0)).cuda()
            y = torch.from_numpy(np.stack([it[1] for it in items], 0)).cuda() > 0
            p = model.predict_device(x) >= 0.5
            tp += int((p & y).sum()); fp += int((p & ~y).sum()); fn += int((~p & y).sum())
    return tp / max(tp + fp + fn, 1)


def benchmark_ISIC2018(model, images_dir, masks_dir, pred_path, h, w, c, batch_size=64, create_images=True,
                       print_results=False):
    """functions.py:1078-1151: batch-64 predict, > 0.5, optional PNG dump, per-image IoU/Dice rounded to 4,
    means rounded to 3."""
    os.makedirs(pred_path, exist_ok=True)
    names = os.listdir(images_dir)
    ious, dices = [], []
    with _pool() as pool:
        for i in range(0, len(names), batch_size):
            chunk = names[i:i + batch_size]
            imgs = list(pool.map(lambda n: read_png(os.path.join(images_dir, n), c), chunk))
            gts = list(pool.map(lambda n: read_png(os.path.join(masks_dir, n), 1)[..., 0], chunk))
            probs = model.predict_device(torch.from_numpy(np.stack(imgs, 0)).cuda())
            pred = ((probs > 0.5) * 255).to(torch.uint8).cpu().numpy()[..., 0]
            if create_images:
                list(pool.map(lambda a: write_png(*a), [(os.path.join(pred_path, n), pred[j]) for j, n in enumerate(chunk)]))
            for j, n in enumerate(chunk):
                d = round(float(dice_score_numpy_binary(gts[j], pred[j])), 4)
                u = round(float(get_IoU_binary(gts[j], pred[j])), 4)
                dices.append(d); ious.append(u)
                if print_results:
                    print(f"{n} IoU: {u}    DS: {d}")
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
    files = shard_list(glob.glob(os.path.join(train_images_dir, "*.png")))
    loader = _EpochLoader(files, lambda p: parse_image_ISIC_2018(p, c), BATCH_SIZE, SEED)
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
    return mIoU_val, mIoU_test, mIoU_unl, dice_val, dice_test, dice_unl
