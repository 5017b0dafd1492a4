"""Host-side mirror of the reference's EvalNet call sites for the HeLa IM++ / AIM++ drivers (HeLa/12_HeLa_IM++.py,
HeLa/14_HeLa_aug_IM++.py): same function names, positional arguments, directory layout, file names and CSV format.
Ensemble inference + IM, EvalNet inference / training, morphology and augmentation run on the GPU through libimk.so;
PNG I/O and the CSV bookkeeping stay on the host.  Re-exported by functions.py."""
import csv
import os
import random

import numpy as np
import torch

from . import im as _im
from .augment import augment_batch, draw_params
from .evalnet import EvalNet


def _F():
    from . import functions
    return functions


def _draw_plan(rng, n_images, n_models, n_min_models, n_max_models):
    """per image, in the reference's order of draws (functions.py:3623-3640, 3931-3953): size and members of the
    sub-ensemble, erode and dilate kernel from {0, 3, 5}, the augment-or-not coin"""
    n_max = min(n_max_models, n_models)
    n_min = min(n_min_models, n_max)
    plan = []
    for _ in range(n_images):
        n_sel = rng.randint(n_min, n_max)
        subset = tuple(sorted(rng.sample(range(n_models), n_sel)))
        plan.append((subset, rng.choice([0, 3, 5]), rng.choice([0, 3, 5]), rng.random() < 0.5))
    return plan


def _random_morph(im, plan, idx):
    """erode, then dilate each image's IM with its own kernel (functions.py:3630-3638)"""
    for op, col in (("erode", 1), ("dilate", 2)):
        ks = torch.tensor([plan[i][col] for i in idx], device="cuda")
        for k in (3, 5):
            sel = torch.nonzero(ks == k).flatten()
            if sel.numel():
                im[sel] = _im.morph(im[sel].contiguous(), k, op)
    return im


def _groups(plan):
    g = {}
    for i, p in enumerate(plan):
        g.setdefault(p[0], []).append(i)
    return g


# ---------------------------------------------------------------------------------------------------
# training data of the binary EvalNet from ensemble IM predictions (functions.py:3572-3670)
# ---------------------------------------------------------------------------------------------------
def create_training_data_evalnet_im_binary(models, h, w, c, images_path, masks_path, main_output_path, num_loops,
                                           n_min_models=2, n_max_models=4, rgb=True, brightness_range_alpha=(0.6, 1.4),
                                           brightness_range_beta=(-20, 20), max_blur=3, max_noise=20, free_rotation=False,
                                           seed=None):
    """Per loop and labelled image: a random sub-ensemble -> binary IM (>), randomly eroded / dilated -> blocked image +
    mask, IoU of the blocked mask against the ground truth (rounded to 4 decimals) as the label; half of the samples are
    written augmented (geometry on image and mask, photometry on the image).  `{stem}_aug_{loop}.png`, labels.csv."""
    F = _F()
    flip = (not rgb) and c == 3      # rgb=False (functions.py:3611 ff.): the nets see the file's channels in OpenCV's order (BGR)
    rng = random.Random(F.SEED if seed is None else seed)
    np_rng = np.random.default_rng(rng.getrandbits(32))
    iout, mout = os.path.join(main_output_path, "images"), os.path.join(main_output_path, "masks")
    os.makedirs(iout, exist_ok=True)
    os.makedirs(mout, exist_ok=True)
    names = sorted(os.listdir(images_path))
    draw_kw = dict(brightness_range_alpha=brightness_range_alpha, brightness_range_beta=brightness_range_beta,
                   max_blur=max_blur, max_noise=max_noise, free_rotation=free_rotation, rng=rng, np_rng=np_rng)
    rows, ensembles = [], {}
    with F._pool() as pool:
        for nl in range(num_loops):
            plan = _draw_plan(rng, len(names), len(models), n_min_models, n_max_models)
            loop_rows = {}
            for subset, idx_all in _groups(plan).items():
                if subset not in ensembles:
                    ensembles[subset] = F.EnsembleIM([models[j] for j in subset])
                for s in range(0, len(idx_all), F.INFER_BATCH):
                    idx = idx_all[s:s + F.INFER_BATCH]
                    x = torch.from_numpy(F.read_png_stack(pool, [os.path.join(images_path, names[i]) for i in idx], c)).cuda()
                    gt = torch.from_numpy(F.read_png_stack(pool, [os.path.join(masks_path, names[i]) for i in idx], 1)).cuda()
                    r = ensembles[subset].run(x.flip(-1).contiguous() if flip else x, F.THRESHOLD, False, False, False)
                    im = _random_morph(r["im"], plan, idx)
                    masks, img = r["masks"], x.clone()
                    _im.block_apply(im, img, masks)
                    pred, g = masks[:, 0] > 0, gt[..., 0] > 0
                    inter = (pred & g).sum(dim=(1, 2)).cpu().numpy()
                    union = (pred | g).sum(dim=(1, 2)).cpu().numpy()
                    ious = inter / (union + 1e-7)
                    mk = masks.permute(0, 2, 3, 1).contiguous()
                    aug = torch.tensor([plan[i][3] for i in idx], device="cuda")
                    sel = torch.nonzero(aug).flatten()
                    if sel.numel():      # functions.py:3657-3658
                        o, om = augment_batch(img[sel].contiguous(), mk[sel].contiguous(), draw_params(int(sel.numel()), **draw_kw))
                        img[sel], mk[sel] = o, om
                    img_np, mk_np = img.cpu().numpy(), mk.cpu().numpy()
                    jobs = []
                    for row, i in enumerate(idx):
                        out_name = f"{names[i][:-4]}_aug_{nl}.png"
                        loop_rows[i] = (out_name, round(float(ious[row]), 4))
                        jobs.append((os.path.join(iout, out_name), img_np[row]))
                        jobs.append((os.path.join(mout, out_name), mk_np[row, :, :, 0]))
                    F.write_pngs_async(jobs)      # on the writer pool: the next batch's decode, network pass and augmentation run meanwhile
            rows += [loop_rows[i] for i in range(len(names))]
    F.flush_writes()
    with open(os.path.join(main_output_path, "labels.csv"), "a", encoding="utf-8", newline="") as f:
        wr = csv.writer(f, delimiter=";")
        for row in rows:
            wr.writerow(row)


# ---------------------------------------------------------------------------------------------------
# multiclass: label helpers (functions.py:4328-4358, 4400-4459) and training data (functions.py:3773-3877)
# ---------------------------------------------------------------------------------------------------
def compute_classwise_IoU(pred, gt, num_classes):
    """functions.py:4328-4358: IoU per class present in gt, rounded to 4 decimals; class 0 scores 1 as soon as the
    prediction has a class-0 pixel and gt has none of them (the prefill), else its IoU like the others."""
    pred, gt = np.asarray(pred), np.asarray(gt)
    iou_list = [0] * num_classes
    if (pred == 0).sum() > 0:
        iou_list[0] = 1
    for cls in range(num_classes):
        if cls in gt:
            inter = np.logical_and(gt == cls, pred == cls).sum()
            union = np.logical_or(gt == cls, pred == cls).sum()
            if union > 0:
                iou_list[cls] = round(inter / union, 4)
    return iou_list


def compute_classwise_detection(mask, num_classes):
    """functions.py:4400-4421: class present on more than 1 % of the pixels"""
    mask = np.asarray(mask)
    return [1 if (mask == cls).sum() > mask.size * 0.01 else 0 for cls in range(num_classes)]


def compute_classwise_detection_im(pred_mask, num_classes, gt_class_counts, threshold):
    """functions.py:4424-4459: class detected if its pixel count in the IM-blocked mask is at least `threshold` of its
    ground-truth count, or at least 10 % of the image; class 0 as soon as one pixel of it is left."""
    pred_mask = np.asarray(pred_mask)
    total = pred_mask.size
    out = [0] * num_classes
    for cls in range(num_classes):
        n = (pred_mask == cls).sum()
        ratio = 0 if gt_class_counts[cls] == 0 else n / gt_class_counts[cls]
        if cls == 0 and n > 0:
            out[cls] = 1
        elif ratio >= threshold:
            out[cls] = 1
        elif n / total >= 0.1:
            out[cls] = 1
    return out


def create_training_data_evalnet_miou_im_multiclass(models, h, w, c, num_classes, images_path, masks_path, main_output_path,
                                                    num_loops, n_min_models=2, n_max_models=4, rgb=True,
                                                    brightness_range_alpha=(0.8, 1.2), brightness_range_beta=(-10, 10),
                                                    max_blur=1, max_noise=10, free_rotation=False, seed=None):
    """Per loop and labelled image: a random sub-ensemble -> argmax agreement IM, randomly eroded / dilated -> blocked
    image + label map; labels = class-wise IoU of the blocked prediction against the ground truth and class-wise
    detection of the IM-blocked ground truth (threshold 0.3); half of the samples are written augmented."""
    F = _F()
    flip = (not rgb) and c == 3      # rgb=False (functions.py:3611 ff.): the nets see the file's channels in OpenCV's order (BGR)
    rng = random.Random(F.SEED if seed is None else seed)
    np_rng = np.random.default_rng(rng.getrandbits(32))
    iout, mout = os.path.join(main_output_path, "images"), os.path.join(main_output_path, "masks")
    os.makedirs(iout, exist_ok=True)
    os.makedirs(mout, exist_ok=True)
    names = sorted(os.listdir(images_path))
    draw_kw = dict(brightness_range_alpha=brightness_range_alpha, brightness_range_beta=brightness_range_beta,
                   max_blur=max_blur, max_noise=max_noise, free_rotation=free_rotation, rng=rng, np_rng=np_rng)
    rows, ensembles = [], {}
    with F._pool() as pool:
        for nl in range(num_loops):
            plan = _draw_plan(rng, len(names), len(models), n_min_models, n_max_models)
            loop_rows = {}
            for subset, idx_all in _groups(plan).items():
                if subset not in ensembles:
                    ensembles[subset] = F.EnsembleIM([models[j] for j in subset])
                for s in range(0, len(idx_all), F.INFER_BATCH):
                    idx = idx_all[s:s + F.INFER_BATCH]
                    x = torch.from_numpy(F.read_png_stack(pool, [os.path.join(images_path, names[i]) for i in idx], c)).cuda()
                    gts = list(pool.map(lambda i: F.read_png(os.path.join(masks_path, names[i]), 1)[..., 0], idx))
                    r = ensembles[subset].run(x.flip(-1).contiguous() if flip else x, F.THRESHOLD, False, False, False)
                    im = _random_morph(r["im"], plan, idx)
                    masks, img = r["masks"], x.clone()
                    _im.block_apply(im, img, masks)
                    pred_np, im_np = masks[:, 0].cpu().numpy(), im.cpu().numpy()
                    mk = masks.permute(0, 2, 3, 1).contiguous()
                    sel = torch.nonzero(torch.tensor([plan[i][3] for i in idx], device="cuda")).flatten()
                    if sel.numel():      # functions.py:3859-3860
                        o, om = augment_batch(img[sel].contiguous(), mk[sel].contiguous(), draw_params(int(sel.numel()), **draw_kw))
                        img[sel], mk[sel] = o, om
                    img_np, mk_np = img.cpu().numpy(), mk.cpu().numpy()
                    jobs = []
                    for row, i in enumerate(idx):
                        gt = gts[row]
                        ious = compute_classwise_IoU(pred_np[row], gt, num_classes)
                        counts = np.zeros(num_classes)
                        bins = np.bincount(gt.ravel(), minlength=num_classes)
                        counts[:len(bins)] += bins[:num_classes] if len(bins) > num_classes else bins
                        gt_blocked = gt.copy()
                        gt_blocked[im_np[row] > 0] = 0
                        det = compute_classwise_detection_im(gt_blocked, num_classes, counts, 0.3)
                        out_name = f"{names[i][:-4]}_aug_{nl}.png"
                        loop_rows[i] = (out_name, *ious, *det)
                        jobs.append((os.path.join(iout, out_name), img_np[row]))
                        jobs.append((os.path.join(mout, out_name), mk_np[row, :, :, 0]))
                    F.write_pngs_async(jobs)      # on the writer pool: the next batch's decode, network pass and augmentation run meanwhile
            rows += [loop_rows[i] for i in range(len(names))]
    F.flush_writes()
    with open(os.path.join(main_output_path, "labels.csv"), "a", encoding="utf-8", newline="") as f:
        wr = csv.writer(f, delimiter=";")
        for row in rows:
            wr.writerow(row)


# ---------------------------------------------------------------------------------------------------
# training data of the mIoU EvalNet from ensemble IM predictions (functions.py:3881-4006)
# ---------------------------------------------------------------------------------------------------
def create_training_data_evalnet_miou_im_hela(models, h, w, c, main_input_path, main_output_path, num_loops, n_min_models=2,
                                              n_max_models=4, brightness_range_alpha=(0.8, 1.2),
                                              brightness_range_beta=(-10, 10), max_blur=1, max_noise=10,
                                              free_rotation=False, seed=None):
    """Per loop and labelled image: a random sub-ensemble (n_min..n_max models) -> three binary IMs (>=) -> combined IM,
    randomly eroded / dilated with a kernel from {0, 3, 5} -> blocked brightfield + masks written as
    `{stem}_aug_{loop}.png`, labels.csv row (name; IoU alive, dead, pos against the ground truth; detection flags).

    Two behaviours of the reference that are kept because they shape EvalNet's inputs:
      * `final_mask * 255` on the uint8 {0,255} masks wraps to {0,1} (functions.py:3939-3942): the masks are written, and
        later read by the EvalNet generator, with values 0 / 1;
      * the randomly augmented copy (functions.py:3983-3990) is overwritten by the plain one under the same name
        (:3993-3996), so only the plain files exist afterwards -- the augmentation is not computed here.
    The sub-ensembles of a loop are batched by model subset."""
    F = _F()
    rng = random.Random(F.SEED if seed is None else seed)
    sub = ("brightfield", "alive", "dead", "mod_position")
    din = {k: os.path.join(main_input_path, k) for k in sub}
    dout = {k: os.path.join(main_output_path, k) for k in sub}
    for d in dout.values():
        os.makedirs(d, exist_ok=True)
    names = sorted(os.listdir(din["brightfield"]))
    rows = []
    ensembles = {}
    with F._pool() as pool:
        for nl in range(num_loops):
            plan = _draw_plan(rng, len(names), len(models), n_min_models, n_max_models)
            loop_rows = {}
            groups = _groups(plan)
            for subset, idx_all in groups.items():
                if subset not in ensembles:
                    ensembles[subset] = F.EnsembleIM([models[j] for j in subset])
                ens = ensembles[subset]
                for s in range(0, len(idx_all), F.INFER_BATCH):
                    idx = idx_all[s:s + F.INFER_BATCH]
                    rd = lambda k: torch.from_numpy(np.stack(list(pool.map(
                        lambda i: F.read_png(os.path.join(din[k], names[i]), 1), idx)), 0)).cuda()
                    bf, gt = rd("brightfield"), torch.cat([rd("alive"), rd("dead"), rd("mod_position")], 3)
                    r = ens.run(bf, 0.5, True, False, False)
                    im = _random_morph(r["im"], plan, idx)     # erode first, then dilate (functions.py:3945-3953)
                    masks = r["masks"]
                    bf = bf.clone()
                    _im.block_apply(im, bf, masks)
                    pred = masks > 0
                    g = gt.permute(0, 3, 1, 2) > 0
                    inter = (pred & g).sum(dim=(2, 3)).cpu().numpy()
                    union = (pred | g).sum(dim=(2, 3)).cpu().numpy()
                    gt_nz = g.sum(dim=(2, 3)).cpu().numpy()
                    ious = inter / (union + 1e-7)                       # get_IoU_binary, functions.py:1767-1788
                    det = np.stack([gt_nz[:, 0] >= h * w * 0.01, gt_nz[:, 1] >= h * w * 0.01, gt_nz[:, 2] >= h * w * 0.001], 1)
                    bf_np = bf.cpu().numpy()
                    m_np = (masks > 0).to(torch.uint8).cpu().numpy()    # {0,1}: the uint8 wrap of `mask * 255`
                    jobs = []
                    for row, i in enumerate(idx):
                        out_name = f"{names[i][:-4]}_aug_{nl}.png"
                        loop_rows[i] = (out_name, ious[row, 0], ious[row, 1], ious[row, 2], int(det[row, 0]), int(det[row, 1]),
                                        int(det[row, 2]))
                        jobs.append((os.path.join(dout["brightfield"], out_name), bf_np[row, :, :, 0]))
                        for k, key in enumerate(("alive", "dead", "mod_position")):
                            jobs.append((os.path.join(dout[key], out_name), m_np[row, k]))
                    F.write_pngs_async(jobs)      # on the writer pool: the next batch's decode, network pass and augmentation run meanwhile
            rows += [loop_rows[i] for i in range(len(names))]
    F.flush_writes()
    with open(os.path.join(main_output_path, "labels.csv"), "a", encoding="utf-8", newline="") as f:
        wr = csv.writer(f, delimiter=";")
        for row in rows:
            wr.writerow(row)


# ---------------------------------------------------------------------------------------------------
# EvalNet training (functions.py:4673-4722 with the generator of :4823-4882)
# ---------------------------------------------------------------------------------------------------
def _read_labels(main_path, num_classes=3):
    rows = []
    with open(os.path.join(main_path, "labels.csv"), encoding="utf-8", newline="") as f:
        for r in csv.reader(f, delimiter=";"):
            if r:
                rows.append((r[0], np.asarray(r[1:1 + 2 * num_classes], dtype=np.float32)))
    return rows


def _cached_evalnet_set(kind, main_path, load):
    """the EvalNet candidates of a run train on the same two directories: decode them once (device decode cache, keyed by the
    directory, its labels.csv and what `kind` of set it is read as)"""
    F = _F()
    st = os.stat(os.path.join(main_path, "labels.csv"))
    key = ("evalnet", kind, os.path.abspath(main_path), st.st_size, st.st_mtime_ns)
    with F._CACHE_LOCK:
        hit = F._DECODE_CACHE.get(key)
        if hit is not None:
            return tuple(F._used_here(hit))
        out = list(load())
        F._uploaded()
        F._decode_cache_put(key, out)
        return tuple(out)


def _load_evalnet_set(main_path, rows, pool):
    """brightfield (grey) [N,H,W,1], masks alive|dead|mod_position [N,H,W,3] (raw uint8 values), labels [N,6] on the device"""
    F = _F()

    def one(r):
        mask_name = r[0]
        image_name = mask_name.split("___")[0] + ".png" if "___" in mask_name else mask_name   # functions.py:4863-4866
        bf = F.read_png(os.path.join(main_path, "brightfield", image_name), 1)
        gt = np.concatenate([F.read_png(os.path.join(main_path, k, mask_name), 1) for k in ("alive", "dead", "mod_position")], 2)
        return bf, gt

    items = list(pool.map(one, rows))
    xa = torch.from_numpy(np.stack([i[0] for i in items], 0)).cuda()
    xb = torch.from_numpy(np.stack([i[1] for i in items], 0)).cuda()
    y = torch.from_numpy(np.stack([r[1] for r in rows], 0)).cuda()
    return xa, xb, y


def _evaluate_evalnet(model, xa, xb, y, batch_size, steps, k):
    """model.evaluate(generator, steps): inference-mode forward over `steps` batches -> (total, mse, bce, mae, acc)"""
    rows = []
    for s in range(steps):
        sl = slice(s * batch_size, (s + 1) * batch_size)
        out = model.predict_device(xa[sl].contiguous(), xb[sl].contiguous()).double()
        t = y[sl].double()
        p_iou, p_det, t_iou, t_det = out[:, :k], out[:, k:], t[:, :k], t[:, k:]
        pc = p_det.clamp(1e-7, 1 - 1e-7)
        rows.append(torch.stack([((p_iou - t_iou) ** 2).mean(), -(t_det * pc.log() + (1 - t_det) * (1 - pc).log()).mean(),
                                 (p_iou - t_iou).abs().mean(), ((p_det > 0.5).double() == t_det).double().mean()]))
    tot = np.zeros(4)
    for r in (torch.stack(rows).cpu().numpy() if rows else []):      # one transfer; the batches' float64 values summed in order
        tot += r
    mse, bce, mae, acc = tot / max(steps, 1)
    return mse + bce, mse, bce, mae, acc


def save_evalnet(model, path):
    """ModelCheckpoint's model.save(path) for the EvalNet; IMK_MODEL_FORMAT=keras_h5: a Keras save_weights HDF5 file (keras_h5.py)"""
    if os.environ.get("IMK_MODEL_FORMAT", "safetensors") == "keras_h5":
        from .keras_h5 import save_keras_evalnet_weights
        return save_keras_evalnet_weights(model, path)
    from safetensors.torch import save_file
    p = model.plan
    meta = {"net": "evalnet", "h": str(p.h), "w": str(p.w), "ca": str(p.ca), "cb": str(p.cb), "n_out": str(p.n_out),
            "alpha": repr(p.alpha), "two_heads": str(int(p.two_heads)), "normalize_a": str(p.cfg.normalize_a),
            "normalize_b": str(p.cfg.normalize_b), "b_onehot": str(int(p.b_onehot))}
    save_file({k: v.contiguous() for k, v in model.state_dict().items()}, path, metadata=meta)


def load_evalnet(path, device="cuda"):
    """load_model for an EvalNet file: this package's safetensors, or a Keras HDF5 checkpoint of evalnet.get_evalnet / get_evalnet_miou"""
    from . import h5lite
    if h5lite.is_hdf5(path):
        from .keras_h5 import load_keras_evalnet
        return load_keras_evalnet(path, device=device)
    from safetensors import safe_open
    with safe_open(path, framework="pt") as f:
        meta = f.metadata()
        sd = {k: f.get_tensor(k) for k in f.keys()}
    m = EvalNet(int(meta["h"]), int(meta["w"]), int(meta["ca"]), int(meta["cb"]), int(meta["n_out"]), float(meta["alpha"]),
                bool(int(meta["two_heads"])), bool(int(meta["normalize_a"])), bool(int(meta["normalize_b"])), seed=0, device=device,
                b_onehot=bool(int(meta.get("b_onehot", "0"))))
    m.load_state_dict(sd)
    return m


def _fit_evalnet(model, train_set, val_set, filepath_h5, batch_size, epochs, evaluate, monitor, seed):
    """model.fit(train_generator, validation_data=val_generator, steps_per_epoch=len//batch, validation_steps=len//batch,
    callbacks=[ModelCheckpoint(save_best_only, monitor, mode='min')]) then load_model + evaluate.  `evaluate(model, set,
    steps)` returns the metric tuple, `monitor` indexes it.  Every rank of a multi-GPU launch trains the same model from
    the same seed (the sets are small)."""
    F = _F()
    rng = np.random.default_rng(F.SEED if seed is None else seed)
    xa, xb, y = train_set
    n_train, n_val = xa.shape[0], val_set[0].shape[0]
    steps, val_steps = n_train // batch_size, n_val // batch_size
    vperm = torch.as_tensor(rng.permutation(n_val), device="cuda")          # the validation generator shuffles too
    val_set = tuple(t[vperm] for t in val_set)
    model.init_train_state()
    best = float("inf")
    order = []
    for ep in range(epochs):
        for _ in range(steps):
            if not order:      # dataframe.sample(frac=1): a new shuffled pass, one short batch at its end (functions.py:4794-4799)
                perm = torch.as_tensor(rng.permutation(n_train), device="cuda")
                pa, pb, py = xa[perm], xb[perm], y[perm]
                order = [(i, min(i + batch_size, n_train)) for i in range(0, n_train, batch_size)]
            lo, hi = order.pop(0)
            model.fwd_bwd(pa[lo:hi], pb[lo:hi], py[lo:hi])
            model.adamw_step(F.LR, F.WD)
        val = evaluate(model, val_set, val_steps)[monitor]
        if val < best:
            best = val
            if F._rank_world()[0] == 0:
                save_evalnet(model, filepath_h5)
    d = F._dist()
    if d:
        d.barrier()
    return tuple(float(v) for v in evaluate(load_evalnet(filepath_h5), val_set, val_steps))


def train_evalnet_miou_model_hela(model, train_main_path, val_main_path, filepath_h5, batch_size, epochs, seed=None):
    """functions.py:4673-4722: AdamW(LR, WD), loss ['mse', 'binary_crossentropy'], best epoch by val_loss (min), then
    the best model evaluated on the validation generator.  Returns (total_loss, iou_loss, detection_loss, iou_mae,
    detection_acc)."""
    F = _F()
    k = model.plan.n_out
    with F._pool() as pool:
        tr = _cached_evalnet_set(("hela", k), train_main_path, lambda: _load_evalnet_set(train_main_path, _read_labels(train_main_path, k), pool))
        va = _cached_evalnet_set(("hela", k), val_main_path, lambda: _load_evalnet_set(val_main_path, _read_labels(val_main_path, k), pool))
    ev = lambda m, st, steps: _evaluate_evalnet(m, st[0], st[1], st[2], batch_size, steps, k)
    return _fit_evalnet(model, tr, va, filepath_h5, batch_size, epochs, ev, 0, seed)


def _load_binary_evalnet_set(main_path, pool):
    """images (RGB) [N,H,W,3], masks (grey) [N,H,W,1], labels [N,1] (functions.py:4778-4820)"""
    F = _F()
    rows = []
    with open(os.path.join(main_path, "labels.csv"), encoding="utf-8", newline="") as f:
        for r in csv.reader(f, delimiter=";"):
            if r:
                rows.append((r[0], float(r[1])))

    def one(r):
        mask_name = r[0]
        image_name = mask_name.split("___")[0] + ".png" if "___" in mask_name else mask_name
        return (F.read_png(os.path.join(main_path, "images", image_name), 3), F.read_png(os.path.join(main_path, "masks", mask_name), 1))

    items = list(pool.map(one, rows))
    return (torch.from_numpy(np.stack([i[0] for i in items], 0)).cuda(), torch.from_numpy(np.stack([i[1] for i in items], 0)).cuda(),
            torch.tensor([[r[1]] for r in rows], dtype=torch.float32, device="cuda"))


def train_evalnet_ISIC_2018(model, train_main_path, val_main_path, filepath_h5, batch_size, epochs, seed=None):
    """functions.py:4464-4506: loss 'mean_squared_error', metric 'mean_absolute_error', best epoch by
    val_mean_absolute_error (min).  Returns (mse, mae) of the best model on the validation generator."""
    F = _F()
    with F._pool() as pool:
        tr = _cached_evalnet_set("binary", train_main_path, lambda: _load_binary_evalnet_set(train_main_path, pool))
        va = _cached_evalnet_set("binary", val_main_path, lambda: _load_binary_evalnet_set(val_main_path, pool))

    def ev(m, st, steps):
        rows = []
        for s in range(steps):
            sl = slice(s * batch_size, (s + 1) * batch_size)
            d = m.predict_device(st[0][sl].contiguous(), st[1][sl].contiguous()).double() - st[2][sl].double()
            rows.append(torch.stack([(d ** 2).mean(), d.abs().mean()]))
        tot = np.zeros(2)
        for r in (torch.stack(rows).cpu().numpy() if rows else []):      # one transfer; summed on the host in batch order
            tot += r
        return tuple(tot / max(steps, 1))

    return _fit_evalnet(model, tr, va, filepath_h5, batch_size, epochs, ev, 1, seed)


def _load_multiclass_evalnet_set(main_path, num_classes, pool):
    """images (RGB) [N,H,W,3], class-id masks [N,H,W,1] (one-hot on the device), labels [N,2K] (functions.py:4940-4984)"""
    F = _F()
    rows = _read_labels(main_path, num_classes)

    def one(r):
        mask_name = r[0]
        image_name = mask_name.split("___")[0] + ".png" if "___" in mask_name else mask_name
        return (F.read_png(os.path.join(main_path, "images", image_name), 3), F.read_png(os.path.join(main_path, "masks", mask_name), 1))

    items = list(pool.map(one, rows))
    return (torch.from_numpy(np.stack([i[0] for i in items], 0)).cuda(), torch.from_numpy(np.stack([i[1] for i in items], 0)).cuda(),
            torch.from_numpy(np.stack([r[1] for r in rows], 0)).cuda())


def train_evalnet_miou_model_multiclass(model, h, w, train_main_path, val_main_path, filepath_h5, batch_size, num_classes, epochs,
                                        seed=None):
    """functions.py:4726-4775: like the HeLa variant, masks are label maps fed as one-hot stacks.  Returns (total_loss,
    iou_loss, detection_loss, iou_mae, detection_acc)."""
    F = _F()
    if not model.plan.b_onehot or model.plan.n_out != num_classes:
        raise ValueError("needs a get_evalnet_miou(..., inputB_channels=num_classes) model with a one-hot input B")
    with F._pool() as pool:
        tr = _cached_evalnet_set(("multi", num_classes), train_main_path, lambda: _load_multiclass_evalnet_set(train_main_path, num_classes, pool))
        va = _cached_evalnet_set(("multi", num_classes), val_main_path, lambda: _load_multiclass_evalnet_set(val_main_path, num_classes, pool))
    ev = lambda m, st, steps: _evaluate_evalnet(m, st[0], st[1], st[2], batch_size, steps, num_classes)
    return _fit_evalnet(model, tr, va, filepath_h5, batch_size, epochs, ev, 0, seed)


# ---------------------------------------------------------------------------------------------------
# EvalNet-weighted augmentation of the pseudo-labelled set (functions.py:5837-5941)
# ---------------------------------------------------------------------------------------------------
def num_augs_from_miou(miou, min_threshold, max_threshold):
    """functions.py:5921-5930"""
    step = (max_threshold - min_threshold) / 5
    if miou > max_threshold:
        n = 5
    elif miou > min_threshold:
        n = 1 + int((miou - min_threshold) / step)
    else:
        n = 1
    return min(n, 5)


def create_augment_images_and_masks_with_evalnet_ensemble_hela(evalnets, h, w, c, min_threshold, max_threshold,
                                                               main_input_path, main_output_path,
                                                               brightness_range_alpha=(0.6, 1.4),
                                                               brightness_range_beta=(-20, 20), max_blur=3, max_noise=20,
                                                               free_rotation=True):
    """Every pseudo-labelled sample gets 1..5 augmented copies `{stem}___{j}.png`, the number growing with the mean
    IoU the EvalNet ensemble predicts for it (classes whose mean detection score is below 0.5 do not count).
    EvalNet sees the masks as 0 / 1 here (mask / 255, functions.py:5881-5883)."""
    F = _F()
    sub = ("brightfield", "alive", "dead", "mod_position")
    din = {k: os.path.join(main_input_path, k) for k in sub}
    dout = {k: os.path.join(main_output_path, k) for k in sub}
    for d in dout.values():
        os.makedirs(d, exist_ok=True)
    mine = F.shard_list(os.listdir(din["brightfield"]))
    draw_kw = dict(brightness_range_alpha=brightness_range_alpha, brightness_range_beta=brightness_range_beta,
                   max_blur=max_blur, max_noise=max_noise, free_rotation=free_rotation)
    with F._pool() as pool:
        for s in range(0, len(mine), F.INFER_BATCH):
            chunk = mine[s:s + F.INFER_BATCH]
            rd = lambda k: torch.from_numpy(F.read_png_stack(pool, [os.path.join(din[k], n) for n in chunk], 1)).cuda()
            bf = rd("brightfield")
            m255 = torch.cat([rd("alive"), rd("dead"), rd("mod_position")], 3)
            m01 = (m255.float() / 255.0).round().clamp(0, 255).to(torch.uint8)      # what predict() receives: mask / 255
            outs = torch.stack([e.predict_device(bf, m01) for e in evalnets], 0).double().mean(0).cpu().numpy()
            k = outs.shape[1] // 2
            n_augs = []
            for row in outs:
                valid = [row[ci] for ci in range(k) if row[k + ci] >= 0.5]
                n_augs.append(num_augs_from_miou(sum(valid) / len(valid) if valid else 0, min_threshold, max_threshold))
            n_augs = torch.tensor(n_augs, device="cuda")
            jobs = []
            for j in range(5):
                sel = torch.nonzero(n_augs > j).flatten()
                if not sel.numel():
                    break
                o, om = augment_batch(bf[sel].contiguous(), m255[sel].contiguous(), draw_params(int(sel.numel()), **draw_kw))
                o, om = o.cpu().numpy(), ((om >= 128).to(torch.uint8) * 255).cpu().numpy()   # (mask/255 >= 0.5) * 255
                for row, i in enumerate(sel.tolist()):
                    name = f"{chunk[i][:-4]}___{j}.png"
                    jobs.append((os.path.join(dout["brightfield"], name), o[row, :, :, 0]))
                    for ci, key in enumerate(("alive", "dead", "mod_position")):
                        jobs.append((os.path.join(dout[key], name), om[row, :, :, ci]))
            F.write_pngs_async(jobs)      # on the writer pool: the next batch's decode, network pass and augmentation run meanwhile
    F.flush_writes()
    if F._dist():
        F._dist().barrier()


def create_augment_images_and_masks_with_evalnet_ensemble_binary(evalnets, h, w, c, min_threshold, max_threshold,
                                                                 main_input_path, main_output_path,
                                                                 brightness_range_alpha=(0.6, 1.4),
                                                                 brightness_range_beta=(-20, 20), max_blur=3, max_noise=20,
                                                                 free_rotation=True, rgb=True):
    """functions.py:5684-5757: 1..5 augmented copies `{stem}___{j}.png` of every pseudo-labelled pair, the number growing
    with the IoU the EvalNet ensemble predicts for (image, mask)."""
    F = _F()
    flip = (not rgb) and c == 3      # rgb=False (functions.py:3611 ff.): the nets see the file's channels in OpenCV's order (BGR)
    iin, min_ = os.path.join(main_input_path, "images"), os.path.join(main_input_path, "masks")
    iout, mout = os.path.join(main_output_path, "images"), os.path.join(main_output_path, "masks")
    os.makedirs(iout, exist_ok=True)
    os.makedirs(mout, exist_ok=True)
    mine = F.shard_list(os.listdir(iin))
    draw_kw = dict(brightness_range_alpha=brightness_range_alpha, brightness_range_beta=brightness_range_beta,
                   max_blur=max_blur, max_noise=max_noise, free_rotation=free_rotation)
    with F._pool() as pool:
        for s in range(0, len(mine), F.INFER_BATCH):
            chunk = mine[s:s + F.INFER_BATCH]
            x = torch.from_numpy(F.read_png_stack(pool, [os.path.join(iin, n) for n in chunk], c)).cuda()
            m = torch.from_numpy(F.read_png_stack(pool, [os.path.join(min_, n) for n in chunk], 1)).cuda()
            xin = x.flip(-1).contiguous() if flip else x
            mean_iou = torch.stack([e.predict_device(xin, m) for e in evalnets], 0).double().mean(0)[:, 0].cpu().numpy()
            n_augs = torch.tensor([num_augs_from_miou(v, min_threshold, max_threshold) for v in mean_iou], device="cuda")
            jobs = []
            for j in range(5):
                sel = torch.nonzero(n_augs > j).flatten()
                if not sel.numel():
                    break
                o, om = augment_batch(x[sel].contiguous(), m[sel].contiguous(), draw_params(int(sel.numel()), **draw_kw))
                o, om = o.cpu().numpy(), om.cpu().numpy()
                for row, i in enumerate(sel.tolist()):
                    name = f"{chunk[i][:-4]}___{j}.png"
                    jobs.append((os.path.join(iout, name), o[row]))
                    jobs.append((os.path.join(mout, name), om[row, :, :, 0]))
            F.write_pngs_async(jobs)      # on the writer pool: the next batch's decode, network pass and augmentation run meanwhile
    F.flush_writes()
    if F._dist():
        F._dist().barrier()


def create_augment_images_and_masks_with_gt(main_gt_input_path, min_threshold, max_threshold, main_input_path, main_output_path,
                                            brightness_range_alpha=(0.6, 1.4), brightness_range_beta=(-20, 20), max_blur=3,
                                            max_noise=20, free_rotation=False, rgb=True):
    """functions.py:6057-6121 (SUIM/16_SUIM_GT_IM++.py: a "perfect EvalNet"): the number of augmented copies `{stem}___{j}.png`
    of a pseudo-labelled pair follows the IoU (get_IoU_multi_unique) of its pseudo-label against the ground-truth mask with the
    IM pixels blanked.  `rgb` only chooses the channel order the reference hands to nobody here (the augmented image is made
    from the file's own order): accepted, no effect."""
    F = _F()
    iin, min_, imin = (os.path.join(main_input_path, k) for k in ("images", "masks", "im"))
    iout, mout = os.path.join(main_output_path, "images"), os.path.join(main_output_path, "masks")
    os.makedirs(iout, exist_ok=True)
    os.makedirs(mout, exist_ok=True)
    mine = F.shard_list(os.listdir(iin))
    draw_kw = dict(brightness_range_alpha=brightness_range_alpha, brightness_range_beta=brightness_range_beta,
                   max_blur=max_blur, max_noise=max_noise, free_rotation=free_rotation)
    with F._pool() as pool:
        for s in range(0, len(mine), F.INFER_BATCH):
            chunk = mine[s:s + F.INFER_BATCH]
            rd = lambda d, ch: F.read_png_stack(pool, [os.path.join(d, n) for n in chunk], ch)
            x_np, m_np, im_np, gt_np = rd(iin, 3), rd(min_, 1), rd(imin, 1), rd(main_gt_input_path, 1)
            n_augs = []
            for i in range(len(chunk)):
                gt = gt_np[i, :, :, 0].copy()
                gt[im_np[i, :, :, 0] > 0] = 0                                  # :6102
                n_augs.append(num_augs_from_miou(F.get_IoU_multi_unique(m_np[i, :, :, 0], gt), min_threshold, max_threshold))
            x, m = torch.from_numpy(x_np).cuda(), torch.from_numpy(m_np).cuda()
            n_augs = torch.tensor(n_augs, device="cuda")
            jobs = []
            for j in range(5):
                sel = torch.nonzero(n_augs > j).flatten()
                if not sel.numel():
                    break
                o, om = augment_batch(x[sel].contiguous(), m[sel].contiguous(), draw_params(int(sel.numel()), **draw_kw))
                o, om = o.cpu().numpy(), om.cpu().numpy()
                for row, i in enumerate(sel.tolist()):
                    name = f"{chunk[i][:-4]}___{j}.png"
                    jobs.append((os.path.join(iout, name), o[row]))
                    jobs.append((os.path.join(mout, name), om[row, :, :, 0]))
            F.write_pngs_async(jobs)      # on the writer pool: the next batch's decode, network pass and augmentation run meanwhile
    F.flush_writes()
    if F._dist():
        F._dist().barrier()


def create_augment_images_and_masks_with_evalnet_ensemble_multiclass(evalnets, h, w, c, num_classes, min_threshold,
                                                                     max_threshold, main_input_path, main_output_path,
                                                                     brightness_range_alpha=(0.6, 1.4),
                                                                     brightness_range_beta=(-20, 20), max_blur=3,
                                                                     max_noise=20, free_rotation=False, rgb=True):
    """functions.py:5946-6035: the number of augmented copies follows the mean predicted IoU over the classes > 0 whose
    mean detection score is at least 0.5."""
    F = _F()
    flip = (not rgb) and c == 3      # rgb=False (functions.py:3611 ff.): the nets see the file's channels in OpenCV's order (BGR)
    iin, min_ = os.path.join(main_input_path, "images"), os.path.join(main_input_path, "masks")
    iout, mout = os.path.join(main_output_path, "images"), os.path.join(main_output_path, "masks")
    os.makedirs(iout, exist_ok=True)
    os.makedirs(mout, exist_ok=True)
    mine = F.shard_list(os.listdir(iin))
    draw_kw = dict(brightness_range_alpha=brightness_range_alpha, brightness_range_beta=brightness_range_beta,
                   max_blur=max_blur, max_noise=max_noise, free_rotation=free_rotation)
    with F._pool() as pool:
        for s in range(0, len(mine), F.INFER_BATCH):
            chunk = mine[s:s + F.INFER_BATCH]
            x = torch.from_numpy(F.read_png_stack(pool, [os.path.join(iin, n) for n in chunk], c)).cuda()
            m = torch.from_numpy(F.read_png_stack(pool, [os.path.join(min_, n) for n in chunk], 1)).cuda()
            xin = x.flip(-1).contiguous() if flip else x
            outs = torch.stack([e.predict_device(xin, m) for e in evalnets], 0).double().mean(0).cpu().numpy()
            n_augs = []
            for row in outs:
                valid = [row[ci] for ci in range(1, num_classes) if row[num_classes + ci] >= 0.5]
                n_augs.append(num_augs_from_miou(sum(valid) / len(valid) if valid else 0.0, min_threshold, max_threshold))
            n_augs = torch.tensor(n_augs, device="cuda")
            jobs = []
            for j in range(5):
                sel = torch.nonzero(n_augs > j).flatten()
                if not sel.numel():
                    break
                o, om = augment_batch(x[sel].contiguous(), m[sel].contiguous(), draw_params(int(sel.numel()), **draw_kw))
                o, om = o.cpu().numpy(), om.cpu().numpy()
                for row, i in enumerate(sel.tolist()):
                    name = f"{chunk[i][:-4]}___{j}.png"
                    jobs.append((os.path.join(iout, name), o[row]))
                    jobs.append((os.path.join(mout, name), om[row, :, :, 0]))
            F.write_pngs_async(jobs)      # on the writer pool: the next batch's decode, network pass and augmentation run meanwhile
    F.flush_writes()
    if F._dist():
        F._dist().barrier()
