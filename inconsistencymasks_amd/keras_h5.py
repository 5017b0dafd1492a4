"""The reference's Keras checkpoints <-> this package's U-Net (ISIC_2018/09_ISIC_2018_IM.py:74-76, 131-135;
functions.py:217: every U-Net is saved by ModelCheckpoint as a full-model Keras HDF5 file and read back with load_model).

* `load_keras_model(path)`: a `*.h5` written by Keras (full model or `save_weights`) -> UNet, read with the package's own HDF5
  reader (h5lite.py: no h5py / TensorFlow needed).  The Keras layers of `unet.get_unet` are matched to the layer names of
  include/imk.h (`imk_unet_layer_info`: "in.c", "e1.c3", ..., "d9.bnb", "out") by CREATION ORDER: Keras numbers its auto-named
  layers (`conv2d_17`, `batch_normalization_9`, ...) in the order unet.py:46-67 creates them, which is the order of
  `oracle/unet_oracle.layer_table` and of the flat parameter vector.  Kernels stay HWIO, exactly as Keras stores them.
  Input size and output activation come from the file's `model_config` attribute (a weights-only file has none: pass them).
* `save_keras_weights(model, path)`: the layout of Keras' `model.save_weights("x.h5")` (root attributes layer_names / backend /
  keras_version, one group per layer with `weight_names` and the datasets `<layer>/<weight>:0`), layer names as a fresh Keras
  session numbers them.  A TensorFlow user restores it with `get_unet(...).load_weights(path)`: the HDF5 loader goes by the
  ORDER of the weight-bearing layers (names only matter with by_name=True), and both orders are the creation order.
* `keras_weight_list` / `state_dict_from_weight_list`: the `model.get_weights()` order, for .npz interchange.

The HDF5 container code is pinned to files written by h5py / libhdf5 (tests/golden/h5_*.h5); the Keras LAYOUT inside (attribute and
group names) is a restatement of Keras 2.x `saving/hdf5_format.py` conventions -- TensorFlow cannot be installed in the build image,
so no file written by Keras itself has been read yet."""
import json
import re

import numpy as np

from . import h5lite


def layer_table(c_in, n_out, alpha):
    """(name, kind, k, cin, cout) in unet.py creation order (unet.py:46-67) -- the order of imk_unet_layer_info."""
    f = lambda v: int(v * alpha)
    c16, c32, c64, c128, c256 = f(16), f(32), f(64), f(128), f(256)
    t = [("in.c", "conv", 1, c_in, c16), ("in.bn", "bn", 0, c16, c16)]
    for i, (ci, co) in enumerate([(c16, c16), (c16, c32), (c32, c64), (c64, c128)], start=1):
        t += [(f"e{i}.c3", "conv", 3, ci, co), (f"e{i}.c1", "conv", 1, co, co), (f"e{i}.bn", "bn", 0, co, co)]
    t += [("b.c3", "conv", 3, c128, c256), ("b.c1", "conv", 1, c256, c128), ("b.bn", "bn", 0, c128, c128)]
    for j, ci, f1, f2 in [(6, c128, c128, c64), (7, c64, c64, c32), (8, c32, c32, c16), (9, c16, c16, c16)]:
        t += [(f"d{j}.ca", "conv", 1, ci, f1), (f"d{j}.bna", "bn", 0, f1, f1), (f"d{j}.c3", "conv", 3, f1, f1),
              (f"d{j}.c1", "conv", 1, f1, f2), (f"d{j}.bnb", "bn", 0, f2, f2)]
    return t + [("out", "conv", 1, c16, n_out)]


def _suffix(name):
    m = re.search(r"_(\d+)$", name)
    return int(m.group(1)) if m else 0


def match_keras_layers(keras_names, table):
    """our layer name -> Keras layer name.  keras_names: every layer of the HDF5 file that owns weights."""
    convs = sorted([n for n in keras_names if re.fullmatch(r"conv2d(_\d+)?", n)], key=_suffix)
    bns = sorted([n for n in keras_names if re.fullmatch(r"batch_normalization(_\d+)?", n)], key=_suffix)
    ours_conv = [t[0] for t in table if t[1] == "conv" and t[0] != "out"]
    ours_bn = [t[0] for t in table if t[1] == "bn"]
    if len(convs) != len(ours_conv) or len(bns) != len(ours_bn) or "out" not in keras_names:
        raise ValueError(f"not a unet.get_unet model: {len(convs)} Conv2D (+ 'out': {'out' in keras_names}) and {len(bns)} "
                         f"BatchNormalization layers, expected {len(ours_conv)} + 'out' and {len(ours_bn)}")
    m = dict(zip(ours_conv, convs))
    m.update(zip(ours_bn, bns))
    m["out"] = "out"
    return m


def state_dict_from_keras(weights_of, table):
    """weights_of: Keras layer name -> {'kernel','bias'} or {'gamma','beta','moving_mean','moving_variance'} (numpy).
    Returns our state dict (name.w HWIO / .b / .gamma / .beta / .mean / .var), shapes checked against the table."""
    names = match_keras_layers(list(weights_of), table)
    sd = {}
    for name, kind, k, ci, co in table:
        w = weights_of[names[name]]
        if kind == "conv":
            kern, bias = np.asarray(w["kernel"], np.float32), np.asarray(w["bias"], np.float32)
            if kern.shape != (k, k, ci, co) or bias.shape != (co,):
                raise ValueError(f"{name} <- {names[name]}: kernel {kern.shape}, expected {(k, k, ci, co)}")
            sd[name + ".w"], sd[name + ".b"] = kern, bias
        else:
            for ours, theirs in (("gamma", "gamma"), ("beta", "beta"), ("mean", "moving_mean"), ("var", "moving_variance")):
                a = np.asarray(w[theirs], np.float32)
                if a.shape != (co,):
                    raise ValueError(f"{name} <- {names[name]}: {theirs} {a.shape}, expected {(co,)}")
                sd[f"{name}.{ours}"] = a
    return sd


def keras_weight_list(sd, table):
    """our state dict -> the list `model.get_weights()` returns for the reference's get_unet model"""
    out = []
    for name, kind, *_ in table:
        keys = (".w", ".b") if kind == "conv" else (".gamma", ".beta", ".mean", ".var")
        out += [np.asarray(sd[name + k], np.float32) for k in keys]
    return out


def state_dict_from_weight_list(arrays, table):
    it = iter(arrays)
    weights_of = {}
    ci = bi = 0
    for name, kind, *_ in table:
        if kind == "conv":
            kn = "out" if name == "out" else ("conv2d" if ci == 0 else f"conv2d_{ci}")
            ci += name != "out"
            weights_of[kn] = {"kernel": next(it), "bias": next(it)}
        else:
            kn = "batch_normalization" if bi == 0 else f"batch_normalization_{bi}"
            bi += 1
            weights_of[kn] = {"gamma": next(it), "beta": next(it), "moving_mean": next(it), "moving_variance": next(it)}
    return state_dict_from_keras(weights_of, table)


def infer_config(first_kernel_shape, out_kernel_shape):
    """(c_in, n_out, alpha) from the stem's kernel [1,1,c_in,int(16 alpha)] and the head's [1,1,int(16 alpha),n_out]"""
    c_in, c16 = int(first_kernel_shape[2]), int(first_kernel_shape[3])
    alpha = c16 / 16.0
    for cand in (0.25, 0.5, 0.75, 1.0, 1.25, 1.5, 1.75, 2.0, 3.0, 4.0):
        if int(16 * cand) == c16:
            alpha = cand
            break
    return c_in, int(out_kernel_shape[3]), alpha


def _read_keras(path):
    """-> (Keras layer name -> {weight name: array}, parsed model_config or None, this package's imk_config attribute or None)"""
    f = h5lite.File(path)
    g = f["model_weights"] if "model_weights" in f else f
    cfg = f.attrs.get("model_config")
    if cfg is not None:
        cfg = json.loads(cfg.decode("utf-8") if isinstance(cfg, bytes) else cfg)
    own = f.attrs.get("imk_config")                        # written by save_keras_weights: Keras ignores it
    if own is not None:
        own = json.loads(own.decode("utf-8") if isinstance(own, bytes) else own)
    names = h5lite.load_attr_list(g, "layer_names") or g.keys()
    weights_of = {}
    for lname in names:
        lg = g[lname]
        wnames = h5lite.load_attr_list(lg, "weight_names")
        found = {}
        if wnames:
            for wn in wnames:
                found[wn.split("/")[-1].split(":")[0]] = np.asarray(lg[wn][...])
        else:
            for wn, ds in lg.visit_datasets():
                found[wn.split("/")[-1].split(":")[0]] = np.asarray(ds[...])
        if found:
            weights_of[lname] = found
    return weights_of, cfg, own


def _config_layers(cfg):
    return cfg.get("config", {}).get("layers", []) if cfg else []


def read_keras_h5(path):
    """-> (Keras layer name -> {weight name: array}, output activation or None, [H, W, C] of the InputLayer or None)"""
    weights_of, cfg, own = _read_keras(path)
    act = shape = None
    for l in _config_layers(cfg):
        c = l.get("config", {})
        if c.get("name") == "out":
            act = c.get("activation")
        if l.get("class_name") == "InputLayer" and c.get("batch_input_shape"):
            shape = [int(v) for v in c["batch_input_shape"][1:]]
    if own is not None and own.get("net", "unet") == "unet":
        act, shape = own.get("act_out"), [int(own["h"]), int(own["w"]), int(own["c_in"])]
    return weights_of, act, shape


def keras_h5_kind(path):
    """'unet' / 'evalnet' / None: which of the reference's two networks a Keras HDF5 file holds (by its layer names)"""
    f = h5lite.File(path)
    g = f["model_weights"] if "model_weights" in f else f
    names = set(h5lite.load_attr_list(g, "layer_names") or g.keys())
    if "out" in names:
        return "unet"
    if {"iou", "detection"} <= names or any(re.fullmatch(r"dense(_\d+)?", n) for n in names):
        return "evalnet"
    return None


def state_dict_from_keras_h5(path):
    """-> (state dict, dict(h, w, c_in, n_out, alpha, act_out)); h / w / act_out are None where the file does not say"""
    weights_of, act, shape = read_keras_h5(path)
    convs = sorted([n for n in weights_of if re.fullmatch(r"conv2d(_\d+)?", n)], key=_suffix)
    if not convs or "out" not in weights_of:
        raise ValueError(f"{path}: not a unet.get_unet checkpoint (layers with weights: {', '.join(sorted(weights_of)[:8])} ...)")
    c_in, n_out, alpha = infer_config(weights_of[convs[0]]["kernel"].shape, weights_of["out"]["kernel"].shape)
    sd = state_dict_from_keras(weights_of, layer_table(c_in, n_out, alpha))
    if shape is not None and shape[2] != c_in:
        raise ValueError(f"{path}: InputLayer has {shape[2]} channels, the first convolution {c_in}")
    return sd, {"h": shape[0] if shape else None, "w": shape[1] if shape else None, "c_in": c_in, "n_out": n_out, "alpha": alpha,
                "act_out": act if act in ("sigmoid", "softmax") else None}


def load_keras_model(path, height=None, width=None, act_out=None, device="cuda"):
    """tf.keras.models.load_model(path) for a checkpoint the reference wrote: -> UNet with the file's weights"""
    import torch
    from .unet import UNet
    sd, cfg = state_dict_from_keras_h5(path)
    h, w, act = height or cfg["h"], width or cfg["w"], act_out or cfg["act_out"]
    if h is None or w is None or act is None:
        raise ValueError(f"{path}: a weights-only HDF5 file does not record the input size / output activation -- pass "
                         "height=, width=, act_out='sigmoid'|'softmax'")
    m = UNet(int(h), int(w), cfg["c_in"], cfg["n_out"], cfg["alpha"], act, seed=0, device=device)
    m.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()})
    return m


def keras_layer_names(table):
    """our layer name -> the name a fresh Keras session gives it (conv2d, conv2d_1, ..., batch_normalization, ..., out)"""
    out, ci, bi = {}, 0, 0
    for name, kind, *_ in table:
        if kind == "conv":
            out[name] = "out" if name == "out" else ("conv2d" if ci == 0 else f"conv2d_{ci}")
            ci += name != "out"
        else:
            out[name] = "batch_normalization" if bi == 0 else f"batch_normalization_{bi}"
            bi += 1
    return out


def save_keras_weights(model_or_sd, path, c_in=None, n_out=None, alpha=None, own=None):
    """Write the U-Net's weights in the layout of Keras' model.save_weights(path) (HDF5 format).  A model (rather than a bare
    state dict) also leaves its geometry in the root attribute `imk_config` (JSON), so that load_keras_model needs no arguments."""
    if hasattr(model_or_sd, "state_dict"):
        m = model_or_sd
        sd = {k: v.numpy() for k, v in m.state_dict().items()}
        c_in, n_out, alpha = m.plan.c_in, m.plan.n_out, m.plan.alpha
        own = {"net": "unet", "h": m.plan.h, "w": m.plan.w, "c_in": c_in, "n_out": n_out, "alpha": alpha, "act_out": m.plan.act_out}
    else:
        sd = {k: np.asarray(v) for k, v in model_or_sd.items()}
        if c_in is None:
            c_in, n_out, alpha = infer_config(sd["in.c.w"].shape, sd["out.w"].shape)
    table = layer_table(c_in, n_out, alpha)
    kn = keras_layer_names(table)
    tree, order = {}, []
    for name, kind, *_ in table:
        k = kn[name]
        order.append(k.encode())
        ws = ([("kernel:0", ".w"), ("bias:0", ".b")] if kind == "conv" else
              [("gamma:0", ".gamma"), ("beta:0", ".beta"), ("moving_mean:0", ".mean"), ("moving_variance:0", ".var")])
        tree[k] = {h5lite.ATTRS: {"weight_names": [f"{k}/{w}".encode() for w, _ in ws]},
                   k: {w: np.asarray(sd[name + s], np.float32) for w, s in ws}}
    tree[h5lite.ATTRS] = {"layer_names": order, "backend": b"tensorflow", "keras_version": b"2.10.0"}
    if own:
        tree[h5lite.ATTRS]["imk_config"] = json.dumps(own)
    h5lite.write(path, tree)


# ---------------------------------------------------------------------------------------------------------------- EvalNet (evalnet.py:24-73)
def evalnet_layer_table(ca, cb, n_out, alpha, two_heads):
    """(name, kind, k, cin, cout) in evalnet.py's creation order -- the order of the plan's flat parameter vector: tower A (input block,
    conv block), tower B, five merged blocks, the Dense head(s) as 1x1 'convs' on the pooled features"""
    f = lambda v: int(v * alpha)
    f0 = f(16)
    t = []
    for tw, cin in (("a", ca), ("b", cb)):
        t += [(f"{tw}.in.c", "conv", 1, cin, f0), (f"{tw}.in.bn", "bn", 0, f0, f0),
              (f"{tw}.c3", "conv", 3, f0, f0), (f"{tw}.c1", "conv", 1, f0, f0), (f"{tw}.bn", "bn", 0, f0, f0)]
    prev = 2 * f0
    for i, v in enumerate((16, 32, 64, 128, 256), start=1):
        t += [(f"m{i}.c3", "conv", 3, prev, f(v)), (f"m{i}.c1", "conv", 1, f(v), f(v)), (f"m{i}.bn", "bn", 0, f(v), f(v))]
        prev = f(v)
    return t + ([("iou", "conv", 1, prev, n_out), ("detection", "conv", 1, prev, n_out)] if two_heads else [("dense", "conv", 1, prev, n_out)])


_HEADS = ("dense", "iou", "detection")


def evalnet_keras_layer_names(table):
    out, ci, bi = {}, 0, 0
    for name, kind, *_ in table:
        if name in _HEADS:
            out[name] = name
        elif kind == "conv":
            out[name] = "conv2d" if ci == 0 else f"conv2d_{ci}"
            ci += 1
        else:
            out[name] = "batch_normalization" if bi == 0 else f"batch_normalization_{bi}"
            bi += 1
    return out


def evalnet_state_dict_from_keras_h5(path):
    """-> (state dict, dict(h, w, ca, cb, n_out, alpha, two_heads, normalize_a, normalize_b)); entries the file does not record are None"""
    weights_of, cfg, own = _read_keras(path)
    convs = sorted([n for n in weights_of if re.fullmatch(r"conv2d(_\d+)?", n)], key=_suffix)
    bns = sorted([n for n in weights_of if re.fullmatch(r"batch_normalization(_\d+)?", n)], key=_suffix)
    two = "iou" in weights_of and "detection" in weights_of
    dense = [n for n in weights_of if re.fullmatch(r"dense(_\d+)?", n)]
    if len(convs) != 16 or len(bns) != 9 or not (two or len(dense) == 1):
        raise ValueError(f"{path}: not an evalnet.get_evalnet / get_evalnet_miou checkpoint ({len(convs)} Conv2D, {len(bns)} BatchNormalization, "
                         f"heads {sorted(set(weights_of) - set(convs) - set(bns))})")
    k_a, k_b = weights_of[convs[0]]["kernel"], weights_of[convs[3]]["kernel"]
    head = weights_of["iou" if two else dense[0]]["kernel"]
    ca, cb, n_out = int(k_a.shape[2]), int(k_b.shape[2]), int(head.shape[-1])
    _, _, alpha = infer_config(k_a.shape, (1, 1, 1, 1))
    table = evalnet_layer_table(ca, cb, n_out, alpha, two)
    it_c, it_b = iter(convs), iter(bns)
    sd = {}
    for name, kind, k, ci, co in table:
        if name in _HEADS:
            w = weights_of[name if two else dense[0]]
            kern = np.asarray(w["kernel"], np.float32).reshape(1, 1, ci, co)          # Dense kernel [features, units]
            sd[name + ".w"], sd[name + ".b"] = kern, np.asarray(w["bias"], np.float32)
        elif kind == "conv":
            w = weights_of[next(it_c)]
            kern = np.asarray(w["kernel"], np.float32)
            if kern.shape != (k, k, ci, co):
                raise ValueError(f"{path}: {name}: kernel {kern.shape}, expected {(k, k, ci, co)}")
            sd[name + ".w"], sd[name + ".b"] = kern, np.asarray(w["bias"], np.float32)
        else:
            w = weights_of[next(it_b)]
            for ours, theirs in (("gamma", "gamma"), ("beta", "beta"), ("mean", "moving_mean"), ("var", "moving_variance")):
                sd[f"{name}.{ours}"] = np.asarray(w[theirs], np.float32)
    meta = {"h": None, "w": None, "ca": ca, "cb": cb, "n_out": n_out, "alpha": alpha, "two_heads": two, "normalize_a": None, "normalize_b": None}
    layers = _config_layers(cfg)
    if layers:
        # the two Inputs in model_config["config"]["input_layers"] order (A first, evalnet.py:45); a Lambda fed by an input = its x / 255
        order = [e[0] for e in cfg["config"].get("input_layers", [])] or [l["config"]["name"] for l in layers if l.get("class_name") == "InputLayer"]
        shapes = {l["config"]["name"]: l["config"].get("batch_input_shape") for l in layers if l.get("class_name") == "InputLayer"}
        if len(order) == 2 and all(shapes.get(n) for n in order):
            meta["h"], meta["w"] = int(shapes[order[0]][1]), int(shapes[order[0]][2])
            fed = set()
            for l in layers:
                if l.get("class_name") == "Lambda":
                    for node in l.get("inbound_nodes", []):
                        for e in node:
                            fed.add(e[0])
            meta["normalize_a"], meta["normalize_b"] = order[0] in fed, order[1] in fed
    if own is not None and own.get("net") == "evalnet":
        meta.update({k: own[k] for k in ("h", "w", "normalize_a", "normalize_b") if k in own})
        meta["b_onehot"] = bool(own.get("b_onehot", False))
    return sd, meta


def load_keras_evalnet(path, height=None, width=None, normalize_a=None, normalize_b=None, device="cuda"):
    """tf.keras.models.load_model(path) for an EvalNet checkpoint of the reference (ISIC_2018/12_ISIC_2018_IM++.py): -> EvalNet"""
    import torch
    from .evalnet import EvalNet
    sd, m = evalnet_state_dict_from_keras_h5(path)
    pick = lambda a, b: a if a is not None else b
    h, w, na, nb = pick(height, m["h"]), pick(width, m["w"]), pick(normalize_a, m["normalize_a"]), pick(normalize_b, m["normalize_b"])
    if None in (h, w, na, nb):
        raise ValueError(f"{path}: a weights-only HDF5 file does not record the input size / which inputs are divided by 255 -- pass "
                         "height=, width=, normalize_a=, normalize_b=")
    net = EvalNet(int(h), int(w), m["ca"], m["cb"], m["n_out"], m["alpha"], m["two_heads"], bool(na), bool(nb), seed=0, device=device,
                  b_onehot=bool(m.get("b_onehot", False)))
    net.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()})
    return net


def save_keras_evalnet_weights(model, path):
    """The EvalNet's weights in the layout of Keras' save_weights (HDF5); Dense kernels as [features, units]."""
    p = model.plan
    sd = {k: v.numpy() for k, v in model.state_dict().items()}
    table = evalnet_layer_table(p.ca, p.cb, p.n_out, p.alpha, p.two_heads)
    kn = evalnet_keras_layer_names(table)
    tree = {}
    # `layer_names` in the order of Keras' model.layers for this graph (layers by depth, creation order inside a depth): the two towers
    # interleave -- the order load_weights(by_name=False) zips with; by_name=True does not depend on it
    towers = [f"{t}.{l}" for l in ("in.c", "in.bn", "c3", "c1", "bn") for t in ("a", "b")]
    order = [kn[n].encode() for n in towers + [n for n, *_ in table if n not in towers]]
    for name, kind, k, ci, co in table:
        kname = kn[name]
        if kind == "conv":
            kern = np.asarray(sd[name + ".w"], np.float32)
            ws = {"kernel:0": kern.reshape(ci, co) if name in _HEADS else kern, "bias:0": np.asarray(sd[name + ".b"], np.float32)}
        else:
            ws = {"gamma:0": sd[name + ".gamma"], "beta:0": sd[name + ".beta"], "moving_mean:0": sd[name + ".mean"],
                  "moving_variance:0": sd[name + ".var"]}
        tree[kname] = {h5lite.ATTRS: {"weight_names": [f"{kname}/{w}".encode() for w in ws]},
                       kname: {w: np.asarray(a, np.float32) for w, a in ws.items()}}
    own = {"net": "evalnet", "h": p.h, "w": p.w, "ca": p.ca, "cb": p.cb, "n_out": p.n_out, "alpha": p.alpha, "two_heads": bool(p.two_heads),
           "normalize_a": bool(p.cfg.normalize_a), "normalize_b": bool(p.cfg.normalize_b), "b_onehot": bool(p.b_onehot)}
    tree[h5lite.ATTRS] = {"layer_names": order, "backend": b"tensorflow", "keras_version": b"2.10.0", "imk_config": json.dumps(own)}
    h5lite.write(path, tree)
