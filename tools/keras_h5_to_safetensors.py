#!/usr/bin/env python3
"""Interchange between the reference's Keras checkpoints and this repository's safetensors model files.

The reference saves every U-Net with ModelCheckpoint as a full-model Keras HDF5 file (`*.h5`,
ISIC_2018/09_ISIC_2018_IM.py:74-76, 131-135; functions.py:217).  h5py / TensorFlow are not available in the build
image, so this converter is meant to run OFFLINE on a machine that has `h5py` (TensorFlow is not needed):

    python tools/keras_h5_to_safetensors.py model.h5 model.safetensors.h5 --height 256 --width 256
    python tools/keras_h5_to_safetensors.py --to-keras-npz model.safetensors.h5 weights.npz

* HDF5 -> safetensors: the Keras layers of `unet.get_unet` are matched to the layer names of include/imk.h
  (`imk_layer_info`: "in.c", "e1.c3", ..., "d9.bnb", "out") by CREATION ORDER: Keras numbers its auto-named layers
  (`conv2d_17`, `batch_normalization_9`, ...) in the order unet.py:46-67 creates them, which is the order of
  `oracle/unet_oracle.layer_table` and of the flat parameter vector.  Kernels stay HWIO, exactly as Keras stores them.
* safetensors -> `model.get_weights()` order (.npz, arrays named w000, w001, ...): a TensorFlow user restores it with
  `model.set_weights([d[k] for k in sorted(d.files)])` (zero-padded names) on a model built by the reference's `get_unet`.
  The same .npz is accepted as INPUT (`--from-keras-npz`), for checkpoints dumped with `np.savez(..., *model.get_weights())`.

The mapping logic is pure Python / numpy and is unit-tested without h5py (tests/test_cpu_api.py)."""
import argparse
import json
import re
import sys

import numpy as np


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


def read_keras_h5(path):
    """Keras layer name -> weights dict, and the output activation, from a full-model or weights-only HDF5 file"""
    import h5py
    with h5py.File(path, "r") as f:
        g = f["model_weights"] if "model_weights" in f else f
        act = None
        cfg = f.attrs.get("model_config")
        if cfg is not None:
            cfg = json.loads(cfg.decode() if isinstance(cfg, bytes) else cfg)
            for l in cfg["config"]["layers"]:
                if l["config"].get("name") == "out":
                    act = l["config"].get("activation")
        weights_of = {}
        for lname in g:
            found = {}
            g[lname].visititems(lambda n, o: found.__setitem__(n.split("/")[-1].split(":")[0], np.array(o))
                                if isinstance(o, h5py.Dataset) else None)
            if found:
                weights_of[lname] = found
    return weights_of, act


def write_safetensors(sd, path, h, w, c_in, n_out, alpha, act_out):
    import torch
    from safetensors.torch import save_file
    meta = {"h": str(h), "w": str(w), "c_in": str(c_in), "n_out": str(n_out), "alpha": repr(float(alpha)), "act_out": act_out}
    save_file({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, path, metadata=meta)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--act-out", choices=("sigmoid", "softmax"), help="output activation if the file does not say")
    ap.add_argument("--to-keras-npz", action="store_true", help="safetensors -> model.get_weights() order (.npz)")
    ap.add_argument("--from-keras-npz", action="store_true", help="src is an .npz of model.get_weights()")
    a = ap.parse_args(argv)
    if a.to_keras_npz:
        from safetensors import safe_open
        with safe_open(a.src, framework="np") as f:
            meta = f.metadata()
            sd = {k: f.get_tensor(k) for k in f.keys()}
        table = layer_table(int(meta["c_in"]), int(meta["n_out"]), float(meta["alpha"]))
        np.savez(a.dst, **{f"w{i:03d}": v for i, v in enumerate(keras_weight_list(sd, table))})
        return 0
    if a.from_keras_npz:
        d = np.load(a.src)
        # np.savez(path, *model.get_weights()) names the arrays arr_0 ... arr_113: order by the NUMERIC suffix (a plain sort
        # puts arr_10 in front of arr_2); this tool's own zero-padded w000 ... names sort the same either way
        arrays = [d[k] for k in sorted(d.files, key=lambda k: int(re.search(r"(\d+)$", k).group(1)))]
        c_in, n_out, alpha = infer_config(arrays[0].shape, arrays[-2].shape)
        table = layer_table(c_in, n_out, alpha)
        sd, act = state_dict_from_weight_list(arrays, table), a.act_out
    else:
        weights_of, act = read_keras_h5(a.src)
        convs = sorted([n for n in weights_of if re.fullmatch(r"conv2d(_\d+)?", n)], key=_suffix)
        c_in, n_out, alpha = infer_config(weights_of[convs[0]]["kernel"].shape, weights_of["out"]["kernel"].shape)
        table = layer_table(c_in, n_out, alpha)
        sd, act = state_dict_from_keras(weights_of, table), (a.act_out or act)
    if act not in ("sigmoid", "softmax"):
        print("output activation unknown: pass --act-out sigmoid|softmax", file=sys.stderr)
        return 2
    write_safetensors(sd, a.dst, a.height, a.width, c_in, n_out, alpha, act)
    return 0


if __name__ == "__main__":
    sys.exit(main())
