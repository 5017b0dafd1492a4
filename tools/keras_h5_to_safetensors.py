#!/usr/bin/env python3
"""Interchange between the reference's Keras checkpoints and this repository's safetensors model files.

The reference saves every U-Net with ModelCheckpoint as a full-model Keras HDF5 file (`*.h5`,
ISIC_2018/09_ISIC_2018_IM.py:74-76, 131-135; functions.py:217).  The package reads those files itself
(`functions.load_model` sniffs the HDF5 signature; inconsistencymasks_amd/keras_h5.py + h5lite.py, no h5py needed); this is the
command-line form of the same conversions:

    python tools/keras_h5_to_safetensors.py model.h5 model.safetensors.h5 --height 256 --width 256
    python tools/keras_h5_to_safetensors.py --to-keras-npz model.safetensors.h5 weights.npz

* HDF5 -> safetensors: the Keras layers of `unet.get_unet` are matched to the layer names of include/imk.h
  (`imk_layer_info`: "in.c", "e1.c3", ..., "d9.bnb", "out") by CREATION ORDER: Keras numbers its auto-named layers
  (`conv2d_17`, `batch_normalization_9`, ...) in the order unet.py:46-67 creates them, which is the order of
  `oracle/unet_oracle.layer_table` and of the flat parameter vector.  Kernels stay HWIO, exactly as Keras stores them.
* safetensors -> `model.get_weights()` order (.npz, arrays named w000, w001, ...): a TensorFlow user restores it with
  `model.set_weights([d[k] for k in sorted(d.files)])` (zero-padded names) on a model built by the reference's `get_unet`.
  The same .npz is accepted as INPUT (`--from-keras-npz`), for checkpoints dumped with `np.savez(..., *model.get_weights())`.

    python tools/keras_h5_to_safetensors.py --to-keras-h5 model.safetensors.h5 weights.h5     # Keras save_weights layout

The mapping logic lives in inconsistencymasks_amd/keras_h5.py (unit-tested in tests/test_cpu_api.py, tests/test_cpu_h5lite.py)."""
import argparse
import os
import re
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inconsistencymasks_amd.keras_h5 import (infer_config, keras_weight_list, layer_table, match_keras_layers,   # noqa: E402,F401
                                             save_keras_weights, state_dict_from_keras, state_dict_from_keras_h5,
                                             state_dict_from_weight_list)


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
    ap.add_argument("--to-keras-h5", action="store_true", help="safetensors -> Keras save_weights HDF5 (get_unet(...).load_weights)")
    a = ap.parse_args(argv)
    if a.to_keras_h5:
        from safetensors import safe_open
        with safe_open(a.src, framework="np") as f:
            meta = f.metadata()
            sd = {k: f.get_tensor(k) for k in f.keys()}
        save_keras_weights(sd, a.dst, int(meta["c_in"]), int(meta["n_out"]), float(meta["alpha"]))
        return 0
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
        sd, cfg = state_dict_from_keras_h5(a.src)
        c_in, n_out, alpha, act = cfg["c_in"], cfg["n_out"], cfg["alpha"], (a.act_out or cfg["act_out"])
        if cfg["h"] is not None and (a.height, a.width) == (256, 256):
            a.height, a.width = cfg["h"], cfg["w"]
    if act not in ("sigmoid", "softmax"):
        print("output activation unknown: pass --act-out sigmoid|softmax", file=sys.stderr)
        return 2
    write_safetensors(sd, a.dst, a.height, a.width, c_in, n_out, alpha, act)
    return 0


if __name__ == "__main__":
    sys.exit(main())
