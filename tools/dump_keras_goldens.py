#!/usr/bin/env python3
"""Pin the floating-point path to the real reference: run where TensorFlow >= 2.10 and tensorflow_addons exist (NOT in the
build container, NOT on the GPU box -- neither has them), with the reference checkout on disk:

    python tools/dump_keras_goldens.py --reference /path/to/InconsistencyMasks --out tests/golden

For each case it builds the reference's OWN model (`from unet import get_unet`, unet.py:46-67) under the policy every
hot-path script sets (`mixed_float16`, ISIC_2018/09_ISIC_2018_IM.py:16), gives it non-trivial BatchNorm statistics and
biases, and records
  * the weights in `model.get_weights()` order,
  * a seeded uint8 batch and `model.predict` on it (functions.py:3157),
  * the output of every Conv2D layer on that batch (`act_<our layer name>`: the post-ReLU, pre-BatchNorm tensors this
    repository stores, mapped by creation order like the weights) -- a first mismatch can then be localised layer by layer,
  * the weights after `--steps` steps of `model.fit` with `tfa.optimizers.AdamW(learning_rate=LR, weight_decay=WD)` and the
    script's loss ('mse' / CategoricalCrossentropy()), batch by batch in file order (functions.py:207-218), and the
    per-step losses,
into `tests/golden/keras_<case>.npz` (a few hundred KB each: commit them).  `tests/test_gpu_keras_goldens.py` then checks
the HIP path AND oracle/unet_oracle.py against these numbers; without the files that test is skipped with this reason, and
the U-Net part of the oracle stays "parity unpinned" (DESIGN.md section 6).
Only arrays and a JSON string of versions are written: no reference source text."""
import argparse
import json
import os
import sys

import numpy as np

CASES = {      # small shapes: the fixtures stay small and the GPU test fast; widths cover alpha 0.5 and 1
    "isic": dict(h=64, w=64, c=3, k=1, alpha=0.5, act="sigmoid", loss="mse", batch=4),
    "suim": dict(h=48, w=64, c=3, k=9, alpha=1.0, act="softmax", loss="cce", batch=2),
    "hela": dict(h=32, w=48, c=1, k=3, alpha=1.0, act="sigmoid", loss="mse", batch=3),
}
LR, WD = 0.003, 0.0001      # config.ini:10-11


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", required=True, help="checkout of MichaelVorndran/InconsistencyMasks (for its unet.py)")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--cases", default=",".join(CASES))
    a = ap.parse_args(argv)
    import tensorflow as tf
    import tensorflow_addons as tfa
    from tensorflow.keras import mixed_precision
    mixed_precision.set_global_policy("mixed_float16")
    sys.path.insert(0, a.reference)
    cwd = os.getcwd()
    os.chdir(a.reference)            # the reference's modules read config.ini relative to the working directory
    from unet import get_unet
    os.chdir(cwd)
    os.makedirs(a.out, exist_ok=True)
    for name in a.cases.split(","):
        cs = CASES[name]
        tf.keras.utils.set_random_seed(1234)
        model = get_unet(cs["h"], cs["w"], cs["c"], cs["k"], cs["alpha"], "relu", cs["act"])
        rng = np.random.default_rng(7)
        w0 = []
        for v, arr in zip(model.weights, model.get_weights()):      # kernels stay he_normal; the rest becomes non-trivial
            n = v.name
            if "moving_mean" in n: arr = (0.4 + 0.1 * rng.standard_normal(arr.shape)).astype(np.float32)
            elif "moving_variance" in n: arr = (0.5 + 0.5 * rng.random(arr.shape)).astype(np.float32)
            elif "gamma" in n: arr = (0.8 + 0.4 * rng.random(arr.shape)).astype(np.float32)
            elif "beta" in n: arr = (0.1 * rng.standard_normal(arr.shape)).astype(np.float32)
            elif "bias" in n: arr = (0.05 * rng.standard_normal(arr.shape)).astype(np.float32)
            w0.append(arr)
        model.set_weights(w0)
        b, steps = cs["batch"], a.steps
        yy, xx = np.mgrid[0:cs["h"], 0:cs["w"]]
        x = (127 + 80 * np.sin(xx / 7.0)[None, :, :, None] * np.cos(yy / 5.0)[None, :, :, None]
             + rng.integers(-30, 30, (b * steps, cs["h"], cs["w"], cs["c"]))).clip(0, 255).astype(np.uint8)
        if cs["loss"] == "mse":
            y = (rng.random((b * steps, cs["h"], cs["w"], cs["k"])) > 0.6).astype(np.uint8)
            if name == "hela":
                y[..., 2] *= 3                                        # parse_image_hela: position channel x 3
            target, loss = y.astype(np.float32), "mse"
        else:
            y = rng.integers(0, cs["k"], (b * steps, cs["h"], cs["w"])).astype(np.uint8)
            target, loss = np.eye(cs["k"], dtype=np.float32)[y], tf.keras.losses.CategoricalCrossentropy()
        probs = model.predict(x[:b], batch_size=b, verbose=0).astype(np.float32)
        # every conv's output (what libimk stores: post-ReLU, pre-BatchNorm), keyed by OUR layer names through the creation-order map
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import keras_h5_to_safetensors as K
        table = K.layer_table(cs["c"], cs["k"], cs["alpha"])
        name_map = K.match_keras_layers([l.name for l in model.layers if l.weights], table)
        convs = [(ours, model.get_layer(kn)) for ours, kn in name_map.items() if ours != "out" and not ours.endswith(("bn", "bna", "bnb"))]
        probe = tf.keras.Model(model.input, [l.output for _, l in convs])
        acts = probe.predict(x[:b], batch_size=b, verbose=0)
        act_out = {f"act_{ours}": np.asarray(a, np.float32) for (ours, _), a in zip(convs, acts)}
        model.compile(optimizer=tfa.optimizers.AdamW(learning_rate=LR, weight_decay=WD), loss=loss)
        losses = []
        for s in range(steps):       # one optimizer step per call, batches in order (fit(shuffle=False) over one batch each)
            h = model.fit(x[s * b:(s + 1) * b], target[s * b:(s + 1) * b], batch_size=b, epochs=1, shuffle=False, verbose=0)
            losses.append(float(h.history["loss"][0]))
        w1 = model.get_weights()
        meta = {"case": name, **cs, "steps": steps, "lr": LR, "wd": WD, "tensorflow": tf.__version__,
                "tensorflow_addons": tfa.__version__, "policy": "mixed_float16", "numpy": np.__version__}
        out = {f"w0_{i:03d}": v for i, v in enumerate(w0)}
        out.update({f"w1_{i:03d}": v for i, v in enumerate(w1)})
        out.update(act_out)
        out.update(x=x, y=y, probs=probs, losses=np.asarray(losses, np.float64), meta=np.asarray(json.dumps(meta)))
        path = os.path.join(a.out, f"keras_{name}.npz")
        np.savez_compressed(path, **out)
        print("wrote", path, {k: meta[k] for k in ("tensorflow", "tensorflow_addons")}, "losses", losses)
    return 0


if __name__ == "__main__":
    sys.exit(main())
