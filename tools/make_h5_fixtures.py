#!/opt/conda/bin/python3.9
"""HDF5 fixtures for inconsistencymasks_amd/h5lite.py, written by the REAL HDF5 library through h5py.

The build image has no h5py on its main interpreter, but its Anaconda tree does (/opt/conda/bin/python3.9: h5py 3.3.0 on
libhdf5 1.10.6).  Run THIS script with that interpreter; it writes

  tests/golden/h5_keras_full_model.h5   a file laid out the way Keras 2.x `model.save(path)` / ModelCheckpoint lays out a
                                       full model (the reference's checkpoints: ISIC_2018/09_ISIC_2018_IM.py:74-76,
                                       functions.py:217): root attributes keras_version / backend / model_config /
                                       training_config, groups model_weights/<layer>/<layer>/<weight>:0 with the
                                       layer_names / weight_names attributes, optimizer_weights.  The network is the
                                       reference's get_unet at alpha 0.25, 32x32x3 input, 2 softmax outputs, with
                                       seeded random weights -- the LAYOUT is a restatement of Keras' hdf5_format
                                       conventions (TensorFlow is not installable here), the BYTES are h5py's.
  tests/golden/h5_keras_full_model.npz  the same arrays, by "<layer>/<weight>" -- what a reader has to return
  tests/golden/h5_cases.h5              container features a reader meets in the wild: chunked + gzip + shuffle +
                                       fletcher32 datasets, compact data, integer / float64 / big-endian types, scalar /
                                       variable-length-string / fixed-string-array attributes, groups with 300 members
                                       (several symbol-table nodes under a B-tree)
  tests/golden/h5_cases.npz             the expected values
  tests/golden/h5_keras_evalnet.h5/.npz the same for evalnet.get_evalnet_miou (two towers: `layer_names` in model.layers order, i.e. interleaved;
                                       Lambda / Input layers with their inbound nodes in model_config)
  tests/golden/h5_latest.h5             the same library told to use its LATEST format (superblock 3, version-2 object headers,
                                       link messages, version-4 layouts; a 20-member group in dense storage): what the reader
                                       takes of it and what it must refuse by name

Nothing here is read at test time except the four output files."""
import json
import os
import sys

import h5py
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "tools"))


def unet_keras_layers(c_in, n_out, alpha, h, w, act):
    """[(keras layer name, class, [(weight name, shape)], config)] in unet.py:4-67 creation order, pooling / upsampling /
    add / activation layers included (Keras writes a group for every layer; those without weights get an empty
    weight_names attribute)."""
    f = lambda v: int(v * alpha)
    c16, c32, c64, c128, c256 = f(16), f(32), f(64), f(128), f(256)
    L, n = [], {"conv2d": 0, "batch_normalization": 0, "activation": 0, "max_pooling2d": 0, "up_sampling2d": 0, "add": 0}

    def name(kind):
        i = n[kind]
        n[kind] += 1
        return kind if i == 0 else f"{kind}_{i}"

    def conv(k, ci, co, nm=None, activation="linear"):
        nm = nm or name("conv2d")
        L.append((nm, "Conv2D", [("kernel:0", (k, k, ci, co)), ("bias:0", (co,))],
                  {"name": nm, "filters": co, "kernel_size": [k, k], "padding": "same", "activation": activation}))

    def bn(c):
        nm = name("batch_normalization")
        L.append((nm, "BatchNormalization", [("gamma:0", (c,)), ("beta:0", (c,)), ("moving_mean:0", (c,)), ("moving_variance:0", (c,))],
                  {"name": nm, "momentum": 0.99, "epsilon": 0.001}))

    def plain(kind, cls):
        nm = name(kind)
        L.append((nm, cls, [], {"name": nm}))

    L.append(("input_1", "InputLayer", [], {"name": "input_1", "batch_input_shape": [None, h, w, c_in], "dtype": "float32"}))
    conv(1, c_in, c16); bn(c16); plain("activation", "Activation")
    for ci, co in [(c16, c16), (c16, c32), (c32, c64), (c64, c128)]:
        conv(3, ci, co); plain("activation", "Activation"); conv(1, co, co); bn(co); plain("activation", "Activation")
        plain("max_pooling2d", "MaxPooling2D")
    conv(3, c128, c256); plain("activation", "Activation"); conv(1, c256, c128); bn(c128); plain("activation", "Activation")
    for ci, f1, f2 in [(c128, c128, c64), (c64, c64, c32), (c32, c32, c16), (c16, c16, c16)]:
        plain("up_sampling2d", "UpSampling2D"); conv(1, ci, f1); bn(f1); plain("activation", "Activation"); plain("add", "Add")
        conv(3, f1, f1); plain("activation", "Activation"); conv(1, f1, f2); bn(f2); plain("activation", "Activation")
    conv(1, c16, n_out, nm="out", activation=act)
    return L


def write_keras_full_model(path, npz_path):
    rs = np.random.RandomState(20241004)
    layers = unet_keras_layers(3, 2, 0.25, 32, 32, "softmax")
    expect = {}
    with h5py.File(path, "w") as f:
        f.attrs["keras_version"] = "2.10.0"                      # h5py 3: a str attribute is a variable-length UTF-8 string
        f.attrs["backend"] = "tensorflow"
        cfg = {"class_name": "Functional", "config": {"name": "model", "layers": [
            {"class_name": cls, "config": c, "name": nm, "inbound_nodes": []} for nm, cls, _, c in layers]}}
        f.attrs["model_config"] = json.dumps(cfg)
        f.attrs["training_config"] = json.dumps({"loss": "dice_loss", "metrics": None, "optimizer_config": {
            "class_name": "Addons>AdamW", "config": {"learning_rate": 0.001, "weight_decay": 1e-4}}})
        g = f.create_group("model_weights")
        g.attrs["layer_names"] = np.array([nm.encode() for nm, *_ in layers])        # numpy 'S' array: fixed-length strings
        g.attrs["backend"] = b"tensorflow"                                          # bytes: a fixed-length string scalar
        g.attrs["keras_version"] = b"2.10.0"
        for nm, cls, ws, _ in layers:
            lg = g.create_group(nm)
            lg.attrs["weight_names"] = np.array([f"{nm}/{wn}".encode() for wn, _ in ws]) if ws else np.zeros((0,), "S1")
            for wn, shape in ws:
                if wn.startswith("moving_variance"):
                    a = rs.uniform(0.5, 1.5, shape).astype(np.float32)
                else:
                    a = (rs.standard_normal(shape) * 0.3).astype(np.float32)
                lg.create_dataset(f"{nm}/{wn}", data=a)              # -> model_weights/<layer>/<layer>/<weight>:0
                expect[f"{nm}/{wn}"] = a
        og = f.create_group("optimizer_weights")
        og.attrs["weight_names"] = np.array([b"AdamW/iter:0", b"AdamW/conv2d/kernel/m:0"])
        og.create_dataset("AdamW/iter:0", data=np.int64(1234))
        og.create_dataset("AdamW/conv2d/kernel/m:0", data=rs.standard_normal((1, 1, 3, 4)).astype(np.float32))
    np.savez(npz_path, **expect)


def write_keras_evalnet(path, npz_path):
    """evalnet.get_evalnet_miou (evalnet.py:48-73) at alpha 0.5, 64x64 (the smallest the six poolings allow), inputs of 3 and 2 channels, normalize_A only: the full-model layout
    with `layer_names` in the order of Keras' model.layers (by depth: the two towers interleave), Lambda / Input layers in model_config"""
    rs = np.random.RandomState(20241005)
    f16, chans = 8, [8, 16, 32, 64, 128]
    layers = [("input_1", "InputLayer", [], {"name": "input_1", "batch_input_shape": [None, 64, 64, 3]}, []),
              ("input_2", "InputLayer", [], {"name": "input_2", "batch_input_shape": [None, 64, 64, 2]}, []),
              ("lambda", "Lambda", [], {"name": "lambda"}, [[["input_1", 0, 0, {}]]])]
    conv = lambda nm, k, ci, co, inb: (nm, "Conv2D", [("kernel:0", (k, k, ci, co)), ("bias:0", (co,))], {"name": nm, "filters": co}, [[[inb, 0, 0, {}]]])
    bn = lambda nm, c, inb: (nm, "BatchNormalization", [("gamma:0", (c,)), ("beta:0", (c,)), ("moving_mean:0", (c,)), ("moving_variance:0", (c,))],
                            {"name": nm}, [[[inb, 0, 0, {}]]])
    # model.layers order: depth first, creation order inside a depth
    layers += [conv("conv2d", 1, 3, f16, "lambda"), conv("conv2d_3", 1, 2, f16, "input_2"), bn("batch_normalization", f16, "conv2d"),
               bn("batch_normalization_2", f16, "conv2d_3"), conv("conv2d_1", 3, f16, f16, "batch_normalization"),
               conv("conv2d_4", 3, f16, f16, "batch_normalization_2"), conv("conv2d_2", 1, f16, f16, "conv2d_1"), conv("conv2d_5", 1, f16, f16, "conv2d_4"),
               bn("batch_normalization_1", f16, "conv2d_2"), bn("batch_normalization_3", f16, "conv2d_5"),
               ("max_pooling2d", "MaxPooling2D", [], {"name": "max_pooling2d"}, [[["batch_normalization_1", 0, 0, {}]]]),
               ("max_pooling2d_1", "MaxPooling2D", [], {"name": "max_pooling2d_1"}, [[["batch_normalization_3", 0, 0, {}]]]),
               ("concatenate", "Concatenate", [], {"name": "concatenate"}, [[["max_pooling2d", 0, 0, {}], ["max_pooling2d_1", 0, 0, {}]]])]
    prev, ci, nc, nb = "concatenate", 2 * f16, 6, 4
    for i, co in enumerate(chans):
        layers += [conv(f"conv2d_{nc}", 3, ci, co, prev), conv(f"conv2d_{nc + 1}", 1, co, co, f"conv2d_{nc}"), bn(f"batch_normalization_{nb}", co, f"conv2d_{nc + 1}"),
                   (f"max_pooling2d_{i + 2}", "MaxPooling2D", [], {"name": f"max_pooling2d_{i + 2}"}, [[[f"batch_normalization_{nb}", 0, 0, {}]]])]
        prev, ci, nc, nb = f"max_pooling2d_{i + 2}", co, nc + 2, nb + 1
    layers += [("global_average_pooling2d", "GlobalAveragePooling2D", [], {"name": "global_average_pooling2d"}, [[[prev, 0, 0, {}]]])]
    for nm in ("iou", "detection"):
        layers.append((nm, "Dense", [("kernel:0", (chans[-1], 2)), ("bias:0", (2,))], {"name": nm, "units": 2, "activation": "sigmoid"},
                       [[["global_average_pooling2d", 0, 0, {}]]]))
    expect = {}
    with h5py.File(path, "w") as f:
        f.attrs["keras_version"] = "2.10.0"
        f.attrs["backend"] = "tensorflow"
        f.attrs["model_config"] = json.dumps({"class_name": "Functional", "config": {
            "name": "model_2", "layers": [{"class_name": cls, "config": c, "name": nm, "inbound_nodes": inb} for nm, cls, _, c, inb in layers],
            "input_layers": [["input_1", 0, 0], ["input_2", 0, 0]], "output_layers": [["iou", 0, 0], ["detection", 0, 0]]}})
        g = f.create_group("model_weights")
        g.attrs["layer_names"] = np.array([nm.encode() for nm, *_ in layers])
        g.attrs["backend"] = b"tensorflow"
        g.attrs["keras_version"] = b"2.10.0"
        for nm, cls, ws, _, _ in layers:
            lg = g.create_group(nm)
            lg.attrs["weight_names"] = np.array([f"{nm}/{wn}".encode() for wn, _ in ws]) if ws else np.zeros((0,), "S1")
            for wn, shape in ws:
                a = (rs.uniform(0.5, 1.5, shape) if wn.startswith("moving_variance") else rs.standard_normal(shape) * 0.3).astype(np.float32)
                lg.create_dataset(f"{nm}/{wn}", data=a)
                expect[f"{nm}/{wn}"] = a
    np.savez(npz_path, **expect)


def write_cases(path, npz_path):
    rs = np.random.RandomState(7)
    exp = {}
    with h5py.File(path, "w") as f:
        a = rs.standard_normal((37, 29)).astype(np.float32)
        f.create_dataset("chunked_gzip_shuffle", data=a, chunks=(8, 16), compression="gzip", shuffle=True)
        exp["chunked_gzip_shuffle"] = a
        b = rs.randint(-1000, 1000, (5, 7, 3)).astype(np.int16)
        f.create_dataset("chunked_plain", data=b, chunks=(2, 3, 3))
        exp["chunked_plain"] = b
        c = rs.standard_normal((19,)).astype(np.float64)
        f.create_dataset("chunked_fletcher", data=c, chunks=(5,), fletcher32=True)
        exp["chunked_fletcher"] = c
        d = np.arange(6, dtype=np.uint8).reshape(2, 3)
        f.create_dataset("contiguous_u8", data=d)
        exp["contiguous_u8"] = d
        e = rs.standard_normal((4, 4)).astype(">f4")
        f.create_dataset("big_endian_f4", data=e)
        exp["big_endian_f4"] = e.astype(np.float32)
        f.create_dataset("scalar_i64", data=np.int64(-5))
        exp["scalar_i64"] = np.int64(-5)
        f.create_dataset("empty_f4", shape=(0, 3), dtype="f4")
        exp["empty_f4"] = np.zeros((0, 3), np.float32)
        # compact layout: h5py exposes it only through the low-level API
        sp = h5py.h5s.create_simple((3,))
        pl = h5py.h5p.create(h5py.h5p.DATASET_CREATE)
        pl.set_layout(h5py.h5d.COMPACT)
        ds = h5py.h5d.create(f.id, b"compact_i32", h5py.h5t.NATIVE_INT32, sp, dcpl=pl)
        v = np.array([11, -22, 33], np.int32)
        ds.write(h5py.h5s.ALL, h5py.h5s.ALL, v)
        exp["compact_i32"] = v
        f.attrs["a_vlen_str"] = "variable-length äö"
        f.attrs["a_fixed_str"] = np.bytes_(b"fixed")
        f.attrs["a_f8"] = 2.5
        f.attrs["a_i32_vec"] = np.array([1, 2, 3], np.int32)
        f.attrs["a_S_array"] = np.array([b"alpha", b"be", b"gamma_delta"])
        f.attrs["a_vlen_array"] = np.array(["one", "three"], dtype=h5py.string_dtype())
        big = f.create_group("many")
        for i in range(300):                                         # 300 members: > one symbol node, a 2-level B-tree at K = 4 / 16
            big.create_dataset(f"member_{i:03d}", data=np.float32(i))
        deep = f.create_group("a/b/c")
        deep.create_dataset("leaf", data=np.array([1.5, 2.5], np.float32))
        deep.attrs["where"] = "a/b/c"
        exp["a/b/c/leaf"] = np.array([1.5, 2.5], np.float32)
    np.savez(npz_path, **{k.replace("/", "|"): v for k, v in exp.items()})


def write_latest(path):
    with h5py.File(path, "w", libver="latest") as f:
        f.attrs["x"] = "hello"
        g = f.create_group("g")
        g.create_dataset("d", data=np.arange(5, dtype="f4"))
        g.attrs["n"] = 3
        f.create_dataset("c", data=np.arange(12, dtype="i4").reshape(3, 4))
        f.create_dataset("k", data=np.arange(64, dtype="f4").reshape(8, 8), chunks=(4, 4))
        many = f.create_group("many")
        for i in range(20):
            many.create_dataset(f"m{i}", data=np.float32(i))


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    write_latest(os.path.join(GOLD, "h5_latest.h5"))
    write_keras_evalnet(os.path.join(GOLD, "h5_keras_evalnet.h5"), os.path.join(GOLD, "h5_keras_evalnet.npz"))
    write_keras_full_model(os.path.join(GOLD, "h5_keras_full_model.h5"), os.path.join(GOLD, "h5_keras_full_model.npz"))
    write_cases(os.path.join(GOLD, "h5_cases.h5"), os.path.join(GOLD, "h5_cases.npz"))
    for n in ("h5_keras_full_model.h5", "h5_keras_full_model.npz", "h5_cases.h5", "h5_cases.npz", "h5_latest.h5", "h5_keras_evalnet.h5", "h5_keras_evalnet.npz"):
        print(n, os.path.getsize(os.path.join(GOLD, n)), "bytes; h5py", h5py.__version__, "libhdf5", h5py.version.hdf5_version)
