"""inconsistencymasks_amd/h5lite.py + keras_h5.py (the reference's model files are Keras HDF5 checkpoints).

The reader is checked against files the real HDF5 library wrote through h5py (tests/golden/h5_*.h5, made by
tools/make_h5_fixtures.py under /opt/conda/bin/python3.9); the writer against its own reader here and -- where that
interpreter exists (it does in the build image and on the GPU boxes) -- against h5py reading the file."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
CONDA_PY = "/opt/conda/bin/python3.9"

from inconsistencymasks_amd import h5lite as H  # noqa: E402


def test_reads_what_libhdf5_wrote():
    f = H.File(os.path.join(GOLD, "h5_cases.h5"))
    assert f.superblock_version == 0
    exp = np.load(os.path.join(GOLD, "h5_cases.npz"))
    for k in exp.files:
        d = f[k.replace("|", "/")]
        a = d[...]
        assert np.asarray(a).dtype == exp[k].dtype and np.array_equal(a, exp[k]), k
        assert tuple(d.shape) == exp[k].shape
    assert f.attrs["a_vlen_str"] == "variable-length äö"            # variable-length UTF-8: through the global heap
    assert f.attrs["a_fixed_str"] == b"fixed"
    assert f.attrs["a_f8"] == 2.5 and list(f.attrs["a_i32_vec"]) == [1, 2, 3]
    assert list(f.attrs["a_S_array"]) == [b"alpha", b"be", b"gamma_delta"]
    assert list(f.attrs["a_vlen_array"]) == ["one", "three"]
    many = f["many"]                                                 # 300 members: several symbol nodes under a B-tree
    assert len(many) == 300 and many.keys() == [f"member_{i:03d}" for i in range(300)]
    assert all(float(many[f"member_{i:03d}"][...]) == i for i in (0, 7, 8, 150, 299))
    assert f["a/b/c"].attrs["where"] == "a/b/c" and "a/b/c/leaf" in f and "a/b/x" not in f
    with pytest.raises(KeyError):
        f["a/b/c/leaf/deeper"]
    assert H.is_hdf5(os.path.join(GOLD, "h5_cases.h5")) and not H.is_hdf5(os.path.join(GOLD, "h5_cases.npz"))


def test_latest_format_taken_or_refused_by_name():
    f = H.File(os.path.join(GOLD, "h5_latest.h5"))
    assert f.superblock_version == 3
    assert f.attrs["x"] == "hello" and f["g"].attrs["n"] == 3
    assert np.array_equal(f["g/d"][...], np.arange(5, dtype=np.float32))
    assert np.array_equal(f["c"][...], np.arange(12, dtype=np.int32).reshape(3, 4))
    with pytest.raises(H.H5Unsupported, match="dense storage"):
        f["many"].keys()
    with pytest.raises(H.H5Unsupported, match="layout"):
        f["k"][...]


def test_matlab_v73_file_with_user_block():
    p = "/usr/local/lib/python3.10/dist-packages/scipy/io/matlab/tests/data/testhdf5_7.4_GLNX86.mat"
    if not os.path.exists(p):
        pytest.skip("scipy's test data are not installed")
    f = H.File(p)                                                    # written by MATLAB: a 512-byte user block in front
    assert np.allclose(np.asarray(f["testdouble"][...]).ravel(), np.arange(9) * np.pi / 4)


def test_keras_full_model_fixture_to_state_dict():
    from inconsistencymasks_amd import keras_h5 as K
    path = os.path.join(GOLD, "h5_keras_full_model.h5")
    exp = np.load(os.path.join(GOLD, "h5_keras_full_model.npz"))
    weights_of, act, shape = K.read_keras_h5(path)
    assert act == "softmax" and shape == [32, 32, 3]
    assert len(weights_of) == 38 and sum(len(v) for v in weights_of.values()) == len(exp.files) == 104
    for k in exp.files:
        lay, w = k.split("/")
        assert np.array_equal(weights_of[lay][w.split(":")[0]], exp[k]), k
    sd, cfg = K.state_dict_from_keras_h5(path)
    assert cfg == {"h": 32, "w": 32, "c_in": 3, "n_out": 2, "alpha": 0.25, "act_out": "softmax"}
    table = K.layer_table(3, 2, 0.25)
    kn = K.keras_layer_names(table)
    assert kn["in.c"] == "conv2d" and kn["e1.c3"] == "conv2d_1" and kn["d9.c1"] == "conv2d_22" and kn["d9.bnb"] == "batch_normalization_13"
    for name, kind, k, ci, co in table:
        if kind == "conv":
            assert sd[name + ".w"].shape == (k, k, ci, co) and np.array_equal(sd[name + ".w"], exp[kn[name] + "/kernel:0"])
            assert np.array_equal(sd[name + ".b"], exp[kn[name] + "/bias:0"])
        else:
            assert np.array_equal(sd[name + ".var"], exp[kn[name] + "/moving_variance:0"])
            assert np.array_equal(sd[name + ".gamma"], exp[kn[name] + "/gamma:0"])
    # the optimizer's slots are there and ignored
    assert int(H.File(path)["optimizer_weights/AdamW/iter:0"][...]) == 1234


def _tree():
    rs = np.random.RandomState(1)
    return {H.ATTRS: {"layer_names": [b"conv2d", b"batch_normalization", b"out"], "backend": b"tensorflow", "keras_version": "2.10.0",
                      "f": np.float64(1.5), "iv": np.array([1, 2, 3], np.int32)},
            "conv2d": {H.ATTRS: {"weight_names": [b"conv2d/kernel:0", b"conv2d/bias:0"]},
                       "conv2d": {"kernel:0": rs.standard_normal((3, 3, 4, 8)).astype(np.float32), "bias:0": np.zeros(8, np.float32)}},
            "batch_normalization": {H.ATTRS: {"weight_names": np.zeros((0,), "S1")}},
            "out": {"x": H.Dataset(np.arange(6, dtype=np.int64).reshape(2, 3), {"unit": "px"}), "h": np.float16(2.5),
                    "e": np.zeros((0, 4), np.float32), "u8": np.arange(5, dtype=np.uint8),
                    "t": np.arange(24, dtype=np.float32).reshape(2, 3, 4).transpose(2, 0, 1)},
            "many": {f"m{i:03d}": np.float32(i) for i in range(100)}}


def test_writer_round_trip(tmp_path):
    tree = _tree()
    p = str(tmp_path / "w.h5")
    H.write(p, tree)
    f = H.File(p)
    assert f.keys() == ["batch_normalization", "conv2d", "many", "out"]
    assert H.load_attr_list(f, "layer_names") == ["conv2d", "batch_normalization", "out"]
    assert f.attrs["backend"] == b"tensorflow" and f.attrs["keras_version"] == "2.10.0" and f.attrs["f"] == 1.5
    assert np.array_equal(f["conv2d/conv2d/kernel:0"][...], tree["conv2d"]["conv2d"]["kernel:0"])
    assert H.load_attr_list(f["conv2d"], "weight_names") == ["conv2d/kernel:0", "conv2d/bias:0"]
    assert H.load_attr_list(f["batch_normalization"], "weight_names") == [] and len(f["batch_normalization"]) == 0
    assert np.array_equal(f["out/x"][...], np.arange(6).reshape(2, 3)) and f["out/x"].attrs["unit"] == "px"
    assert f["out/h"][...] == np.float16(2.5) and f["out/h"].shape == () and f["out/e"].shape == (0, 4)
    assert np.array_equal(f["out/t"][...], tree["out"]["t"]) and len(f["many"]) == 100 and float(f["many/m042"][...]) == 42
    with pytest.raises(H.H5Error, match="64 KB"):
        H.write(str(tmp_path / "big.h5"), {H.ATTRS: {"big": np.zeros(20000, np.float32)}})


@pytest.mark.skipif(not os.path.exists(CONDA_PY), reason="no interpreter with h5py in this image")
def test_writer_output_read_by_h5py(tmp_path):
    """the real library reads what h5lite.write produced: values, attributes, the Keras save_weights layout"""
    from inconsistencymasks_amd import keras_h5 as K
    tree = _tree()
    p, q = str(tmp_path / "w.h5"), str(tmp_path / "k.h5")
    H.write(p, tree)
    np.save(str(tmp_path / "kernel.npy"), tree["conv2d"]["conv2d"]["kernel:0"])
    rs = np.random.RandomState(3)
    table = K.layer_table(1, 3, 0.5)
    sd = {}
    for name, kind, k, ci, co in table:
        for s, shape in ((".w", (k, k, ci, co)), (".b", (co,))) if kind == "conv" else ((".gamma", (co,)), (".beta", (co,)), (".mean", (co,)), (".var", (co,))):
            sd[name + s] = rs.standard_normal(shape).astype(np.float32)
    K.save_keras_weights(sd, q)
    np.savez(str(tmp_path / "sd.npz"), **sd)
    script = f'''
import json, sys, numpy as np, h5py
out = {{}}
with h5py.File({p!r}, "r") as f:
    out["keys"] = sorted(f)
    out["layer_names"] = [v.decode() for v in f.attrs["layer_names"]]
    out["kernel_equal"] = bool(np.array_equal(f["conv2d/conv2d/kernel:0"][...], np.load({str(tmp_path / "kernel.npy")!r})))
    out["x"] = f["out/x"][...].tolist(); out["unit"] = f["out/x"].attrs["unit"].decode()
    out["h"] = float(f["out/h"][()]); out["h_shape"] = list(f["out/h"].shape); out["e_shape"] = list(f["out/e"].shape)
    out["t_equal"] = bool(np.array_equal(f["out/t"][...], np.arange(24, dtype=np.float32).reshape(2, 3, 4).transpose(2, 0, 1)))
    out["many"] = len(f["many"]); out["m42"] = float(f["many/m042"][()])
    n = []
    f.visit(n.append); out["visited"] = len(n)
sd = np.load({str(tmp_path / "sd.npz")!r})
with h5py.File({q!r}, "r") as f:                       # Keras' load_weights_from_hdf5_group, restated
    names = [v.decode() for v in f.attrs["layer_names"]]
    got = []
    for nm in names:
        g = f[nm]
        got += [np.asarray(g[w.decode()]) for w in g.attrs["weight_names"]]
    out["n_layers"], out["n_arrays"] = len(names), len(got)
    out["first"], out["last"] = names[0], names[-1]
    out["backend"] = f.attrs["backend"].decode()
order = []
for k in sd.files:
    order.append(k)
out["all_found"] = all(any(np.array_equal(sd[k], a) for a in got) for k in sd.files)
print(json.dumps(out))
'''
    r = subprocess.run([CONDA_PY, "-W", "ignore", "-c", script], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["keys"] == ["batch_normalization", "conv2d", "many", "out"] and out["layer_names"] == ["conv2d", "batch_normalization", "out"]
    assert out["kernel_equal"] and out["t_equal"] and out["x"] == [[0, 1, 2], [3, 4, 5]] and out["unit"] == "px"
    assert out["h"] == 2.5 and out["h_shape"] == [] and out["e_shape"] == [0, 4] and out["many"] == 100 and out["m42"] == 42.0
    assert out["visited"] == 4 + 2 + 1 + 5 + 100
    assert out["n_layers"] == 38 and out["n_arrays"] == 104 and out["first"] == "conv2d" and out["last"] == "out"
    assert out["backend"] == "tensorflow" and out["all_found"]
    # the library's own tools walk every object of both files: h5repack rewrites them, h5diff finds the copies identical
    repack, diff = "/opt/conda/bin/h5repack", "/opt/conda/bin/h5diff"
    if os.path.exists(repack) and os.path.exists(diff):
        for src in (p, q):
            dst = src + ".repacked"
            r = subprocess.run([repack, src, dst], capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-1000:]
            r = subprocess.run([diff, src, dst], capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-1000:]


def test_keras_weights_file_round_trip_and_get_weights_order(tmp_path):
    """save_keras_weights -> state_dict_from_keras_h5 is the identity; the arrays come out in model.get_weights() order"""
    from inconsistencymasks_amd import keras_h5 as K
    rs = np.random.RandomState(5)
    table = K.layer_table(3, 9, 1.25)
    sd = {}
    for name, kind, k, ci, co in table:
        for s, shape in ((".w", (k, k, ci, co)), (".b", (co,))) if kind == "conv" else ((".gamma", (co,)), (".beta", (co,)), (".mean", (co,)), (".var", (co,))):
            sd[name + s] = rs.standard_normal(shape).astype(np.float32)
    p = str(tmp_path / "k.h5")
    K.save_keras_weights(sd, p, own={"h": 64, "w": 96, "c_in": 3, "n_out": 9, "alpha": 1.25, "act_out": "softmax"})
    back, cfg = K.state_dict_from_keras_h5(p)
    assert cfg == {"h": 64, "w": 96, "c_in": 3, "n_out": 9, "alpha": 1.25, "act_out": "softmax"}
    assert set(back) == set(sd) and all(np.array_equal(back[k], sd[k]) for k in sd)
    f = H.File(p)
    flat = []
    for nm in H.load_attr_list(f, "layer_names"):
        flat += [f[nm][w][...] for w in H.load_attr_list(f[nm], "weight_names")]
    want = K.keras_weight_list(sd, table)
    assert len(flat) == len(want) and all(np.array_equal(a, b) for a, b in zip(flat, want))
    # a weights-only file without the package's own attribute does not say the input size
    K.save_keras_weights(sd, p)
    _, cfg = K.state_dict_from_keras_h5(p)
    assert cfg["h"] is None and cfg["act_out"] is None and cfg["alpha"] == 1.25


def test_keras_evalnet_fixture_to_state_dict(tmp_path):
    """evalnet.get_evalnet_miou's checkpoint layout (two towers: `layer_names` in model.layers order, i.e. interleaved): the mapping goes
    by creation order all the same, Dense kernels become 1x1 'conv' weights, the normalisation flags come from the Lambda layers' inputs"""
    from inconsistencymasks_amd import keras_h5 as K
    path = os.path.join(GOLD, "h5_keras_evalnet.h5")
    exp = np.load(os.path.join(GOLD, "h5_keras_evalnet.npz"))
    assert K.keras_h5_kind(path) == "evalnet" and K.keras_h5_kind(os.path.join(GOLD, "h5_keras_full_model.h5")) == "unet"
    sd, m = K.evalnet_state_dict_from_keras_h5(path)
    assert m == {"h": 64, "w": 64, "ca": 3, "cb": 2, "n_out": 2, "alpha": 0.5, "two_heads": True, "normalize_a": True, "normalize_b": False}
    table = K.evalnet_layer_table(3, 2, 2, 0.5, True)
    kn = K.evalnet_keras_layer_names(table)
    assert kn["a.in.c"] == "conv2d" and kn["b.in.c"] == "conv2d_3" and kn["m1.c3"] == "conv2d_6" and kn["m5.bn"] == "batch_normalization_8"
    n = 0
    for name, kind, k, ci, co in table:
        if name in ("iou", "detection"):
            assert sd[name + ".w"].shape == (1, 1, ci, co) and np.array_equal(sd[name + ".w"].reshape(ci, co), exp[name + "/kernel:0"])
            assert np.array_equal(sd[name + ".b"], exp[name + "/bias:0"])
            n += 2
        elif kind == "conv":
            assert np.array_equal(sd[name + ".w"], exp[kn[name] + "/kernel:0"]) and np.array_equal(sd[name + ".b"], exp[kn[name] + "/bias:0"])
            n += 2
        else:
            for ours, theirs in (("gamma", "gamma"), ("beta", "beta"), ("mean", "moving_mean"), ("var", "moving_variance")):
                assert np.array_equal(sd[f"{name}.{ours}"], exp[f"{kn[name]}/{theirs}:0"])
            n += 4
    assert n == len(exp.files) == len(sd)
    with pytest.raises(ValueError, match="not a unet.get_unet"):
        K.state_dict_from_keras_h5(path)
    with pytest.raises(ValueError, match="not an evalnet"):
        K.evalnet_state_dict_from_keras_h5(os.path.join(GOLD, "h5_keras_full_model.h5"))


def test_mutated_files_end_in_h5error_not_in_a_hang(tmp_path):
    """A damaged checkpoint must raise H.H5Error (a ValueError) -- not hang, not exhaust the stack, not leak numpy's or zlib's own
    exception types.  Round 5: mutating the four fixtures found group links back into their own ancestry (visit_datasets recursed for
    ever), a link with the empty name (looked up as the group itself), object-header continuations and B-tree nodes pointing at
    themselves, and zlib / numpy errors coming out raw.  600 seeded mutations (bit flips, 8 random bytes, truncations), each walked in
    full (every dataset read, every attribute decoded) under a 10 s alarm."""
    import glob
    import signal
    seeds = [open(f, "rb").read() for f in sorted(glob.glob(os.path.join(GOLD, "*.h5")))]
    assert len(seeds) >= 4
    rng = np.random.default_rng(5)

    class Hang(Exception):
        pass

    def on_alarm(sig, frame):
        raise Hang()

    old = signal.signal(signal.SIGALRM, on_alarm)
    outcomes = {"ok": 0, "H5Error": 0}
    try:
        for it in range(600):
            s = bytearray(seeds[it % len(seeds)])
            k = it % 3
            if k == 0:
                for _ in range(int(rng.integers(1, 6))):
                    s[int(rng.integers(0, len(s)))] ^= 1 << int(rng.integers(0, 8))
            elif k == 1:
                for _ in range(int(rng.integers(1, 4))):
                    o = int(rng.integers(0, len(s) - 8))
                    s[o:o + 8] = rng.integers(0, 256, 8).astype("uint8").tobytes()
            else:
                s = s[:int(rng.integers(8, len(s)))]
            p = str(tmp_path / "m.h5")
            with open(p, "wb") as fh:
                fh.write(s)
            signal.alarm(10)
            try:
                with H.File(p) as f:
                    for name, ds in f.visit_datasets():
                        ds.read()
                        dict(ds.attrs)
                    dict(f.attrs)
                outcomes["ok"] += 1
            except H.H5Error:
                outcomes["H5Error"] += 1
            finally:
                signal.alarm(0)
    finally:
        signal.signal(signal.SIGALRM, old)
    assert outcomes["ok"] > 50 and outcomes["H5Error"] > 50, outcomes      # both kinds of file were in the sample
