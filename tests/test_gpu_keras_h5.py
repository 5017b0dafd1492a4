"""functions.load_model / save_model on Keras HDF5 checkpoints (ISIC_2018/09_ISIC_2018_IM.py:74-76): the file the
HDF5 library wrote in Keras' full-model layout (tests/golden/h5_keras_full_model.h5) goes straight into the HIP path, and a
trained model goes out as a Keras save_weights file and comes back bit for bit."""
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)


def test_load_model_takes_a_keras_full_model_checkpoint():
    from inconsistencymasks_amd import functions as F
    from inconsistencymasks_amd import keras_h5 as K
    from oracle import unet_oracle as U
    path = os.path.join(GOLD, "h5_keras_full_model.h5")
    m = F.load_model(path, custom_objects={"dice_loss": None})
    assert (m.plan.h, m.plan.w, m.plan.c_in, m.plan.n_out, m.plan.alpha, m.plan.act_out) == (32, 32, 3, 2, 0.25, "softmax")
    sd, _ = K.state_dict_from_keras_h5(path)
    mine = m.state_dict()
    assert set(mine) == set(sd) and all(np.array_equal(mine[k].numpy(), sd[k]) for k in sd)
    x = np.random.default_rng(0).integers(0, 256, (4, 32, 32, 3), dtype=np.uint8)
    got = m.predict_device(torch.from_numpy(x).cuda()).cpu().numpy()
    ref = U.forward({k: torch.from_numpy(v) for k, v in sd.items()}, x, 3, 2, 0.25, "softmax", emulate_fp16=True).numpy()
    assert got.shape == ref.shape == (4, 32, 32, 2)
    assert np.abs(got - ref).max() <= 3e-2                            # the tolerance of tests/test_gpu_unet.py


def test_save_model_as_keras_weights_and_back(tmp_path, monkeypatch):
    from inconsistencymasks_amd import functions as F
    from inconsistencymasks_amd import h5lite as H
    from inconsistencymasks_amd.unet import UNet
    m = UNet(64, 96, 3, 1, 0.5, "sigmoid", seed=11)
    x = torch.randint(0, 256, (8, 64, 96, 3), dtype=torch.uint8, device="cuda")
    y = (torch.rand((8, 64, 96, 1), device="cuda") > 0.6).to(torch.uint8)
    m.init_train_state()
    for _ in range(3):                                                # moving statistics and weights away from their initial values
        m.train_step(x, y, 0, 3e-3, 1e-4)
    p_st, p_h5 = str(tmp_path / "a.h5"), str(tmp_path / "b.h5")
    F.save_model(m, p_st)
    monkeypatch.setenv("IMK_MODEL_FORMAT", "keras_h5")
    F.save_model(m, p_h5)
    assert not H.is_hdf5(p_st) and H.is_hdf5(p_h5)
    f = H.File(p_h5)
    assert H.load_attr_list(f, "layer_names")[:3] == ["conv2d", "batch_normalization", "conv2d_1"] and f.attrs["backend"] == b"tensorflow"
    a, b = F.load_model(p_st), F.load_model(p_h5)
    assert (b.plan.h, b.plan.w, b.plan.act_out) == (64, 96, "sigmoid")
    assert torch.equal(a.params, b.params) and torch.equal(a.params, m.params)
    assert torch.equal(a.predict_device(x), b.predict_device(x))


def test_isic_generation_with_hdf5_model_files(tmp_path):
    """IMK_MODEL_FORMAT=keras_h5: the subset baseline and one IM generation write, rename and reload their models as real HDF5
    files (ModelCheckpoint -> load_model -> top-K rename, ISIC_2018/09_ISIC_2018_IM.py:74-76, 131-135); the results agree with the
    default format's to the last digit (same seeds, and the weights survive either container bit for bit)."""
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_driver import CONFIG, SETUP
    from inconsistencymasks_amd import h5lite as H
    rows = {}
    for fmt in ("safetensors", "keras_h5"):
        work = tmp_path / fmt
        work.mkdir()
        base = work / "data"
        cfg = work / "config.ini"
        cfg.write_text(CONFIG.format(base=base))
        env = {**os.environ, "IM_CONFIG": str(cfg), "IM_RUNIDS": "1", "IM_NS": "2", "IM_GENS": "0", "IM_CANDIDATES": "0,1",
               "IMK_MODEL_FORMAT": fmt}
        setup = SETUP.format(root=ROOT).split("import torch\nx = torch.from_numpy")[0]      # the data only
        subprocess.run([sys.executable, "-c", setup], env=env, check=True, cwd=work)
        for script in ("03_ISIC_2018_subset.py", "09_ISIC_2018_IM.py"):
            r = subprocess.run([sys.executable, os.path.join(ROOT, "ISIC_2018", script)], env=env, cwd=work, capture_output=True, text=True)
            assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        models = sorted(os.listdir(base / "models"))
        assert "ISIC_2018_subset_1_topK_1.h5" in models and any("_IM_1_n2_gen0_" in m and "topK_1" in m for m in models)
        assert all(H.is_hdf5(str(base / "models" / m)) == (fmt == "keras_h5") for m in models)
        stem = "ISIC_2018_IM_1_n2_gen0_e0_d0_bi_True_bo_True"
        rows[fmt] = (base / "csv" / f"results_{stem}.csv").read_text()
    assert rows["safetensors"] == rows["keras_h5"]


def test_load_evalnet_takes_a_keras_checkpoint_and_round_trips(tmp_path, monkeypatch):
    """evalnet.get_evalnet_miou's Keras checkpoint (tests/golden/h5_keras_evalnet.h5: h5py's bytes, two towers) through load_evalnet and
    the compat tf.keras.models.load_model; save_evalnet under IMK_MODEL_FORMAT=keras_h5 and back, bit for bit"""
    from inconsistencymasks_amd import evalnet_functions as EF
    from inconsistencymasks_amd import h5lite as H
    from inconsistencymasks_amd import keras_h5 as K
    from oracle import evalnet_oracle as E
    path = os.path.join(GOLD, "h5_keras_evalnet.h5")
    m = EF.load_evalnet(path)
    p = m.plan
    assert (p.h, p.w, p.ca, p.cb, p.n_out, p.alpha, p.two_heads, p.cfg.normalize_a, p.cfg.normalize_b) == (64, 64, 3, 2, 2, 0.5, True, 1, 0)
    sd, _ = K.evalnet_state_dict_from_keras_h5(path)
    rng = np.random.default_rng(3)
    xa = rng.integers(0, 256, (4, 64, 64, 3), dtype=np.uint8)
    xb = (rng.random((4, 64, 64, 2)) > 0.5).astype(np.uint8)
    got = np.concatenate(m.predict([xa, xb]), 1)
    ref, _ = E.forward({k: torch.from_numpy(v) for k, v in sd.items()}, xa, xb, True, True, False, emulate_fp16=True)
    assert got.shape == (4, 4) and np.abs(got - ref.numpy()).max() <= 2e-2          # tests/test_gpu_evalnet.py's tolerance
    sys.path.insert(0, os.path.join(ROOT, "inconsistencymasks_amd", "compat"))
    import tensorflow as tf                                                          # the compat namespace
    m2 = tf.keras.models.load_model(path)
    assert type(m2).__name__ == "EvalNet" and torch.equal(m2.params, m.params)
    u = tf.keras.models.load_model(os.path.join(GOLD, "h5_keras_full_model.h5"))
    assert type(u).__name__ == "UNet"
    monkeypatch.setenv("IMK_MODEL_FORMAT", "keras_h5")
    q = str(tmp_path / "evalnet.h5")
    EF.save_evalnet(m, q)
    assert H.is_hdf5(q)
    names = H.load_attr_list(H.File(q), "layer_names")
    assert names[:4] == ["conv2d", "conv2d_3", "batch_normalization", "batch_normalization_2"] and names[-2:] == ["iou", "detection"]
    back = EF.load_evalnet(q)
    assert torch.equal(back.params, m.params) and (back.plan.cfg.normalize_a, back.plan.cfg.normalize_b) == (1, 0)
    assert np.array_equal(np.concatenate(back.predict([xa, xb]), 1), got)
