"""The drop-in's REAL import surface (missing 2): for every reference script that has a counterpart under ISIC_2018/
HeLa/ SUIM/ Cityscapes/, every name the reference script takes from `functions`, `unet`, `evalnet`, `paths`, the per-dataset class
mappings and the TensorFlow namespace must resolve against the repo-root shims + inconsistencymasks_amd/compat.  Build container only
(the reference is absent on the GPU box); the test reads NAMES out of the reference scripts' syntax trees and copies nothing.
Reference: functions.py:105-160 (`ignore_im_*`), Cityscapes/13_Cityscapes_aug_IM+.py:6, HeLa/12_HeLa_IM++.py:5, HeLa/14_HeLa_aug_IM++.py:5."""
import ast
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
DATASETS = ("ISIC_2018", "HeLa", "SUIM", "Cityscapes")
SHIMMED = {"functions", "unet", "evalnet", "paths", "SUIM_class_mapping", "Cityscapes_class_mapping"}
TF_ROOTS = {"tf", "tensorflow", "mixed_precision", "keras", "K"}


def _dotted(node):
    parts = []
    while isinstance(node, ast.Attribute):
        parts.append(node.attr)
        node = node.value
    if isinstance(node, ast.Name):
        parts.append(node.id)
        return ".".join(reversed(parts))
    return None


def surface(path):
    """-> (from-imports {module: {names}}, attribute chains rooted at an imported module alias {alias: {dotted}}, alias -> module)"""
    tree = ast.parse(open(path, encoding="utf-8", errors="replace").read())
    froms, alias = {}, {}
    for n in ast.walk(tree):
        if isinstance(n, ast.ImportFrom) and n.module:
            froms.setdefault(n.module, set()).update(a.name for a in n.names)
            for a in n.names:
                alias[a.asname or a.name] = f"{n.module}.{a.name}"
        elif isinstance(n, ast.Import):
            for a in n.names:
                alias[a.asname or a.name.split(".")[0]] = a.name if a.asname else a.name.split(".")[0]
    chains = {}
    for n in ast.walk(tree):
        if isinstance(n, ast.Attribute):
            d = _dotted(n)
            if d and d.split(".")[0] in alias:
                chains.setdefault(d.split(".")[0], set()).add(d)
    return froms, chains, alias


def pairs():
    out = []
    for ds in DATASETS:
        for f in sorted(os.listdir(os.path.join(ROOT, ds))):
            if f.endswith(".py") and os.path.exists(os.path.join(REF, ds, f)):
                out.append((ds, f))
    return out


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference is only present in the build container")
def test_every_name_the_in_scope_reference_scripts_import_resolves():
    ps = pairs()
    assert len(ps) >= 29, ps
    wanted = {}     # "module:name" or "chain:<dotted, module-qualified>" -> scripts that need it
    for ds, f in ps:
        froms, chains, alias = surface(os.path.join(REF, ds, f))
        for mod, names in froms.items():
            top = mod.split(".")[0]
            if top in SHIMMED or top in ("tensorflow", "tensorflow_addons"):
                for n in names:
                    wanted.setdefault(f"from:{mod}:{n}", []).append(f"{ds}/{f}")
        for a, ds_ in chains.items():
            target = alias[a].split(".")[0]
            if target in SHIMMED or target in ("tensorflow", "tensorflow_addons"):
                # keep the longest chains only (tf.keras.models.load_model implies tf.keras.models)
                longest = {d for d in ds_ if not any(o != d and o.startswith(d + ".") for o in ds_)}
                for d in longest:
                    wanted.setdefault(f"chain:{alias[a]}:{d.split('.', 1)[1]}", []).append(f"{ds}/{f}")
    assert any(k.endswith(":ignore_im_categorical_crossentropy") for k in wanted)      # the gap round 5's hand-written script could not see
    assert any(k.startswith("chain:paths:") for k in wanted) and any("load_model" in k for k in wanted)
    probe = r"""
import importlib, json, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
for ds in %r:
    sys.path.insert(0, %r + "/" + ds)          # the class-mapping modules sit beside the scripts
wanted = json.load(sys.stdin)
missing = []
for key in wanted:
    kind, mod, name = key.split(":", 2)
    try:
        parts = mod.split(".")
        m = importlib.import_module(parts[0])
        for part in parts[1:]:                  # a dotted "module" may be an attribute of its parent (tf.keras.mixed_precision)
            m = getattr(m, part) if hasattr(m, part) else importlib.import_module(m.__name__ + "." + part)
    except Exception as e:
        missing.append(f"{key} ({type(e).__name__}: {e})"); continue
    obj = m
    try:
        for part in name.split("."):
            if part == "*":
                break
            try:
                obj = getattr(obj, part)
            except AttributeError:
                obj = importlib.import_module(obj.__name__ + "." + part)      # `from tensorflow.keras import mixed_precision`
    except Exception as e:
        missing.append(f"{key} ({type(e).__name__})")
print(json.dumps(missing))
""" % (ROOT, os.path.join(ROOT, "inconsistencymasks_amd", "compat"), list(DATASETS), ROOT)
    # attribute chains go as deep as a METHOD of a returned object (model.predict, os.path.join ...): only module-rooted chains were
    # collected, and a chain's tail past a call is not in it (ast.Attribute over a Call has no Name root)
    r = subprocess.run([sys.executable, "-c", probe], input=json.dumps(sorted(wanted)), capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    missing = json.loads(r.stdout.strip().splitlines()[-1])
    assert not missing, "\n".join(f"{m}  <- {wanted.get(m.split(' (')[0], '?')}" for m in missing)


def test_ignore_im_losses_are_importable_and_refused_by_the_trainers():
    sys.path.insert(0, ROOT)
    import numpy as np
    from inconsistencymasks_amd import functions as F
    cce, dice = F.ignore_im_categorical_crossentropy(), F.ignore_im_dice_loss_multiclass()
    # known answers on a shape where the reference's `y_true[:, 0]` broadcast is defined: [B, H, K] with H == 1
    y = np.zeros((2, 1, 3), np.float32); y[0, 0, 1] = 1; y[1, 0, 0] = 1
    p = np.full((2, 1, 3), 1 / 3, np.float32)
    # cce per pixel = ln 3; mask = 1 - y[:, 0] = [[1,0,1],[0,1,1]] broadcast against loss [2,1] -> mean over [2,2,3]... numpy semantics
    expect = (np.log(3.0) * (1 - y[:, 0])).mean()
    assert abs(float(cce(y, p)) - expect) < 1e-6
    yt, yp = y[:, :, 1:], p[:, :, 1:]
    d = (2 * (yt * yp).sum((1, 2)) + 1e-7) / (yt.sum((1, 2)) + yp.sum((1, 2)) + 1e-7)
    assert abs(float(dice(y, p)) - (1 - d).mean()) < 1e-6
    for loss in (cce, dice, F.ignore_im_dice_loss_multiclass):
        with pytest.raises(NotImplementedError):
            F.train_multiclass(*([None] * 10), loss, *([None] * 9))


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference is only present in the build container")
def test_generated_colour_tables_equal_the_reference_modules():
    """SUIM/SUIM_class_mapping.py and Cityscapes/Cityscapes_class_mapping.py here GENERATE their tables from the public palettes' rules;
    every name of the reference modules must come out equal."""
    import importlib.util

    def load(path, name):
        spec = importlib.util.spec_from_file_location(name, path)
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m
    for ds, mod in (("SUIM", "SUIM_class_mapping"), ("Cityscapes", "Cityscapes_class_mapping")):
        mine, ref = load(os.path.join(ROOT, ds, mod + ".py"), "mine_" + mod), load(os.path.join(REF, ds, mod + ".py"), "ref_" + mod)
        names = [n for n in vars(ref) if n.isupper()]
        assert len(names) >= 3
        for n in names:
            assert getattr(mine, n) == getattr(ref, n), (mod, n)
