"""GPU: benchmark_hela (functions.py:1155-1260 of the reference) against a restatement of the reference's loop on the host: thresholds,
get_IoU_binary on all three maps, mod_pos_size, get_pos_contours + get_cell_count on prediction and ground truth -- from the model's
probabilities, with the numpy / scipy geometry of oracle/hela_geometry.py -- and the PNG files it leaves behind."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dataset(root, n, seed):
    from inconsistencymasks_amd import functions as F
    yy, xx = np.mgrid[0:64, 0:64]
    for k in ("brightfield", "alive", "dead", "mod_position"):
        os.makedirs(os.path.join(root, k), exist_ok=True)
    for i in range(n):
        r = np.random.default_rng(seed * 1009 + i)
        bf = np.full((64, 64), 120, np.int64) + r.integers(-8, 8, (64, 64))
        alive, dead, pos = (np.zeros((64, 64), np.uint8) for _ in range(3))
        for _ in range(int(r.integers(0, 6))):
            cy, cx, rad, is_dead = int(r.integers(6, 58)), int(r.integers(6, 58)), int(r.integers(4, 9)), int(r.integers(0, 2))
            cell = (yy - cy) ** 2 + (xx - cx) ** 2 < rad * rad
            bf[cell] += 70 if is_dead else -60
            (dead if is_dead else alive)[cell] = 255
            pos[(yy - cy) ** 2 + (xx - cx) ** 2 < 10] = 255
        name = f"c_{i:03d}.png"
        F.write_png(os.path.join(root, "brightfield", name), bf.clip(0, 255).astype(np.uint8))
        F.write_png(os.path.join(root, "alive", name), alive)
        F.write_png(os.path.join(root, "dead", name), dead)
        F.write_png(os.path.join(root, "mod_position", name), pos)


@pytest.mark.parametrize("mod_position", [True, False])
def test_benchmark_hela_equals_the_reference_loop_on_the_host(tmp_path, mod_position):
    from inconsistencymasks_amd import functions as F
    from inconsistencymasks_amd.unet import get_unet
    from oracle import hela_geometry as G
    gt, pred = str(tmp_path / "gt"), str(tmp_path / "pred")
    _dataset(gt, 70, 3)                                      # 70: a full batch of 64 and a short one
    names = sorted(os.listdir(os.path.join(gt, "brightfield")))
    rd = lambda k, n: F.read_png(os.path.join(gt, k, n), 1)[..., 0]
    model = get_unet(64, 64, 1, 3, 1.0, "relu", "sigmoid", seed=4)
    x = torch.from_numpy(np.stack([F.read_png(os.path.join(gt, "brightfield", n), 1) for n in names])).cuda()
    y = torch.from_numpy(np.stack([np.stack([rd("alive", n) // 255, rd("dead", n) // 255, rd("mod_position", n) // 255 * 3], -1)
                                   for n in names])).cuda()
    for _ in range(60):                                      # enough training for non-trivial maps (some cells found, some not)
        model.train_step(x[:32].contiguous(), y[:32].contiguous(), 0, 3e-3, 1e-4)
    model.repack()
    got = F.benchmark_hela(model, gt, pred, 64, 64, 1, mod_position=mod_position)
    F.flush_writes()
    again = F.benchmark_hela(model, gt, pred, 64, 64, 1, mod_position=mod_position)      # second call: cached decoded set + GT counts
    F.flush_writes()
    assert again == got

    probs = model.predict_device(x).cpu().numpy()
    mious, mious_ad, delta = [], [], 0
    sub = "mod_position" if mod_position else "position"
    found = 0
    for j, n in enumerate(names):
        a_u, d_u, p_u = [((probs[j, ..., k] > 0.5) * 255).astype(np.uint8) for k in range(3)]
        if mod_position:
            p_u = G.mod_pos_size(p_u)
        ga, gd, gp = rd("alive", n), rd("dead", n), rd("mod_position", n)
        ia, idd, ip = (round(float(F.get_IoU_binary(g, p)), 4) for g, p in ((ga, a_u), (gd, d_u), (gp, p_u)))
        mious.append((ia + idd + ip) / 3); mious_ad.append((ia + idd) / 2)
        pp = G.get_pos_contours(p_u)
        found += len(pp)
        pa, pd, _ = G.get_cell_count(pp, a_u, d_u)
        qa, qd, _ = G.get_cell_count(G.get_pos_contours(gp), ga, gd)
        delta += abs(pa - qa) + abs(pd - qd)
        for k, arr in (("alive", a_u), ("dead", d_u), (sub, p_u)):
            assert np.array_equal(F.read_png(os.path.join(pred, k, n), 1)[..., 0], arr), (k, n)
    want = (round(float(np.sum(mious) / len(mious)), 3), round(float(np.sum(mious_ad) / len(mious_ad)), 3),
            round(delta / len(mious), 3))
    assert got == want
    assert found > 0 and 0 < want[0] < 1                    # the comparison had something to compare
