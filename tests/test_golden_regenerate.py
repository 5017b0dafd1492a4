"""The pin of the integer IM chain, re-derived: tests/golden/make_golden.py imports the REAL reference (/root/reference/functions.py, its
cv2 / tensorflow imports stubbed, every exercised function pure numpy) and writes the fixtures the parity tests compare against.  Where
the reference is present -- the build container; never the GPU box -- this test runs the generator into a scratch directory and requires
every array of every fixture to equal the committed one (NaN-aware: the probability stacks hold NaN / inf / 0.5 +- ulp cases on purpose).
A committed fixture that the reference no longer reproduces, or a generator edited without regenerating, fails here.
SURVEY 8c; functions.py:3104-3238, 2832-2891, 2988-3070, 4328-4459, 1463-1478."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
REFERENCE = "/root/reference/functions.py"
FIXTURES = ("im_binary", "im_hela", "im_multiclass", "writer_isic", "writer_multi", "metrics", "augment", "evalnet_labels")


@pytest.mark.skipif(not os.path.exists(REFERENCE), reason="the reference is only present in the build container")
def test_fixtures_regenerate_identically_from_the_reference(tmp_path):
    env = {**os.environ, "IMK_GOLDEN_OUT": str(tmp_path)}
    p = subprocess.run([sys.executable, os.path.join(GOLD, "make_golden.py")], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    made = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(str(tmp_path), "*.npz")))
    assert made == sorted(FIXTURES), made
    for name in FIXTURES:
        new, old = np.load(os.path.join(str(tmp_path), name + ".npz")), np.load(os.path.join(GOLD, name + ".npz"))
        assert set(new.files) == set(old.files), name
        for k in old.files:
            a, b = old[k], new[k]
            assert a.dtype == b.dtype and a.shape == b.shape, (name, k)
            assert np.array_equal(a, b, equal_nan=True) if a.dtype.kind in "fc" else np.array_equal(a, b), (name, k)
