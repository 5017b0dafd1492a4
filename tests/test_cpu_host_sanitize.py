"""libimk's host C++ (csrc/imk_png.cpp, csrc/imk_geom.cpp) under AddressSanitizer + UBSan on the CPU (GPU sanitizers are not available on
the pool): tools/host_sanitize.py builds the two files with g++ -fsanitize=address,undefined and drives the PNG decoder with valid and
mutated files, the encoder at its extremes and the HeLa geometry on degenerate masks.  Skipped where g++ or its sanitizer runtime is
missing.  (Reference paths replaced: cv2.imread / cv2.imwrite, functions.py:2846, 2885-2887; get_pos_contours / mod_pos_size /
get_cell_count, functions.py:6181-6371.)"""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _have_asan():
    if not shutil.which("g++"):
        return False
    p = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True)
    return p.returncode == 0 and os.path.isabs(p.stdout.strip()) and os.path.exists(p.stdout.strip())


@pytest.mark.skipif(not _have_asan(), reason="g++ with libasan is not installed")
def test_host_code_is_clean_under_asan_and_ubsan():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "host_sanitize.py"), "4000"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    assert "host sanitizer run clean" in p.stdout
