"""HeLa labelled-subset baseline on MI355X: counterpart of the reference driver HeLa/03_HeLa_subset.py (same loops, model / CSV
names, top-K rename); it produces the `*_subset_{runid}_topK_{j}.h5` ensemble that generation 0 of the IM drivers loads.
The loop body lives in inconsistencymasks_amd/subset_driver.py."""
import os
import sys

sys.path.append(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from inconsistencymasks_amd.subset_driver import run  # noqa: E402

if __name__ == "__main__":
    run("HeLa")
