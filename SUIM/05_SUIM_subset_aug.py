"""SUIM augmented labelled-subset baseline (ALDT) on MI355X -- the generation-0 ensemble of the AIM+ driver: counterpart of the
reference driver SUIM/05_SUIM_subset_aug.py (same loops and file / model / CSV names); the loop body lives in
inconsistencymasks_amd/subset_driver.py."""
import os
import sys

sys.path.append(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from inconsistencymasks_amd.subset_driver import run  # noqa: E402

if __name__ == "__main__":
    run("SUIM", aug=True)
