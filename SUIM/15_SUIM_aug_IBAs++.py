"""SUIM AIM++ generations (EvalNet-weighted augmentation of the IM pseudo-labels; the EvalNet sees the label map as a
one-hot stack) on MI355X: counterpart of the reference driver SUIM/15_SUIM_aug_IBAs++.py (same loops, schedules, file /
model / CSV names); the loop body lives in inconsistencymasks_amd/impp_driver.py."""
import os
import sys

sys.path.append(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from inconsistencymasks_amd.impp_driver import run  # noqa: E402

if __name__ == "__main__":
    run("SUIM", aug=True)
