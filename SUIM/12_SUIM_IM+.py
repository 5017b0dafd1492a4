"""SUIM IM+ generations (IM pseudo-labels + Noisy-Student augmentation + growing U-Net width) on MI355X: counterpart of
the reference driver SUIM/12_SUIM_IM+.py (same loops, schedules, file / model / CSV names); the loop body lives in
inconsistencymasks_amd/im_driver.py."""
import os
import sys

sys.path.append(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from inconsistencymasks_amd.im_driver import run  # noqa: E402

if __name__ == "__main__":
    run("SUIM", approach="IM_plus")
