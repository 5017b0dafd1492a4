"""ISIC_2018 AIM+ generations (IM+ on top of the augmented-subset baseline: augmented labelled set, un-augmented IM pairs kept) on
MI355X: counterpart of the reference driver ISIC_2018/13_ISIC_2018_aug_IM+.py (same loops, schedules, file / model / CSV names); the loop body lives in
inconsistencymasks_amd/im_driver.py."""
import os
import sys

sys.path.append(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from inconsistencymasks_amd.im_driver import run  # noqa: E402

if __name__ == "__main__":
    run("ISIC_2018", approach="aug_IM_plus")
