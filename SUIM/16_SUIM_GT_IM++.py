"""SUIM IM++ with a "perfect EvalNet" (the augmentation count follows the pseudo-label's IoU against the ground truth) on MI355X:
counterpart of the reference driver SUIM/16_SUIM_GT_IM++.py (same loops, schedules, file / model / CSV names); the loop body
lives in inconsistencymasks_amd/impp_driver.py."""
import os
import sys

sys.path.append(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from inconsistencymasks_amd.impp_driver import run  # noqa: E402

if __name__ == "__main__":
    run("SUIM", gt=True)
