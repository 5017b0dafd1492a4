"""`from functions import train_ISIC_2018, create_pseudo_labels_im_ISIC_2018, dice_loss` -- the import line of the
reference's per-dataset scripts (ISIC_2018/09_ISIC_2018_IM.py:5) resolves to the MI355X implementation."""
from inconsistencymasks_amd.functions import *  # noqa: F401,F403
from inconsistencymasks_amd.functions import (BATCH_SIZE, LR, NUM_EPOCHS, NUM_EPOCHS_CS, SEED, THRESHOLD, WD,  # noqa: F401
                                              config)
