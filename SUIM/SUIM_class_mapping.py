"""`from SUIM_class_mapping import COLOR_TO_CLASS_MAPPING_SUIM` (the reference's SUIM scripts, e.g. SUIM/10_SUIM_IM.py:5): the colour ->
class id table of the `*_color.png` dumps, generated from the public SUIM palette's rule (inconsistencymasks_amd/im_driver.color_mapping:
3-bit RGB codes shifted by one behind the light-grey IM class 0).  SUIM/SUIM_class_mapping.py:4-14, 29-39."""
import os
import sys

sys.path.append(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from inconsistencymasks_amd.im_driver import color_mapping  # noqa: E402

COLOR_TO_CLASS_MAPPING_SUIM = color_mapping("SUIM", 9)
COLOR_TO_CLASS_MAPPING_SUIM_ORG = {c: k - 1 for c, k in COLOR_TO_CLASS_MAPPING_SUIM.items() if k > 0}      # the dataset's own 8 classes
CLASS_DESCRIPTION = dict(enumerate(("IM", "Background (waterbody)", "Human divers", "Aquatic plants and sea-grass", "Wrecks and ruins",
                                    "Robots (AUVs/ROVs/instruments)", "Reefs and invertebrates", "Fish and vertebrates",
                                    "Sea-floor and rocks")))
