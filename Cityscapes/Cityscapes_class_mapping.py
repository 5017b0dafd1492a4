"""`from Cityscapes_class_mapping import COLOR_TO_CLASS_MAPPING_CITYSCAPES` (the reference's Cityscapes scripts, e.g.
Cityscapes/09_Cityscapes_IM.py:5): the colour -> class id table of the `*_color.png` dumps, generated from the rule the public palette
follows (inconsistencymasks_amd/im_driver.color_mapping: PASCAL-VOC bit-interleaved colours, blue carrying bit 0, class 0 = IM, plus
the licence-plate colour -> -1).  Cityscapes/Cityscapes_class_mapping.py:4-80."""
import os
import sys

sys.path.append(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from inconsistencymasks_amd.im_driver import color_mapping  # noqa: E402

COLOR_TO_CLASS_MAPPING_CITYSCAPES = color_mapping("Cityscapes", 35)
# the dataset's own 34 label ids (no IM class in front): the same colours one id lower (the last colour of the 35-entry table drops out)
_BY_ID = {k: c for c, k in COLOR_TO_CLASS_MAPPING_CITYSCAPES.items()}
COLOR_TO_CLASS_MAPPING_CITYSCAPES_ORG = {**{_BY_ID[k]: k for k in range(34)}, _BY_ID[-1]: -1}
CLASS_DESCRIPTION = {**dict(enumerate(("IM", "Unlabeled", "Ego vehicle", "Rectification border", "Out of roi", "Static", "Dynamic", "Ground",
                                       "Road", "Sidewalk", "Parking", "Rail track", "Building", "Wall", "Fence", "Guard rail", "Bridge",
                                       "Tunnel", "Pole", "Polegroup", "Traffic light", "Traffic sign", "Vegetation", "Terrain", "Sky",
                                       "Person", "Rider", "Car", "Truck", "Bus", "Caravan", "Trailer", "Train", "Motorcycle", "Bicycle"))),
                     -1: "License plate"}
