"""`from unet import get_unet` (ISIC_2018/09_ISIC_2018_IM.py:6) resolves to the MI355X implementation."""
from inconsistencymasks_amd.unet import UNet, get_unet  # noqa: F401
