"""`from evalnet import get_evalnet, get_evalnet_miou` (ISIC_2018/10_ISIC_2018_evalnet.py:6) resolves to the MI355X
implementation."""
from inconsistencymasks_amd.evalnet import EvalNet, get_evalnet, get_evalnet_miou  # noqa: F401
