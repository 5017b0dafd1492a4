"""`import paths` (ISIC_2018/09_ISIC_2018_IM.py:13): the reference's directory constants, derived from config.ini."""
from inconsistencymasks_amd.paths import *  # noqa: F401,F403
from inconsistencymasks_amd import paths as _p

globals().update({k: v for k, v in vars(_p).items() if k.isupper()})
