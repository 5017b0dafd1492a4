"""Directory constants of the hot path, same names as the reference's paths.py (paths.py:10-45 for ISIC),
derived from BASE_DIR in config.ini.  Only what the IM drivers touch is defined."""
import configparser
import os

config = configparser.ConfigParser()
config.read(os.environ.get("IM_CONFIG", "config.ini"))


def _dataset(prefix, section):
    if section not in config:
        return
    base = config[section]["BASE_DIR"]
    g = globals()
    g[f"{prefix}_BASE_DIR"] = base
    for split in ("TRAIN_LABELED", "VAL", "TEST", "TRAIN_UNLABELED", "TRAIN_FULL"):
        d = os.path.join(base, split.lower())
        g[f"{prefix}_{split}_IMAGES_DIR"] = os.path.join(d, "images")
        g[f"{prefix}_{split}_MASKS_DIR"] = os.path.join(d, "masks")
        g[f"{prefix}_{split}_DIR"] = d
    g[f"{prefix}_MODEL_DIR"] = os.path.join(base, "models")
    g[f"{prefix}_CSV_DIR"] = os.path.join(base, "csv")


_dataset("ISIC_2018", "ISIC_2018")
_dataset("SUIM", "SUIM")
_dataset("CITYSCAPES", "CITYSCAPES")
_dataset("HELA", "HELA")
