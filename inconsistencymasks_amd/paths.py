"""Directory constants with the reference's names (paths.py:10-201), generated from BASE_DIR in config.ini by the rule
the reference's table follows: `{DATASET}_{SPLIT}[_MAIN]_DIR = BASE_DIR/{split}` and below it `images` / `masks`
(HeLa: `brightfield` / `alive` / `dead` / `pos` / `mod_position`), plus `models` and `csv`.  Every processed-dataset
constant of the reference exists here with the same value; the `*_ORG_*` constants of the dataset-preparation scripts
(out of scope) are not defined."""
import configparser
import os

config = configparser.ConfigParser()
config.read(os.environ.get("IM_CONFIG", "config.ini"))

_SPLITS = ("TRAIN_FULL", "TRAIN_LABELED", "TRAIN_LABELED_AUG", "TRAIN_UNLABELED", "VAL", "TEST")
_HELA_SUBS = (("BRIGHTFIELD", "brightfield"), ("ALIVE", "alive"), ("DEAD", "dead"), ("POS", "pos"),
              ("MOD_POS", "mod_position"))


def _dataset(prefix, section):
    if section not in config:
        return
    base = config[section]["BASE_DIR"]
    g = globals()
    g[f"{prefix}_BASE_DIR"] = base
    for split in _SPLITS:
        d = os.path.join(base, split.lower())
        g[f"{prefix}_{split}_DIR"] = d
        g[f"{prefix}_{split}_MAIN_DIR"] = d
        if prefix == "HELA":
            for name, sub in _HELA_SUBS:
                g[f"{prefix}_{split}_{name}_DIR"] = os.path.join(d, sub)
        else:
            g[f"{prefix}_{split}_IMAGES_DIR"] = os.path.join(d, "images")
            g[f"{prefix}_{split}_MASKS_DIR"] = os.path.join(d, "masks")
    g[f"{prefix}_MODEL_DIR"] = os.path.join(base, "models")
    g[f"{prefix}_CSV_DIR"] = os.path.join(base, "csv")


_dataset("ISIC_2018", "ISIC_2018")
_dataset("SUIM", "SUIM")
_dataset("CITYSCAPES", "CITYSCAPES")
_dataset("HELA", "HELA")
