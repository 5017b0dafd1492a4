"""The five TensorFlow names the reference's hot-path scripts use (see inconsistencymasks_amd/compat/__init__.py)."""
import contextlib

from . import keras  # noqa: F401

__version__ = "0.0-imk-compat"


@contextlib.contextmanager
def device(name):
    """`with tf.device('/gpu:0'):` (ISIC_2018/09_ISIC_2018_IM.py:47) -> torch.cuda.device(index)."""
    import torch
    idx = int(str(name).rsplit(":", 1)[-1]) if ":" in str(name) else 0
    if "gpu" in str(name).lower() and torch.cuda.is_available():
        with torch.cuda.device(idx):
            yield
    else:
        yield
