"""MI355X-native Inconsistency-Mask hot path: host-side mirror of the reference's Python API
(unet.get_unet, functions.get_im_prediction_* / create_pseudo_labels_im_* / train_*) on top of
libimk.so (hand-written HIP for gfx950, C ABI in include/imk.h)."""
import os as _os

# The HIP runtime multiplexes all streams of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4).  This package uses the
# caller's stream + 1-2 side streams, torch.distributed's RCCL backend adds its own: on 4 queues two of them share one and
# "concurrent" kernels run in line (measured: -14 % on a training epoch, profiles/README.md round 3).  Effective only if the
# package is imported before the process makes its first HIP call (importing torch alone does not make one).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

__version__ = "0.1.0"
