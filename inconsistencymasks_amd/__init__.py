"""MI355X-native Inconsistency-Mask hot path: host-side mirror of the reference's Python API
(unet.get_unet, functions.get_im_prediction_* / create_pseudo_labels_im_* / train_*) on top of
libimk.so (hand-written HIP for gfx950, C ABI in include/imk.h)."""
__version__ = "0.1.0"
