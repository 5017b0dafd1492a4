"""Compatibility namespace for running the reference's own per-dataset scripts unchanged.

The scripts touch TensorFlow through exactly five names (grep over ISIC_2018/03,09-14, HeLa/03,09-14, SUIM/04,10-15,
Cityscapes/03,09-14): `tf.device`, `tf.keras.models.load_model`, `tf.keras.backend.clear_session`,
`tf.keras.losses.CategoricalCrossentropy` and `tensorflow.keras.mixed_precision.set_global_policy`.  Putting this
directory on PYTHONPATH (`PYTHONPATH=<repo>:<repo>/inconsistencymasks_amd/compat`) provides a `tensorflow` module with
those five names mapped onto libimk.so-backed objects, next to the top-level `functions` / `unet` / `evalnet` / `paths`
modules of this repository -- SURVEY section 8b's "thin compatibility namespace".  Nothing else of TensorFlow exists
here, and nothing in the product imports it."""
