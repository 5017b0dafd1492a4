"""tf.keras.{models.load_model, backend.clear_session, losses.CategoricalCrossentropy, mixed_precision.set_global_policy}"""
import types


def _load_model(filepath, custom_objects=None, compile=True):
    """tf.keras.models.load_model(path, custom_objects=...) (ISIC_2018/09_ISIC_2018_IM.py:75; EvalNet files at
    ISIC_2018/12_ISIC_2018_IM++.py): a Keras HDF5 checkpoint of get_unet goes to the HDF5 reader; otherwise the safetensors
    metadata says which network the file holds."""
    from inconsistencymasks_amd import h5lite
    from inconsistencymasks_amd.functions import load_model
    if h5lite.is_hdf5(filepath):
        from inconsistencymasks_amd.keras_h5 import keras_h5_kind
        if keras_h5_kind(filepath) == "evalnet":
            from inconsistencymasks_amd.evalnet_functions import load_evalnet
            return load_evalnet(filepath)
        return load_model(filepath, custom_objects=custom_objects)
    from safetensors import safe_open
    with safe_open(filepath, framework="pt") as f:
        meta = f.metadata() or {}
    if meta.get("net") == "evalnet":
        from inconsistencymasks_amd.evalnet_functions import load_evalnet
        return load_evalnet(filepath)
    return load_model(filepath, custom_objects=custom_objects)


def _clear_session():
    """tf.keras.backend.clear_session() (ISIC_2018/09_ISIC_2018_IM.py:120): drop cached device buffers."""
    import gc
    import torch
    gc.collect()
    if torch.cuda.is_available():
        torch.cuda.empty_cache()


class CategoricalCrossentropy:
    """tf.keras.losses.CategoricalCrossentropy() as passed to train_multiclass (SUIM/10_SUIM_IM.py:112): a marker; the
    loss itself is fused into the head kernel (loss_kind 1)."""
    name = "categorical_crossentropy"


def _set_global_policy(name):
    """mixed_precision.set_global_policy('mixed_float16') (ISIC_2018/09_ISIC_2018_IM.py:16): that policy is the only one
    the kernels implement (fp16 tensors, fp32 variables, fp32 head)."""
    if name != "mixed_float16":
        raise NotImplementedError(f"policy {name!r}: the MI355X kernels implement mixed_float16 only")


models = types.SimpleNamespace(load_model=_load_model)
backend = types.SimpleNamespace(clear_session=_clear_session)
losses = types.SimpleNamespace(CategoricalCrossentropy=CategoricalCrossentropy)
mixed_precision = types.SimpleNamespace(set_global_policy=_set_global_policy)
