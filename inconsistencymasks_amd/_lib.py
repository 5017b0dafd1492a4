"""ctypes binding of libimk.so (include/imk.h).  There is no fallback: if the library is missing or an
entry point is absent, importing this module raises."""
import ctypes
import os

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IMK_LIB_PATH", os.path.join(PKG, "libimk.so"))   # override: A/B builds of the library

c_int, c_float, c_void_p, c_int64 = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_int64


class UnetCfg(ctypes.Structure):
    _fields_ = [("h", c_int), ("w", c_int), ("c_in", c_int), ("n_out", c_int),
                ("ch", c_int * 5), ("act_out", c_int)]


class EvalnetCfg(ctypes.Structure):
    _fields_ = [("h", c_int), ("w", c_int), ("ca", c_int), ("cb", c_int), ("n_out", c_int), ("two_heads", c_int),
                ("normalize_a", c_int), ("normalize_b", c_int), ("ch", c_int * 5), ("b_onehot", c_int)]


class AugParams(ctypes.Structure):
    _fields_ = [("flip_v", c_int), ("flip_h", c_int), ("rot", c_int), ("bright_on", c_int), ("alpha", c_float),
                ("beta", c_float), ("blur_k", c_int), ("noise_max", c_int), ("seed", ctypes.c_uint32)]


class LayerInfo(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 16), ("kind", c_int), ("ksize", c_int), ("cin", c_int), ("cout", c_int),
                ("off_w", c_int64), ("off_b", c_int64), ("off_mean", c_int64), ("off_var", c_int64)]


# name -> (restype, argtypes); must list every symbol include/imk.h declares
SIGNATURES = {
    "imk_version": (c_int, []),
    "imk_error_string": (ctypes.c_char_p, [c_int]),
    "imk_im_binary": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_int,
                              c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                              c_void_p, c_void_p, c_void_p]),
    "imk_im_multiclass": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int,
                                  c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p]),
    "imk_morph": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "imk_block_apply": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "imk_unet_plan_create": (c_int, [ctypes.POINTER(UnetCfg), ctypes.POINTER(c_void_p)]),
    "imk_unet_plan_destroy": (None, [c_void_p]),
    "imk_unet_param_count": (c_int, [c_void_p, ctypes.POINTER(c_int64), ctypes.POINTER(c_int64)]),
    "imk_unet_num_layers": (c_int, [c_void_p]),
    "imk_unet_layer_info": (c_int, [c_void_p, c_int, ctypes.POINTER(LayerInfo)]),
    "imk_unet_packed_bytes": (c_int64, [c_void_p]),
    "imk_unet_pack_weights": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "imk_unet_workspace_bytes": (c_int64, [c_void_p, c_int, c_int]),
    "imk_unet_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_void_p]),
    "imk_unet_tensor_info": (c_int, [c_void_p, c_int, c_int, c_int, c_int, ctypes.POINTER(c_int64),
                                     ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int),
                                     ctypes.POINTER(c_int)]),
    "imk_unet_forward_im_workspace_bytes": (c_int64, [c_void_p, c_int, c_int, c_int]),
    "imk_unet_forward_im": (c_int, [c_void_p, c_int, ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p),
                                    c_void_p, c_int, c_float, c_int, c_void_p, c_int, c_int,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_int64, c_void_p]),
    "imk_unet_state_bytes": (c_int64, [c_void_p]),
    "imk_unet_state_init": (c_int, [c_void_p, c_void_p, c_void_p]),
    "imk_unet_fwd_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                 c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "imk_evalnet_plan_create": (c_int, [ctypes.POINTER(EvalnetCfg), ctypes.POINTER(c_void_p)]),
    "imk_evalnet_workspace_bytes": (c_int64, [c_void_p, c_int, c_int]),
    "imk_evalnet_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64,
                                    c_void_p]),
    "imk_evalnet_tensor_info": (c_int, [c_void_p, c_int, c_int, c_int, c_int, ctypes.POINTER(c_int64),
                                        ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int),
                                        ctypes.POINTER(c_int)]),
    "imk_evalnet_fwd_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "imk_gather_pairs": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "imk_augment": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "imk_eval_binary": (c_int, [c_void_p, c_float, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "imk_eval_multiclass": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "imk_eval_soft_out_doubles": (c_int64, [c_int]),
    "imk_eval_soft_sums": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "imk_unet_plan_debug": (c_int, [c_void_p, c_int, c_int]),
    "imk_unet_plan_set_bn_momentum": (c_int, [c_void_p, c_float]),
    "imk_unet_plan_get_bn_momentum": (c_int, [c_void_p, ctypes.POINTER(c_float)]),
    "imk_prof_create": (c_int, [c_int, ctypes.POINTER(c_void_p)]),
    "imk_prof_destroy": (None, [c_void_p]),
    "imk_prof_bind": (c_int, [c_void_p]),
    "imk_prof_unbind": (c_int, [c_void_p]),
    "imk_prof_totals_enable": (c_int, [c_void_p, c_int]),
    "imk_prof_totals_dump": (c_int64, [c_void_p, ctypes.c_char_p, c_int64]),
    "imk_prof_mark": (c_int, [c_int, c_void_p]),
    "imk_runtime_warnings": (c_int, []),
    "imk_unet_plan_side_stream": (c_int, [c_void_p, c_int, ctypes.POINTER(c_void_p)]),
    "imk_prof_set_period": (c_int, [c_void_p, c_int]),
    "imk_prof_collect": (c_int, [c_void_p, ctypes.POINTER(c_int64), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                 ctypes.POINTER(ctypes.c_double)]),
    "imk_png_info": (c_int, [ctypes.c_char_p, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "imk_png_read_file": (c_int, [ctypes.c_char_p, c_int, c_void_p, c_int64, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "imk_png_decode": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int64, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "imk_png_encode": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int64, ctypes.POINTER(c_int64)]),
    "imk_png_write_file": (c_int, [ctypes.c_char_p, c_void_p, c_int, c_int, c_int, c_int]),
    "imk_pos_contours": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int]),
    "imk_mod_pos_size": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "imk_cell_count": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "imk_unet_adamw_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                    c_float, c_float, c_float, c_float, c_float, c_void_p]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the product path.")
    # PyTorch-ROCm ships its own HIP / HSA runtime (torch/lib/libamdhip64.so): it has to be the one in the process before
    # libimk.so's dependency on libamdhip64 is resolved -- with /opt/rocm's copy loaded first, torch brings a second runtime
    # and one of the two fails with "no ROCm-capable device is detected" at its first call.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


class ImkError(RuntimeError):
    pass


def check(code, what=""):
    if code != 0:
        msg = lib.imk_error_string(code)
        raise ImkError(f"{what}: imk error {code} ({msg.decode() if msg else '?'})")
