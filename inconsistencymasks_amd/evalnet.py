"""Host-side mirror of the reference's evalnet.py on top of libimk.so.

`get_evalnet(i_height, i_width, inputA_channels, inputB_channels, alpha=2, ...)` (evalnet.py:24) and
`get_evalnet_miou(...)` (evalnet.py:49) keep the reference signatures and return an `EvalNet` whose
`.predict([A, B])` takes uint8 NHWC batches like `tf.keras.Model.predict` and returns what the Keras models return:
one float32 [B,1] array, or the list [iou [B,Cb], detection [B,Cb]].  `get_evalnet_miou_v2` (evalnet.py:76, used by
no shipped script) is not provided.  torch is used for device memory only.
"""
import ctypes

import numpy as np
import torch

from ._lib import EvalnetCfg, check, lib
from .unet import Plan, UNet, _stream


class EvalPlan(Plan):
    """imk_evalnet_plan_create: the same C plan object as the U-Net's (parameter layout, packing, optimizer state)."""

    _ws_fn = "imk_evalnet_workspace_bytes"

    def __init__(self, h, w, ca, cb, n_out, alpha, two_heads, normalize_a, normalize_b, b_onehot=False):
        ch = [int(v * alpha) for v in (16, 32, 64, 128, 256)]          # evalnet.py:28-41
        self.cfg = EvalnetCfg(h, w, ca, cb, n_out, int(two_heads), int(normalize_a), int(normalize_b), (ctypes.c_int * 5)(*ch),
                              int(b_onehot))
        self.h, self.w, self.ca, self.cb, self.n_out, self.alpha = h, w, ca, cb, n_out, alpha
        self.two_heads, self.b_onehot = bool(two_heads), bool(b_onehot)
        self._p = ctypes.c_void_p()
        check(lib.imk_evalnet_plan_create(ctypes.byref(self.cfg), ctypes.byref(self._p)), "imk_evalnet_plan_create")
        self._describe()


class EvalNet(UNet):
    N_STATS = 8     # loss, overflow flag, loss scale, step, loss of head 0 (mse), loss of head 1 (bce), 2 spare

    def __init__(self, h, w, ca, cb, n_out, alpha, two_heads, normalize_a=True, normalize_b=True, seed=None, device="cuda",
                 b_onehot=False):
        self._init_from_plan(EvalPlan(h, w, ca, cb, n_out, alpha, two_heads, normalize_a, normalize_b, b_onehot), seed, device,
                             dense=("dense", "iou", "detection"))
        self.n_heads = 2 if two_heads else 1

    # ---- inference ----------------------------------------------------------------------------------
    def _as_u8(self, x, c):
        t = torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x)
        if t.dim() == 3:
            t = t[None]
        if t.dtype != torch.uint8:
            t = t.round().clamp(0, 255).to(torch.uint8)
        if t.shape[1:] != (self.plan.h, self.plan.w, c):
            raise ValueError(f"expected [B,{self.plan.h},{self.plan.w},{c}], got {tuple(t.shape)}")
        return t.to(self.device).contiguous()

    def _as_input_b(self, x):
        """input B as the kernels take it: the uint8 mask stack [B,H,W,Cb], or with b_onehot the class-id map [B,H,W,1]
        (a one-hot stack [B,H,W,K], which is what the reference's call sites pass, is folded back to class ids)"""
        if not self.plan.b_onehot:
            return self._as_u8(x, self.plan.cb)
        t = torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x)
        if t.dim() == 4 and t.shape[-1] == self.plan.cb and self.plan.cb > 1:
            t = t.argmax(-1)
        if t.dim() == 2:
            t = t[None]
        if t.dim() == 3:
            t = t[..., None]
        return self._as_u8(t, 1)

    def predict_device(self, xa_u8, xb_u8):
        """uint8 device tensors [B,H,W,Ca], [B,H,W,Cb] (b_onehot: class ids [B,H,W,1]) -> float32 [B, n_heads*n_out]."""
        self.ready_for_inference()
        b = xa_u8.shape[0]
        ws = self.workspace(b, 0)
        out = torch.empty((b, self.n_heads * self.plan.n_out), dtype=torch.float32, device=self.device)
        check(lib.imk_evalnet_forward(self.plan.ptr, self.params.data_ptr(), self.packed.data_ptr(), xa_u8.data_ptr(),
                                      xb_u8.data_ptr(), b, out.data_ptr(), ws.data_ptr(), ws.numel(), _stream()),
              "imk_evalnet_forward")
        return out

    def predict(self, x, batch_size=32, verbose=0):
        """Keras-like: [A, B] numpy batches in; [B,1] (get_evalnet) or [iou, detection] (get_evalnet_miou) out."""
        xa, xb = self._as_u8(x[0], self.plan.ca), self._as_input_b(x[1])
        outs = [self.predict_device(xa[i:i + batch_size], xb[i:i + batch_size]) for i in range(0, xa.shape[0], batch_size)]
        o = torch.cat(outs, 0).cpu().numpy()
        k = self.plan.n_out
        return [o[:, :k], o[:, k:]] if self.n_heads == 2 else o

    def intermediate(self, layer_name, batch, mode=0, which=0):
        idx = [l["name"] for l in self.plan.layers].index(layer_name)
        off, h, w, c, cs = ctypes.c_int64(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        check(lib.imk_evalnet_tensor_info(self.plan.ptr, batch, mode, idx, which, ctypes.byref(off), ctypes.byref(h),
                                          ctypes.byref(w), ctypes.byref(c), ctypes.byref(cs)), "imk_evalnet_tensor_info")
        ws = [v for k, v in self._ws.items() if k[0] == batch and k[1] == mode][0]
        n = batch * h.value * w.value * cs.value
        t = ws[off.value:off.value + 2 * n].view(torch.float16).reshape(batch, h.value, w.value, cs.value)
        return t[..., :c.value].float().cpu()

    # ---- training ---------------------------------------------------------------------------------------
    def fwd_bwd(self, xa_u8, xb_u8, y):
        """One forward/backward on device batches (y float32 [B, n_heads*n_out]); fills self.grads / self.stats and
        returns the training-mode outputs [B, n_heads*n_out]."""
        if self.train_state is None:
            self.init_train_state()
        if not self._packed_ok:
            self.repack()
        self._fold_ok = False
        b = xa_u8.shape[0]
        ws = self.workspace(b, 1)
        out = torch.empty((b, self.n_heads * self.plan.n_out), dtype=torch.float32, device=self.device)
        check(lib.imk_evalnet_fwd_bwd(self.plan.ptr, self.params.data_ptr(), self.packed.data_ptr(),
                                      self.train_state.data_ptr(), xa_u8.data_ptr(), xb_u8.data_ptr(), y.data_ptr(), b,
                                      out.data_ptr(), self.grads.data_ptr(), self.stats.data_ptr(), ws.data_ptr(),
                                      ws.numel(), _stream()), "imk_evalnet_fwd_bwd")
        return out

    def train_step(self, xa_u8, xb_u8, y, lr, wd):
        out = self.fwd_bwd(xa_u8, xb_u8, y)
        self.adamw_step(lr, wd)
        return out


def _check_defaults(actifu, ksi, kernel_ini):
    if actifu != "relu" or ksi != 3 or kernel_ini != "he_normal":
        raise NotImplementedError("only the defaults every reference script uses: relu, 3x3, he_normal")


def get_evalnet(i_height, i_width, inputA_channels, inputB_channels, alpha=2, actifu="relu", ksi=3, kernel_ini="he_normal",
                normalize_A=True, normalize_B=True, seed=None, device="cuda"):
    """evalnet.py:24 -- one sigmoid unit."""
    _check_defaults(actifu, ksi, kernel_ini)
    return EvalNet(i_height, i_width, inputA_channels, inputB_channels, 1, alpha, False, normalize_A, normalize_B, seed, device)


def get_evalnet_miou(i_height, i_width, inputA_channels, inputB_channels, alpha=2, actifu="relu", ksi=3,
                     kernel_ini="he_normal", normalize_A=True, normalize_B=False, seed=None, device="cuda", onehot_B=None):
    """evalnet.py:49 -- heads 'iou' and 'detection', inputB_channels sigmoid units each.  onehot_B: input B is the
    one-hot stack of a class-id map (the multiclass scripts, functions.py:4978); it is then passed as class ids and
    expanded on the device.  Default: True when inputB_channels > 4 (SUIM: 9 classes), False for mask stacks (HeLa: 3)."""
    _check_defaults(actifu, ksi, kernel_ini)
    onehot = inputB_channels > 4 if onehot_B is None else bool(onehot_B)
    if onehot and normalize_B:
        raise NotImplementedError("a one-hot input is not divided by 255 in any reference script")
    return EvalNet(i_height, i_width, inputA_channels, inputB_channels, inputB_channels, alpha, True, normalize_A,
                   normalize_B, seed, device, b_onehot=onehot)
